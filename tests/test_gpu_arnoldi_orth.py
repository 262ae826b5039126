"""The Gram-Schmidt of an Arnoldi step (Sparse.hs:655-667) as ONE persistent launch (csrc/sla_arnoldi_orth.hip, option arn_orth, default on for single-rank
contexts whose rows fit 8192 per CU): against the three-launch flow (dots | update | normalisation) and against the oracle -- H, Q, the breakdown
behaviour, GMRES iterates -- at block shapes that exercise short last blocks, odd row counts (a trailing single row), one and many workgroups."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sla():
    import sla_amd
    return sla_amd


def _arnoldi(sla, dims, rp, ci, va, b, kn, orth):
    ctx = sla.Context(0).set_options(arn_orth=orth)
    A = sla.fromCSR(dims, rp, ci, va, ctx)
    Q, H = sla.arnoldi(A, sla.fromVector(b, ctx), kn)
    launches = int(ctx.get_option("arn_orth_launches"))
    del A
    ctx.close()
    return Q, H, launches


@pytest.mark.parametrize("n,kn", [(2048, 5), (2049, 7), (5000, 12), (70001, 31), (300000, 9), (1300001, 6)])
def test_fused_step_against_the_launch_flow_and_the_oracle(sla, n, kn):
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.banded_nonsym(n)
    Ao = orc.Csr(n, n, rp, ci, va)
    b = np.random.default_rng(n).standard_normal(n)
    Q1, H1, l1 = _arnoldi(sla, dims, rp, ci, va, b, kn, 1)
    Q0, H0, l0 = _arnoldi(sla, dims, rp, ci, va, b, kn, 0)
    assert l1 == kn and l0 == 0                                            # one persistent launch per step / none
    assert H1.shape == H0.shape == (kn + 1, kn) and Q1.shape == Q0.shape == (n, kn + 1)
    assert np.abs(H1 - H0).max() <= 1e-11 * np.abs(H0).max() and np.abs(Q1 - Q0).max() <= 1e-10
    assert np.abs(Q1.T @ Q1 - np.eye(kn + 1)).max() <= 1e-10
    if n <= 300000:
        rc, Qo, Ho, k = orc.arnoldi(Ao, b, kn)
        assert np.abs(H1 - Ho).max() <= 1e-10 * np.abs(Ho).max() and np.abs(Q1 - Qo).max() <= 1e-9
    else:                                                                   # checkArnoldi's identity (LibSpec.hs:642-653) on a sample of rows
        rows = np.random.default_rng(1).integers(0, n, 2000)
        AQ = np.stack([orc.spmv(Ao, Q1[:, j])[rows] for j in range(kn)], axis=1)
        assert np.abs(AQ - (Q1 @ H1)[rows]).max() <= 1e-10 * np.abs(H1).max()


def test_columns_beyond_the_fused_range_take_the_launch_flow(sla):
    from sla_amd import workloads as wl
    n, kn = 4096, 34                                                        # 35 basis columns > the 32 one launch handles
    dims, (rp, ci, va) = wl.banded_nonsym(n)
    b = np.random.default_rng(2).standard_normal(n)
    Q1, H1, l1 = _arnoldi(sla, dims, rp, ci, va, b, kn, 1)
    assert l1 == 0
    rc, Qo, Ho, k = orc.arnoldi(orc.Csr(n, n, rp, ci, va), b, kn)
    assert np.abs(H1 - Ho).max() <= 1e-10 * np.abs(Ho).max()


def test_breakdown_like_the_launch_flow(sla):
    """A Krylov space that ends early: a diagonal matrix with three distinct values -- the fourth step's vector is (numerically) in the span, the
    reference stops with the columns it has (Sparse.hs:665-667).  Same number of columns, same H, from both flows and the oracle."""
    n = 6000
    d = np.array([2.0, 3.0, 5.0])[np.arange(n) % 3]
    rp, ci = np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int64)
    b = np.ones(n)
    Q1, H1, l1 = _arnoldi(sla, (n, n), rp, ci, d, b, 8, 1)
    Q0, H0, l0 = _arnoldi(sla, (n, n), rp, ci, d, b, 8, 0)
    rc, Qo, Ho, k = orc.arnoldi(orc.Csr(n, n, rp, ci, d), b, 8)
    assert H1.shape == H0.shape == (4, 3), (H1.shape, H0.shape)              # three distinct eigenvalues: the space is exhausted with the third column
    assert np.abs(H1 - H0).max() <= 1e-11 * np.abs(H0).max() and l1 >= 3
    # (the oracle's h_{4,3} is rounding noise of the same size as the 1e-12 threshold and may land on either side of it: compared on the leading block)
    assert Ho.shape[1] >= 3 and np.abs(H1[:3, :3] - Ho[:3, :3]).max() <= 1e-9 * np.abs(Ho[:3, :3]).max()


def test_gmres_through_the_fused_step(sla):
    from sla_amd import workloads as wl
    n = 200000
    dims, (rp, ci, va) = wl.banded_nonsym(n)
    Ao = orc.Csr(n, n, rp, ci, va)
    b = orc.spmv(Ao, np.ones(n))
    out = {}
    for orth in (1, 0):
        ctx = sla.Context(0).set_options(arn_orth=orth)
        A = sla.fromCSR(dims, rp, ci, va, ctx)
        x, info = sla.gmres(A, sla.fromVector(b, ctx), sla.fromVector(np.zeros(n), ctx), restart=30, return_info=True)
        out[orth] = (x.toDenseListSV(), info, int(ctx.get_option("arn_orth_launches")))
        del A
        ctx.close()
    (x1, i1, l1), (x0, i0, l0) = out[1], out[0]
    assert l1 > 0 and l0 == 0 and i1["converged"] and i0["converged"] and i1["iters"] == i0["iters"]
    assert np.linalg.norm(x1 - x0) <= 1e-10 * np.linalg.norm(x0)
    assert np.linalg.norm(orc.spmv(Ao, x1) - b) <= i1["tol"] * (1 + 1e-9)
    rc, xo, it_o, res_o, r0_o = orc.gmres(Ao, b, np.zeros(n), restart=30, max_restarts=10)
    assert i1["iters"] == it_o and np.linalg.norm(x1 - xo) <= 1e-9 * np.linalg.norm(xo)


def test_a_lost_workgroup_falls_back_to_the_launch_flow(sla):
    """arn_orth_fault = 1: the last workgroup of every fused step leaves at once, the others' first barrier times out (~2 s), the step flags
    SLA_FLAG_SYNC_TIMEOUT, the later steps of the run return at once -- and the host repeats the whole run on the launch flow and keeps it for the context."""
    from sla_amd import workloads as wl
    n, kn = 40000, 6
    dims, (rp, ci, va) = wl.banded_nonsym(n)
    b = np.random.default_rng(8).standard_normal(n)
    ctx = sla.Context(0).set_options(arn_orth=1, arn_orth_fault=1)
    A = sla.fromCSR(dims, rp, ci, va, ctx)
    Q, H = sla.arnoldi(A, sla.fromVector(b, ctx), kn)
    assert int(ctx.get_option("arn_orth_fallbacks")) == 1 and ctx.get_option("arn_orth") == "0"
    rc, Qo, Ho, k = orc.arnoldi(orc.Csr(n, n, rp, ci, va), b, kn)
    assert np.abs(H - Ho).max() <= 1e-10 * np.abs(Ho).max() and np.abs(Q - Qo).max() <= 1e-9
    Q2, H2 = sla.arnoldi(A, sla.fromVector(b, ctx), kn)                     # the context stays on the launch flow: no second timeout
    assert int(ctx.get_option("arn_orth_fallbacks")) == 1 and np.array_equal(H2, H)
    del A
    ctx.close()
