"""BASELINE.json configs 2, 3a, 3b, 4, 5 at FULL size against the CPU oracle.

Each test lowers the full-size matrix through the C ABI, multiplies a seeded random x on the device and compares
with `orc.spmv` (the reference's ascending left fold, Common.hs:247-260 / IntM.hs:78-108) row by row:

  * forms that fold a row in one lane (wdia, wdia-vv: the stencil / banded configs 2, 4, 5) must be BIT-EXACT --
    at size this exercises what the small tests cannot: the plane-tiled `sched[]` walk, the guard-slack reads at both
    ends of x, the 32-bit byte offsets of the record gathers;
  * config 3a (10 M rows, 33 random columns per row: the tile form): the default (CU-wide slices, every row owned by one wavefront)
    folds every row entry by entry in ascending column order: BIT-EXACT; the opt-in tile_relaxed = 1 adds a row's products by LDS
    atomics in timing order -- |dy_i| <= nnz_i * eps * sum_j |a_ij x_j| -- at the same size;
  * config 3b (LDS panels at 2000 entries per row) reduces a row in wavefront segments:
    |dy_i| <= nnz_i * eps * sum_j |a_ij x_j| (the bound of SURVEY 8(a) A1).

Then two `bicgstabStep`s of the oracle (Sparse.hs:972-981) are compared with the device state record at 1e-9.
"""
import gc

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


@pytest.fixture(scope="module")
def sla():
    import sla_amd
    return sla_amd


def _check(sla, dims, rp, ci, va, expect_form, bit_exact, b_mode, tag, ctx=None, cgs=False):
    n = dims[0]
    ctx = ctx or sla.default_context()
    A = sla.fromCSR(dims, rp, ci, va, ctx)
    info = A.kernel_info()
    assert expect_form in info, (tag, info)
    if expect_form.startswith("wdia"):   # (round 4: a value-indexed matrix once got column-panel views on top -- 6 passes per (#>), K1 45 -> 817 us)
        assert "col_panels" not in info, (tag, info)
    Ao = orc.Csr(n, n, rp, ci, va)
    rng = np.random.default_rng(20260928)
    x = rng.standard_normal(n)
    y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
    yo = orc.spmv(Ao, x)
    if bit_exact:
        assert np.array_equal(y, yo), (tag, info, int(np.count_nonzero(y != yo)))
    else:
        bound = np.diff(rp) * EPS * orc.spmv(orc.Csr(n, n, rp, ci, np.abs(va)), np.abs(x))
        bad = np.abs(y - yo) > bound
        assert not bad.any(), (tag, info, int(bad.sum()), float(np.abs(y - yo).max()))
        assert np.linalg.norm(y - yo) <= 1e-14 * np.linalg.norm(yo)
    # two reference steps: state records agree to 1e-9 (x, r, p of Sparse.hs:959-960)
    if b_mode == "ones":
        b = np.add.reduceat(va, rp[:-1])                       # b = A . 1
    else:
        b = orc.spmv(Ao, np.random.default_rng(7).standard_normal(n))   # b = A x*, x* ~ N(0,1) seed 7 (config 3)
    x0 = np.zeros(n)
    so = orc.BicgstabState(Ao, b, x0)
    sd = sla.bicgsInit(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
    so.step(b, 2)                                              # r0hat = b - A 0 = b
    sd.step(2)
    for name, dev, ref in (("x", sd._xBicgstab, so.x), ("r", sd._rBicgstab, so.r), ("p", sd._pBicgstab, so.p)):
        d = dev.toDenseListSV()
        assert np.linalg.norm(d - ref) <= 1e-9 * np.linalg.norm(ref), (tag, name)
    if cgs:   # two cgsStep's (Sparse.hs:928-939) on the same lowered matrix: the CGS epilogues (EPI_AXPY_DOT) of this SpMV form at full size
        so, sc = orc.CgsState(Ao, b, x0), sla.cgsInit(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
        so.step(b, 2)
        sc.step(2)
        for name, dev, ref in (("x", sc._x, so.x), ("r", sc._r, so.r), ("p", sc._p, so.p), ("u", sc._u, so.u)):
            assert np.linalg.norm(dev.toDenseListSV() - ref) <= 1e-9 * np.linalg.norm(ref), (tag, "cgs", name)
        del sc
    del sd, A
    gc.collect()


def test_config2_poisson_1m_bit_exact_vs_oracle(sla):
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.poisson2d(1000, 1000)
    _check(sla, dims, rp, ci, va, "wdia", True, "ones", "config2")


def test_config4_laplace3d_10m_bit_exact_vs_oracle(sla):
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.laplace3d(216, 216, 216)
    assert dims[0] == 10077696
    # the plane-march lowering is what the headline times: a silent fall-back to another wave-sliced variant must fail here
    _check(sla, dims, rp, ci, va, "wdia+march", True, "ones", "config4")


def test_config4_plain_csr_10m_bit_exact_vs_oracle(sla):
    """The kernel behind the metric's "CSR SpMV achieved HBM GB/s" (bench.py's general_csr block: the 216^3 Laplacian stored as plain
    CSR -- f64 values, i32 columns, i32 row pointers -- options wdia = 0 vdict = 0 diag = 0) at full size: spmv_wave_kernel folds a row in
    one lane, ascending -- the reference's left fold (Common.hs:247-260) bit for bit -- then two bicgstabSteps at 1e-9."""
    from sla_amd import workloads as wl
    ctx = sla.Context(0).set_options(wdia=0, vdict=0, diag=0)
    dims, (rp, ci, va) = wl.laplace3d(216, 216, 216)
    _check(sla, dims, rp, ci, va, "stream+wave", True, "ones", "config4 plain CSR", ctx)
    ctx.close()


def test_config4_laplace3d_10m_reference_split_flow_vs_oracle(sla):
    """The default flow fuses K4 and K5 (the test above; since round 3 on sharded contexts too); the reference's own split stays
    selectable.  The same full-size check on a context with the option bicg_fuse45=0."""
    from sla_amd import workloads as wl
    ctx = sla.Context(0).set_option("bicg_fuse45", 0)
    dims, (rp, ci, va) = wl.laplace3d(216, 216, 216)
    _check(sla, dims, rp, ci, va, "wdia+march", True, "ones", "config4 split", ctx)
    ctx.close()


def test_config4_k2_folded_into_k3_at_full_size(sla):
    """BASELINE config 4 as the headline runs it: three launches per bicgstabStep (K2 folded into K3: s = r - alpha Ap, Sparse.hs:975-976,
    built in the staged windows and rebuilt by the K4+K5 sweep, never stored).  Five steps at 216^3 against the four-launch flow
    (bicg_fuse23 = 0).  Every ROW is the same bits either way; the fused dot products are summed per (tile, run) task, and at this size the
    folded instantiation holds 3 workgroups per CU where K3 holds 4, so its planes are cut into other runs and the four sums are grouped
    differently.  With both flows cut for 3 per CU (wd_march_occ = 3) x, r, p must agree bit for bit; the default cut within 1e-12."""
    from sla_amd import _lib, workloads as wl
    dims, (rp, ci, va) = wl.laplace3d(216, 216, 216)
    n = dims[0]
    b = np.add.reduceat(va, rp[:-1])
    out, launches = {}, {}
    for key, opts in (("folded, 3 per CU", {"bicg_fuse23": 1, "wd_march_occ": 3}), ("four launches, 3 per CU", {"bicg_fuse23": 0, "wd_march_occ": 3}),
                      ("folded, default", {"bicg_fuse23": 1})):
        ctx = sla.Context(0).set_options(**opts)
        A = sla.fromCSR(dims, rp, ci, va, ctx)
        assert "wdia+march" in A.kernel_info().split()[0], A.kernel_info()
        st = sla.bicgsInit(A, sla.fromVector(b, ctx), sla.fromVector(np.zeros(n), ctx))
        ctx.prof_start(_lib.KERNEL_ALL, 64)
        st.step(5)
        ctx.prof_stop()
        launches[key] = (ctx.prof_query(_lib.KERNEL_BICG_K2)[0], ctx.prof_query(_lib.KERNEL_SPMV_DOT2)[0], ctx.prof_query(_lib.KERNEL_BICG_K45)[0])
        out[key] = [v.toDenseListSV().copy() for v in (st._xBicgstab, st._rBicgstab, st._pBicgstab)]
        del st, A
        ctx.close()
        gc.collect()
    assert launches["folded, 3 per CU"] == (0, 5, 5) and launches["four launches, 3 per CU"] == (5, 5, 5) and launches["folded, default"] == (0, 5, 5), launches
    for u, v in zip(out["folded, 3 per CU"], out["four launches, 3 per CU"]):
        assert np.array_equal(u.view(np.uint64), v.view(np.uint64)), np.abs(u - v).max()
    for u, v in zip(out["folded, default"], out["four launches, 3 per CU"]):
        assert np.linalg.norm(u - v) <= 1e-12 * np.linalg.norm(v), np.linalg.norm(u - v) / np.linalg.norm(v)


def test_config5_banded_2m_bit_exact_vs_oracle(sla):
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.banded_nonsym(2000000)
    _check(sla, dims, rp, ci, va, "wdia-vv", True, "ones", "config5")


def test_config3a_random_spd_10m_vs_oracle(sla):
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.random_spd(10000000, 16, 42)
    assert dims[0] == 10000000 and rp[-1] == 329999456
    # ... and CGS, the second half of config 3 ("CGS vs BiCGSTAB"), at the same 10 M rows through the tile form's fused epilogues
    _check(sla, dims, rp, ci, va, "exact_fold=1 cu_slices=1 row_owned=1", True, "xstar", "config3a (default: the reference's fold bit for bit)", cgs=True)
    ctx = sla.Context(0).set_option("tile_relaxed", 1)   # the opt-in: relaxed order, within the rounding bound
    _check(sla, dims, rp, ci, va, "exact_fold=0", False, "xstar", "config3a relaxed order", ctx, cgs=True)
    ctx.close()


def test_config3a_cgs_two_steps_vs_oracle_1m(sla):
    """CGS on config 3a's construction (1 M rows: the oracle's two steps stay in seconds)."""
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.random_spd(1000000, 16, 42)
    n = dims[0]
    A, Ao = sla.fromCSR(dims, rp, ci, va), orc.Csr(n, n, rp, ci, va)
    b = orc.spmv(Ao, np.random.default_rng(7).standard_normal(n))
    x0 = np.zeros(n)
    so, sd = orc.CgsState(Ao, b, x0), sla.cgsInit(A, sla.fromVector(b), sla.fromVector(x0))
    so.step(b, 2)
    sd.step(2)
    for name, dev, ref in (("x", sd._x, so.x), ("r", sd._r, so.r), ("p", sd._p, so.p), ("u", sd._u, so.u)):
        assert np.linalg.norm(dev.toDenseListSV() - ref) <= 1e-9 * np.linalg.norm(ref), name


def test_config3b_dense_rows_200k_vs_oracle(sla):
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.random_spd(200000, 1000, 42)
    assert dims[0] == 200000 and 1990 <= rp[-1] / dims[0] <= 2001      # 1 % density honoured
    _check(sla, dims, rp, ci, va, "ldspanels", False, "xstar", "config3b")


def test_config5_arnoldi_2m_kn30_vs_oracle(sla):
    """`arnoldi aa b 30` (Sparse.hs:630-667) on config 5's matrix at FULL size (2 M rows, the basis GMRES(30) builds) against the
    oracle's classical Gram-Schmidt with left-fold sums.  Thirty steps of un-reorthogonalised Gram-Schmidt amplify the difference
    between the two summation orders: measured (round 4) max|H - Ho| = 2.3e-10 max|H|, growing from 1e-13 in column 0.  Most of it
    is the ORACLE's: a 2 M-term left fold carries ~sqrt(n) eps relative to sum|q_i w_i|, the device's two-stage tree far less -- checked
    on h_00 against a long-double evaluation, where the device must be at least as close as the oracle.  Bounds: H to 1e-9 max|H|,
    basis entries to 1e-8, basis orthonormal to 1e-9."""
    from sla_amd import workloads as wl
    n, kn = 2000000, 30
    dims, (rp, ci, va) = wl.banded_nonsym(n)
    A, Ao = sla.fromCSR(dims, rp, ci, va), orc.Csr(n, n, rp, ci, va)
    assert "wdia-vv" in A.kernel_info(), A.kernel_info()
    b = np.add.reduceat(va, rp[:-1])                    # b = A . 1, the right-hand side of config 5
    Q, H = sla.arnoldi(A, sla.fromVector(b), kn)
    rc, Qo, Ho, k = orc.arnoldi(Ao, b, kn)
    assert rc == orc.OK and k == kn and H.shape == Ho.shape == (kn + 1, kn) and Q.shape == Qo.shape == (n, kn + 1)
    dh = np.abs(H - Ho).max() / np.abs(Ho).max()
    assert dh <= 1e-9, dh
    assert np.abs(H[:, 0] - Ho[:, 0]).max() <= 1e-12 * np.abs(Ho).max()          # the first column: before any amplification
    dq = np.abs(Q - Qo).max()
    assert dq <= 1e-8, dq
    # h_00 = q0 . (A q0) in extended precision: whose sum is the accurate one?
    q0 = Qo[:, 0].copy()
    w = orc.spmv(Ao, q0)
    exact = float(np.dot(q0.astype(np.longdouble), w.astype(np.longdouble)))
    assert abs(H[0, 0] - exact) <= abs(Ho[0, 0] - exact) + 4 * np.finfo(np.float64).eps * abs(exact), (H[0, 0], Ho[0, 0], exact)
    del Qo
    G = Q.T @ Q
    assert np.abs(G - np.eye(kn + 1)).max() <= 1e-9
    del A, Q
    gc.collect()
