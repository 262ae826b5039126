"""sla_ctx_create_multi: ONE caller, several devices.  On a 1-GPU box the device id is repeated, which makes the
library use its loopback communicator between the rank contexts (RCCL refuses two ranks per GPU); on a multi-GPU node the
same test runs over RCCL with distinct devices.  The whole reference surface is driven through the parent context exactly like
a single-device one -- whole matrices / vectors in and out -- and compared with the oracle and with a single-device context."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sla():
    import sla_amd
    return sla_amd


@pytest.mark.parametrize("nranks", [2, 3])
def test_one_caller_many_devices(sla, nranks):
    from sla_amd import workloads as wl
    ndev = sla.Context.device_count()
    ids = list(range(nranks)) if ndev >= nranks else [0] * nranks
    ctx = sla.Context.multi(ids)
    assert ctx.comm_ranks() == nranks and ctx.row_range(1000) == (0, 1000)
    dims, (rp, ci, va) = wl.laplace3d(24, 20, 22)
    n = dims[0]
    A = sla.fromCSR(dims, rp, ci, va, ctx)
    Ao = orc.Csr(n, n, rp, ci, va)
    assert "x_exchange=window" in A.kernel_info()
    rpx, cix, vax = A.csr()                                             # whole matrix back (concatenated row blocks)
    assert np.array_equal(rpx, rp) and np.array_equal(cix, ci) and np.array_equal(vax, va)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(n)
    y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
    assert np.array_equal(y, orc.spmv(Ao, x))                           # sharded (#>) = the whole-matrix left fold, bit for bit
    yt = sla.vecMat(sla.fromVector(x, ctx), A).toDenseListSV()
    assert np.allclose(yt, orc.spmv(orc.transpose(Ao), x), rtol=1e-13, atol=1e-13)
    d = sla.dot(sla.fromVector(x, ctx), sla.fromVector(y, ctx))
    assert abs(d - orc.dot(x, y)) <= 1e-11 * abs(orc.dot(x, y))
    assert abs(sla.norm2(sla.fromVector(x, ctx)) - orc.norm2(x)) <= 1e-12 * orc.norm2(x)
    b = orc.spmv(Ao, np.ones(n))
    ctx1 = sla.Context(0)
    A1 = sla.fromCSR(dims, rp, ci, va, ctx1)
    for meth in (sla.BICGSTAB_, sla.CGS_, sla.CGNE_):
        xs, info = sla.linSolve0(meth, A, sla.fromVector(b, ctx), sla.fromVector(np.zeros(n), ctx), return_info=True)
        x1, info1 = sla.linSolve0(meth, A1, sla.fromVector(b, ctx1), sla.fromVector(np.zeros(n), ctx1), return_info=True)
        assert info["converged"] == info1["converged"] and abs(info["iters"] - info1["iters"]) <= 3, (meth, info, info1)
        if info["converged"]:
            assert np.linalg.norm(orc.spmv(Ao, xs.toDenseListSV()) - b) <= info["tol"] * (1 + 1e-9)
    # the state records: pure steps through clone, fields downloaded whole
    s0 = sla.bicgsInit(A, sla.fromVector(b, ctx), sla.fromVector(np.zeros(n), ctx))
    s2 = sla.bicgstabStep(A, None, sla.bicgstabStep(A, None, s0))
    so = orc.BicgstabState(Ao, b, np.zeros(n))
    so.step(b, 2)
    assert np.linalg.norm(s2._xBicgstab.toDenseListSV() - so.x) <= 1e-9 * np.linalg.norm(so.x)
    assert np.array_equal(s0._xBicgstab.toDenseListSV(), np.zeros(n))
    # arnoldi: Q assembled from the ranks' row blocks, H from rank 0
    Q, H = sla.arnoldi(A, sla.fromVector(b, ctx), 6)
    rc, Qo, Ho, k = orc.arnoldi(Ao, b, 6)
    assert Q.shape == Qo.shape and np.abs(H - Ho).max() <= 1e-9 * np.abs(Ho).max() and np.abs(Q - Qo).max() <= 1e-8
    xg = sla.linSolve(A, sla.fromVector(b, ctx)).toDenseListSV()        # (<\>): GMRES from 0.1 * ones
    assert np.linalg.norm(orc.spmv(Ao, xg) - b) <= 1e-4 * np.linalg.norm(b) + 1e-6
    # what is not sharded says so
    with pytest.raises(sla.SlaError) as e:
        sla.matMat(A, A)
    assert "multi-device" in str(e.value)
    with pytest.raises(sla.SlaError):
        sla.mSsorPre(A, 1.0)
    del s0, s2, A
    ctx.close()
