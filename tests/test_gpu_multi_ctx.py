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
    # bcgStep (the extension of Sparse.hs:886-909): (#>) and (<#) per step, the five fields downloaded whole
    g0 = sla.bcgInit(A, sla.fromVector(b, ctx), sla.fromVector(np.zeros(n), ctx))
    g2 = sla.bcgStep(A, sla.bcgStep(A, g0))
    go = orc.BcgState(Ao, b, np.zeros(n))
    go.step(2)
    for got, want in ((g2._xBcg, go.x), (g2._rBcg, go.r), (g2._rHatBcg, go.rhat), (g2._pBcg, go.p), (g2._pHatBcg, go.phat)):
        assert np.linalg.norm(got.toDenseListSV() - want) <= 1e-9 * max(np.linalg.norm(want), 1e-300)
    assert np.array_equal(g0._pHatBcg.toDenseListSV(), b)               # the record the pure step started from keeps its value
    del g0, g2
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


def test_options_are_typed_per_context_and_reach_every_rank(sla):
    """sla_ctx_set_option: the knob table without the environment.  Two contexts in one process hold different settings; a lowering
    knob applies to matrices created afterwards; a multi-device parent hands the option to every rank context."""
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.laplace3d(20, 20, 20)
    a, b = sla.Context(0), sla.Context(0)
    a.set_options(wdia=0, vdict=0, diag=0)
    assert a.get_option("wdia") == "0" and b.get_option("wdia") == "1"
    Aa, Ab = sla.fromCSR(dims, rp, ci, va, a), sla.fromCSR(dims, rp, ci, va, b)
    assert "algo=stream" in Aa.kernel_info() and "algo=wdia" in Ab.kernel_info()
    x = np.random.default_rng(1).standard_normal(dims[0])
    assert np.array_equal(sla.matVec(Aa, sla.fromVector(x, a)).toDenseListSV(), sla.matVec(Ab, sla.fromVector(x, b)).toDenseListSV())
    with pytest.raises(sla.SlaError):
        a.set_option("no_such_knob", 1)
    with pytest.raises(sla.SlaError):
        a.set_option("tile_shift", 99)
    assert a.set_option("x_exchange", "window").get_option("x_exchange") == "window"
    # the knobs of round 3: defaults, ranges
    assert b.get_option("wd_march") == "1" and b.get_option("wd_march_occ") == "4" and b.get_option("lp_copy") == "1"
    assert b.get_option("tile_relaxed") == "0" and b.get_option("tile_rowown") == "-1"   # no default form is order-relaxed (end of round 6)
    assert int(b.get_option("vec_policy")) == 0x2bff          # every BiCGSTAB stream but the p store, CGS's q and u stores: past the caches
    for name, bad in (("wd_march", 3), ("wd_march_occ", 5), ("lp_copy", 2), ("vec_policy", 1 << 15)):
        with pytest.raises(sla.SlaError):
            a.set_option(name, bad)
    assert a.set_options(wd_march=0, vec_policy=0x3fb).get_option("vec_policy") == str(0x3fb)
    m = sla.Context.multi([0, 0])
    m.set_option("bicg_ghost", 0)
    assert m.get_option("bicg_ghost") == "0"
    del Aa, Ab
    for c in (a, b, m):
        c.close()


def test_single_device_handles_do_not_mix_with_bundles(sla):
    """a bundle of a multi-device context carries n but no device pointer: combining it with a single-device matrix or
    vector passed the dimension checks and faulted on the GPU.  Every such combination is refused with SLA_ERR_INVALID."""
    from sla_amd import _lib, workloads as wl
    import ctypes as C
    lib = _lib.lib()
    dims, (rp, ci, va) = wl.laplace3d(8, 8, 8)
    n = dims[0]
    one, many = sla.Context(0), sla.Context.multi([0, 0])
    A1 = sla.fromCSR(dims, rp, ci, va, one)
    v1, w1 = sla.DeviceVector(one, n, np.ones(n)), sla.DeviceVector(one, n)
    vm, wm = sla.DeviceVector(many, n, np.ones(n)), sla.DeviceVector(many, n)
    d = C.c_double()
    info = _lib.SolveInfo()
    st = C.c_void_p()
    calls = [lambda: lib.sla_spmv(A1.h, vm.h, w1.h), lambda: lib.sla_spmv(A1.h, v1.h, wm.h), lambda: lib.sla_spmv_t(A1.h, vm.h, w1.h),
             lambda: lib.sla_dot(v1.h, vm.h, C.byref(d)), lambda: lib.sla_axpby(1.0, v1.h, 1.0, vm.h), lambda: lib.sla_vec_copy(v1.h, vm.h),
             lambda: lib.sla_solver_init(4, A1.h, vm.h, w1.h, C.byref(st)),
             lambda: lib.sla_linsolve0(4, A1.h, v1.h, w1.h, None, wm.h, C.byref(info)),
             lambda: lib.sla_gmres(A1.h, vm.h, w1.h, 10, None, w1.h, C.byref(info)),
             lambda: lib.sla_dot(vm.h, v1.h, C.byref(d)), lambda: lib.sla_axpby(1.0, vm.h, 1.0, v1.h)]
    for i, f in enumerate(calls):
        assert f() == _lib.ERR_INVALID, i
    other = sla.Context(0)
    vo = sla.DeviceVector(other, n, np.ones(n))
    assert lib.sla_spmv(A1.h, vo.h, w1.h) == _lib.ERR_INVALID          # a vector of another single-device context
    assert lib.sla_spmv(A1.h, v1.h, w1.h) == _lib.OK
    del A1, v1, w1, vm, wm, vo
    for c in (one, many, other):
        c.close()
