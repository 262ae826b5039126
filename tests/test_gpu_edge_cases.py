"""Edge cases of the C ABI on the GPU: empty / ragged / rectangular inputs, degenerate sizes, misuse."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sla():
    import sla_amd
    return sla_amd


def test_empty_matrix_and_zero_vectors(sla):
    A = sla.fromListSM((5, 5), [])
    assert A.nnz() == 0 and A.csr()[0].tolist() == [0] * 6
    y = sla.matVec(A, sla.onesSV(5))
    assert y.nnz() == 0 and y.toDenseListSV().tolist() == [0.0] * 5       # no row keys at all
    assert not A.isDiagonalSM()
    z = sla.zeroSV(4)
    assert sla.dot(z, sla.onesSV(4)) == 0.0 and sla.norm2(z) == 0.0


def test_one_by_one_and_diagonal(sla):
    A = sla.fromListSM((1, 1), [(0, 0, 4.0)])
    assert A.isDiagonalSM()
    x = sla.linSolve0(sla.BICGSTAB_, A, sla.mkSpVR(1, [2.0]), sla.mkSpVR(1, [0.0]))
    assert x.toDenseListSV().tolist() == [0.5]


def test_rectangular_matvec_and_vecmat(sla):
    rng = np.random.default_rng(3)
    m, n, nnz = 70, 1300, 2500
    r, c, v = rng.integers(0, m, nnz), rng.integers(0, n, nnz), rng.standard_normal(nnz)
    A = sla.fromCOO((m, n), r, c, v)
    rc, Ao = orc.coo_to_csr(m, n, r, c, v)
    x, w = rng.standard_normal(n), rng.standard_normal(m)
    assert np.allclose(sla.matVec(A, sla.fromVector(x)).toDenseListSV(), orc.spmv(Ao, x), rtol=1e-13, atol=1e-13)
    assert np.allclose(sla.vecMat(sla.fromVector(w), A).toDenseListSV(), orc.spmv(orc.transpose(Ao), w), rtol=1e-13, atol=1e-13)
    with pytest.raises(sla.MatVecSizeMismatchException):
        sla.matVec(A, sla.onesSV(m))


def test_rows_at_block_limits(sla):
    # rows of exactly 1024 / 1025 entries (LDS row-block limit) and a 256-row block of single entries
    rng = np.random.default_rng(9)
    n = 4000
    rows, cols, vals = [], [], []
    for i, k in ((0, 1024), (1, 1025), (2, 1), (3, 1023), (4, 2049)):
        cj = np.sort(rng.choice(n, size=k, replace=False))
        rows.append(np.full(k, i)); cols.append(cj); vals.append(rng.standard_normal(k))
    for i in range(5, 5 + 600):
        rows.append([i]); cols.append([int(rng.integers(0, n))]); vals.append([float(rng.standard_normal())])
    r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    A = sla.fromCOO((n, n), r, c, v)
    rc, Ao = orc.coo_to_csr(n, n, r, c, v)
    x = rng.standard_normal(n)
    y, yo = sla.matVec(A, sla.fromVector(x)).toDenseListSV(), orc.spmv(Ao, x)
    assert np.allclose(y, yo, rtol=1e-12, atol=1e-12)
    assert np.array_equal(y[5:605], yo[5:605])


def test_duplicate_policy_sum_extension(sla):
    A = sla.fromCOO((2, 2), [0, 0, 1], [1, 1, 0], [1.0, 2.0, 5.0], dup_policy=1)
    assert A.toDense().tolist() == [[0.0, 3.0], [5.0, 0.0]]


def test_non_canonical_csr_rejected(sla):
    with pytest.raises(sla.SlaError):
        sla.fromCSR((2, 2), [0, 2, 2], [1, 0], [1.0, 2.0])          # descending columns
    with pytest.raises(sla.IndexOutOfBounds):
        sla.fromCSR((2, 2), [0, 1, 2], [0, 5], [1.0, 2.0])


def test_null_handles_and_bad_arguments(sla):
    L = sla._lib.lib()
    assert L.sla_spmv(None, None, None) == sla._lib.ERR_INVALID
    assert L.sla_solver_step(None, 1) == sla._lib.ERR_INVALID
    out = C.c_void_p()
    assert L.sla_ctx_create(99, C.byref(out)) == sla._lib.ERR_INVALID
    A = sla.fromListSM((3, 3), [(0, 0, 1.0), (1, 2, 2.0), (2, 1, 1.0)])
    with pytest.raises(sla.SlaError):
        sla.arnoldi(A, sla.onesSV(3), 0)
    with pytest.raises(sla.MatVecSizeMismatchException):
        sla.cgsInit(A, sla.onesSV(4), sla.onesSV(3))


def test_nan_propagation_like_reference(sla):
    # iterating past convergence gives 0/0 = NaN in the reference (README.md:220); no exception here either
    A = sla.fromListSM((3, 3), [(0, 0, 2), (1, 0, 4), (1, 1, 3), (1, 2, 2), (2, 2, 5)])
    s = sla.bicgsInit(A, sla.fromListDenseSV(3, [3, 2, 5]), sla.fromListSV(3, []))
    s.step(20)
    x = s._xBicgstab.toDenseListSV()
    assert x.shape == (3,)                                            # NaN or converged: just no crash
    x, info = sla.linSolve0(sla.BICGSTAB_, A, sla.fromListDenseSV(3, [3, 2, 5]), sla.fromListSV(3, []), return_info=True)
    assert info["converged"] and info["iters"] <= 3


def test_rerun_is_bit_identical(sla):
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.banded_nonsym(30000)
    A = sla.fromCSR(dims, rp, ci, va)
    b = sla.fromVector(np.random.default_rng(1).standard_normal(dims[0]))
    x0 = sla.fromVector(np.zeros(dims[0]))
    a = sla.linSolve0(sla.BICGSTAB_, A, b, x0).toDenseListSV()
    c = sla.linSolve0(sla.BICGSTAB_, A, b, x0).toDenseListSV()
    assert np.array_equal(a, c)                                       # deterministic two-stage reductions


def test_solve_opts_struct_size_window(sla):
    """a caller from before struct_size existed has max_iters as its first member; the reference's default nits = 200 must not
    pass for a size (it did: 200 > 48, <= 48 + 256, 200 % 8 == 0 -- 48 bytes were then read from a 32-byte struct).  Sizes that pass: the
    library's own layout and a newer header's larger one, 8-byte granularity, at most 64 bytes more."""
    import ctypes as C
    from sla_amd import _lib
    A = sla.fromListSM((3, 3), [(0, 0, 2.0), (0, 1, -1.0), (1, 0, -1.0), (1, 1, 2.0), (1, 2, -1.0), (2, 1, -1.0), (2, 2, 2.0)])
    b, x0 = sla.fromListDenseSV(3, [1, 0, 1]).device(), sla.fromListSV(3, []).device()
    out = sla.DeviceVector(A.ctx, 3)
    own = C.sizeof(_lib.SolveOpts)
    for size, ok in ((own, True), (own + 8, True), (own + 64, True), (own + 72, False), (200, False), (own - 8, False), (0, False), (own + 4, False)):
        o, info = _lib.SolveOpts(), _lib.SolveInfo()
        o.struct_size = size
        rc = _lib.lib().sla_linsolve0(int(sla.BICGSTAB_), A.h, b.h, x0.h, C.byref(o), out.h, C.byref(info))
        assert (rc == 0) is ok, (size, rc)
        if not ok:
            assert rc == _lib.ERR_INVALID and b"struct_size" in _lib.lib().sla_last_error()
    info = _lib.SolveInfo()
    info.struct_size = 200
    assert _lib.lib().sla_linsolve0(int(sla.BICGSTAB_), A.h, b.h, x0.h, None, out.h, C.byref(info)) == _lib.ERR_INVALID


def test_dual_spmv_flow_equals_three_sweep_flow(sla):
    """linSolve0 with the true residual fused into the next K1 (default) must return exactly what the
    three-SpMV-per-iteration flow returns: same iterate, same iteration count, same residual."""
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.laplace3d(30, 20, 25)
    n = dims[0]
    b = np.random.default_rng(4).standard_normal(n)
    res = []
    for dual in ("1", "0"):
        ctx = sla.Context(0).set_option("dual_spmv", dual)
        A = sla.fromCSR(dims, rp, ci, va, ctx)
        for meth in (sla.BICGSTAB_, sla.CGS_):
            for ce in (16, 5, 1):
                x, info = sla.linSolve0(meth, A, sla.fromVector(b, ctx), sla.fromVector(np.zeros(n), ctx),
                                        return_info=True, check_every=ce)
                res.append((dual, int(meth), ce, x.toDenseListSV(), info["iters"], info["resnorm"], info["converged"]))
        del A
        ctx.close()
    half = len(res) // 2
    ref = {}
    for (dual, meth, ce, x, it, rn, cv) in res:
        key = meth
        if key not in ref:
            ref[key] = (x, it, rn, cv)
        assert cv and it == ref[key][1] and np.array_equal(x, ref[key][0]) and rn == ref[key][2], (dual, meth, ce, it)


def test_consecutive_long_rows_wave_per_row_and_block_per_row(sla):
    # rows of 1025..16384 entries are grouped 4 per row block (one wavefront each); longer rows get the
    # whole workgroup; mixed with short rows in between
    rng = np.random.default_rng(21)
    n = 30000
    lens = [1500, 3000, 1100, 2000, 5000, 20000, 1200, 3, 0, 1025, 16384, 16385, 7]
    rows, cols, vals = [], [], []
    for i, k in enumerate(lens):
        if k:
            cj = np.sort(rng.choice(n, size=k, replace=False))
            rows.append(np.full(k, i)); cols.append(cj); vals.append(rng.standard_normal(k))
    r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    m = len(lens)
    A = sla.fromCOO((m, n), r, c, v)
    rc, Ao = orc.coo_to_csr(m, n, r, c, v)
    x = rng.standard_normal(n)
    y, yo = sla.matVec(A, sla.fromVector(x)).toDenseListSV(), orc.spmv(Ao, x)
    bound = np.diff(Ao.rowptr) * np.finfo(float).eps * orc.spmv(orc.Csr(m, n, Ao.rowptr, Ao.colidx, np.abs(Ao.val)), np.abs(x))
    assert np.all(np.abs(y - yo) <= bound + 1e-300)


@pytest.mark.parametrize("policy", [0, 1])
def test_device_coo_sort_equals_host_builder_and_oracle(sla, policy):
    """Large triple lists are sorted / deduplicated on the GPU (rocPRIM radix sort); the result must be
    bit-identical to the host builder and, for last-wins, to the oracle's fromListSM restatement."""
    rng = np.random.default_rng(77)
    m, n, nnz = 5000, 7000, 200000
    r, c = rng.integers(0, m, nnz), rng.integers(0, n, nnz)
    r[:500], c[:500] = r[500:1000], c[500:1000]          # plenty of duplicates, some of them triples
    r[1000:1200], c[1000:1200] = r[:200], c[:200]
    v = rng.standard_normal(nnz)
    outs = []
    for thr in ("1", str(1 << 40)):                       # device path, host path
        ctx = sla.Context(0).set_option("device_coo_min", thr)
        A = sla.fromCOO((m, n), r, c, v, ctx, dup_policy=policy)
        outs.append(tuple(a.copy() for a in A.csr()))
        del A
        ctx.close()
    for a, b_ in zip(*outs):
        assert np.array_equal(a, b_)
    if policy == 0:
        rc, Ao = orc.coo_to_csr(m, n, r, c, v)
        assert np.array_equal(outs[0][0], Ao.rowptr) and np.array_equal(outs[0][1], Ao.colidx) and np.array_equal(outs[0][2], Ao.val)
    ctx = sla.Context(0).set_option("device_coo_min", 1)
    with pytest.raises(sla.IndexOutOfBounds):
        sla.fromCOO((4, 4), [0, 4], [0, 0], [1.0, 1.0], ctx)
    ctx.close()


def test_column_panel_spmv_equals_plain_path(sla):
    """Irregular matrices whose x does not fit the L2 are swept in column panels (y += A_p x in ascending panel
    order, each pass continuing the row's running sum).  Forced here with tiny panels: the result must equal the
    un-panelled path bit for bit on short rows, and the solvers must behave identically."""
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.random_spd(3000, 3, 11)          # ~7 entries per row, random columns
    n = dims[0]
    Ao = orc.Csr(n, n, rp, ci, va)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(n)
    b = orc.spmv(Ao, rng.standard_normal(n))
    res = {}
    for mode, opts in (("panels", {"panel_cols": 500}), ("plain", {"panels": 0})):
        ctx = sla.Context(0).set_options(**opts)
        A = sla.fromCSR(dims, rp, ci, va, ctx)
        assert ("colpanels" in A.kernel_info()) == (mode == "panels"), A.kernel_info()
        y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
        yt = sla.vecMat(sla.fromVector(x, ctx), A).toDenseListSV()
        out = [y, yt]
        for meth in (sla.BICGSTAB_, sla.CGS_, sla.CGNE_):
            xs, info = sla.linSolve0(meth, A, sla.fromVector(b, ctx), sla.fromVector(np.zeros(n), ctx), return_info=True)
            out += [xs.toDenseListSV(), info["iters"], info["resnorm"]]
        Q, H = sla.arnoldi(A, sla.fromVector(b, ctx), 8)
        out += [H]
        res[mode] = out
        del A
        ctx.close()
    assert np.array_equal(res["panels"][0], orc.spmv(Ao, x))                # the reference's left fold, bit for bit
    assert np.array_equal(res["panels"][0], res["plain"][0]) and np.array_equal(res["panels"][1], res["plain"][1])
    # the fused inner products are grouped by the LAST panel's row blocks, so solver scalars differ in the last
    # bits between the two paths: compare to solver tolerance, not bitwise
    for a, b_ in zip(res["panels"][2:], res["plain"][2:]):
        if isinstance(a, np.ndarray):
            assert np.linalg.norm(a - b_) <= 1e-8 * max(np.linalg.norm(b_), 1e-30)
        elif isinstance(a, int):
            assert abs(a - b_) <= 1
        else:
            assert abs(a - b_) <= 1e-6 * max(abs(b_), 1e-6)


def test_column_panels_with_long_and_mid_rows(sla):
    rng = np.random.default_rng(31)
    m = n = 6000
    lens = rng.choice([0, 2, 9, 40, 300, 1500, 5000], size=m, p=[0.05, 0.5, 0.3, 0.1, 0.03, 0.015, 0.005])
    rows, cols, vals = [], [], []
    for i, k in enumerate(lens):
        if k:
            cj = rng.choice(n, size=int(k), replace=False)
            rows.append(np.full(int(k), i)); cols.append(cj); vals.append(rng.standard_normal(int(k)))
    r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    rc, Ao = orc.coo_to_csr(m, n, r, c, v)
    x = rng.standard_normal(n)
    yo = orc.spmv(Ao, x)
    bound = (np.diff(Ao.rowptr) + 8) * np.finfo(float).eps * orc.spmv(orc.Csr(m, n, Ao.rowptr, Ao.colidx, np.abs(Ao.val)), np.abs(x))
    # (52 entries per row on average and x fits one LDS panel: the lowering prefers the LDS-panel form by default)
    for lpanel, form in (("0", "colpanels"), ("1", "ldspanels")):
        ctx = sla.Context(0).set_options(panel_cols=700, lpanel=lpanel)
        A = sla.fromCOO((m, n), r, c, v, ctx)
        assert form in A.kernel_info()
        y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
        assert np.all(np.abs(y - yo) <= bound + 1e-300), form
        del A
        ctx.close()


def test_64_bit_row_pointer_kernels(sla):
    """Matrices with more than 2^31 - 1 stored entries switch every general kernel to its int64 row-pointer
    instantiation (the value-indexed forms are 32-bit only and step aside).  the option force_rp64=1 runs those instantiations
    at test sizes: stencil (stream + dictionary codes + x window), irregular short rows (stream, column panels), dense
    rows (LDS panels) and rows at the row-block limits, (#>), (<#) and the solver epilogues against the oracle."""
    from sla_amd import workloads as wl
    rng = np.random.default_rng(41)

    def dense_rows(m, n, k):
        rp = np.arange(m + 1, dtype=np.int64) * k
        ci = np.concatenate([np.sort(rng.choice(n, size=k, replace=False)) for _ in range(m)]).astype(np.int64)
        return (m, n), (rp, ci, rng.uniform(-1, 1, m * k))

    def limits():
        n, rows, cols, vals = 4000, [], [], []
        for i, k in ((0, 1024), (1, 1025), (2, 1), (3, 2049), (5, 3000)):
            cj = np.sort(rng.choice(n, size=k, replace=False))
            rows.append(np.full(k, i)); cols.append(cj); vals.append(rng.standard_normal(k))
        rc, A = orc.coo_to_csr(n, n, np.concatenate(rows), np.concatenate(cols), np.concatenate(vals))
        return (n, n), (A.rowptr, A.colidx, A.val)

    cases = {
        "laplace3d": (wl.laplace3d(13, 9, 11), {}, "algo=stream+diagdict "),
        "laplace3d, dictionary codes + x window": (wl.laplace3d(13, 9, 11), {"xwin": 2}, "algo=stream+diagdict+xwin"),
        "laplace3d, plain stream": (wl.laplace3d(13, 9, 11), {"diag": 0, "xwin": 0}, "algo=stream "),
        "random_spd short rows": (wl.random_spd(3000, 4, 3), {}, "algo=stream"),
        "laplace3d, x window (narrow loads)": (wl.laplace3d(13, 9, 11), {"diag": 0, "stream_wide": 0}, "algo=stream+xwin"),
        "random_spd, narrow loads": (wl.random_spd(3000, 4, 3), {"stream_wide": 0}, "algo=stream"),
        "random_spd, column panels": (wl.random_spd(5000, 6, 4), {"panel_cols": 600}, "colpanels"),
        "dense rows, LDS panels": (dense_rows(900, 20000, 120), {}, "ldspanels"),
        "row-block limits": (limits(), {"lpanel": 0}, "algo=stream"),
        "scalar kernel": (wl.random_spd(2000, 5, 6), {"spmv_algo": "scalar"}, "scalar"),
    }
    for name, ((dims, csr), opts, form) in cases.items():
        opts = dict(opts, wdia=0, vdict=0, stream_wave=0)       # (the 32-bit run takes the same general form as the 64-bit one: spmv_wave_kernel is 32-bit only)
        m, n = dims
        Ao = orc.Csr(m, n, *csr)
        x, w = rng.standard_normal(n), rng.standard_normal(m)
        want, want_t = orc.spmv(Ao, x), orc.spmv(orc.transpose(Ao), w)
        res = {}
        for rp64 in ("1", "0"):
            ctx = sla.Context(0).set_options(force_rp64=rp64, **opts)
            A = sla.fromCSR(dims, *csr, ctx)
            info = A.kernel_info()
            assert form in info + " " and ("rowptr=i64" in info) == (rp64 == "1"), (name, info)
            y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
            yt = sla.vecMat(sla.fromVector(w, ctx), A).toDenseListSV()
            out = [y, yt]
            if m == n and "limits" not in name and "dense" not in name:
                b = orc.spmv(Ao, np.ones(n))
                for meth in (sla.BICGSTAB_, sla.CGS_, sla.CGNE_):
                    xs, inf = sla.linSolve0(meth, A, sla.fromVector(b, ctx), sla.fromVector(np.zeros(n), ctx), return_info=True)
                    out += [xs.toDenseListSV(), inf["iters"]]
            res[rp64] = out
            del A
            ctx.close()
        assert np.allclose(res["1"][0], want, rtol=1e-12, atol=1e-12), name
        assert np.allclose(res["1"][1], want_t, rtol=1e-12, atol=1e-12), name
        for a, b_ in zip(res["1"], res["0"]):       # the index width changes nothing else: same bits, same iteration counts
            assert np.array_equal(a, b_) if isinstance(a, np.ndarray) else a == b_, name


def test_column_panel_views_keep_their_own_row_pointer_width(sla):
    """a matrix with more than 2^31 - 1 entries has 64-bit row pointers while each of its column-panel
    views, holding a fraction of the entries, uses 32-bit ones; the panel passes must run the instantiation of the VIEW.
    The option force_rp64=2 forces the 64-bit width on the parent only, which reproduces that mix at test size."""
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.random_spd(3000, 3, 11)
    n = dims[0]
    Ao = orc.Csr(n, n, rp, ci, va)
    x = np.random.default_rng(6).standard_normal(n)
    b = orc.spmv(Ao, np.random.default_rng(7).standard_normal(n))
    ctx = sla.Context(0).set_options(panel_cols=500, force_rp64=2)
    A = sla.fromCSR(dims, rp, ci, va, ctx)
    assert "colpanels" in A.kernel_info() and "rowptr=i64" in A.kernel_info(), A.kernel_info()
    assert np.array_equal(sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV(), orc.spmv(Ao, x))
    xs, info = sla.linSolve0(sla.BICGSTAB_, A, sla.fromVector(b, ctx), sla.fromVector(np.zeros(n), ctx), return_info=True)
    assert info["converged"] and np.linalg.norm(orc.spmv(Ao, xs.toDenseListSV()) - b) <= info["tol"] * (1 + 1e-9)


def test_step_graph_replay_is_bit_identical_to_stream_launches(sla):
    """sla_solver_step replays pairs of steps as a captured HIP graph at launch-bound sizes: same kernels, same arguments, same
    order => the iterates must equal the stream-launched ones bit for bit, from even and odd starting parity, for BiCGSTAB and
    CGS, with leftovers, on a clone (which captures its own graph) and after the graph was built."""
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.poisson2d(60, 50)
    n = dims[0]
    b = np.add.reduceat(va, rp[:-1])
    out = {}
    for mode in ("1", "0"):
        ctx = sla.Context(0).set_option("step_graph", mode)
        A = sla.fromCSR(dims, rp, ci, va, ctx)
        res = []
        for init, xf in ((sla.bicgsInit, "_xBicgstab"), (sla.cgsInit, "_x")):
            s = init(A, sla.fromVector(b, ctx), sla.fromVector(np.zeros(n), ctx))
            s.step(7)                         # even start: 3 replays + 1 plain step
            res.append(getattr(s, xf).toDenseListSV())
            s.step(6)                         # odd start: 1 plain step, 2 replays, 1 plain
            res.append(getattr(s, xf).toDenseListSV())
            t = s.clone().step(5)
            res.append(getattr(t, xf).toDenseListSV())
            s.step(1)
            s.step(4)
            res.append(getattr(s, xf).toDenseListSV())
        out[mode] = res
        ctx.close()
    for a, b_ in zip(out["1"], out["0"]):
        assert np.array_equal(a, b_)
    assert np.isfinite(out["1"][3]).all() and np.linalg.norm(out["1"][3] - 1.0) < np.linalg.norm(np.ones(n))     # and it is heading for x* = 1


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_canonical_arrays_written_on_the_device_are_the_callers(sla, mode):
    """Round 4: a value-indexed matrix (constant-coefficient stencil) gets the part of its canonical col / val arrays that has not crossed
    PCIe by the time the analysis is through written on the device from its 1-byte codes (sla_lower.cpp: vd_expand_kernel).  The
    export, the plain-CSR kernels and the transposed product must see the caller's arrays bit for bit whichever way they got there:
    canon_device = 0: all uploaded; 1 (default; 2 is accepted as a synonym): the upload waits for the pair analysis's decision, so nothing
    of col / val crosses PCIe and all of it is written on the device."""
    from sla_amd import workloads as wl
    for dims, (rp, ci, va) in (wl.laplace3d(40, 36, 33), wl.poisson2d(300, 211)):
        ctx = sla.Context(0).set_options(canon_device=mode)
        A = sla.fromCSR(dims, rp, ci, va, ctx)
        assert "wdia" in A.kernel_info() or "vdict" in A.kernel_info(), A.kernel_info()
        info = A.lower_info()
        key = "canonical entries over PCIe (fraction)"
        if mode >= 1:
            assert info.get(key) == 0.0, info
        if mode == 0:
            assert key not in info, info
        rp2, ci2, va2 = A.csr()
        assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and np.array_equal(va2.view(np.uint64), va.view(np.uint64))
        x = np.random.default_rng(5).standard_normal(dims[1])
        ctx.set_options(wdia=0, vdict=0, diag=0)                          # the plain-CSR kernel on the generated arrays
        xd, y = sla.DeviceVector(ctx, dims[1], x), sla.DeviceVector(ctx, dims[0])
        sla._lib.check(sla._lib.lib().sla_spmv(A.h, xd.h, y.h))
        assert "stream" in A.kernel_info(), A.kernel_info()
        assert np.array_equal(y.to_host(), orc.spmv(orc.Csr(dims[0], dims[1], rp, ci, va), x))


@pytest.mark.parametrize("lanes,xfer", [(1, 1), (3, 1), (8, 1), (4, 0)])
def test_large_copies_through_the_pinned_lanes_are_exact(sla, lanes, xfer):
    """Copies of >= 24 MiB between the caller's pageable arrays and the device are staged by the library itself (csrc/sla_xfer.cpp:
    8 MiB chunks dealt round-robin to `xfer_lanes` host threads with two pinned slots each).  Sizes that are no multiple of a chunk,
    of the lane count or of anything else must arrive bit for bit, up and down, for vectors and for a matrix's arrays (export),
    whatever the lane count -- and with the staging switched off."""
    ctx = sla.Context(0).set_options(xfer=xfer, xfer_lanes=lanes)
    rng = np.random.default_rng(lanes * 10 + xfer)
    for n in (3 * (1 << 20) + 1, 9 * (1 << 20) + 12345):           # 24 MiB + 8 B; 72.09 MiB
        x = rng.standard_normal(n)
        v = sla.DeviceVector(ctx, n, x)
        assert np.array_equal(v.to_host().view(np.uint64), x.view(np.uint64)), (n, lanes, xfer)
        del v
    # a matrix that is NOT value-indexed (its canonical arrays do cross PCIe): 4 M rows x 3 random entries, values at random
    n = 4_000_000
    cols = np.sort(rng.integers(0, n, (n, 3)), axis=1)
    cols[:, 1] += (cols[:, 1] == cols[:, 0])
    cols[:, 2] = np.maximum(cols[:, 2], cols[:, 1] + 1)
    keep = cols[:, 2] < n
    cols[~keep] = np.array([0, 1, 2])
    rp = np.arange(0, 3 * n + 1, 3, dtype=np.int64)
    ci = cols.astype(np.int64).ravel()
    va = rng.standard_normal(3 * n)
    A = sla.fromCSR((n, n), rp, ci, va, ctx)
    rp2, ci2, va2 = A.csr()
    assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and np.array_equal(va2.view(np.uint64), va.view(np.uint64))


@pytest.mark.parametrize("rp64", [0, 1])
def test_device_transpose_equals_the_host_transpose(sla, rp64):
    """transposeSM of a lowered matrix (the first (<#) / cgneStep builds it): as a device sort by (column, row) (round 4, option
    transpose_device) and by the host path (export + counting sort).  (<#) through either must be the oracle's fold over the
    transposed matrix (to the row-length bound; bit for bit BETWEEN the two paths) -- rectangular shapes, empty rows and columns, a stencil, 64-bit row pointers."""
    from sla_amd import workloads as wl
    rng = np.random.default_rng(31 + rp64)

    def ragged(m, n, maxlen):
        lens = rng.integers(0, maxlen + 1, m)
        lens[rng.random(m) < 0.2] = 0
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ci = np.concatenate([np.sort(rng.choice(n - n // 5, size=int(k), replace=False)) for k in lens] + [np.zeros(0, np.int64)]).astype(np.int64)   # (the last fifth of the columns: empty)
        return (m, n), (rp, ci, rng.standard_normal(len(ci)))

    for dims, (rp, ci, va) in (ragged(700, 2900, 40), ragged(5000, 300, 12), wl.laplace3d(17, 13, 11), wl.poisson2d(90, 41)):
        Ao = orc.Csr(dims[0], dims[1], rp, ci, va)
        x = rng.standard_normal(dims[0])
        want = orc.spmv(orc.transpose(Ao), x)
        cnt = np.bincount(ci, minlength=dims[1])
        bound = (cnt + 8) * np.finfo(float).eps * orc.spmv(orc.transpose(orc.Csr(dims[0], dims[1], rp, ci, np.abs(va))), np.abs(x)) + 1e-300
        got = {}
        for dev in (2, 0):
            ctx = sla.Context(0).set_options(transpose_device=dev, force_rp64=rp64)
            A = sla.fromCSR(dims, rp, ci, va, ctx)
            y = sla.DeviceVector(ctx, dims[1])
            xd = sla.DeviceVector(ctx, dims[0], x)
            sla._lib.check(sla._lib.lib().sla_spmv_t(A.h, xd.h, y.h))
            got[dev] = y.to_host()
            # (the transposes of the ragged cases have rows of hundreds of entries: the long-row kernels are not a strict left fold)
            assert np.all(np.abs(got[dev] - want) <= bound), (dims, dev, rp64, float(np.abs(got[dev] - want).max()))
            assert np.all(got[dev][cnt == 0] == 0.0)
            del A, y, xd
            ctx.close()
        # the same arrays from both paths: the same kernel gives the same bits
        assert np.array_equal(got[2].view(np.uint64), got[0].view(np.uint64)), (dims, rp64)


def test_process_exit_with_background_work_in_flight():
    """A process that exits while the library still has background threads running (the pinned copy lanes of a new context being
    built; large host buffers being released) must exit cleanly: the context is never destroyed here, as in a caller that leaks it.
    (Before the exit handler of sla_xfer.cpp: a segfault inside the HIP runtime's own teardown, exit code 139.)
    Runs by default since round 5 (a dozen short-lived GPU processes, well under a minute); tools/exit_race.sh is the same check by hand."""
    import subprocess
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sparse-linear-algebra_amd")
    cases = {
        "context alive at exit": "import sla_amd as sla\nc = sla.Context(0)\n",
        "context closed right away": "import sla_amd as sla\nc = sla.Context(0)\nc.close()\n",
        "matrix from triples, then exit": (
            "import sla_amd as sla, numpy as np\n"
            "c = sla.Context(0)\n"
            "n = 2000000\n"
            "r = np.repeat(np.arange(n, dtype=np.int64), 3)\n"
            "k = (r + np.tile(np.array([0, 7, 1000003], dtype=np.int64), n)) % n\n"
            "M = sla.fromCOO((n, n), r, k, np.ones(3 * n), ctx=c)\n"),
    }
    env = dict(os.environ, PYTHONPATH=pkg + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for name, src in cases.items():
        for rep in range(4):
            out = subprocess.run([sys.executable, "-c", src], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
            assert out.returncode == 0, (name, rep, out.returncode, out.stderr[-400:])
