"""The value-indexed SpMV forms (spmv_wdia_kernel: wave-sliced (offset, value) records in SGPRs, two rows per
lane; spmv_wdia_lds_kernel: uniform records + x windows staged in LDS for stencils of <= 8 pairs, forced at these sizes
with the context option wd_lds=2; spmv_vdict_kernel: one byte per entry) against the oracle's left fold, bit for bit, and against each other
and the general kernels -- including the shapes that stress them: odd row counts, ragged patterns where only
one row of a lane pair holds an entry, more than 8 records per slice, several values on one diagonal, rows at
the matrix edge, -0.0 / Inf operands, and every fused epilogue through the solver steps."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sla():
    import sla_amd
    return sla_amd


def _stencil(n, offsets, value_of, keep=None):
    """CSR of a banded matrix: entry (i, i + off) = value_of(i, t) for each offset t, dropped where keep(i, t) is False."""
    from sla_amd import workloads as wl
    offsets = sorted(offsets)

    def valid(rows, t):
        c = rows + offsets[t]
        ok = (c >= 0) & (c < n)
        if keep is not None:
            ok &= keep(rows, t)
        return ok

    return (n, n), wl._stencil_rows(0, n, offsets, valid, lambda rows, t: value_of(rows, offsets[t]))


def _cases():
    from sla_amd import workloads as wl
    rng = np.random.default_rng(5)
    drop = rng.random((4099, 16)) < 0.3
    return {
        "laplace3d 14x11x13": wl.laplace3d(14, 11, 13),
        "poisson2d 37x29 (odd n)": wl.poisson2d(37, 29),
        "tridiag n=1": _stencil(1, [-1, 0, 1], lambda r, o: np.full(len(r), 2.0 if o == 0 else -1.0)),
        "tridiag n=129": _stencil(129, [-1, 0, 1], lambda r, o: np.full(len(r), 2.0 if o == 0 else -1.0)),
        # 13 diagonals: more than one 8-record chunk per slice
        "13 diagonals n=3001": _stencil(3001, [-700, -64, -9, -3, -2, -1, 0, 1, 2, 5, 63, 128, 900],
                                         lambda r, o: np.full(len(r), 20.0 if o == 0 else -1.0 - 0.125 * (o % 7))),
        # ragged: 30 % of the entries missing at random, so lane pairs often hold an entry in one row only
        "ragged 11 diagonals n=4099": _stencil(4099, [-300, -17, -4, -2, -1, 0, 1, 3, 16, 250, 1025],
                                                lambda r, o: np.full(len(r), 9.0 if o == 0 else 0.5 + (o % 3)),
                                                keep=lambda r, t: ~drop[r, t] | (t == 5)),
        # <= 8 pairs (the LDS-window kernel's domain), three windows, 30 % of the entries missing, odd row count
        "ragged 7 diagonals n=4099": _stencil(4099, [-1300, -40, -1, 0, 1, 40, 1300],
                                               lambda r, o: np.full(len(r), 9.0 if o == 0 else 0.5 + (o % 3)),
                                               keep=lambda r, t: ~drop[r, t] | (t == 3)),
        # five far-apart diagonals: five windows, 1285 staged pairs -- the LDS-window kernel's 6-load instantiation (> 1024 pairs)
        "5 far diagonals n=7001": _stencil(7001, [-3000, -1500, 0, 1500, 3000], lambda r, o: np.full(len(r), 7.0 if o == 0 else -1.0 - 0.5 * (o > 0)),
                                          keep=lambda r, t: (r % 11 != 3) | (t == 2)),
        # three different values along each diagonal (row mod 3): several records share an offset
        "3 values per diagonal n=2500": _stencil(2500, [-50, -1, 0, 1, 50],
                                                  lambda r, o: (8.0 if o == 0 else -1.0) * (1.0 + 0.25 * (r % 3))),
        # signed zeros and a row of explicit zeros must survive (they are values like any other)
        "explicit +-0.0 n=777": _stencil(777, [-2, 0, 2], lambda r, o: np.where(r % 5 == 0, -0.0 if o else 0.0, 3.0 + o)),
    }


def _oracle_csr(dims, csr):
    rp, ci, va = csr
    return orc.Csr(dims[0], dims[1], rp, ci, va)


@pytest.mark.parametrize("name", list(_cases()))
def test_value_indexed_forms_fold_like_the_reference(sla, name):
    dims, csr = _cases()[name]
    n = dims[0]
    Ao = _oracle_csr(dims, csr)
    rng = np.random.default_rng(11)
    x = rng.standard_normal(n)
    x[rng.integers(0, n, max(1, n // 50))] = -0.0
    want = orc.spmv(Ao, x)
    want_t = orc.spmv(orc.transpose(Ao), x)
    got = {}
    lds_cases = ("laplace3d 14x11x13", "poisson2d 37x29 (odd n)", "tridiag n=1", "tridiag n=129", "ragged 7 diagonals n=4099",
                 "5 far diagonals n=7001", "explicit +-0.0 n=777")
    for form, opts in (("wdia", {"wd_lds": 0}), ("wdia+ldswin", {"wd_lds": 2}), ("vdict", {"wdia": 0}),
                       ("diag", {"wdia": 0, "vdict": 0}),
                       ("stream", {"wdia": 0, "vdict": 0, "diag": 0, "xwin": 0})):
        ctx = sla.Context(0).set_options(**opts)      # (typed per-context options: sla_ctx_set_option)
        A = sla.fromCSR(dims, *csr, ctx)
        if form == "wdia+ldswin":
            # the LDS-window kernel takes stencils of <= 8 (offset, value) pairs; the others stay on the gather kernel
            assert ("ldswin" in A.kernel_info().split()[0]) == (name in lds_cases), (name, A.kernel_info())
            if name == "5 far diagonals n=7001":     # (the instantiation with six staging loads per lane)
                assert "windows=5" in A.kernel_info() and int(A.kernel_info().split("win_pairs=")[1].split()[0]) > 1024, A.kernel_info()
            form = "wdia"
        if form in ("wdia", "vdict"):
            assert form in A.kernel_info().split()[0], (form, A.kernel_info())
        y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
        yt = sla.vecMat(sla.fromVector(x, ctx), A).toDenseListSV()
        got[form] = y
        if form in ("wdia", "vdict") or len(csr[1]) <= 8 * n:
            # every row is the ascending left fold with separately rounded a*x and +: identical bits, signed zeros included
            assert np.array_equal(y.view(np.uint64), want.view(np.uint64)), (name, form, np.abs(y - want).max())
            assert np.array_equal(yt.view(np.uint64), want_t.view(np.uint64)), (name, form, "transpose")
        else:   # the general kernels sum rows of more than 8 entries (on average) by wavefront segments: rounding-level differences
            assert np.allclose(y, want, rtol=1e-13, atol=1e-13) and np.allclose(yt, want_t, rtol=1e-13, atol=1e-13)
        del A
        ctx.close()


@pytest.mark.parametrize("name", ["laplace3d 14x11x13", "ragged 11 diagonals n=4099", "ragged 7 diagonals n=4099",
                                  "5 far diagonals n=7001", "3 values per diagonal n=2500"])
def test_value_indexed_epilogues_through_the_solvers(sla, name):
    """K1/K3 (dot, dot2), the true-residual sweep, CGS's and CGNE's fused updates, r0 = b - A x0: same iterates as
    the general kernels (the per-row results are bit-identical; only partial-sum grouping differs)."""
    dims, csr = _cases()[name]
    n = dims[0]
    Ao = _oracle_csr(dims, csr)
    b = orc.spmv(Ao, np.linspace(-1.0, 2.0, n))
    out = {}
    for form, opts in (("wdia", {"wd_lds": 0}), ("ldswin", {"wd_lds": 2}), ("vdict", {"wdia": 0}),
                       ("stream", {"wdia": 0, "vdict": 0, "diag": 0})):
        ctx = sla.Context(0).set_options(**opts)
        A = sla.fromCSR(dims, *csr, ctx)
        for meth in (sla.BICGSTAB_, sla.CGS_, sla.CGNE_):
            x, info = sla.linSolve0(meth, A, sla.fromVector(b, ctx), sla.fromVector(np.full(n, 0.25), ctx), return_info=True)
            out[(form, int(meth))] = (x.toDenseListSV(), info["iters"], info["resnorm"])
        del A
        ctx.close()
    for meth, ometh in ((sla.BICGSTAB_, orc.BICGSTAB_), (sla.CGS_, orc.CGS_), (sla.CGNE_, orc.CGNE_)):
        rc, xo, it_o, res_o, r0_o = orc.linsolve0(ometh, Ao, b, np.full(n, 0.25))
        ref = out[("stream", int(meth))]
        for form in ("wdia", "ldswin", "vdict"):
            x, it, rn = out[(form, int(meth))]
            assert abs(it - ref[1]) <= 1 and abs(it - it_o) <= 2, (name, form, meth, it, ref[1], it_o)
            if it_o < 200:
                assert np.linalg.norm(orc.spmv(Ao, x) - b) <= max(1e-6, 1e-4 * r0_o) * (1 + 1e-9)
            if it == ref[1]:       # same number of steps: the iterates agree up to the partial-sum grouping, amplified by the solver
                scale = np.abs(ref[0]).max() + 1e-300
                assert np.abs(x - ref[0]).max() <= 1e-5 * scale, (name, form, meth)


@pytest.mark.parametrize("wd_lds", [0, 2])
def test_value_indexed_propagates_inf_and_nan_only_where_the_reference_does(sla, wd_lds):
    """Lanes masked off in a slice must not touch x at all: an Inf next to a missing neighbour stays out of that row."""
    n = 640
    dims, csr = _stencil(n, [-1, 0, 1], lambda r, o: np.full(len(r), 2.0 if o == 0 else -1.0),
                         keep=lambda r, t: ~((r % 64 == 10) & (t == 2)))       # rows 10, 74, ... have no (i, i+1) entry
    Ao = _oracle_csr(dims, csr)
    x = np.ones(n)
    x[11] = np.inf           # row 10 does not reference x[11]; rows 11 and 12 do
    x[300] = np.nan
    A = sla.fromCSR(dims, *csr, sla.Context(0).set_option("wd_lds", wd_lds))
    assert "wdia" in A.kernel_info() and ("ldswin" in A.kernel_info()) == (wd_lds == 2)
    y = sla.matVec(A, sla.fromVector(x, A.ctx)).toDenseListSV()
    want = orc.spmv(Ao, x)
    assert np.isfinite(y[10]) and np.isinf(y[11]) and np.isinf(y[12])
    assert np.array_equal(np.isnan(y), np.isnan(want)) and np.array_equal(y[~np.isnan(y)], want[~np.isnan(want)])


@pytest.mark.parametrize("wd_lds", [0, 2])
def test_value_indexed_randomised_patterns(sla, wd_lds):
    """(wd_lds = 2: the patterns of <= 8 pairs go through the LDS-window kernel)
    60 seeded random banded matrices (square and rectangular, 1..900 rows, up to 14 diagonals anywhere in the matrix,
    1-4 distinct values per diagonal, random holes, some with an empty leading / trailing block of rows): whatever form
    the lowering picks, (#>) and (<#) are the oracle's left fold bit for bit and the forms agree with each other."""
    ctx = sla.Context(0).set_option("wd_lds", wd_lds)
    rng = np.random.default_rng(2024)
    picked = {"wdia": 0, "vdict": 0, "other": 0, "ldswin": 0}
    for case in range(60):
        m = int(rng.integers(1, 900))
        n = m if case % 3 else int(rng.integers(1, 900))
        nd = int(rng.integers(1, 15))
        offs = np.unique(rng.integers(-min(m, 400), min(n, 400) + 1, nd))
        pal = {int(o): rng.choice([-2.0, -1.0, -0.5, 0.25, 1.0, 3.0, 6.0, -0.0], size=int(rng.integers(1, 5))) for o in offs}
        hole = rng.random() * 0.5
        lead, trail = (int(rng.integers(0, m // 3 + 1)), int(rng.integers(0, m // 3 + 1))) if case % 5 == 0 else (0, 0)
        rows, cols, vals = [], [], []
        for i in range(lead, m - trail):
            for o in offs:
                j = i + int(o)
                if 0 <= j < n and rng.random() >= hole:
                    rows.append(i), cols.append(j), vals.append(float(pal[int(o)][(i * 7 + j) % len(pal[int(o)])]))
        r, c, v = np.array(rows, np.int64), np.array(cols, np.int64), np.array(vals)
        rc, Ao = orc.coo_to_csr(m, n, r, c, v)
        assert rc == orc.OK
        x = rng.standard_normal(n)
        xt = rng.standard_normal(m)
        want, want_t = orc.spmv(Ao, x), orc.spmv(orc.transpose(Ao), xt)
        A = sla.fromCSR((m, n), Ao.rowptr, Ao.colidx, Ao.val, ctx)
        algo = A.kernel_info().split()[0]
        picked["wdia" if "wdia" in algo else "vdict" if "vdict" in algo else "other"] += 1
        picked["ldswin"] += "ldswin" in algo
        y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
        yt = sla.vecMat(sla.fromVector(xt, ctx), A).toDenseListSV()
        exact = "wdia" in algo or "vdict" in algo or len(v) <= 8 * m
        if exact:
            assert np.array_equal(y.view(np.uint64), want.view(np.uint64)), (case, algo, m, n, offs)
        else:
            assert np.allclose(y, want, rtol=1e-13, atol=1e-13), (case, algo)
        assert np.allclose(yt, want_t, rtol=1e-13, atol=1e-13), (case, algo, "transpose")
    assert picked["wdia"] >= 20, picked      # the generator is meant to exercise the value-indexed forms
    assert (picked["ldswin"] >= 5) == (wd_lds == 2) and (picked["ldswin"] == 0) == (wd_lds == 0), picked


def _vv_cases():
    from sla_amd import workloads as wl
    rng = np.random.default_rng(77)
    drop = rng.random((5003, 16)) < 0.25
    noise = rng.standard_normal((5003, 16))
    return {
        "banded_nonsym 4001 (config 5 structure)": wl.banded_nonsym(4001),
        "11 diagonals, random values, n=3001": _stencil(3001, [-700, -64, -9, -2, -1, 0, 1, 2, 63, 128, 900],
                                                        lambda r, o: (12.0 if o == 0 else -1.0) * (1.0 + 0.1 * noise[r, (o % 13)])),
        "ragged 9 diagonals, random values, odd n=5003": _stencil(5003, [-300, -17, -2, -1, 0, 1, 3, 250, 1025],
                                                                   lambda r, o: 7.0 * (o == 0) + noise[r, o % 11],
                                                                   keep=lambda r, t: ~drop[r, t] | (t == 4)),
    }


@pytest.mark.parametrize("name", list(_vv_cases()))
def test_variable_coefficient_wave_sliced_form(sla, name):
    """Banded / stencil structure with arbitrary values: the wave-sliced kernel with per-row value blocks ("wdia-vv") --
    the oracle's fold bit for bit, and the solvers agree with the general kernels."""
    dims, csr = _vv_cases()[name]
    n = dims[0]
    Ao = _oracle_csr(dims, csr)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(n)
    want, want_t = orc.spmv(Ao, x), orc.spmv(orc.transpose(Ao), x)
    b = orc.spmv(Ao, np.linspace(-1.0, 1.0, n))
    sols = {}
    for form, opts in (("wdia-vv", {}), ("general", {"wdia_vv": 0})):
        ctx = sla.Context(0).set_options(**opts)
        A = sla.fromCSR(dims, *csr, ctx)
        algo = A.kernel_info().split()[0]
        assert ("wdia-vv" in algo) == (form == "wdia-vv"), (form, algo)
        y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
        yt = sla.vecMat(sla.fromVector(x, ctx), A).toDenseListSV()
        if form == "wdia-vv" or len(csr[1]) <= 8 * n:
            assert np.array_equal(y.view(np.uint64), want.view(np.uint64)), (name, form)
            assert np.array_equal(yt.view(np.uint64), want_t.view(np.uint64)), (name, form, "transpose")
        else:
            assert np.allclose(y, want, rtol=1e-13, atol=1e-13)
        for meth in (sla.BICGSTAB_, sla.CGS_, sla.CGNE_):
            xs, info = sla.linSolve0(meth, A, sla.fromVector(b, ctx), sla.fromVector(np.zeros(n), ctx), return_info=True)
            sols[(form, int(meth))] = (xs.toDenseListSV(), info["iters"])
        del A
        ctx.close()
    for meth in (sla.BICGSTAB_, sla.CGS_, sla.CGNE_):
        (xa, ia), (xb, ib) = sols[("wdia-vv", int(meth))], sols[("general", int(meth))]
        assert abs(ia - ib) <= 1, (name, meth, ia, ib)
        if ia == ib:
            assert np.abs(xa - xb).max() <= 1e-5 * (np.abs(xb).max() + 1e-300)


@pytest.mark.parametrize("values", ["constant", "arbitrary"])
def test_rows_without_entries_still_run_the_epilogues(sla, values):
    """Whole 128-row slices without a stored entry (and single empty rows): r0 = b - A x0, the residual, CGS's and CGNE's
    fused updates are defined there too (y_i = 0).  One step of every solver against the oracle."""
    n = 900
    rng = np.random.default_rng(8)
    empty = np.zeros(n, bool)
    empty[:300] = True            # two whole slices and a bit
    empty[640:700] = True
    empty[rng.integers(300, 640, 20)] = True
    dims, csr = _stencil(n, [-40, -1, 0, 1, 3, 64],
                         (lambda r, o: np.full(len(r), 9.0 if o == 0 else -1.0)) if values == "constant"
                         else (lambda r, o: (9.0 if o == 0 else 0.0) + np.cos(r * 0.37 + o)),
                         keep=lambda r, t: ~empty[r])
    Ao = _oracle_csr(dims, csr)
    A = sla.fromCSR(dims, *csr)
    assert ("wdia-vv" if values == "arbitrary" else "wdia") in A.kernel_info().split()[0]
    b, x0 = rng.standard_normal(n), rng.standard_normal(n)
    r0 = b - orc.spmv(Ao, x0)
    for init, fld_r, fld_x, ocls in ((sla.bicgsInit, "_rBicgstab", "_xBicgstab", orc.BicgstabState), (sla.cgsInit, "_r", "_x", orc.CgsState),
                                     (sla.cgneInit, "_rCgne", "_xCgne", orc.CgneState)):
        st = init(A, sla.fromVector(b), sla.fromVector(x0))
        assert np.array_equal(getattr(st, fld_r).toDenseListSV(), r0)              # EPI_SUB on empty rows: r_i = b_i
        os_ = ocls(Ao, b, x0)
        st.step(1)
        os_.step(1) if ocls is orc.CgneState else os_.step(r0, 1)
        for fld, want in ((fld_x, os_.x), (fld_r, os_.r)):
            got = getattr(st, fld).toDenseListSV()
            assert np.allclose(got, want, rtol=1e-10, atol=1e-10 * (np.abs(want).max() + 1)), (values, fld)
    x, info = sla.linSolve0(sla.BICGSTAB_, A, sla.fromVector(b), sla.fromVector(x0), return_info=True, max_iters=3)
    res = orc.spmv(Ao, x.toDenseListSV()) - b                                       # EPI_RES counts the empty rows' b_i
    assert abs(info["resnorm"] - np.linalg.norm(res)) <= 1e-9 * np.linalg.norm(res)


def _march_cases():
    from sla_amd import workloads as wl
    rng = np.random.default_rng(9)
    drop = rng.random((9000, 8)) < 0.3
    return {
        # D = 36 * 30 = 1080: tiles of 512, 512, 56 rows; 9 planes
        "laplace3d 36x30x9": wl.laplace3d(36, 30, 9),
        # D = 40 * 64 = 2560 = 5 whole tiles, 7 planes
        "laplace3d 40x64x7": wl.laplace3d(40, 64, 7),
        # 2-D 5-point stencil on a 1100-wide grid: D = 1100, in-plane window +-1
        "poisson2d 1100x13": wl.poisson2d(1100, 13),
        # ragged, odd row count, last plane partial (4099 = 3 * 1300 + 199), 30 % of the entries missing
        "ragged 7 diagonals n=4099": _stencil(4099, [-1300, -40, -1, 0, 1, 40, 1300],
                                               lambda r, o: np.full(len(r), 9.0 if o == 0 else 0.5 + (o % 3)),
                                               keep=lambda r, t: ~drop[r, t] | (t == 3)),
        # no (i, i + 2) entry in a third of the rows, in-plane offsets up to +-250 (window of 1014 elements), n = 7 D + 1
        "wide in-plane window n=8401": _stencil(8401, [-1200, -250, -2, 0, 2, 250, 1200],
                                                 lambda r, o: np.full(len(r), 11.0 if o == 0 else -1.0 - 0.25 * (abs(o) % 5)),
                                                 keep=lambda r, t: (r % 3 != 1) | (t != 4)),
    }


@pytest.mark.parametrize("name", list(_march_cases()))
@pytest.mark.parametrize("grid", [0, 8])
def test_plane_march_form_folds_like_the_reference(sla, name, grid):
    """spmv_wdia_march_kernel (a workgroup walks a run of planes of a 3-D stencil, one staged window per step): every row the
    reference's left fold bit for bit, (<#) through the transposed matrix's own lowering, and the fused epilogues through the
    solvers against the LDS-window form.  grid = 8: eight workgroups, so each walks several (tile, run) tasks."""
    dims, csr = _march_cases()[name]
    n = dims[0]
    Ao = _oracle_csr(dims, csr)
    rng = np.random.default_rng(13)
    x = rng.standard_normal(n)
    x[rng.integers(0, n, max(1, n // 50))] = -0.0
    want = orc.spmv(Ao, x)
    want_t = orc.spmv(orc.transpose(Ao), x)
    b = orc.spmv(Ao, np.linspace(-1.0, 2.0, n))
    out = {}
    for march in (2, 0):
        ctx = sla.Context(0).set_options(wd_lds=2, wd_march=march)
        if grid:
            ctx.set_option("spmv_grid", grid)
        A = sla.fromCSR(dims, *csr, ctx)
        assert ("wdia+march" in A.kernel_info().split()[0]) == (march == 2), A.kernel_info()
        y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
        yt = sla.vecMat(sla.fromVector(x, ctx), A).toDenseListSV()
        assert np.array_equal(y.view(np.uint64), want.view(np.uint64)), (name, march, np.abs(y - want).max())
        assert np.array_equal(yt.view(np.uint64), want_t.view(np.uint64)), (name, march, "transpose")
        for meth in (sla.BICGSTAB_, sla.CGS_, sla.CGNE_):
            xs, info = sla.linSolve0(meth, A, sla.fromVector(b, ctx), sla.fromVector(np.full(n, 0.25), ctx), return_info=True)
            out[(march, int(meth))] = (xs.toDenseListSV(), info["iters"], info["resnorm"])
        del A
        ctx.close()
    for meth in (sla.BICGSTAB_, sla.CGS_, sla.CGNE_):
        (xm, itm, _), (xl, itl, _) = out[(2, int(meth))], out[(0, int(meth))]
        assert abs(itm - itl) <= 1, (name, meth, itm, itl)
        if itm == itl:      # same rows, differently grouped partial sums, amplified by the solver
            assert np.abs(xm - xl).max() <= 2e-4 * (np.abs(xl).max() + 1e-300), (name, meth)
        if itm < 200:
            assert np.linalg.norm(orc.spmv(Ao, xm) - b) <= 1e-4 * np.linalg.norm(orc.spmv(Ao, np.full(n, 0.25)) - b) * (1 + 1e-9) + 1e-6


def test_plane_march_form_step_for_step_against_the_oracle(sla):
    """Two BiCGSTAB steps (fused sweep: K1 dot, K3 with four sums and the window operand) and two CGS steps on the march form
    against the oracle's steps."""
    from sla_amd import workloads as wl
    dims, csr = wl.laplace3d(36, 30, 9)
    n = dims[0]
    Ao = _oracle_csr(dims, csr)
    b = orc.spmv(Ao, np.linspace(-1.0, 2.0, n))
    x0 = np.full(n, 0.25)
    ctx = sla.Context(0).set_options(wd_lds=2, wd_march=2)
    A = sla.fromCSR(dims, *csr, ctx)
    assert "wdia+march" in A.kernel_info()
    r0hat = b - orc.spmv(Ao, x0)
    for fuse in (1, 0):     # the fused K4 + K5 sweep (K3 with four sums) and the reference's split
        ctx.set_option("bicg_fuse45", fuse)
        so, sd = orc.BicgstabState(Ao, b, x0), sla.bicgsInit(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
        so.step(r0hat, 2)
        sd.step(2)
        for got, want in ((sd._xBicgstab, so.x), (sd._rBicgstab, so.r), (sd._pBicgstab, so.p)):
            assert np.linalg.norm(got.toDenseListSV() - want) <= 1e-11 * np.linalg.norm(want), fuse
    co, cd = orc.CgsState(Ao, b, x0), sla.cgsInit(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
    co.step(r0hat, 2)
    cd.step(2)
    for got, want in ((cd._x, co.x), (cd._r, co.r), (cd._p, co.p)):
        assert np.linalg.norm(got.toDenseListSV() - want) <= 1e-11 * np.linalg.norm(want)
    del A
    ctx.close()


@pytest.mark.parametrize("name", list(_march_cases()))
@pytest.mark.parametrize("grid", [0, 8])
def test_plane_march_k2_folded_into_k3_same_bits(sla, name, grid):
    """BiCGSTAB on the plane-march form, one rank: K2 (alpha ; s = r - alpha Ap, Sparse.hs:975-976) is folded into K3 -- s is built while
    the x windows are staged and never stored, the fused K4 + K5 sweep rebuilds it from r and Ap (option bicg_fuse23, default 1).  Same
    alpha, same multiply-add, same fold: x, r, p after 1, 2 and 7 steps are bit-identical to the four-launch flow, the kernel table shows
    three launches per step, and linSolve0 returns the same iterate."""
    from sla_amd import _lib
    dims, csr = _march_cases()[name]
    n = dims[0]
    Ao = _oracle_csr(dims, csr)
    b = orc.spmv(Ao, np.linspace(-1.0, 2.0, n))
    x0 = np.full(n, 0.25)
    states, sols, launches = {}, {}, {}
    for f23 in (1, 0):
        ctx = sla.Context(0).set_options(wd_lds=2, wd_march=2, bicg_fuse23=f23, onchip=0)   # (onchip = 0: these are tests of the LAUNCH flow's kernels)
        if grid:
            ctx.set_option("spmv_grid", grid)
        A = sla.fromCSR(dims, *csr, ctx)
        assert "wdia+march" in A.kernel_info().split()[0], A.kernel_info()
        st = sla.bicgsInit(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
        snaps = []
        ctx.prof_start(_lib.KERNEL_ALL, 64)
        for k in (1, 1, 5):
            st.step(k)
            snaps.append([v.toDenseListSV().copy() for v in (st._xBicgstab, st._rBicgstab, st._pBicgstab)])
        ctx.prof_stop()
        launches[f23] = (ctx.prof_query(_lib.KERNEL_BICG_K2)[0], ctx.prof_query(_lib.KERNEL_SPMV_DOT2)[0], ctx.prof_query(_lib.KERNEL_BICG_K45)[0])
        states[f23] = snaps
        xs, info = sla.linSolve0(sla.BICGSTAB_, A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx), return_info=True)
        sols[f23] = (xs.toDenseListSV(), info["iters"], info["resnorm"])
        del A, st
        ctx.close()
    assert launches[1] == (0, 7, 7) and launches[0] == (7, 7, 7), launches
    for a, c in zip(states[1], states[0]):
        for u, v in zip(a, c):
            assert np.array_equal(u.view(np.uint64), v.view(np.uint64)), (name, grid, np.abs(u - v).max())
    assert sols[1][1] == sols[0][1] and sols[1][2] == sols[0][2], (sols[1][1:], sols[0][1:])
    assert np.array_equal(sols[1][0].view(np.uint64), sols[0][0].view(np.uint64))
    # and against the oracle's two steps (the reference's expressions: mul then subtract -- tolerance as in the test above)
    so = orc.BicgstabState(Ao, b, x0)
    so.step(b - orc.spmv(Ao, x0), 2)
    for got, want in zip(states[1][1], (so.x, so.r, so.p)):
        assert np.linalg.norm(got - want) <= 1e-11 * np.linalg.norm(want)


@pytest.mark.parametrize("name", list(_march_cases()))
@pytest.mark.parametrize("grid", [0, 8])
def test_plane_march_cgs_c2_folded_into_c3_same_bits(sla, name, grid):
    """cgsStep (Sparse.hs:928-939) on the plane-march form, one rank: C2 is folded away -- C3 builds u + q (q = u - alpha A p) in its staged
    windows, one sweep after it does C2's x update and C4's u, p (cgs_c24_kernel); q and u + q are never stored.  x, r, p, u after 1, 2
    and 7 steps bit-identical to the four-launch flow (bicg_fuse23 = 0), three launches per step, linSolve0 CGS_ returns the same iterate."""
    from sla_amd import _lib
    dims, csr = _march_cases()[name]
    n = dims[0]
    Ao = _oracle_csr(dims, csr)
    b = orc.spmv(Ao, np.linspace(-1.0, 2.0, n))
    x0 = np.full(n, 0.25)
    states, sols, launches = {}, {}, {}
    for f23 in (1, 0):
        ctx = sla.Context(0).set_options(wd_lds=2, wd_march=2, bicg_fuse23=f23, onchip=0)   # (onchip = 0: these are tests of the LAUNCH flow's kernels)
        if grid:
            ctx.set_option("spmv_grid", grid)
        A = sla.fromCSR(dims, *csr, ctx)
        assert "wdia+march" in A.kernel_info().split()[0], A.kernel_info()
        st = sla.cgsInit(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
        snaps = []
        ctx.prof_start(_lib.KERNEL_ALL, 64)
        for k in (1, 1, 5):
            st.step(k)
            snaps.append([v.toDenseListSV().copy() for v in (st._x, st._r, st._p, st._u)])
        ctx.prof_stop()
        launches[f23] = (ctx.prof_query(_lib.KERNEL_CGS_C2)[0], ctx.prof_query(_lib.KERNEL_SPMV_DOT2)[0], ctx.prof_query(_lib.KERNEL_CGS_C4)[0])
        states[f23] = snaps
        xs, info = sla.linSolve0(sla.CGS_, A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx), return_info=True)
        sols[f23] = (xs.toDenseListSV(), info["iters"], info["resnorm"])
        del A, st
        ctx.close()
    assert launches[1] == (0, 7, 7) and launches[0] == (7, 7, 7), launches
    for a, c in zip(states[1], states[0]):
        for u, v in zip(a, c):
            assert np.array_equal(u.view(np.uint64), v.view(np.uint64)), (name, grid, np.abs(u - v).max())
    assert sols[1][1] == sols[0][1] and sols[1][2] == sols[0][2], (sols[1][1:], sols[0][1:])
    assert np.array_equal(sols[1][0].view(np.uint64), sols[0][0].view(np.uint64))
    co = orc.CgsState(Ao, b, x0)
    co.step(b - orc.spmv(Ao, x0), 2)
    for got, want in zip(states[1][1], (co.x, co.r, co.p, co.u)):
        assert np.linalg.norm(got - want) <= 1e-10 * np.linalg.norm(want)


def _gather_cases():
    from sla_amd import workloads as wl
    return {
        # constant coefficients, gather kernel (spmv_wdia_kernel): 2-D 5-point grid, odd row count; a 3-D grid of 12 planes
        "poisson2d 301x211": (wl.poisson2d(301, 211), "algo=wdia", {"wd_lds": 0}),
        "laplace3d 40x40x12": (wl.laplace3d(40, 40, 12), "algo=wdia", {"wd_lds": 0}),
        # variable coefficients (per-row value blocks): the non-symmetric banded matrix of config 5
        "banded 20011": (wl.banded_nonsym(20011), "algo=wdia-vv", {}),
    }


@pytest.mark.parametrize("name", list(_gather_cases()))
@pytest.mark.parametrize("grid", [0, 8])
def test_gather_kernel_k2_folded_into_k3_same_bits(sla, name, grid):
    """bicg_fuse23 on the gather kernel of the wave-sliced forms (constant and variable coefficients, <= 4 M rows): every gather of K3 loads
    the row pair of r and of Ap and combines them with K2's multiply-add; x, r, p after 1, 2 and 7 steps bit-identical to the four-launch
    flow, three launches per step, linSolve0 returns the same iterate."""
    from sla_amd import _lib
    (dims, csr), algo, opts = _gather_cases()[name]
    n = dims[0]
    Ao = _oracle_csr(dims, csr)
    b = orc.spmv(Ao, np.linspace(-1.0, 2.0, n))
    x0 = np.full(n, 0.25)
    states, sols, launches = {}, {}, {}
    for f23 in (1, 0):
        ctx = sla.Context(0).set_options(bicg_fuse23=f23, onchip=0, **opts)   # (the launch flow's gather kernel: not the on-chip step)
        if grid:
            ctx.set_option("spmv_grid", grid)
        A = sla.fromCSR(dims, *csr, ctx)
        assert A.kernel_info().split()[0] == algo, A.kernel_info()
        st = sla.bicgsInit(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
        snaps = []
        ctx.prof_start(_lib.KERNEL_ALL, 64)
        for k in (1, 1, 5):
            st.step(k)
            snaps.append([v.toDenseListSV().copy() for v in (st._xBicgstab, st._rBicgstab, st._pBicgstab)])
        ctx.prof_stop()
        launches[f23] = (ctx.prof_query(_lib.KERNEL_BICG_K2)[0], ctx.prof_query(_lib.KERNEL_SPMV_DOT2)[0], ctx.prof_query(_lib.KERNEL_BICG_K45)[0])
        states[f23] = snaps
        xs, info = sla.linSolve0(sla.BICGSTAB_, A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx), return_info=True)
        sols[f23] = (xs.toDenseListSV(), info["iters"], info["resnorm"])
        del A, st
        ctx.close()
    assert launches[1] == (0, 7, 7) and launches[0] == (7, 7, 7), launches
    for a, c in zip(states[1], states[0]):
        for u, v in zip(a, c):
            assert np.array_equal(u.view(np.uint64), v.view(np.uint64)), (name, grid, np.abs(u - v).max())
    assert sols[1][1] == sols[0][1] and sols[1][2] == sols[0][2], (sols[1][1:], sols[0][1:])
    assert np.array_equal(sols[1][0].view(np.uint64), sols[0][0].view(np.uint64))
    so = orc.BicgstabState(Ao, b, x0)
    so.step(b - orc.spmv(Ao, x0), 2)
    for got, want in zip(states[1][1], (so.x, so.r, so.p)):
        assert np.linalg.norm(got - want) <= 1e-10 * np.linalg.norm(want)


def test_plane_march_randomised_patterns(sla):
    """40 seeded random 5- / 7-pair stencils with one pair at -D and one at +D (D even, 1024..3000; in-plane offsets up to +-254: a window of at most 512 pairs; row
    counts that are not multiples of D; random holes; several values share no offset): the march form must be taken, and (#>), (<#)
    are the oracle's left fold bit for bit under random grid caps (several tasks per workgroup) and occupancies."""
    rng = np.random.default_rng(777)
    taken = 0
    for case in range(40):
        D = 2 * int(rng.integers(512, 1501))
        npairs = 5 if case % 2 else 7
        inner = sorted(set(int(v) for v in rng.integers(1, 255, npairs // 2 - 1)))
        while len(inner) < npairs // 2 - 1:
            inner = sorted(set(inner + [int(rng.integers(1, 255))]))
        offsets = [-D] + [-o for o in reversed(inner)] + [0] + inner + [D]
        n = int(rng.integers(2 * D, 7 * D)) + int(rng.integers(0, 2))
        vals = {o: float(rng.choice([-2.0, -1.0, -0.5, 0.25, 3.0])) for o in offsets}
        vals[0] = 9.0
        hole = float(rng.random() * 0.4)
        drop = rng.random((n, len(offsets))) < hole
        dims, csr = _stencil(n, offsets, lambda r, o: np.full(len(r), vals[o]),
                             keep=lambda r, t: ~drop[r, t] | (np.asarray(offsets)[t] == 0))
        Ao = _oracle_csr(dims, csr)
        x = rng.standard_normal(n)
        want, want_t = orc.spmv(Ao, x), orc.spmv(orc.transpose(Ao), x)
        ctx = sla.Context(0).set_options(wd_lds=2, wd_march=2, wd_march_occ=int(rng.integers(1, 5)))
        if case % 3 == 0:
            ctx.set_option("spmv_grid", 8 * int(rng.integers(1, 9)))
        A = sla.fromCSR(dims, *csr, ctx)
        info = A.kernel_info()
        taken += "wdia+march" in info
        assert "wdia+march" in info, (case, D, offsets, info)
        y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
        yt = sla.vecMat(sla.fromVector(x, ctx), A).toDenseListSV()
        assert np.array_equal(y.view(np.uint64), want.view(np.uint64)), (case, D, n, offsets, info)
        assert np.array_equal(yt.view(np.uint64), want_t.view(np.uint64)), (case, D, n, offsets, "transpose")
        # the fused residual and dot epilogues against the general kernels' sums (same rows, other grouping)
        b = orc.spmv(Ao, np.ones(n))
        r = sla.matVec(A, sla.fromVector(np.ones(n), ctx)).toDenseListSV()
        assert np.array_equal(r, b)
        del A
        ctx.close()
    assert taken == 40
