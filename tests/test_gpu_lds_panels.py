"""The LDS-panel SpMV form (spmv_lpanel_kernel + lpanel_finish_kernel: x cut into 16384-column panels held in LDS, one
wavefront per (row, panel) segment, per-panel partial sums added in ascending panel order) against the oracle and
against the stream kernel it replaces on matrices with dense rows: several panels, a single panel, rectangular
shapes, empty / one-entry / very long rows next to each other, non-finite operands, and every fused epilogue through
the solvers.  Rows this long are summed by wavefront segments in every GPU form, so parity is the rounding bound of
SURVEY 8(a) row A1 (|dy_i| <= gamma_k sum_j |a_ij x_j|), not bit equality."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sla():
    import sla_amd
    return sla_amd


def _dense_rows(m, n, per_row, seed, ragged=False, dominant=False):
    """m x n CSR with ~per_row uniformly random distinct columns per row (ascending), values in [-1, 1)."""
    rng = np.random.default_rng(seed)
    rp = [0]
    cols, vals = [], []
    for i in range(m):
        k = per_row
        if ragged:
            k = (0, 1, 3, per_row, 4 * per_row, 700)[i % 6]      # empty, short and > 256-entry segments side by side
        k = min(k, n)
        c = np.sort(rng.choice(n, size=k, replace=False)) if k else np.zeros(0, np.int64)
        v = rng.uniform(-1.0, 1.0, k)
        if dominant and i < n:
            if i not in c:
                c = np.sort(np.append(c, i))
                v = np.append(v, 0.0)
            v[np.searchsorted(c, i)] = float(k) + 1.0
        cols.append(c.astype(np.int64))
        vals.append(v)
        rp.append(rp[-1] + len(c))
    return (m, n), (np.array(rp, np.int64), np.concatenate(cols), np.concatenate(vals))


def _bound(csr, x, m):
    """gamma_k * sum |a_ij x_j| per row, k = entries of the row (plus the panel additions)."""
    rp, ci, va = csr
    absum = np.add.reduceat(np.append(np.abs(va * x[ci]), 0.0), np.minimum(rp[:-1], len(va)))[:m]
    absum[np.diff(rp) == 0] = 0.0
    return (np.diff(rp) + 64) * 1.2e-16 * absum + 1e-300


CASES = {
    "3 panels, square 40000 x 40000, 100 per row": lambda: _dense_rows(40000, 40000, 100, 1),
    "4 panels, wide 700 x 60000, 400 per row": lambda: _dense_rows(700, 60000, 400, 2),
    "1 panel, tall 9000 x 3000, 64 per row": lambda: _dense_rows(9000, 3000, 64, 3),
    "ragged rows (0, 1, 3, 80, 320, 700 entries), 2 panels": lambda: _dense_rows(3001, 20000, 80, 4, ragged=True),
    "panel edge: n = 16384 + 1": lambda: _dense_rows(2000, 16385, 200, 5),
    "fewer rows than one chunk: 7 x 40000, 3000 per row": lambda: _dense_rows(7, 40000, 3000, 6),
    "one row, one panel: 1 x 500, 400 entries": lambda: _dense_rows(1, 500, 400, 7),
    "16-lane groups: 5000 x 30000, 40 per row": lambda: _dense_rows(5000, 30000, 40, 8),
}


@pytest.mark.parametrize("name", list(CASES))
def test_lds_panels_match_the_oracle_and_the_stream_kernel(sla, name):
    dims, csr = CASES[name]()
    m, n = dims
    Ao = orc.Csr(m, n, *csr)
    rng = np.random.default_rng(8)
    x, xt = rng.standard_normal(n), rng.standard_normal(m)
    want, want_t = orc.spmv(Ao, x), orc.spmv(orc.transpose(Ao), xt)
    got = {}
    for form, opts in (("ldspanels", {}), ("ldspanels row-major", {"lp_copy": 0}), ("stream", {"lpanel": 0})):
        ctx = sla.Context(0).set_options(**opts)
        A = sla.fromCSR(dims, *csr, ctx)
        assert ("ldspanels" in A.kernel_info()) == form.startswith("ldspanels"), (name, A.kernel_info())
        if form.startswith("ldspanels"):    # (the default streams a panel-major second copy of the entries)
            assert ("entries=panel-major-copy" in A.kernel_info()) == (form == "ldspanels"), A.kernel_info()
        y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
        yt = sla.vecMat(sla.fromVector(xt, ctx), A).toDenseListSV()
        assert np.all(np.abs(y - want) <= _bound(csr, x, m)), (name, form, np.abs(y - want).max())
        assert np.allclose(yt, want_t, rtol=1e-12, atol=1e-12), (name, form, "transpose")
        got[form] = y
        del A
        ctx.close()
    assert np.all(np.abs(got["ldspanels"] - got["stream"]) <= 2 * _bound(csr, x, m))
    # same segments; the copy is read in pairs of entries per lane, the row-major arrays one entry per lane: wavefront-segmented sums
    # either way (the rounding bound of SURVEY 8(a) A1), grouped differently
    assert np.all(np.abs(got["ldspanels"] - got["ldspanels row-major"]) <= 2 * _bound(csr, x, m))
    empty = np.diff(csr[0]) == 0
    assert np.all(got["ldspanels"][empty] == 0.0)            # a row key with an empty row map gives 0.0 (Common.hs:242-250)


def test_lds_panels_touch_only_the_referenced_entries_of_x(sla):
    """Non-finite entries of x reach exactly the rows that reference them (a whole panel of x sits in LDS, but a lane
    only reads the columns of its own entries)."""
    dims, csr = _dense_rows(1500, 20000, 90, 6)
    m, n = dims
    Ao = orc.Csr(m, n, *csr)
    x = np.random.default_rng(9).standard_normal(n)
    x[[5, 16383, 16384, 19999]] = [np.inf, np.nan, -np.inf, np.nan]
    A = sla.fromCSR(dims, *csr)
    assert "ldspanels" in A.kernel_info()
    y = sla.matVec(A, sla.fromVector(x)).toDenseListSV()
    want = orc.spmv(Ao, x)
    assert np.array_equal(np.isnan(y), np.isnan(want)) and np.array_equal(np.isinf(y), np.isinf(want))
    fin = np.isfinite(want)
    assert fin.any() and (~fin).any()
    assert np.array_equal(np.sign(y[np.isinf(y)]), np.sign(want[np.isinf(want)]))
    assert np.allclose(y[fin], want[fin], rtol=1e-12, atol=1e-12)


def test_lds_panels_epilogues_through_the_solvers(sla):
    """K1/K3 (dot, dot2), the true-residual sweep (stand-alone: the fused two-vector sweep is not used with this form),
    CGS's and CGNE's fused updates, r0 = b - A x0, Arnoldi and GMRES on a diagonally dominant matrix with dense rows:
    the same iteration counts as the oracle and the stream kernel, iterates equal up to partial-sum grouping."""
    dims, csr = _dense_rows(18000, 18000, 70, 7, dominant=True)
    n = dims[0]
    Ao = orc.Csr(n, n, *csr)
    b = orc.spmv(Ao, np.linspace(-1.0, 2.0, n))
    x0 = np.full(n, 0.25)
    out = {}
    for form, opts in (("ldspanels", {}), ("stream", {"lpanel": 0})):
        ctx = sla.Context(0).set_options(**opts)
        A = sla.fromCSR(dims, *csr, ctx)
        assert ("ldspanels" in A.kernel_info()) == (form == "ldspanels")
        for meth in (sla.BICGSTAB_, sla.CGS_, sla.CGNE_):
            x, info = sla.linSolve0(meth, A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx), return_info=True)
            out[(form, int(meth))] = (x.toDenseListSV(), info["iters"])
        Q, H = sla.arnoldi(A, sla.fromVector(b, ctx), 5)
        out[(form, "H")] = H
        del A
        ctx.close()
    for meth, ometh in ((sla.BICGSTAB_, orc.BICGSTAB_), (sla.CGS_, orc.CGS_), (sla.CGNE_, orc.CGNE_)):
        rc, xo, it_o, res_o, r0_o = orc.linsolve0(ometh, Ao, b, x0)
        x, it = out[("ldspanels", int(meth))]
        xs, its = out[("stream", int(meth))]
        assert it == its and abs(it - it_o) <= 1, (meth, it, its, it_o)
        assert it_o < 200 and np.linalg.norm(orc.spmv(Ao, x) - b) <= max(1e-6, 1e-4 * r0_o) * (1 + 1e-9)
        assert np.abs(x - xs).max() <= 1e-8 * (np.abs(xs).max() + 1e-300), meth
        assert np.abs(x - xo).max() <= 1e-8 * (np.abs(xo).max() + 1e-300), meth
    rc, Qo, Ho, k = orc.arnoldi(Ao, b, 5)
    Hl = out[("ldspanels", "H")]
    # (single-pass classical Gram-Schmidt amplifies the last-bit differences of the long-row sums from column to column)
    assert Hl.shape == Ho.shape and np.abs(Hl - Ho).max() <= 1e-7 * np.abs(Ho).max()
    assert np.abs(Hl - out[("stream", "H")]).max() <= 1e-7 * np.abs(Ho).max()


def test_lds_panels_randomised_shapes(sla):
    """24 seeded random matrices with dense rows -- 1..5 panels, tall / wide / square, uniform and strongly skewed row
    lengths (most rows short, a few holding most entries), empty leading / trailing rows: (#>) within the rounding
    bound of the oracle's left fold, the epilogue b - A x (EPI_SUB) and the fused residual norm through linSolve0's
    first check included."""
    rng = np.random.default_rng(77)
    taken = 0
    for case in range(24):
        m = int(rng.integers(300, 5000))
        n = int(rng.integers(200, 70000))
        base = int(rng.integers(30, 200)) * (1 + n // 16384)
        if case % 3 == 0:      # skewed: 85 % of the rows short, the rest long
            lens = np.where(rng.random(m) < 0.85, rng.integers(0, 12, m), rng.integers(base * 4, base * 8, m))
        else:
            lens = rng.integers(base // 2, base * 2, m)
        lens = np.minimum(lens, n)
        if case % 4 == 0:
            lens[: m // 10] = 0
            lens[-(m // 7):] = 0
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ci = np.concatenate([np.sort(rng.choice(n, size=int(k), replace=False)) for k in lens] + [np.zeros(0, np.int64)]).astype(np.int64)
        va = rng.uniform(-2.0, 2.0, len(ci))
        Ao = orc.Csr(m, n, rp, ci, va)
        x = rng.standard_normal(n)
        A = sla.fromCSR((m, n), rp, ci, va)
        taken += "ldspanels" in A.kernel_info()
        y = sla.matVec(A, sla.fromVector(x)).toDenseListSV()
        want = orc.spmv(Ao, x)
        assert np.all(np.abs(y - want) <= _bound((rp, ci, va), x, m)), (case, A.kernel_info(), np.abs(y - want).max())
        del A
    assert taken >= 12, taken


# ---- the flat variant (sla_spmv_lflat.hip, round 4): one lane per (panel, row) segment, for segments of 1.5 .. 16 entries ----------------

FLAT_CASES = {
    "4 panels, 60000 x 60000, 24 per row (segments of 6)": lambda: _dense_rows(60000, 60000, 24, 21),
    "7 panels, wide 3000 x 100000, 40 per row": lambda: _dense_rows(3000, 100000, 40, 22),
    "ragged rows (0, 1, 3, 30, 120, 700 entries), 4 panels": lambda: _dense_rows(5003, 60000, 30, 23, ragged=True),
    "fewer rows than a workgroup round: 100 x 80000, 60 per row": lambda: _dense_rows(100, 80000, 60, 24),
    "one row: 1 x 70000, 90 entries": lambda: _dense_rows(1, 70000, 90, 25),
    "tall 120000 x 50000, 8 per row (segments of 2)": lambda: _dense_rows(120000, 50000, 8, 26),
}


@pytest.mark.parametrize("name", list(FLAT_CASES))
def test_flat_lds_panel_form_matches_the_oracle(sla, name):
    """`lflat`: the form between the tile form and the LDS panels.  Device-built panel-major copy, one lane per segment, left fold per
    segment, segment sums folded in ascending panel order by the shared finish kernel: per row within nnz_i eps sum |a_ij x_j| of the
    oracle's single left fold, deterministic, and against the forms it replaces (lflat = 0)."""
    dims, csr = FLAT_CASES[name]()
    m, n = dims
    rp, ci, va = csr
    Ao = orc.Csr(m, n, rp, ci, va)
    x = np.random.default_rng(11).standard_normal(n)
    want = orc.spmv(Ao, x)
    bound = _bound(csr, x, m)
    # (force the form onto every case: the lowering's own window is a mean segment of 1.5 .. 16 entries on a matrix without band structure)
    ctx = sla.Context(0).set_options(lflat=2, lf_min_seg10=1, lpanel=0, diag=0, wdia=0, vdict=0)
    A = sla.fromCSR(dims, rp, ci, va, ctx)
    info = A.kernel_info()
    assert info.startswith("algo=lflat "), (name, info)
    y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
    assert np.all(np.abs(y - want) <= bound), (name, float(np.abs(y - want).max()), int(np.argmax(np.abs(y - want) - bound)))
    assert np.array_equal(y, sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV())
    if m == n:     # (<#) goes through the lazily built transpose, which picks its own form
        w = np.random.default_rng(12).standard_normal(m)
        yt = sla.vecMat(sla.fromVector(w, ctx), A).toDenseListSV()
        assert np.allclose(yt, orc.spmv(orc.transpose(Ao), w), rtol=1e-12, atol=1e-12)
    del A
    ctx.close()
    ctx0 = sla.Context(0).set_options(lflat=0)
    A0 = sla.fromCSR(dims, rp, ci, va, ctx0)
    assert "lflat" not in A0.kernel_info()
    y0 = sla.matVec(A0, sla.fromVector(x, ctx0)).toDenseListSV()
    assert np.all(np.abs(y0 - want) <= bound)
    del A0
    ctx0.close()


def test_flat_lds_panel_form_is_picked_for_medium_rows_and_runs_the_solvers(sla):
    """The lowering's own choice (no forcing): 50000 x 50000 with 24 random entries per row = 4 panels, segments of 6 -> lflat; 8 per row
    (segments of 2) too; 3 per row (segments of 0.75) not.  Then every fused epilogue through bicgstabStep (fused and split), cgsStep,
    cgneStep and linSolve0 against the oracle."""
    n = 50000
    for k, expect in ((24, True), (8, True), (3, False)):
        dims, (rp, ci, va) = _dense_rows(n, n, k, 31 + k, dominant=True)
        A = sla.fromCSR(dims, rp, ci, va)
        assert A.kernel_info().startswith("algo=lflat ") == expect, (k, A.kernel_info())
        del A
    dims, (rp, ci, va) = _dense_rows(n, n, 24, 55, dominant=True)
    Ao = orc.Csr(n, n, rp, ci, va)
    b = orc.spmv(Ao, np.linspace(-1.0, 2.0, n))
    x0 = np.full(n, 0.25)
    for fuse in (1, 0):
        ctx = sla.Context(0).set_options(bicg_fuse45=fuse)
        A = sla.fromCSR(dims, rp, ci, va, ctx)
        assert A.kernel_info().startswith("algo=lflat ")
        so, sd = orc.BicgstabState(Ao, b, x0), sla.bicgsInit(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
        so.step(b - orc.spmv(Ao, x0), 2)
        sd.step(2)
        for nm, dev, ref in (("x", sd._xBicgstab, so.x), ("r", sd._rBicgstab, so.r), ("p", sd._pBicgstab, so.p)):
            assert np.linalg.norm(dev.toDenseListSV() - ref) <= 1e-11 * np.linalg.norm(ref), (fuse, nm)
        del sd
        if fuse:
            sc, sdc = orc.CgsState(Ao, b, x0), sla.cgsInit(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
            sc.step(b - orc.spmv(Ao, x0), 2)
            sdc.step(2)
            assert np.linalg.norm(sdc._x.toDenseListSV() - sc.x) <= 1e-11 * np.linalg.norm(sc.x)
            sn, sdn = orc.CgneState(Ao, b, x0), sla.cgneInit(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
            sn.step(2)
            sdn.step(2)
            assert np.linalg.norm(sdn._xCgne.toDenseListSV() - sn.x) <= 1e-11 * np.linalg.norm(sn.x)
            del sdc, sdn
            xs, inf = sla.linSolve0(sla.BICGSTAB_, A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx), return_info=True)
            rc, xo, it_o, res_o, r0_o = orc.linsolve0(orc.BICGSTAB_, Ao, b, x0)
            assert inf["converged"] and abs(inf["iters"] - it_o) <= 2 and np.linalg.norm(orc.spmv(Ao, xs.toDenseListSV()) - b) <= inf["tol"] * (1 + 1e-9)
        del A
        ctx.close()
