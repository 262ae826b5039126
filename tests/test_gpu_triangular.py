"""SURVEY 8(f).2 on the GPU: triLowerSolve / triUpperSolve (level-scheduled substitution) and the SSOR factors against
the oracle -- bit for bit: every row is the reference's ascending fold, one subtraction, one division."""
import numpy as np
import pytest

from oracle import oracle as orc
from refdata import coo_of, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sla():
    import sla_amd
    return sla_amd


def _csr(dims, r, c, v):
    rc, A = orc.coo_to_csr(dims[0], dims[1], r, c, v)
    assert rc == orc.OK
    return A


@pytest.mark.parametrize("name", ["ltri0", "utri0", "ltri1", "utri1"])
def test_reference_triangular_cases(sla, name):
    e = golden()["triangular"][name]
    dims, r, c, v = coo_of(e)
    T = sla.fromListSM(dims, list(zip(r.tolist(), c.tolist(), v.tolist())))
    b = sla.fromListDenseSV(len(e["b"]), e["b"])
    x = (sla.triUpperSolve if e["upper"] else sla.triLowerSolve)(T, b)
    assert x.toDenseListSV().tolist() == e["x"]
    resid = sla.matVec(T, x).toDenseListSV() - np.array(e["b"])          # the reference's own check (LibSpec.hs:436-458)
    assert np.sqrt((resid ** 2).sum()) <= 1e-12


@pytest.mark.parametrize("kind", ["laplace3d", "banded", "random", "chain", "band3"])
@pytest.mark.parametrize("upper", [False, True])
def test_triangular_solve_is_the_reference_substitution(sla, kind, upper):
    from sla_amd import workloads as wl
    rng = np.random.default_rng(17)
    if kind == "laplace3d":
        dims, (rp, ci, va) = wl.laplace3d(13, 9, 11)            # levels = hyperplanes of the grid
    elif kind == "banded":
        dims, (rp, ci, va) = wl.banded_nonsym(4001)
    elif kind == "random":
        dims, (rp, ci, va) = wl.random_spd(1500, 5, 3)
    elif kind == "band3":                                        # offsets +-1, +-24, +-144 on EVERY row (no grid boundaries): looks like a 24 x 6 x . stencil to
        n = 24 * 6 * 7                                           # the block-local plan, but its brick order would put rows before rows they read -- the
        rows, cols = [], []                                      # plan must notice and keep the sweep order
        for d in (-144, -24, -1, 0, 1, 24, 144):
            i = np.arange(max(0, -d), min(n, n - d))
            rows.append(i)
            cols.append(i + d)
        rows, cols = np.concatenate(rows), np.concatenate(cols)
        order = np.lexsort((cols, rows))
        rows, cols = rows[order], cols[order]
        rp = np.zeros(n + 1, np.int64)
        np.add.at(rp, rows + 1, 1)
        rp = np.cumsum(rp)
        ci, va, dims = cols, rng.uniform(0.5, 1.5, len(cols)) * np.where(cols == rows, 6.0, 1.0), (n, n)
    else:                                                        # bidiagonal: n levels of one row each
        n = 700
        rows = np.repeat(np.arange(n), 3)[1:-1]
        cols = (rows + np.tile([-1, 0, 1], n)[1:-1])
        rp = np.zeros(n + 1, np.int64)
        np.add.at(rp, rows + 1, 1)
        rp = np.cumsum(rp)
        ci, va, dims = cols, rng.uniform(0.5, 1.5, len(cols)) * np.where(cols == rows, 4.0, 1.0), (n, n)
    n = dims[0]
    A = orc.Csr(n, n, rp, ci, va)                                # the whole matrix: the solve uses one triangle of it
    b = rng.standard_normal(n)
    b[rng.integers(0, n, 5)] = 0.0
    rc, want, _ = (orc.tri_upper_solve if upper else orc.tri_lower_solve)(A, b)
    assert rc == orc.OK
    T = sla.fromCSR(dims, rp, ci, va)
    lv, wd = sla.triSolveLevels(T, upper)
    assert 1 <= lv <= n and 1 <= wd <= n
    if kind == "chain":
        assert lv == n and wd == 1
    if kind == "laplace3d":
        assert lv == 13 + 9 + 11 - 2
    bv = sla.DeviceVector(T.ctx, n, b)
    for rep in range(3):                                         # first call captures the graph, later ones replay it
        x = (sla.triUpperSolve if upper else sla.triLowerSolve)(T, bv).to_host()
        assert np.array_equal(x.view(np.uint64), want.view(np.uint64)), (kind, upper, rep, np.abs(x - want).max())
    # the default picks the form by the schedule's shape: deep and narrow with the dependencies inside the blocks -> the block-local launch
    assert int(T.ctx.get_option("tri_mode_used")) == {"chain": 2, "banded": 2, "random": 0, "laplace3d": 0, "band3": 2}[kind], (kind, lv, wd)
    b2 = rng.standard_normal(n)                                  # another right-hand side buffer: re-captured
    rc, want2, _ = (orc.tri_upper_solve if upper else orc.tri_lower_solve)(A, b2)
    x2 = (sla.triUpperSolve if upper else sla.triLowerSolve)(T, sla.DeviceVector(T.ctx, n, b2)).to_host()
    assert np.array_equal(x2, want2)
    # round 5: the same substitution as ONE persistent launch whose rows poll x for the rows they read (option tri_syncfree,
    # csrc/sla_tri.hip): the same bits at every grid size -- the chain (a dependency between neighbouring LANES of one wavefront in every
    # slot) included -- and a spin limit of one poll makes lanes give up: the level schedule then runs instead, same bits again
    ctx = sla.Context(0)
    T2 = sla.fromCSR(dims, rp, ci, va, ctx)
    bv2 = sla.DeviceVector(ctx, n, b)
    ctx.set_options(tri_syncfree=0)                              # (the default picks by the schedule's shape: here the level schedule by name)
    for rep in range(2):
        x0 = (sla.triUpperSolve if upper else sla.triLowerSolve)(T2, bv2).to_host()
        assert np.array_equal(x0.view(np.uint64), want.view(np.uint64)), (kind, upper, "levels", rep)
    for grid, spin in ((1, 200000), (7, 200000), (256, 200000), (256, 1)):
        ctx.set_options(tri_syncfree=1, tri_grid=grid, tri_spin=spin)
        before = int(ctx.get_option("tri_fallbacks"))
        x3 = (sla.triUpperSolve if upper else sla.triLowerSolve)(T2, bv2).to_host()
        assert np.array_equal(x3.view(np.uint64), want.view(np.uint64)), (kind, upper, grid, spin)
        fell_back = int(ctx.get_option("tri_fallbacks")) - before
        assert fell_back == (1 if spin == 1 and lv > 1 else 0) or spin == 1, (kind, grid, spin, fell_back)
    # ... and the block-local form (tri_syncfree = 2): blocks of consecutive rows, one workgroup each with the block's x in LDS, other
    # blocks' rows polled in memory; tiny blocks put most dependencies ACROSS blocks, one big block keeps them all in LDS
    for rows_per_block, grid, spin in ((8, 256, 200000), (8, 3, 200000), (64, 256, 200000), (200, 1, 200000), (16384, 256, 200000), (8, 256, 1)):
        ctx.set_options(tri_syncfree=2, tri_block_rows=rows_per_block, tri_grid=grid, tri_spin=spin)
        before = int(ctx.get_option("tri_fallbacks"))
        for rep in range(2):
            x4 = (sla.triUpperSolve if upper else sla.triLowerSolve)(T2, bv2).to_host()
            assert np.array_equal(x4.view(np.uint64), want.view(np.uint64)), (kind, upper, rows_per_block, grid, spin, rep)
        assert spin == 1 or int(ctx.get_option("tri_fallbacks")) == before, (kind, rows_per_block, grid)
        plan = ctx.get_option("tri_plan")
        if kind == "laplace3d" and rows_per_block in (64, 200):    # 13-row lines: bricks of 2 x 2 / 5 x 3 lines
            assert "bricks=1" in plan, plan
        if kind in ("banded", "chain", "random") or (kind == "band3" and rows_per_block == 200):
            assert "bricks=0" in plan, (kind, plan)               # (band3 in bricks of 4 lines x 2 planes: row (0, 0, odd plane) reads the last line of
                                                                   # the plane before it, which a LATER brick holds -- not a valid order, dropped)
    del T2, bv2
    ctx.close()


def test_needs_pivoting_and_sparsify(sla):
    T = sla.fromListSM((3, 3), [(0, 0, 2.0), (1, 0, 1.0), (2, 1, 1.0), (2, 2, 1e-13)])       # l_11 missing
    with pytest.raises(sla.NeedsPivoting) as ei:
        sla.triLowerSolve(T, sla.fromListDenseSV(3, [1.0, 1.0, 1.0]))
    assert "(1,1)" in str(ei.value)
    U = sla.fromListSM((2, 2), [(0, 0, 2.0), (1, 1, 1e-13)])
    with pytest.raises(sla.NeedsPivoting):
        sla.triUpperSolve(U, sla.fromListDenseSV(2, [1.0, 1.0]))
    # sparsifySV on the way out, unsparsified values inside the recurrence
    L = sla.fromListSM((2, 2), [(0, 0, 1.0), (1, 0, 1e12), (1, 1, 1.0)])
    x = sla.triLowerSolve(L, sla.fromListDenseSV(2, [1e-13, 1.0]))
    assert x.toDenseListSV().tolist() == [0.0, 1.0 - 1e12 * 1e-13] and x.nnz() == 1
    with pytest.raises(sla.MatVecSizeMismatchException):
        sla.triLowerSolve(L, sla.fromListDenseSV(3, [1.0, 1.0, 1.0]))


@pytest.mark.parametrize("omega", [1.0, 1.3])
def test_ssor_factors_and_their_solves(sla, omega):
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.poisson2d(23, 19)
    n = dims[0]
    A = orc.Csr(n, n, rp, ci, va)
    rc, Lo, Ro = orc.ssor_pre(A, omega)
    Ad = sla.fromCSR(dims, rp, ci, va)
    L, R = sla.mSsorPre(Ad, omega)
    for got, want in ((L, Lo), (R, Ro)):
        grp, gci, gva = got.csr()
        assert np.array_equal(grp, want.rowptr) and np.array_equal(gci, want.colidx)
        assert np.array_equal(gva.view(np.uint64), want.val.view(np.uint64))
    # the factors are triangular with a usable diagonal: apply M^-1 = R^-1 L^-1 like a preconditioner would
    b = np.random.default_rng(2).standard_normal(n)
    rc, w, _ = orc.tri_lower_solve(Lo, b)
    rc, z, _ = orc.tri_upper_solve(Ro, w)
    wd = sla.triLowerSolve(L, sla.DeviceVector(L.ctx, n, b))
    zd = sla.triUpperSolve(R, wd).to_host()
    assert np.array_equal(zd, z)


# ---- ilu0Pre (Sparse.hs:696-706) -------------------------------------------------------------------------

def _same_csr(M, Mo):
    rp, ci, va = M.csr()
    assert np.array_equal(rp, Mo.rowptr) and np.array_equal(ci, Mo.colidx) and np.array_equal(va, Mo.val)


def test_ilu0_pre_is_the_filtered_complete_lu_bit_for_bit(sla):
    """exact mode = the reference's definition (complete Doolittle `lu`, then keep aa's stored positions): index structure and
    values equal the oracle's restatement exactly, on the reference's own checkLu matrices (aa0, tm0, tm7: LibSpec.hs:186-194)
    and on random matrices with heavy fill-in."""
    from refdata import coo_of, golden, tridiag_coo
    G = golden()
    cases = [coo_of(G["aa0"]), coo_of(G["lu"]["tm0"]), tridiag_coo(G["lu"]["tm7"]["n"], *G["lu"]["tm7"]["tridiag"])]
    rng = np.random.default_rng(4)
    for n, fill in ((7, 0.5), (40, 0.15), (120, 0.05)):
        nz = int(fill * n * n)
        r, c = rng.integers(0, n, nz), rng.integers(0, n, nz)
        v = rng.uniform(-1, 1, nz)
        r, c, v = np.append(r, np.arange(n)), np.append(c, np.arange(n)), np.append(v, n + rng.uniform(0, 1, n))   # dominant diagonal last (wins)
        cases.append(((n, n), r, c, v))
    for dims, r, c, v in cases:
        A = sla.fromCOO(dims, r, c, v)
        rc, Ao = orc.coo_to_csr(dims[0], dims[1], r, c, v)
        rc, Lo, Uo, _ = orc.ilu0_pre(Ao)
        assert rc == orc.OK
        L, U = sla.ilu0Pre(A)
        _same_csr(L, Lo)
        _same_csr(U, Uo)
        # and it is what the name promises: entries only where aa stores one
        Ap = A.toDense() != 0
        assert not (L.toDense() != 0)[~Ap & ~np.eye(dims[0], dtype=bool)].any() and not (U.toDense() != 0)[~Ap].any()


def test_incomplete_mode_equals_the_exact_one_without_fill_and_scales(sla):
    """exact=False (extension: the recurrences on aa's pattern only) returns the very same factors whenever `lu` creates no
    fill outside the pattern -- tridiagonal, pentadiagonal without gaps, block diagonal with full blocks -- and runs at sizes
    where the complete LU is out of reach; its factors then go through the level-scheduled triangular solves."""
    from refdata import tridiag_coo
    rng = np.random.default_rng(9)
    n = 300
    penta = [(i, j, (6.0 if i == j else rng.uniform(-1, 1))) for i in range(n) for j in range(max(0, i - 2), min(n, i + 3))]
    blocks = [(b * 5 + i, b * 5 + j, (9.0 if i == j else rng.uniform(-1, 1))) for b in range(40) for i in range(5) for j in range(5)]
    for dims, r, c, v in (tridiag_coo(500, -1, 2, -1),
                          ((n, n), *map(np.array, zip(*penta))), ((200, 200), *map(np.array, zip(*blocks)))):
        A = sla.fromCOO(dims, np.asarray(r, np.int64), np.asarray(c, np.int64), np.asarray(v, float))
        Le, Ue = sla.ilu0Pre(A, exact=True)
        Li, Ui = sla.ilu0Pre(A, exact=False)
        for X, Y in ((Le, Li), (Ue, Ui)):
            for a, b in zip(X.csr(), Y.csr()):
                assert np.array_equal(a, b)
    # 1 M-row 5-point Poisson matrix: ILU(0) proper; M^-1 = U^-1 L^-1 applied with sla_tri_solve brings ||b - A x|| down
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.poisson2d(1000, 1000)
    A = sla.fromCSR(dims, rp, ci, va)
    with pytest.raises(sla.SlaError):
        sla.ilu0Pre(A, exact=True)                                           # the complete LU is limited to 4096 rows
    L, U = sla.ilu0Pre(A, exact=False)
    assert L.nnz() == (len(ci) + dims[0]) // 2 and U.nnz() == (len(ci) + dims[0]) // 2    # pattern of A, split at the diagonal
    b = np.add.reduceat(va, rp[:-1])
    bv = sla.DeviceVector(A.ctx, dims[0], b)
    z = sla.triUpperSolve(U, sla.triLowerSolve(L, bv)).to_host()              # z = M^-1 b ~ A^-1 b = 1
    r0, r1 = np.linalg.norm(b), np.linalg.norm(b - orc.spmv(orc.Csr(*dims, rp, ci, va), z))
    assert r1 < 0.75 * r0


def test_ilu0_pre_needs_pivoting(sla):
    A = sla.fromListSM((3, 3), [(0, 0, 1.0), (0, 1, 2.0), (1, 0, 2.0), (1, 1, 4.0), (2, 1, 1.0), (2, 2, 1.0)])   # u_11 = 4 - 2 * 2 = 0
    for exact in (True, False):
        with pytest.raises(sla.NeedsPivoting):
            sla.ilu0Pre(A, exact=exact)
    with pytest.raises(sla.NeedsPivoting):
        sla.ilu0Pre(sla.fromListSM((2, 2), [(0, 1, 1.0), (1, 0, 1.0)]))                                           # u_00 missing
