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


@pytest.mark.parametrize("kind", ["laplace3d", "banded", "random", "chain"])
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
    b2 = rng.standard_normal(n)                                  # another right-hand side buffer: re-captured
    rc, want2, _ = (orc.tri_upper_solve if upper else orc.tri_lower_solve)(A, b2)
    x2 = (sla.triUpperSolve if upper else sla.triLowerSolve)(T, sla.DeviceVector(T.ctx, n, b2)).to_host()
    assert np.array_equal(x2, want2)


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
