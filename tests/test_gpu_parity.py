"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference's golden vectors.
Needs a real MI355X: every test here is marked gpu.

Tolerances (fp64):
  * CSR index construction: bit-exact (integers and values).
  * SpMV: rows reduced by one lane (short rows) are bit-exact with the reference's left fold; rows
    reduced by a wavefront segment / workgroup differ by summation order only:
    |dy_i| <= nnz_i * eps * sum_j |a_ij x_j|.
  * dot / norm2: |d| <= 4 * sqrt(n) * eps * sum |x_i y_i|  (two-stage tree vs left fold).
  * solver states after k <= 5 steps: relative 1e-9 (well-conditioned test matrices);
    linSolve0: same stopping rule, so the returned true residual <= tol and the iteration count
    within +-3 of the oracle's on Krylov-sensitive problems.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from refdata import GOLDEN, coo_of, golden, read_mtx_array, read_mtx_coordinate, tridiag_coo

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps
G = golden()


@pytest.fixture(scope="module")
def sla():
    import sla_amd
    return sla_amd


def both(sla, entry):
    (m, n), r, c, v = coo_of(entry) if isinstance(entry, dict) else entry
    rc, Ao = orc.coo_to_csr(m, n, r, c, v)
    assert rc == orc.OK
    return sla.fromCOO((m, n), r, c, v), Ao


def dense_vec(sla, a):
    return sla.fromVector(np.asarray(a, dtype=np.float64))


def rand_csr(rng, m, n, row_len_fn):
    rows, cols, vals = [], [], []
    for i in range(m):
        k = min(n, int(row_len_fn(i)))
        if k:
            cj = rng.choice(n, size=k, replace=False)
            rows.append(np.full(k, i)); cols.append(cj); vals.append(rng.standard_normal(k))
    return (m, n), np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)


# ---- A0 -------------------------------------------------------------------------------------------

def test_csr_index_parity_bit_exact(sla):
    rng = np.random.default_rng(7)
    for m, n, nnz in [(1, 1, 1), (37, 29, 400), (500, 500, 6000), (64, 5000, 3000)]:
        r, c, v = rng.integers(0, m, nnz), rng.integers(0, n, nnz), rng.standard_normal(nnz)
        A = sla.fromCOO((m, n), r, c, v)
        rc, Ao = orc.coo_to_csr(m, n, r, c, v)
        rp, ci, va = A.csr()
        assert np.array_equal(rp, Ao.rowptr) and np.array_equal(ci, Ao.colidx) and np.array_equal(va, Ao.val)


def test_csr_golden_layouts(sla):
    A = sla.fromListSM((2, 3), [(0, 0, 2), (1, 0, 3), (1, 2, 4), (1, 2, 1)])      # m1': last duplicate wins
    rp, ci, va = A.csr()
    assert rp.tolist() == [0, 1, 3] and ci.tolist() == [0, 0, 2] and va.tolist() == [2.0, 3.0, 1.0]
    A, Ao = both(sla, G["csr_literals"]["m3"])                                     # empty first row
    assert A.csr()[0].tolist() == [0, 0, 2, 3, 4]
    assert A.toListSM()[0] == (3, 1, 6.0)                                          # toListSM is descending


def test_csr_out_of_bounds_raises(sla):
    with pytest.raises(sla.IndexOutOfBounds):
        sla.fromListSM((2, 2), [(0, 0, 1.0), (2, 0, 1.0)])
    with pytest.raises(sla.IndexOutOfBounds):
        sla.fromListSM((2, 2), [(0, -1, 1.0)])


def test_e05r0000_index_and_spmv_parity(sla):
    dims, r, c, v = read_mtx_coordinate(f"{GOLDEN}/e05r0000.mtx")
    A, Ao = both(sla, (dims, r, c, v))
    rp, ci, va = A.csr()
    assert np.array_equal(rp, Ao.rowptr) and np.array_equal(ci, Ao.colidx) and np.array_equal(va, Ao.val)
    rhs = read_mtx_array(f"{GOLDEN}/e05r0000_rhs1.mtx")
    y, yo = sla.matVec(A, dense_vec(sla, rhs)).toDenseListSV(), orc.spmv(Ao, rhs)
    Dabs = orc.spmv(orc.Csr(Ao.m, Ao.n, Ao.rowptr, Ao.colidx, np.abs(Ao.val)), np.abs(rhs))
    klen = np.diff(Ao.rowptr)
    assert np.all(np.abs(y - yo) <= klen * EPS * Dabs)


# ---- A1: (#>) --------------------------------------------------------------------------------------

def test_matvec_golden(sla):
    A, _ = both(sla, G["aa0"])
    assert sla.matVec(A, dense_vec(sla, G["aa0"]["x_true"])).toDenseListSV().tolist() == G["aa0"]["b"]
    assert sla.vecMat(dense_vec(sla, G["aa0"]["x_true"]), A).toDenseListSV().tolist() == G["aa0"]["AT_x_true"]
    A, _ = both(sla, G["readme"])
    assert sla.matVec(A, dense_vec(sla, G["readme"]["x"])).toDenseListSV().tolist() == G["readme"]["b"]
    A, _ = both(sla, G["aa1"])
    assert sla.matVec(A, dense_vec(sla, G["aa1"]["x"])).toDenseListSV().tolist() == G["aa1"]["b"]


def test_matvec_structural_keys_and_dim_mismatch(sla):
    A, _ = both(sla, G["csr_literals"]["m3"])            # row 0 has no entries -> no key 0 in A #> x
    y = sla.matVec(A, sla.onesSV(4))
    assert y.ix.tolist() == [1, 2, 3] and y.dim == 4
    with pytest.raises(sla.MatVecSizeMismatchException):
        sla.matVec(A, sla.onesSV(3))


@pytest.mark.parametrize("shape", ["short", "mid", "long", "mixed", "empty_rows"])
def test_matvec_random_vs_oracle(sla, shape):
    rng = np.random.default_rng({"short": 1, "mid": 2, "long": 3, "mixed": 4, "empty_rows": 5}[shape])
    m = n = 3000
    fn = {"short": lambda i: rng.integers(1, 8), "mid": lambda i: rng.integers(20, 45),
          "long": lambda i: 2500 if i % 500 == 0 else 3,
          "mixed": lambda i: [0, 1, 5, 40, 300, 1500][rng.integers(0, 6)],
          "empty_rows": lambda i: 0 if i % 3 else 4}[shape]
    dims, r, c, v = rand_csr(rng, m, n, fn)
    A, Ao = both(sla, (dims, r, c, v))
    x = rng.standard_normal(n)
    y, yo = sla.matVec(A, dense_vec(sla, x)).toDenseListSV(), orc.spmv(Ao, x)
    bound = np.diff(Ao.rowptr) * EPS * orc.spmv(orc.Csr(m, n, Ao.rowptr, Ao.colidx, np.abs(Ao.val)), np.abs(x))
    assert np.all(np.abs(y - yo) <= bound + 1e-300)
    if shape == "short":                                  # one lane per row: the reference's exact fold
        assert np.array_equal(y, yo)


def test_matvec_stencil_bit_exact(sla):
    # short rows are folded by one lane in the reference's order with separate mul / add roundings: bit-exact.
    # (6.0 and the noisy band values make the products inexact, so an FMA anywhere would show.)
    from sla_amd import workloads as wl
    for dims, (rp, ci, va) in (wl.poisson2d(300, 200), wl.laplace3d(30, 20, 25), wl.banded_nonsym(70000)):
        A, Ao = sla.fromCSR(dims, rp, ci, va), orc.Csr(dims[0], dims[1], rp, ci, va)
        x = np.random.default_rng(3).standard_normal(dims[1])
        assert np.array_equal(sla.matVec(A, dense_vec(sla, x)).toDenseListSV(), orc.spmv(Ao, x)), A.kernel_info()


# ---- A2..A4 ------------------------------------------------------------------------------------------

def test_dot_norm_axpy(sla):
    assert sla.dot(sla.mkSpVR(2, [5, 6]), sla.mkSpVR(2, [5, 6])) == 61          # tv0 <.> tv0 (LibSpec.hs:45-46)
    rng = np.random.default_rng(11)
    for n in [1, 2, 3, 255, 256, 257, 100001]:
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        d = sla.dot(dense_vec(sla, x), dense_vec(sla, y))
        assert abs(d - orc.dot(x, y)) <= 4 * np.sqrt(n) * EPS * np.abs(x * y).sum()
        assert abs(sla.norm2(dense_vec(sla, x)) - orc.norm2(x)) <= 4 * np.sqrt(n) * EPS * orc.norm2(x)
    x = dense_vec(sla, rng.standard_normal(50))
    assert sla.norm2(x - x) == 0.0                                              # LibSpec.hs:43-44
    s = sla.fromListSV(5, [(1, 2.0), (7, 1.0), (1, 9.0)])                        # first duplicate wins, OOB dropped
    assert s.toListSV() == [(1, 2.0)]
    u = s + sla.fromListSV(5, [(3, 1.0)])
    assert u.toListSV() == [(1, 2.0), (3, 1.0)]


# ---- A5..A7: step parity ------------------------------------------------------------------------------

def _spd_problem(n, seed):
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.random_spd(n, k=3, seed=seed)
    x = np.random.default_rng(seed).standard_normal(n)
    return dims, rp, ci, va, x


@pytest.mark.parametrize("n", [5, 64, 2000])
def test_bicgstab_and_cgs_steps_vs_oracle(sla, n):
    dims, rp, ci, va, xs = _spd_problem(n, 10 + n)
    A, Ao = sla.fromCSR(dims, rp, ci, va), orc.Csr(n, n, rp, ci, va)
    b = orc.spmv(Ao, xs)
    x0 = np.full(n, 0.1)
    r0hat = b - orc.spmv(Ao, x0)
    for kind in ("bicgstab", "cgs"):
        so = orc.BicgstabState(Ao, b, x0) if kind == "bicgstab" else orc.CgsState(Ao, b, x0)
        sd = sla.bicgsInit(A, dense_vec(sla, b), dense_vec(sla, x0)) if kind == "bicgstab" else \
            sla.cgsInit(A, dense_vec(sla, b), dense_vec(sla, x0))
        r_dev = (sd._rBicgstab if kind == "bicgstab" else sd._r).toDenseListSV()
        assert np.allclose(r_dev, so.r, rtol=1e-13, atol=1e-13)
        for k in (1, 1, 3):
            so.step(r0hat, k)
            sd.step(k)
            xd = (sd._xBicgstab if kind == "bicgstab" else sd._x).toDenseListSV()
            pd = (sd._pBicgstab if kind == "bicgstab" else sd._p).toDenseListSV()
            assert np.linalg.norm(xd - so.x) <= 1e-9 * np.linalg.norm(so.x), (kind, k)
            # p collapses to rounding noise once the Krylov space is exhausted (n = 5): absolute floor
            assert np.linalg.norm(pd - so.p) <= 1e-7 * np.linalg.norm(so.p) + 1e-12 * np.linalg.norm(r0hat), (kind, k)


def test_cgne_steps_vs_oracle(sla):
    n = 300
    dims, rp, ci, va, xs = _spd_problem(n, 5)
    A, Ao = sla.fromCSR(dims, rp, ci, va), orc.Csr(n, n, rp, ci, va)
    b, x0 = orc.spmv(Ao, xs), np.full(n, 0.1)
    so, sd = orc.CgneState(Ao, b, x0), sla.cgneInit(A, dense_vec(sla, b), dense_vec(sla, x0))
    assert np.allclose(sd._pCgne.toDenseListSV(), so.p, rtol=1e-12, atol=1e-12)
    for k in (1, 2, 4):
        so.step(k); sd.step(k)
        assert np.linalg.norm(sd._xCgne.toDenseListSV() - so.x) <= 1e-9 * np.linalg.norm(so.x)
        assert np.linalg.norm(sd._rCgne.toDenseListSV() - so.r) <= 1e-8 * np.linalg.norm(so.r)


def test_init_state_equalities(sla):
    # specCGS / specBiCGSTAB (LibSpec.hs:240-245, 265-269): r = p = u = b - A x0
    A, Ao = both(sla, G["aa0"])
    b, x0 = np.array(G["aa0"]["b"], float), np.array(G["aa0"]["x0_state"])
    r0 = b - orc.spmv(Ao, x0)
    s = sla.cgsInit(A, dense_vec(sla, b), dense_vec(sla, x0))
    for f in (s._r, s._p, s._u):
        assert np.array_equal(f.toDenseListSV(), r0)
    s = sla.bicgsInit(A, dense_vec(sla, b), dense_vec(sla, x0))
    assert np.array_equal(s._rBicgstab.toDenseListSV(), r0) and np.array_equal(s._pBicgstab.toDenseListSV(), r0)
    assert np.array_equal(s._xBicgstab.toDenseListSV(), x0)


@pytest.mark.parametrize("kind", ["cgs", "bicgstab"])
def test_readme_iterate(sla, kind):
    # README.md:205-241; both methods hit x = [1.5,-2,1] after 3 steps
    R = G["readme"]
    A, _ = both(sla, R)
    b, x0 = sla.fromListDenseSV(3, R["b"]), sla.fromListSV(3, [])
    s = sla.cgsInit(A, b, x0) if kind == "cgs" else sla.bicgsInit(A, b, x0)
    s.step(3)
    x = (s._x if kind == "cgs" else s._xBicgstab).toDenseListSV()
    assert np.linalg.norm(x - np.array(R["x"])) <= 1e-12


# ---- A8: linSolve0 ---------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["aa0", "aa2"])
@pytest.mark.parametrize("method", ["BICGSTAB_", "CGS_", "CGNE_"])
def test_linsolve0_reference_cases(sla, name, method):
    # checkLinSolveR (LibSpec.hs:309-321): x0 = 0.1 * ones ; nearZero (norm2 (x ^-^ xhat))
    A, _ = both(sla, G[name])
    n = A.ncols
    b, xt = sla.mkSpVR(n, G[name]["b"]), sla.mkSpVR(n, G[name]["x_true"])
    xhat = sla.linSolve0(getattr(sla, method), A, b, sla.mkSpVR(n, [0.1] * n))
    assert sla.nearZero(sla.norm2(xt - xhat))


def test_linsolve0_readme_and_errors(sla):
    R = G["readme"]
    A, _ = both(sla, R)
    x = sla.linSolve0(sla.BICGSTAB_, A, sla.fromListDenseSV(3, R["b"]), sla.fromListSV(3, []))
    assert np.linalg.norm(x.toDenseListSV() - np.array(R["x"])) <= 1e-12
    with pytest.raises(sla.MatVecSizeMismatchException):          # Sparse.hs:1022
        sla.linSolve0(sla.BICGSTAB_, A, sla.onesSV(4), sla.onesSV(3))
    for meth in (sla.GMRES_, sla.BCG_):                            # Sparse.hs:1031
        with pytest.raises(sla.IterationException):
            sla.linSolve0(meth, A, sla.onesSV(3), sla.onesSV(3))
    D = sla.fromListSM((3, 3), [(0, 0, 2.0), (1, 1, 4.0), (2, 2, 5.0)])
    assert D.isDiagonalSM() and not A.isDiagonalSM()
    x, info = sla.linSolve0(sla.GMRES_, D, sla.onesSV(3), sla.onesSV(3), return_info=True)   # shortcut precedes the method check
    assert x.toDenseListSV().tolist() == [0.5, 0.25, 0.2] and info["iters"] == 0


@pytest.mark.parametrize("method", ["BICGSTAB_", "CGS_"])
def test_linsolve0_poisson_vs_oracle(sla, method):
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.poisson2d(40, 40)
    n = dims[0]
    A, Ao = sla.fromCSR(dims, rp, ci, va), orc.Csr(n, n, rp, ci, va)
    b = orc.spmv(Ao, np.ones(n))
    x, info = sla.linSolve0(getattr(sla, method), A, dense_vec(sla, b), dense_vec(sla, np.zeros(n)), return_info=True)
    rc, xo, it_o, res_o, r0_o = orc.linsolve0(getattr(orc, method), Ao, b, np.zeros(n))
    xd = x.toDenseListSV()
    assert info["converged"] and abs(info["r0norm"] - r0_o) <= 1e-12 * r0_o
    assert abs(info["tol"] - max(1e-6, 1e-4 * r0_o)) <= 1e-15
    assert np.linalg.norm(orc.spmv(Ao, xd) - b) <= info["tol"] * (1 + 1e-10)
    assert abs(info["resnorm"] - np.linalg.norm(orc.spmv(Ao, xd) - b)) <= 1e-9 * info["tol"]
    assert abs(info["iters"] - it_o) <= 3
    # check_every must not change the answer (the device tests every iteration)
    x1, info1 = sla.linSolve0(getattr(sla, method), A, dense_vec(sla, b), dense_vec(sla, np.zeros(n)),
                              return_info=True, check_every=1)
    assert info1["iters"] == info["iters"] and np.array_equal(x1.toDenseListSV(), xd)


def test_linsolve0_silent_return_at_max_iters(sla):
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.poisson2d(60, 60)
    n = dims[0]
    A = sla.fromCSR(dims, rp, ci, va)
    b = dense_vec(sla, np.random.default_rng(0).standard_normal(n))
    x, info = sla.linSolve0(sla.BICGSTAB_, A, b, sla.fromVector(np.zeros(n)), return_info=True, max_iters=5)
    assert info["iters"] == 5 and not info["converged"] and info["flags"] & 2   # no exception (Sparse.hs:1045)


def test_prop_spd_systems_converge(sla):
    # prop_bicgstab / prop_cgs (LibSpec.hs:969-1009): SPD = M^T M + 2 I, pass = residual <= tol within 100 its
    rng = np.random.default_rng(2024)
    for trial in range(12):
        n = int(rng.integers(3, 40))
        M = np.zeros((n, n))
        for (i, j) in rng.integers(0, n, size=(n, 2)):
            M[i, j] = rng.standard_normal()
        S = M.T @ M + 2.0 * np.eye(n)
        r, c = np.nonzero(S)
        A = sla.fromCOO((n, n), r, c, S[r, c])
        b = S @ rng.standard_normal(n)
        for meth in (sla.BICGSTAB_, sla.CGS_):
            x, info = sla.linSolve0(meth, A, dense_vec(sla, b), sla.fromVector(np.zeros(n)), return_info=True)
            assert info["iters"] <= 100 and np.linalg.norm(S @ x.toDenseListSV() - b) <= info["tol"] * (1 + 1e-9)


# ---- A9 / A10 ----------------------------------------------------------------------------------------------

def test_arnoldi_golden_and_oracle(sla):
    for entry, kn in ((G["arnoldi"]["aa4"], 3),):
        A, Ao = both(sla, entry)
        Q, H = sla.arnoldi(A, sla.onesSV(A.nrows), kn)
        D = A.toDense()
        assert np.linalg.norm(D @ Q[:, :-1] - Q @ H, "fro") <= 1e-12          # checkArnoldi (LibSpec.hs:642-653)
        rc, Qo, Ho, k = orc.arnoldi(Ao, np.ones(A.nrows), kn)
        assert Q.shape == Qo.shape and H.shape == Ho.shape
    t = G["arnoldi"]["tm7"]
    A, Ao = both(sla, tridiag_coo(t["n"], *t["tridiag"]))
    Q, H = sla.arnoldi(A, sla.onesSV(5), t["kn"])
    assert H.shape == (4, 3)                                                    # breakdown at step 3, like the oracle
    assert np.linalg.norm(A.toDense() @ Q[:, :-1] - Q @ H, "fro") <= 1e-12
    with pytest.raises(sla.MatVecSizeMismatchException):                        # Sparse.hs:637
        sla.arnoldi(A, sla.onesSV(4), 2)


def test_arnoldi_vs_oracle_medium(sla):
    from sla_amd import workloads as wl
    n, kn = 5000, 12
    dims, (rp, ci, va) = wl.banded_nonsym(n)
    A, Ao = sla.fromCSR(dims, rp, ci, va), orc.Csr(n, n, rp, ci, va)
    b = np.random.default_rng(5).standard_normal(n)
    Q, H = sla.arnoldi(A, dense_vec(sla, b), kn)
    rc, Qo, Ho, k = orc.arnoldi(Ao, b, kn)
    assert H.shape == Ho.shape == (kn + 1, kn)
    assert np.abs(H - Ho).max() <= 1e-10 * np.abs(Ho).max()
    assert np.abs(Q - Qo).max() <= 1e-9
    assert np.abs(Q.T @ Q - np.eye(kn + 1)).max() <= 1e-10


def test_gmres_and_backslash(sla):
    from sla_amd import workloads as wl
    R = G["readme"]
    A, _ = both(sla, R)
    x = sla.linSolve(A, sla.fromListDenseSV(3, R["b"]))                         # amat <\> b (README.md:183-189)
    assert np.linalg.norm(x.toDenseListSV() - np.array(R["x"])) <= 1e-10
    n = 4000
    dims, (rp, ci, va) = wl.banded_nonsym(n)
    A, Ao = sla.fromCSR(dims, rp, ci, va), orc.Csr(n, n, rp, ci, va)
    b = orc.spmv(Ao, np.ones(n))
    x, info = sla.gmres(A, dense_vec(sla, b), sla.fromVector(np.zeros(n)), restart=30, return_info=True)
    rc, xo, it_o, res_o, r0_o = orc.gmres(Ao, b, np.zeros(n), restart=30, max_restarts=10)
    assert info["converged"] and info["iters"] == it_o
    assert np.linalg.norm(x.toDenseListSV() - xo) <= 1e-9 * np.linalg.norm(xo)
    assert np.linalg.norm(orc.spmv(Ao, x.toDenseListSV()) - b) <= info["tol"] * (1 + 1e-9)


# ---- BASELINE full sizes: size-independent properties -----------------------------------------------------

def test_full_size_poisson_1m(sla):
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.poisson2d(1000, 1000)                               # config 2: nnz = 4 996 000
    n = dims[0]
    assert rp[-1] == 4996000
    A = sla.fromCSR(dims, rp, ci, va)
    rowsum = np.add.reduceat(va, rp[:-1])
    y = sla.matVec(A, sla.onesSV(n)).toDenseListSV()
    assert np.array_equal(y, rowsum)                                            # A 1 = row sums (small integers: exact)
    rng = np.random.default_rng(1234)
    u, v = rng.standard_normal(n), rng.standard_normal(n)
    yu, yv = sla.matVec(A, dense_vec(sla, u)).toDenseListSV(), sla.matVec(A, dense_vec(sla, v)).toDenseListSV()
    yuv = sla.matVec(A, dense_vec(sla, u + 2.0 * v)).toDenseListSV()
    assert np.abs(yuv - (yu + 2.0 * yv)).max() <= 64 * EPS * (np.abs(u).max() + 2 * np.abs(v).max()) * 8   # linearity
    assert abs(np.dot(v, yu) - np.dot(u, yv)) <= 1e-9 * abs(np.dot(v, yu))     # symmetry: v.(A u) = u.(A v)
    x, info = sla.linSolve0(sla.BICGSTAB_, A, dense_vec(sla, rowsum), sla.fromVector(np.zeros(n)), return_info=True)
    r = sla.matVec(A, x).toDenseListSV() - rowsum
    assert np.linalg.norm(r) <= max(info["tol"], 1e-6) * (1 + 1e-9) or info["iters"] == 200
    assert abs(info["resnorm"] - np.linalg.norm(r)) <= 1e-9 * max(info["resnorm"], 1e-30)


def test_full_size_laplace3d_10m(sla):
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.laplace3d(216, 216, 216)                            # config 4 on one GPU
    n = dims[0]
    assert n == 10077696
    A = sla.fromCSR(dims, rp, ci, va)
    rowsum = np.add.reduceat(va, rp[:-1])
    y = sla.matVec(A, sla.onesSV(n)).toDenseListSV()
    assert np.array_equal(y, rowsum)
    s = sla.bicgsInit(A, dense_vec(sla, rowsum), sla.fromVector(np.zeros(n)))
    s.step(3)
    x = s._xBicgstab
    r_true = rowsum - sla.matVec(A, x).toDenseListSV()
    assert np.linalg.norm(r_true - s._rBicgstab.toDenseListSV()) <= 1e-9 * np.linalg.norm(rowsum)   # recurrence = true residual


def test_row_partition_matches_python(sla):
    from sla_amd.partition import row_block
    ctx = sla.default_context()
    for m in (0, 1, 7, 1000, 10077696):
        assert ctx.row_range(m) == row_block(m, 0, 1)


# ---- row-sharded code path rehearsed on one GPU (1-rank RCCL communicator, forced collectives) -------------

def test_forced_collectives_match_single_gpu_path(sla, monkeypatch):
    """Per-rank folding + rank-order summation reproduce the 1-GPU reduction order BIT FOR BIT.  Both contexts run the
    reference's K4 / K5 split (option bicg_fuse45=0): the fused sweep evaluates rho through an identity and publishes four sums as one
    block on a sharded context -- that flow is test_bicgstab_fused_k45_flow_vs_reference_split's and the loopback tests' subject, not
    this test's."""
    from sla_amd import workloads as wl
    monkeypatch.setenv("SLA_FORCE_COLLECTIVES", "1")       # (read when the communicator is built: not a per-context option)
    ctx = sla.Context(0, 0, 1, sla.Context.unique_id()).set_option("bicg_fuse45", 0)
    monkeypatch.delenv("SLA_FORCE_COLLECTIVES")
    plain = sla.Context(0).set_options(bicg_fuse45=0, onchip=0, arn_orth=0)   # (the launch flows on both sides: the comparison is bit for bit)
    dims, (rp, ci, va) = wl.poisson2d(50, 40)
    n = dims[0]
    b = np.add.reduceat(va, rp[:-1])
    outs, cgne = [], []
    for c in (ctx, plain):
        A = sla.fromCSRRows(dims, 0, rp, ci, va, c)
        bv, x0 = sla.fromVector(b, c), sla.fromVector(np.zeros(n), c)
        x, info = sla.linSolve0(sla.BICGSTAB_, A, bv, x0, return_info=True)
        xc, infoc = sla.linSolve0(sla.CGS_, A, bv, x0, return_info=True)
        Q, H = sla.arnoldi(A, bv, 6)
        xg, infog = sla.gmres(A, bv, x0, restart=20, return_info=True)
        outs.append((x.toDenseListSV(), info["iters"], xc.toDenseListSV(), infoc["iters"], H, xg.toDenseListSV(),
                     sla.dot(bv, bv), sla.matVec(A, bv).toDenseListSV(), sla.vecMat(bv, A).toDenseListSV()))
        # CGNE, sharded: partial A^T r + reduce-scatter, unfused N3 (well-conditioned system so that it converges)
        d2, (rp2, ci2, va2) = wl.random_spd(600, 3, 9)
        A2 = sla.fromCSRRows(d2, 0, rp2, ci2, va2, c)
        b2 = sla.fromVector(np.add.reduceat(va2, rp2[:-1]), c)
        xn, infon = sla.linSolve0(sla.CGNE_, A2, b2, sla.fromVector(np.zeros(600), c), return_info=True)
        cgne.append((xn.toDenseListSV(), infon["iters"], infon["converged"]))
    for a, b_ in zip(*outs):
        assert np.array_equal(a, b_)          # per-rank folding + rank-order sum == the 1-GPU reduction order
    assert cgne[0][2] and cgne[1][2] and abs(cgne[0][1] - cgne[1][1]) <= 1
    assert np.linalg.norm(cgne[0][0] - cgne[1][0]) <= 1e-6 * np.linalg.norm(cgne[1][0])
    ctx.close()
    plain.close()


def test_rccl_point_to_point_calls_on_one_rank(sla):
    """ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd as the N > 1 exchanges call them (dlsym'd signatures, the f64 datatype
    constant, the context's stream), with the one rank of a 1-rank communicator as its own peer: what was sent must arrive, for one
    large piece, for many small ones and for a single element.  (A context without a communicator refuses.)"""
    ctx = sla.Context(0, 0, 1, sla.Context.unique_id())
    for count, pieces in ((1, 1), (4096, 1), (1000003, 7), (1 << 22, 64)):
        assert ctx.p2p_selftest(count, pieces) == 0.0, (count, pieces)
    ctx.close()
    with pytest.raises(sla.SlaError):
        sla.Context(0).p2p_selftest(16, 1)


@pytest.mark.parametrize("problem", ["poisson2d 50x40", "laplace3d 14x11x13", "spd 400", "banded_nonsym 4001 (wdia-vv)"])
def test_bicgstab_fused_k45_flow_vs_reference_split(sla, problem):
    """Single-rank BiCGSTAB fuses K4 and K5 (default): rho_{j+1} = s . r0hat - omega (As . r0hat) from K3's sweep instead of
    (s - omega As) . r0hat from K4's.  x, r, p are updated by the reference's formulas either way; rho differs at rounding
    level, which a Krylov iteration amplifies like any other rounding difference.  Bounds from a CPU experiment on these very
    problems (tools/rho_identity_experiment.py, profiles/r02_rho_identity_experiment.txt), which compares the identity with
    the reference's formula under mere regroupings of the dot-product sums: the iterates after 5 / 10 / 20 steps differ by
    1e-16 / 1e-15 / 1e-11 either way; iterations to convergence on poisson2d 50x40 are 61..67 for the reference's formula
    (oracle 63), 60..66 for the identity, and 23 / 6 for everything on the other two.  Hence: 5 steps agree to 1e-10; full
    solves both converge within 4 iterations of the oracle's count (the largest deviation among the 14 CPU variants) and
    within 6 of each other (the width of the regrouping-only range), both meet the residual tolerance; every state also
    against the oracle (which evaluates rho the reference's way)."""
    from sla_amd import workloads as wl
    if problem.startswith("poisson"):
        dims, (rp, ci, va) = wl.poisson2d(50, 40)
    elif problem.startswith("laplace"):
        dims, (rp, ci, va) = wl.laplace3d(14, 11, 13)
    elif problem.startswith("banded"):     # config 5's structure: the variable-coefficient wave-sliced kernel, whose four-sum
        dims, (rp, ci, va) = wl.banded_nonsym(4001)   # instantiation is compiled for one workgroup per CU less
    else:
        dims, rp, ci, va, _ = _spd_problem(400, 77)
    n = dims[0]
    Ao = orc.Csr(n, n, rp, ci, va)
    b = orc.spmv(Ao, np.linspace(-1.0, 2.0, n))
    x0 = np.full(n, 0.1)
    r0hat = b - orc.spmv(Ao, x0)
    got = {}
    for fuse in ("1", "0"):
        c = sla.Context(0).set_option("bicg_fuse45", fuse)
        A = sla.fromCSR(dims, rp, ci, va, c)
        if problem.startswith("banded"):
            assert "wdia-vv" in A.kernel_info(), A.kernel_info()
        sd = sla.bicgsInit(A, sla.fromVector(b, c), sla.fromVector(x0, c))
        sd.step(5)
        st5 = tuple(v.toDenseListSV() for v in (sd._xBicgstab, sd._rBicgstab, sd._pBicgstab))
        x, info = sla.linSolve0(sla.BICGSTAB_, A, sla.fromVector(b, c), sla.fromVector(x0, c), return_info=True)
        got[fuse] = (st5, x.toDenseListSV(), info)
        del sd, A
        c.close()
    so = orc.BicgstabState(Ao, b, x0)
    so.step(r0hat, 5)
    for a, b_, o in zip(got["1"][0], got["0"][0], (so.x, so.r, so.p)):
        assert np.linalg.norm(a - b_) <= 1e-10 * np.linalg.norm(b_), problem
        assert np.linalg.norm(a - o) <= 1e-9 * np.linalg.norm(o) and np.linalg.norm(b_ - o) <= 1e-9 * np.linalg.norm(o), problem
    i1, i0 = got["1"][2], got["0"][2]
    it_o = orc.linsolve0(orc.BICGSTAB_, Ao, b, x0)[2]
    assert i1["converged"] and i0["converged"], (problem, i1, i0)
    assert abs(i1["iters"] - it_o) <= 4 and abs(i0["iters"] - it_o) <= 4 and abs(i1["iters"] - i0["iters"]) <= 6, (problem, i1["iters"], i0["iters"], it_o)
    for k in ("1", "0"):
        assert np.linalg.norm(orc.spmv(Ao, got[k][1]) - b) <= got[k][2]["tol"] * (1 + 1e-9), (problem, k)


def test_matrix_market_ingestion_e05r0000(sla):
    # test/Perf.hs:20-45 loads test/data/e05r0000.mtx (+ rhs) and solves; BCG_ is no longer implemented in the
    # reference, so solve with what is: BICGSTAB_ may stall on this non-symmetric CFD matrix, GMRES must not
    A = sla.readMatrixMarket(f"{GOLDEN}/e05r0000.mtx")
    dims, r, c, v = read_mtx_coordinate(f"{GOLDEN}/e05r0000.mtx")
    rc, Ao = orc.coo_to_csr(dims[0], dims[1], r, c, v)
    rp, ci, va = A.csr()
    assert A.dims == (236, 236) and np.array_equal(rp, Ao.rowptr) and np.array_equal(ci, Ao.colidx) and np.array_equal(va, Ao.val)
    b = sla.readMatrixMarketArray(f"{GOLDEN}/e05r0000_rhs1.mtx")
    assert np.array_equal(b.toDenseListSV(), read_mtx_array(f"{GOLDEN}/e05r0000_rhs1.mtx"))
    x, info = sla.gmres(A, b, sla.fromVector(np.full(236, 0.1)), restart=60, return_info=True, max_iters=2000)
    rco, xo, it_o, res_o, r0_o = orc.gmres(Ao, b.toDenseListSV(), np.full(236, 0.1), restart=60, max_restarts=40)
    res = np.linalg.norm(orc.spmv(Ao, x.toDenseListSV()) - b.toDenseListSV())
    assert abs(info["r0norm"] - r0_o) <= 1e-10 * r0_o
    assert res <= max(info["resnorm"] * (1 + 1e-6), 1e-12) * 1.001 + 1e-12
    with pytest.raises(sla.SlaError):
        sla.readMatrixMarket(f"{GOLDEN}/does_not_exist.mtx")
    with pytest.raises(sla.SlaError):
        sla.readMatrixMarket(f"{GOLDEN}/e05r0000_rhs1.mtx")      # an array file is not a coordinate matrix


def test_jacobi_preconditioner_builders(sla):
    # jacobiPre (Sparse.hs:689-690) and the diagonal #~# that applies it: M = recip <$> extractDiag A
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.banded_nonsym(3000)
    n = dims[0]
    A, Ao = sla.fromCSR(dims, rp, ci, va), orc.Csr(n, n, rp, ci, va)
    M = sla.jacobiPre(A)
    assert M.isDiagonalSM()
    diag = np.array([va[rp[i]:rp[i + 1]][ci[rp[i]:rp[i + 1]] == i][0] for i in range(n)])
    assert np.array_equal(M.csr()[2], 1.0 / diag)
    MA = sla.diagMatMatSparsified(M, A)
    rc, Mo = orc.coo_to_csr(n, n, np.arange(n), np.arange(n), 1.0 / diag)
    rc, Co = orc.matmat(Mo, Ao)                                  # (##) in the oracle, then sparsify
    keep = np.abs(Co.val) > 1e-12
    rows_o = np.repeat(np.arange(n), np.diff(Co.rowptr))[keep]
    rpm, cim, vam = MA.csr()
    assert np.array_equal(cim, Co.colidx[keep]) and np.array_equal(vam, Co.val[keep])
    assert np.array_equal(np.repeat(np.arange(n), np.diff(rpm)), rows_o)
    # left-preconditioned solve converges at least as fast on this diagonally dominant system
    b = orc.spmv(Ao, np.ones(n))
    x, info = sla.linSolve0(sla.BICGSTAB_, A, sla.fromVector(b), sla.fromVector(np.zeros(n)), return_info=True)
    xp, infop = sla.linSolve0(sla.BICGSTAB_, MA, sla.matVec(M, sla.fromVector(b)), sla.fromVector(np.zeros(n)), return_info=True)
    assert info["converged"] and infop["converged"] and infop["iters"] <= info["iters"] + 1
    assert np.abs(xp.toDenseListSV() - 1.0).max() <= 5e-3                 # linSolve0 stops at 1e-4 * ||r0||
    S = sla.fromListSM((2, 2), [(0, 0, 2.0), (0, 1, 1.0), (1, 1, 4.0)])
    with pytest.raises(sla.SlaError):
        sla.diagMatMatSparsified(S, S)                            # left factor not diagonal


def test_pure_step_functions_do_not_alias_and_take_an_explicit_shadow_residual(sla):
    """bicgstabStep aa r0hat s / cgsStep aa rhat s (Sparse.hs:972-981, 928-939) are pure in the reference: the README's
    `iterate (bicgstabStep aa r0hat) s0 !! k` (README.md:222-226) needs every element to stay what it was.  Through
    sla_solver_clone the mirror's three-argument form returns a NEW record; an r0hat other than b - A x0 goes through
    sla_solver_set_shadow and matches the oracle stepping with that r0hat."""
    n = 400
    dims, rp, ci, va, xs = _spd_problem(n, 21)
    A, Ao = sla.fromCSR(dims, rp, ci, va), orc.Csr(n, n, rp, ci, va)
    b, x0 = orc.spmv(Ao, xs), np.full(n, 0.1)
    r0 = b - orc.spmv(Ao, x0)
    rng = np.random.default_rng(3)
    shadow = r0 + 0.05 * rng.standard_normal(n)                      # some other r0hat (not orthogonal to r0)
    for init, step, ostate, xf in ((sla.bicgsInit, sla.bicgstabStep, orc.BicgstabState, "_xBicgstab"), (sla.cgsInit, sla.cgsStep, orc.CgsState, "_x")):
        s0 = init(A, dense_vec(sla, b), dense_vec(sla, x0))
        x_before = getattr(s0, xf).toDenseListSV()
        chain = [s0]
        for _ in range(4):                                           # iterate (step aa r0hat) s0
            chain.append(step(A, None, chain[-1]))
        assert np.array_equal(getattr(s0, xf).toDenseListSV(), x_before)          # s0 is still s0
        so = ostate(Ao, b, x0)
        for k in range(1, 5):
            so.step(r0, 1)
            assert np.linalg.norm(getattr(chain[k], xf).toDenseListSV() - so.x) <= 1e-9 * np.linalg.norm(so.x), k
        assert len({id(c) for c in chain}) == 5 and len({c.h.value for c in chain}) == 5
        inplace = init(A, dense_vec(sla, b), dense_vec(sla, x0)).step(4)            # the in-place fast path agrees bit for bit
        assert np.array_equal(getattr(inplace, xf).toDenseListSV(), getattr(chain[4], xf).toDenseListSV())
        # explicit shadow residual
        s1 = step(A, dense_vec(sla, shadow), step(A, dense_vec(sla, shadow), s0))
        so = ostate(Ao, b, x0)
        so.step(shadow, 2)
        assert np.linalg.norm(getattr(s1, xf).toDenseListSV() - so.x) <= 1e-9 * np.linalg.norm(so.x)
        assert np.linalg.norm(getattr(s1, xf).toDenseListSV() - getattr(chain[2], xf).toDenseListSV()) > 1e-6 * np.linalg.norm(so.x)


@pytest.mark.parametrize("method", ["BICGSTAB_", "CGS_", "CGNE_"])
@pytest.mark.parametrize("check_every", [1, 5, 16])
def test_residual_trace_equals_the_oracles_sequence(sla, method, check_every):
    """sla_solve_opts.history ("per-iteration residual trace buffer on device"; cgsStepDebug, Sparse.hs:942-948,
    is the reference's own per-iteration residual output): the true residual norms runIter evaluates after every step, kept
    on the device and downloaded once.  Against the oracle's step-by-step sequence, for every host polling interval (the
    trace is written by whichever kernel tests the residual: the next step's prologue or the end-of-batch check)."""
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.poisson2d(40, 37)
    n = dims[0]
    A, Ao = sla.fromCSR(dims, rp, ci, va), orc.Csr(n, n, rp, ci, va)
    b = orc.spmv(Ao, np.linspace(-1.0, 2.0, n))
    x, info = sla.linSolve0(getattr(sla, method), A, sla.fromVector(b), sla.fromVector(np.zeros(n)), return_info=True, history=True,
                            check_every=check_every)
    hist = info["history"]
    assert len(hist) == info["iters"] > 3 and info["converged"] == (method != "CGNE_")   # (CGNE stagnates on this system: 200 silent iterations)
    st = {"BICGSTAB_": orc.BicgstabState, "CGS_": orc.CgsState, "CGNE_": orc.CgneState}[method](Ao, b, np.zeros(n))
    seq = []
    for _ in range(len(hist)):
        st.step(b) if method != "CGNE_" else st.step()
        seq.append(np.linalg.norm(orc.spmv(Ao, st.x) - b))
    seq = np.array(seq)
    # Measured on this system (round 3): the first iterations agree to 1e-15; CGS stays within 1e-12 over all 74 iterations, CGNE
    # within 1e-13 over 120; BiCGSTAB amplifies the last-bit differences of the regrouped inner products by ~10x every 4 steps
    # (1e-16 / 1e-10 / 1e-7 / 1e-4 at steps 5 / 20 / 30 / 40 -- with the reference's split flow exactly as with the fused sweep)
    # and still converges within two steps of the oracle.  Hence: 1e-8 over the first 20 steps for all, 1e-9 throughout for CGS.
    rel = np.abs(hist - seq) / seq
    assert rel[:20].max() <= 1e-8, rel[:20].max()
    if method == "CGS_":
        assert rel.max() <= 1e-9
    assert hist[-1] == info["resnorm"] and np.all(hist[:-1] > info["tol"])
    if method != "CGNE_":
        assert hist[-1] <= info["tol"]
        rc, xo, it_o, res_o, r0_o = orc.linsolve0(getattr(orc, method), Ao, b, np.zeros(n))
        assert abs(info["iters"] - it_o) <= 3
    # a short buffer takes the first entries only; without the option nothing is traced
    import ctypes as C
    from sla_amd import _lib
    buf = np.zeros(4)
    o = _lib.SolveOpts(200, 1e-6, 1e-4, check_every, 1, buf.ctypes.data, 4)
    out, inf2 = sla.DeviceVector(A.ctx, n), _lib.SolveInfo()
    bv, zv = sla.DeviceVector(A.ctx, n, b), sla.DeviceVector(A.ctx, n)
    _lib.check(_lib.lib().sla_linsolve0(int(getattr(sla, method)), A.h, bv.h, zv.h, C.byref(o), out.h, C.byref(inf2)))
    assert inf2.history_len == 4 and np.array_equal(buf, hist[:4]) and inf2.iters == info["iters"]
