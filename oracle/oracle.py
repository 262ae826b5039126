"""ctypes loader for the CPU oracle (oracle/sla_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from the product package (sla_amd).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

GMRES_, CGNE_, BCG_, CGS_, BICGSTAB_ = range(5)  # Sparse.hs:1007-1011 constructor order
OK, ERR_DIM, ERR_UNSUPPORTED, ERR_OOB = 0, 1, 2, 3

_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


class _Csr(C.Structure):
    _fields_ = [("m", C.c_int64), ("n", C.c_int64), ("rowptr", C.c_void_p),
                ("colidx", C.c_void_p), ("val", C.c_void_p)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "sla_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_dot.restype = C.c_double
        L.orc_norm2.restype = C.c_double
        L.orc_norm2sq.restype = C.c_double
        _LIB = L
    return _LIB


_LIB_SERIAL = None


def use_omp(flag):
    """Switch to / from liboracle_omp.so, the OpenMP "fair CPU" TIMING build (row-parallel SpMV, parallel
    reductions: not the reference's summation order, never used for parity checks)."""
    global _LIB, _LIB_SERIAL
    if _LIB_SERIAL is None:
        _LIB_SERIAL = lib()
    if not flag:
        _LIB = _LIB_SERIAL
        return _LIB
    so = os.path.join(_HERE, "liboracle_omp.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle_omp.so"])
    L = C.CDLL(so)
    L.orc_dot.restype = C.c_double
    L.orc_norm2.restype = C.c_double
    L.orc_norm2sq.restype = C.c_double
    _LIB = L
    return _LIB


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return C.c_void_p(a.ctypes.data)


class Csr:
    """Host CSR (int64 indices, f64 values) in the reference's canonical layout."""

    def __init__(self, m, n, rowptr, colidx, val):
        self.m, self.n = int(m), int(n)
        self.rowptr, self.colidx, self.val = _i64(rowptr), _i64(colidx), _f64(val)

    @property
    def nnz(self):
        return int(self.rowptr[-1])

    def view(self):
        return _Csr(self.m, self.n, self.rowptr.ctypes.data, self.colidx.ctypes.data,
                    self.val.ctypes.data)


def coo_to_csr(m, n, row, col, val):
    """fromListSM + toCSR layout.  Returns (rc, Csr|None)."""
    row, col, val = _i64(row), _i64(col), _f64(val)
    nnz = len(row)
    rowptr = np.zeros(m + 1, dtype=np.int64)
    colidx = np.zeros(max(nnz, 1), dtype=np.int64)
    vout = np.zeros(max(nnz, 1), dtype=np.float64)
    nout = C.c_int64(0)
    rc = lib().orc_coo_to_csr(C.c_int64(m), C.c_int64(n), C.c_int64(nnz), _p(row), _p(col), _p(val),
                              _p(rowptr), _p(colidx), _p(vout), C.byref(nout))
    if rc != OK:
        return rc, None
    k = nout.value
    return rc, Csr(m, n, rowptr, colidx[:k].copy(), vout[:k].copy())


def cs_ptr(n, sorted_ix):
    ix = _i64(sorted_ix)
    out = np.zeros(n + 1, dtype=np.int64)
    lib().orc_cs_ptr(C.c_int64(n), _p(ix), C.c_int64(len(ix)), _p(out))
    return out


def transpose(A):
    tp = np.zeros(A.n + 1, dtype=np.int64)
    tc = np.zeros(max(A.nnz, 1), dtype=np.int64)
    tv = np.zeros(max(A.nnz, 1), dtype=np.float64)
    lib().orc_csr_transpose(C.c_int64(A.m), C.c_int64(A.n), _p(A.rowptr), _p(A.colidx), _p(A.val),
                            _p(tp), _p(tc), _p(tv))
    return Csr(A.n, A.m, tp, tc[:A.nnz].copy(), tv[:A.nnz].copy())


def to_csc(m, n, row, col, val):
    """toCSC m n ijxv (vector/src/Data/Sparse/Internal/CSC.hs:51-55): STABLE sort of the triplets by column, unzip, column pointer by
    csPtrV.  (The reference passes m to csPtrV, :55 -- right for square matrices only; restated with n, the length a column pointer has.)
    Returns (colptr, rowidx, val).  Index work in numpy (a stable argsort is the reference's merge sort on one key)."""
    row, col, val = _i64(row), _i64(col), _f64(val)
    order = np.argsort(col, kind="stable")
    return cs_ptr(n, col[order]), row[order].copy(), val[order].copy()


def csc_to_csr(m, n, colptr, rowidx, val):
    """CscMatrix arrays -> the canonical CSR of the same matrix: fromCSC0 (CSC.hs:61-77) gives the triplets, toCSR (CSR.hs:74-78) sorts them by
    row; for canonical CSC input (rows ascending inside a column) that is the transpose of the n x m matrix the arrays are the CSR of."""
    return transpose(Csr(n, m, colptr, rowidx, val))


def csb_block_index(dims, beta, i, j):
    """blockIx (vector/src/Data/Sparse/Internal/CSB.hs:88-92): bx + by * nbx with (bx, by) = (i div beta, j div beta) and nbx = ceiling (m / beta)
    (csbParams :72-77) -- the block ROW index runs fastest."""
    nbx = -(-int(dims[0]) // int(beta))
    return _i64(i) // beta + (_i64(j) // beta) * nbx


def to_csb(dims, beta, row, col, val):
    """The CsbMatrix arrays (CSB.hs:62-66: csbVal, csbBlkPtr, csbRowIx, csbColIx) of a triplet list: consBlocks (:103-107) bins the triplets by
    blockIx -- each new element is consed in FRONT of its block's list, so a block holds its elements in reverse input order -- and the blocks are laid
    out in ascending block index with bCoords (:84-85) as block-relative indices.  Returns (blkptr, rowix, colix, val)."""
    row, col, val = _i64(row), _i64(col), _f64(val)
    m, n = int(dims[0]), int(dims[1])
    nblk = (-(-m // beta)) * (-(-n // beta))
    ib = csb_block_index(dims, beta, row, col)
    order = np.argsort(ib[::-1], kind="stable")              # reverse input order inside a block, blocks ascending
    order = (len(row) - 1 - order) if len(row) else order
    blkptr = np.concatenate(([0], np.cumsum(np.bincount(ib, minlength=nblk)))).astype(np.int64)
    return blkptr, (row % beta)[order].copy(), (col % beta)[order].copy(), val[order].copy()


def csb_to_coo(dims, beta, blkptr, rowix, colix, val):
    """The triplets of CsbMatrix arrays in storage order (the definition of the layout, CSB.hs:44-56): element k of block f = bx + by * nbx is
    a (bx * beta + rowix[k], by * beta + colix[k])."""
    blkptr, rowix, colix = _i64(blkptr), _i64(rowix), _i64(colix)
    nbx = -(-int(dims[0]) // int(beta))
    f = np.repeat(np.arange(len(blkptr) - 1, dtype=np.int64), np.diff(blkptr))
    return (f % nbx) * beta + rowix, (f // nbx) * beta + colix, _f64(val)


def is_diagonal(A):
    return bool(lib().orc_is_diagonal(C.c_int64(A.m), _p(A.rowptr), _p(A.colidx)))


def numa_copy(A):
    """A copy of A whose pages are first touched row-parallel by the CURRENT library's thread team (use_omp(True) first):
    the OpenMP timing leg then reads every row block from the NUMA node of the thread that owns it."""
    rp, ci, va = np.empty_like(A.rowptr), np.empty_like(A.colidx), np.empty_like(A.val)
    lib().orc_par_copy_csr(C.c_int64(A.m), _p(A.rowptr), _p(A.colidx), _p(A.val), _p(rp), _p(ci), _p(va))
    return Csr(A.m, A.n, rp, ci, va)


def numa_vec(v):
    v = _f64(v)
    out = np.empty_like(v)
    lib().orc_par_copy_f64(C.c_int64(len(v)), _p(v), _p(out))
    return out


def spmv(A, x):
    x = _f64(x)
    y = np.zeros(A.m, dtype=np.float64)
    lib().orc_spmv(C.c_int64(A.m), _p(A.rowptr), _p(A.colidx), _p(A.val), _p(x), _p(y))
    return y


def spmv_panel_order(A, x, shift, visit):
    """y = A x with every row folded over the column panels of 2^shift columns in the order `visit` (a permutation of the panel ids),
    ascending columns inside a panel: the fold of the product's overlapped all-gather (NOT the reference's unless visit is ascending)."""
    x = _f64(x)
    visit = np.ascontiguousarray(visit, dtype=np.int32)
    pos = np.empty(len(visit), dtype=np.int32)
    pos[visit] = np.arange(len(visit), dtype=np.int32)
    y = np.zeros(A.m, dtype=np.float64)
    lib().orc_spmv_panel_order(C.c_int64(A.m), _p(A.rowptr), _p(A.colidx), _p(A.val), _p(x), _p(y), C.c_int(shift), C.c_int64(len(pos)),
                               C.c_void_p(pos.ctypes.data))
    return y


def dot(x, y):
    x, y = _f64(x), _f64(y)
    return lib().orc_dot(C.c_int64(len(x)), _p(x), _p(y))


def norm2(x):
    x = _f64(x)
    return lib().orc_norm2(C.c_int64(len(x)), _p(x))


def norm2sq(x):
    x = _f64(x)
    return lib().orc_norm2sq(C.c_int64(len(x)), _p(x))


class BicgstabState:
    def __init__(self, A, b, x0):
        self.A, n = A, A.m
        self.x, self.r, self.p = (np.zeros(n) for _ in range(3))
        v = A.view()
        lib().orc_bicgstab_init(C.byref(v), _p(_f64(b)), _p(_f64(x0)), _p(self.x), _p(self.r), _p(self.p))

    def step(self, r0hat, k=1, rho_identity=False):
        """rho_identity=True: beta's numerator as (s . r0hat) - omega (aas . r0hat), the product's fused-sweep formula (NOT the reference's)."""
        v = self.A.view()
        r0hat = _f64(r0hat)
        for _ in range(k):
            lib().orc_bicgstab_step_ex(C.byref(v), _p(r0hat), _p(self.x), _p(self.r), _p(self.p), C.c_int(1 if rho_identity else 0))
        return self


class CgsState:
    def __init__(self, A, b, x0):
        self.A, n = A, A.m
        self.x, self.r, self.p, self.u = (np.zeros(n) for _ in range(4))
        v = A.view()
        lib().orc_cgs_init(C.byref(v), _p(_f64(b)), _p(_f64(x0)), _p(self.x), _p(self.r), _p(self.p), _p(self.u))

    def step(self, rhat, k=1):
        v = self.A.view()
        rhat = _f64(rhat)
        for _ in range(k):
            lib().orc_cgs_step(C.byref(v), _p(rhat), _p(self.x), _p(self.r), _p(self.p), _p(self.u))
        return self


class CgneState:
    def __init__(self, A, b, x0):
        self.A, self.At = A, transpose(A)
        self.x, self.r, self.p = np.zeros(A.n), np.zeros(A.m), np.zeros(A.n)
        v, vt = A.view(), self.At.view()
        lib().orc_cgne_init(C.byref(v), C.byref(vt), _p(_f64(b)), _p(_f64(x0)), _p(self.x), _p(self.r), _p(self.p))

    def step(self, k=1):
        v, vt = self.A.view(), self.At.view()
        for _ in range(k):
            lib().orc_cgne_step(C.byref(v), C.byref(vt), _p(self.x), _p(self.r), _p(self.p))
        return self


class BcgState:
    """bcgInit / bcgStep: the COMMENTED code of Sparse.hs:886-909 (extension; parity unpinned by the reference -- dead code there)."""

    def __init__(self, A, b, x0):
        self.A, self.At = A, transpose(A)
        n = A.m
        self.x, self.r, self.rhat, self.p, self.phat = (np.zeros(n) for _ in range(5))
        v = A.view()
        lib().orc_bcg_init(C.byref(v), _p(_f64(b)), _p(_f64(x0)), _p(self.x), _p(self.r), _p(self.rhat), _p(self.p), _p(self.phat))

    def step(self, k=1):
        v, vt = self.A.view(), self.At.view()
        for _ in range(k):
            lib().orc_bcg_step(C.byref(v), C.byref(vt), _p(self.x), _p(self.r), _p(self.rhat), _p(self.p), _p(self.phat))
        return self


def linsolve0(method, A, b, x0):
    """Returns (rc, x, iters, resnorm, r0norm)."""
    b, x0 = _f64(b), _f64(x0)
    x = np.zeros(max(A.n, A.m), dtype=np.float64)
    it, res, r0 = C.c_int64(0), C.c_double(0), C.c_double(0)
    v = A.view()
    rc = lib().orc_linsolve0(C.c_int(method), C.byref(v), C.c_int64(len(b)), _p(b), _p(x0), _p(x),
                             C.byref(it), C.byref(res), C.byref(r0))
    return rc, x[:A.n], it.value, res.value, r0.value


def arnoldi(A, b, kn):
    """Returns (rc, Q[n, k+1], H[k+1, k], k) trimmed to the steps actually taken."""
    b = _f64(b)
    n = A.n
    Q = np.zeros((kn + 1) * n, dtype=np.float64)
    H = np.zeros((kn + 1) * kn, dtype=np.float64)
    kd = C.c_int64(0)
    v = A.view()
    rc = lib().orc_arnoldi(C.byref(v), C.c_int64(len(b)), _p(b), C.c_int64(kn), _p(Q), _p(H), C.byref(kd))
    if rc != OK:
        return rc, None, None, 0
    k = kd.value
    Qm = Q.reshape(kn + 1, n).T[:, :k + 1].copy()
    Hm = H.reshape(kn, kn + 1).T[:k + 1, :k].copy()
    return rc, Qm, Hm, k


def qr(a, stored=None):
    """qr (Sparse.hs:306-331) of a dense m x n array (m >= n); stored: boolean mask of the IntMap's keys (None: the non-zeros).
    Returns (rc, Q, R) with Q = transpose of the accumulated rotations, like the reference."""
    a = np.asarray(a, dtype=np.float64)
    m, n = a.shape
    af = np.asfortranarray(a)
    qt = np.zeros((m, m), dtype=np.float64, order="F")
    r = np.zeros((m, n), dtype=np.float64, order="F")
    st = None if stored is None else np.asfortranarray(np.asarray(stored, dtype=np.int8))
    lib().orc_qr_dense.restype = C.c_int
    rc = lib().orc_qr_dense(C.c_int64(m), C.c_int64(n), C.c_void_p(af.ctypes.data), None if st is None else C.c_void_p(st.ctypes.data),
                            C.c_void_p(qt.ctypes.data), C.c_void_p(r.ctypes.data))
    return rc, np.ascontiguousarray(qt.T), np.ascontiguousarray(r)


def gmres(A, b, x0, restart=30, max_restarts=10, tol_abs=1e-6, tol_rel=1e-4):
    b, x0 = _f64(b), _f64(x0)
    x = np.zeros(A.n, dtype=np.float64)
    it, res, r0 = C.c_int64(0), C.c_double(0), C.c_double(0)
    v = A.view()
    rc = lib().orc_gmres(C.byref(v), C.c_int64(len(b)), _p(b), _p(x0), C.c_int64(restart),
                         C.c_int64(max_restarts), C.c_double(tol_abs), C.c_double(tol_rel), _p(x),
                         C.byref(it), C.byref(res), C.byref(r0))
    return rc, x, it.value, res.value, r0.value


def matmat(A, B):
    """(##): returns (rc, Csr) structurally dense over rows(A) x cols(B)."""
    cap = max(A.m * B.n, 1)
    cp = np.zeros(A.m + 1, dtype=np.int64)
    cc = np.zeros(cap, dtype=np.int64)
    cv = np.zeros(cap, dtype=np.float64)
    nn = C.c_int64(0)
    va, vb = A.view(), B.view()
    rc = lib().orc_matmat(C.byref(va), C.byref(vb), _p(cp), _p(cc), _p(cv), C.c_int64(cap), C.byref(nn))
    if rc != OK:
        return rc, None
    k = nn.value
    return rc, Csr(A.m, B.n, cp, cc[:k].copy(), cv[:k].copy())


ERR_PIVOT = 5


def _tri(fn, T, b):
    x = np.zeros(T.m, dtype=np.float64)
    bad = C.c_int64(-1)
    v = T.view()
    rc = fn(C.byref(v), _p(_f64(b)), _p(x), C.byref(bad))
    return rc, x, bad.value


def tri_lower_solve(T, b):
    """triLowerSolve (Sparse.hs:750-776): (rc, x, bad_row); rc == ERR_PIVOT when l_ii is missing / near zero."""
    return _tri(lib().orc_tri_lower_solve, T, b)


def tri_upper_solve(T, b):
    """triUpperSolve (Sparse.hs:784-811)."""
    return _tri(lib().orc_tri_upper_solve, T, b)


def ssor_pre(A, omega):
    """mSsorPre (Sparse.hs:712-720): (rc, L, R) with L = (I - omega E) ## reciprocal D, R = D - omega F."""
    cap = int(A.rowptr[-1]) + A.m + 1
    lp, rp = np.zeros(A.m + 1, dtype=np.int64), np.zeros(A.m + 1, dtype=np.int64)
    lc, rc_ = np.zeros(cap, dtype=np.int64), np.zeros(cap, dtype=np.int64)
    lv, rv = np.zeros(cap), np.zeros(cap)
    v = A.view()
    rc = lib().orc_ssor_pre(C.byref(v), C.c_double(omega), _p(lp), _p(lc), _p(lv), _p(rp), _p(rc_), _p(rv))
    if rc != OK:
        return rc, None, None
    return rc, Csr(A.m, A.n, lp, lc[:lp[-1]].copy(), lv[:lp[-1]].copy()), Csr(A.m, A.n, rp, rc_[:rp[-1]].copy(), rv[:rp[-1]].copy())


def lu(A, ilu0=False):
    """lu (Sparse.hs:488-527) / ilu0Pre (:696-706): (rc, L, U, bad_row); rc == ERR_PIVOT when a pivot u_jj fails isNz."""
    n = A.m
    cap = max(n * n, 1)
    lp, up = np.zeros(n + 1, dtype=np.int64), np.zeros(n + 1, dtype=np.int64)
    lc, uc = np.zeros(cap, dtype=np.int64), np.zeros(cap, dtype=np.int64)
    lv, uv = np.zeros(cap), np.zeros(cap)
    bad = C.c_int64(-1)
    v = A.view()
    rc = lib().orc_lu(C.byref(v), C.c_int(1 if ilu0 else 0), _p(lp), _p(lc), _p(lv), _p(up), _p(uc), _p(uv), C.byref(bad))
    if rc != OK:
        return rc, None, None, bad.value
    return rc, Csr(n, n, lp, lc[:lp[-1]].copy(), lv[:lp[-1]].copy()), Csr(n, n, up, uc[:up[-1]].copy(), uv[:up[-1]].copy()), -1


def ilu0_pre(A):
    return lu(A, ilu0=True)
