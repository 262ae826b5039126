"""Host-side mirror of the reference's Numeric.LinearAlgebra.Sparse surface for the hot path, on top of
the C ABI (include/sla_hip.h).  Names follow the reference so that tests read like test/LibSpec.hs:

    fromListSM, fromListDenseSM, fromListSV, fromListDenseSV, mkSpVR, onesSV, zeroSV,
    matVec (#>), vecMat (<#), dot (<.>), norm2, normalize2, SpVector + / - / scalar *  (^+^ ^-^ .*),
    matMat (##), transpose, linSolve0 + LinSolveMethod, cgsInit/cgsStep, bicgsInit/bicgstabStep,
    cgneInit/cgneStep, arnoldi, linSolve (<\\>), gmres.

A SpMatrix is lowered ONCE to the device CSR at construction (sla_csr_from_coo does the sort / dedupe);
SpVectors keep the reference's structural sparsity on the host (sorted index + value arrays) and are
dense on the device.  Reference file:line citations are in include/sla_hip.h and on each function.
"""
import ctypes as C
import enum

import numpy as np

from . import _lib
from ._lib import (IndexOutOfBounds, IterationException, MatVecSizeMismatchException, SlaError,
                   SolveInfo, SolveOpts, check, lib)

_p = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731


class LinSolveMethod(enum.IntEnum):
    """Sparse.hs:1007-1011 (constructor order)."""
    GMRES_ = 0
    CGNE_ = 1
    BCG_ = 2
    CGS_ = 3
    BICGSTAB_ = 4


GMRES_, CGNE_, BCG_, CGS_, BICGSTAB_ = LinSolveMethod


class Context:
    """One GPU (one rank of a row-sharded job)."""

    def __init__(self, device_id=0, rank=0, nranks=1, unique_id=None):
        self.h = C.c_void_p()
        if nranks == 1 and unique_id is None:
            check(lib().sla_ctx_create(device_id, C.byref(self.h)))
        else:
            buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
            check(lib().sla_ctx_create_dist(device_id, rank, nranks, C.cast(buf, C.c_void_p), C.byref(self.h)))
        self.rank, self.nranks = rank, nranks

    @classmethod
    def loopback(cls, rank, nranks, group_key, device_id=0):
        """TEST BACKEND: one rank (= one host thread) of an in-process group on a single GPU."""
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        check(lib().sla_ctx_create_loopback(device_id, rank, nranks, group_key, C.byref(self.h)))
        self.rank, self.nranks = rank, nranks
        return self

    @classmethod
    def multi(cls, device_ids):
        """ONE caller driving len(device_ids) GPUs (sla_ctx_create_multi): matrices / vectors are given and returned whole, the
        library fans every call out to one rank per device.  A repeated device id selects the loopback test backend."""
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        ids = (C.c_int * len(device_ids))(*device_ids)
        check(lib().sla_ctx_create_multi(len(device_ids), C.cast(ids, C.c_void_p), C.byref(self.h)))
        self.rank, self.nranks = 0, 1          # the caller's view: it owns every row
        return self

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        check(lib().sla_dist_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    def row_range(self, m):
        b, e = C.c_int64(), C.c_int64()
        check(lib().sla_ctx_row_range(self.h, m, C.byref(b), C.byref(e)))
        return b.value, e.value

    def sync(self):
        check(lib().sla_ctx_sync(self.h))

    def set_option(self, name, value):
        """Typed entry for the tuning / A-B knobs (sla_ctx_set_option): name = the knob's lower-case name ("wdia", "tile_shift",
        "x_exchange", ...).  Lowering knobs apply to matrices created afterwards.  Returns self (chainable)."""
        check(lib().sla_ctx_set_option(self.h, str(name).encode(), str(value).encode()))
        return self

    def set_options(self, **kw):
        for k, v in kw.items():
            self.set_option(k, v)
        return self

    def p2p_selftest(self, count, pieces=1):
        """Grouped ncclRecv / ncclSend with this rank as its own peer (sla_dist_p2p_selftest): max |sent - arrived| over `count` doubles."""
        err = C.c_double(-1.0)
        check(lib().sla_dist_p2p_selftest(self.h, int(count), int(pieces), C.byref(err)))
        return err.value

    def preflight(self, phase, count=4096):
        """One checked collective across this context's ranks (sla_dist_preflight; every rank calls it): phase 0 ncclAllGather,
        1 the all-gather as one ncclSend / ncclRecv group, 2 the integer max all-reduce.  Returns (max |arrived - sent|, ms)."""
        err, ms = C.c_double(-1.0), C.c_double(0.0)
        check(lib().sla_dist_preflight(self.h, int(phase), int(count), C.byref(err), C.byref(ms)))
        return err.value, ms.value

    def stream_probe(self, reads, writes, n, reps=20):
        """(mean ms, min ms, GB/s at the mean) of a sweep reading `reads` and writing `writes` vectors of n doubles (sla_stream_probe)."""
        mean, mn = C.c_double(), C.c_double()
        check(lib().sla_stream_probe(self.h, reads, writes, int(n), reps, C.byref(mean), C.byref(mn)))
        return mean.value, mn.value, 8.0 * (reads + writes) * (int(n) & ~1) / (mean.value * 1e-3) / 1e9

    def get_option(self, name):
        buf = C.create_string_buffer(512)
        check(lib().sla_ctx_get_option(self.h, str(name).encode(), buf, 512))
        return buf.value.decode()

    @staticmethod
    def binding_violations():
        """SLA_DEBUG_BINDING=1: device work issued by a thread not bound to the context it belongs to (0 in a correct library)."""
        return int(lib().sla_debug_binding_violations())

    def prof_start(self, kernel_id, max_launches):
        check(lib().sla_prof_start(self.h, kernel_id, max_launches))

    def prof_stop(self):
        n, mean, mn = C.c_int(), C.c_double(), C.c_double()
        check(lib().sla_prof_stop(self.h, C.byref(n), C.byref(mean), C.byref(mn)))
        return n.value, mean.value, mn.value

    def prof_query(self, kernel_id):
        """(launches, mean ms, min ms) of one kernel id of the last recording (after prof_stop)."""
        n, mean, mn = C.c_int(), C.c_double(), C.c_double()
        check(lib().sla_prof_query(self.h, kernel_id, C.byref(n), C.byref(mean), C.byref(mn)))
        return n.value, mean.value, mn.value

    def comm_ranks(self):
        """Ranks the communicator behind this context spans (ncclCommCount for RCCL)."""
        n = C.c_int()
        check(lib().sla_ctx_comm_ranks(self.h, C.byref(n)))
        return n.value

    @staticmethod
    def device_count():
        n = C.c_int()
        check(lib().sla_device_count(C.byref(n)))
        return n.value

    def close(self):
        if self.h:
            lib().sla_ctx_destroy(self.h)
            self.h = C.c_void_p()


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def set_default_context(ctx):
    global _default_ctx
    _default_ctx = ctx


# ---- device handles ------------------------------------------------------------------------------------

class DeviceVector:
    """sla_vec_t owner."""

    def __init__(self, ctx, n, host=None, local=False):
        self.ctx, self.n = ctx, int(n)
        self.h = C.c_void_p()
        if host is None:
            check(lib().sla_vec_create(ctx.h, self.n, None, C.byref(self.h)))
        else:
            a = np.ascontiguousarray(host, dtype=np.float64)
            f = lib().sla_vec_create_local if local else lib().sla_vec_create
            check(f(ctx.h, self.n, _p(a), C.byref(self.h)))

    def to_host(self):
        out = np.empty(self.n, dtype=np.float64)
        check(lib().sla_vec_to_host(self.h, _p(out)))
        return out

    def to_host_local(self):
        nl = C.c_int64()
        check(lib().sla_vec_dim(self.h, None, C.byref(nl)))
        out = np.empty(nl.value, dtype=np.float64)
        check(lib().sla_vec_to_host_local(self.h, _p(out)))
        return out

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                lib().sla_vec_destroy(self.h)
        except Exception:
            pass


# ---- SpVector -------------------------------------------------------------------------------------------

class SpVector:
    """SV dim (IntM a) (SpVector.hs:42-43): sorted keys `ix`, values `vals`."""

    def __init__(self, dim, ix, vals, ctx=None):
        self.dim = int(dim)
        self.ix = np.ascontiguousarray(ix, dtype=np.int64)
        self.vals = np.ascontiguousarray(vals, dtype=np.float64)
        self._ctx = ctx                          # resolved lazily: host algebra needs no GPU
        self._dev = None

    @property
    def ctx(self):
        return self._ctx or default_context()

    # -- structure
    def nnz(self):
        return len(self.ix)

    def toDenseListSV(self):                     # SpVector.hs:300
        if len(self.ix) == self.dim:
            return self.vals.copy()
        d = np.zeros(self.dim)
        d[self.ix] = self.vals
        return d

    toVectorDense = toDenseListSV                # SpVector.hs:250

    def toListSV(self):                          # SpVector.hs:294-295
        return list(zip(self.ix.tolist(), self.vals.tolist()))

    def _dense_view(self):
        """Dense values without a scatter when every index is present (the solver inputs / outputs)."""
        if len(self.ix) == self.dim:
            return self.vals
        return self.toDenseListSV()

    def device(self):
        if self._dev is None:
            self._dev = DeviceVector(self.ctx, self.dim, self._dense_view())
        return self._dev

    # -- Eq / Show are structural in the reference
    def __eq__(self, o):
        return (isinstance(o, SpVector) and self.dim == o.dim and np.array_equal(self.ix, o.ix)
                and np.array_equal(self.vals, o.vals))

    def __repr__(self):
        return f"SV ({self.dim}) {self.toListSV()}"

    # -- AdditiveGroup / VectorSpace (SpVector.hs:107-114): unionWith (+), fmap
    def _union(self, o, sign):
        dim = max(self.dim, o.dim)               # liftU2 takes max of the dims (SpVector.hs:63)
        keys = np.union1d(self.ix, o.ix)
        a = np.zeros(len(keys)); b = np.zeros(len(keys))
        ia, ib = np.searchsorted(keys, self.ix), np.searchsorted(keys, o.ix)
        ma = np.zeros(len(keys), bool); mb = np.zeros(len(keys), bool)
        a[ia] = self.vals; ma[ia] = True
        b[ib] = sign * o.vals; mb[ib] = True     # x ^-^ y = x ^+^ negateV y (Class.hs:68-69)
        v = np.where(ma & mb, a + b, np.where(ma, a, b))
        return SpVector(dim, keys, v, self._ctx)

    def __add__(self, o):
        return self._union(o, 1.0)

    def __sub__(self, o):
        return self._union(o, -1.0)

    def __neg__(self):
        return SpVector(self.dim, self.ix, -self.vals, self._ctx)

    def __rmul__(self, a):                       # a .* v
        return SpVector(self.dim, self.ix, float(a) * self.vals, self._ctx)

    def __mul__(self, a):                        # v *. a
        return self.__rmul__(a)


def _dense_spvector(dim, arr, ctx=None):
    return SpVector(dim, np.arange(dim, dtype=np.int64), np.asarray(arr, dtype=np.float64)[:dim], ctx)


def fromListSV(d, iix, ctx=None):
    """fromListSV d iix (SpVector.hs:275-278): foldr insert => the FIRST duplicate wins; out-of-bounds
    entries are silently dropped."""
    seen = {}
    for i, x in iix:
        i = int(i)
        if 0 <= i < d and i not in seen:
            seen[i] = float(x)
    keys = sorted(seen)
    return SpVector(d, keys, [seen[k] for k in keys], ctx)


def fromListDenseSV(d, ll, ctx=None):
    """fromListDenseSV d ll (SpVector.hs:194-195)."""
    ll = list(ll)[:d]
    return SpVector(d, np.arange(len(ll)), ll, ctx)


def mkSpVR(d, ll, ctx=None):
    """mkSpVR d ll = SV d (mkIm ll) (SpVector.hs:183, IntM.hs:114-115): keys 0..len-1, zeros kept."""
    ll = list(ll)
    return SpVector(d, np.arange(len(ll)), ll, ctx)


def fromVector(arr, ctx=None):
    """fromVector (SpVector.hs:240-243): every entry becomes a key."""
    arr = np.asarray(arr, dtype=np.float64)
    return _dense_spvector(len(arr), arr, ctx)


def onesSV(d, ctx=None):
    return _dense_spvector(d, np.ones(d), ctx)


def zeroSV(d, ctx=None):
    return SpVector(d, [], [], ctx)


# ---- SpMatrix -------------------------------------------------------------------------------------------

class SpMatrix:
    """SM (r,c) (IntM (IntM a)) (SpMatrix.hs:52-54), lowered once to the device CSR."""

    def __init__(self, dims, handle, ctx):
        self.dims = (int(dims[0]), int(dims[1]))
        self.h = handle
        self.ctx = ctx
        self._host = None

    @property
    def nrows(self):
        return self.dims[0]

    @property
    def ncols(self):
        return self.dims[1]

    def csr(self):
        """(rowptr, colidx, val) of this rank's row block, int64 / f64, copied back from the device."""
        if self._host is None:
            nnz, rows = C.c_int64(), C.c_int64()
            check(lib().sla_csr_dims(self.h, None, None, C.byref(nnz), C.byref(rows)))
            rp = np.zeros(rows.value + 1, dtype=np.int64)
            ci = np.zeros(max(nnz.value, 1), dtype=np.int64)
            va = np.zeros(max(nnz.value, 1), dtype=np.float64)
            check(lib().sla_csr_export(self.h, _p(rp), _p(ci), _p(va)))
            self._host = (rp, ci[:nnz.value], va[:nnz.value])
        return self._host

    def csc(self):
        """(colptr, rowidx, val): the CscMatrix arrays of this matrix (CSC.hs:17-24), copied back from the device (single-device contexts)."""
        nnz = C.c_int64()
        check(lib().sla_csr_dims(self.h, None, None, C.byref(nnz), None))
        cp = np.zeros(self.ncols + 1, dtype=np.int64)
        ri = np.zeros(max(nnz.value, 1), dtype=np.int64)
        va = np.zeros(max(nnz.value, 1), dtype=np.float64)
        check(lib().sla_csr_export_csc(self.h, _p(cp), _p(ri), _p(va)))
        return cp, ri[:nnz.value], va[:nnz.value]

    def nnz(self):
        return int(self.csr()[0][-1])

    def isDiagonalSM(self):                       # SpMatrix.hs:411-415
        out = C.c_int()
        check(lib().sla_csr_is_diagonal(self.h, C.byref(out)))
        return bool(out.value)

    def toListSM(self):
        """toListSM (SpMatrix.hs:251-253) yields DESCENDING (row, col) order (a consing left fold)."""
        rp, ci, va = self.csr()
        out = [(i, int(ci[k]), float(va[k])) for i in range(len(rp) - 1) for k in range(rp[i], rp[i + 1])]
        return out[::-1]

    def toDense(self):
        rp, ci, va = self.csr()
        D = np.zeros(self.dims)
        for i in range(len(rp) - 1):
            D[i, ci[rp[i]:rp[i + 1]]] = va[rp[i]:rp[i + 1]]
        return D

    def kernel_info(self):
        buf = C.create_string_buffer(1024)
        check(lib().sla_csr_kernel_info(self.h, buf, 1024))
        return buf.value.decode()

    def props(self):
        """Typed properties (sla_csr_get_props): fold (0 exact left fold / 1 fixed regrouping of long rows / 2 relaxed order, not reproducible
        bit for bit), x_exchange (0 one rank / 1 all-gather / 2 window), nranks, rows_local, nnz_local, rowptr_bits."""
        from ._lib import CsrProps
        p = CsrProps()
        check(lib().sla_csr_get_props(self.h, C.byref(p)))
        return {k: getattr(p, k) for k, _ in CsrProps._fields_ if k not in ("struct_size", "reserved")}

    def exchange_plan(self):
        """Row-sharded matrices: (send_len, recv_len) -- doubles this rank sends to / receives from each peer in ONE exchange of a (#>) input."""
        nr = self.props()["nranks"]
        s, r = np.zeros(nr, dtype=np.int64), np.zeros(nr, dtype=np.int64)
        check(lib().sla_csr_exchange_plan(self.h, s.ctypes.data, r.ctypes.data, nr))
        return s, r

    def lower_info(self):
        """Wall-clock phases of this matrix's lowering in milliseconds (sla_csr_lower_info): {phase: ms}."""
        buf = C.create_string_buffer(2048)
        check(lib().sla_csr_lower_info(self.h, buf, 2048))
        out = {}
        for tok in buf.value.decode().split(";"):
            if "=" in tok:
                k, v = tok.rsplit("=", 1)
                try:
                    out[k] = out.get(k, 0.0) + float(v)
                except (TypeError, ValueError):   # a note, e.g. "tile form not taken=layer boundaries (dense rows)"
                    out[k] = v
        return out

    def __eq__(self, o):                          # structural, like the derived Eq
        if not isinstance(o, SpMatrix) or self.dims != o.dims:
            return False
        a, b = self.csr(), o.csr()
        return all(np.array_equal(x, y) for x, y in zip(a, b))

    def __matmul__(self, o):
        return matMat(self, o) if isinstance(o, SpMatrix) else matVec(self, o)

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                lib().sla_csr_destroy(self.h)
        except Exception:
            pass


def fromListSM(dims, triples, ctx=None):
    """fromListSM (m,n) iix (SpMatrix.hs:218-224): last duplicate wins, out-of-bounds raises."""
    ctx = ctx or default_context()
    t = list(triples)
    r = np.array([a[0] for a in t], dtype=np.int64)
    c = np.array([a[1] for a in t], dtype=np.int64)
    v = np.array([a[2] for a in t], dtype=np.float64)
    return fromCOO(dims, r, c, v, ctx)


def fromCOO(dims, rows, cols, vals, ctx=None, dup_policy=0):
    ctx = ctx or default_context()
    r = np.ascontiguousarray(rows, dtype=np.int64)
    c = np.ascontiguousarray(cols, dtype=np.int64)
    v = np.ascontiguousarray(vals, dtype=np.float64)
    h = C.c_void_p()
    check(lib().sla_csr_from_coo(ctx.h, int(dims[0]), int(dims[1]), len(r), _p(r), _p(c), _p(v), dup_policy, C.byref(h)))
    return SpMatrix(dims, h, ctx)


def fromCSR(dims, rowptr, colidx, vals, ctx=None):
    """Already-canonical CSR (ascending columns, no duplicates); the whole matrix on every rank."""
    ctx = ctx or default_context()
    rp = np.ascontiguousarray(rowptr, dtype=np.int64)
    ci = np.ascontiguousarray(colidx, dtype=np.int64)
    va = np.ascontiguousarray(vals, dtype=np.float64)
    h = C.c_void_p()
    check(lib().sla_csr_from_csr(ctx.h, int(dims[0]), int(dims[1]), _p(rp), _p(ci), _p(va), C.byref(h)))
    return SpMatrix(dims, h, ctx)


def fromCSC(dims, colptr, rowidx, vals, ctx=None):
    """CscMatrix arrays (vector/src/Data/Sparse/Internal/CSC.hs:17-24; toCSC :51-55) -> the lowered matrix; the CSC side stays attached
    as its transpose.  Single-device contexts."""
    ctx = ctx or default_context()
    cp = np.ascontiguousarray(colptr, dtype=np.int64)
    ri = np.ascontiguousarray(rowidx, dtype=np.int64)
    va = np.ascontiguousarray(vals, dtype=np.float64)
    if len(cp) != int(dims[1]) + 1:
        raise ValueError("colptr must have ncols + 1 entries")
    h = C.c_void_p()
    check(lib().sla_csr_from_csc(ctx.h, int(dims[0]), int(dims[1]), _p(cp), _p(ri), _p(va), C.byref(h)))
    return SpMatrix(dims, h, ctx)


def fromCSB(dims, beta, blkptr, rowix, colix, vals, ctx=None):
    """CsbMatrix arrays (vector/src/Data/Sparse/Internal/CSB.hs:38-70): blocks of edge beta in blockIx order (:88-92: the block row runs
    fastest), block-relative indices."""
    ctx = ctx or default_context()
    bp = np.ascontiguousarray(blkptr, dtype=np.int64)
    ri = np.ascontiguousarray(rowix, dtype=np.int64)
    ci = np.ascontiguousarray(colix, dtype=np.int64)
    va = np.ascontiguousarray(vals, dtype=np.float64)
    nblk = (-(-int(dims[0]) // int(beta))) * (-(-int(dims[1]) // int(beta))) if beta > 0 else 0
    if len(bp) != nblk + 1:
        raise ValueError("blkptr must have ceil(m / beta) * ceil(n / beta) + 1 entries")
    h = C.c_void_p()
    check(lib().sla_csr_from_csb(ctx.h, int(dims[0]), int(dims[1]), int(beta), _p(bp), _p(ri), _p(ci), _p(va), C.byref(h)))
    return SpMatrix(dims, h, ctx)


def fromCSRRows(dims, row_begin, rowptr_local, colidx, vals, ctx=None):
    """This rank's row block only (global column indices)."""
    ctx = ctx or default_context()
    rp = np.ascontiguousarray(rowptr_local, dtype=np.int64)
    ci = np.ascontiguousarray(colidx, dtype=np.int64)
    va = np.ascontiguousarray(vals, dtype=np.float64)
    h = C.c_void_p()
    check(lib().sla_csr_from_csr_rows(ctx.h, int(dims[0]), int(dims[1]), int(row_begin), len(rp) - 1,
                                      _p(rp), _p(ci), _p(va), C.byref(h)))
    return SpMatrix(dims, h, ctx)


def readMatrixMarket(path, ctx=None, dup_policy=0):
    """`matrix coordinate real general` file -> SpMatrix with the loader semantics of test/Perf.hs:20-45
    (1-based -> 0-based, file order into fromListSM, no symmetric expansion)."""
    ctx = ctx or default_context()
    h = C.c_void_p()
    check(lib().sla_csr_from_matrix_market(ctx.h, str(path).encode(), dup_policy, C.byref(h)))
    m, n = C.c_int64(), C.c_int64()
    check(lib().sla_csr_dims(h, C.byref(m), C.byref(n), None, None))
    return SpMatrix((m.value, n.value), h, ctx)


def readMatrixMarketArray(path, ctx=None):
    """`matrix array` file -> dense SpVector (the right-hand side of test/Perf.hs)."""
    ctx = ctx or default_context()
    h = C.c_void_p()
    check(lib().sla_vec_from_matrix_market(ctx.h, str(path).encode(), C.byref(h)))
    n = C.c_int64()
    check(lib().sla_vec_dim(h, C.byref(n), None))
    out = np.empty(n.value, dtype=np.float64)
    check(lib().sla_vec_to_host(h, _p(out)))
    lib().sla_vec_destroy(h)
    return fromVector(out, ctx)


def fromListDenseSM(m, ll, ctx=None):
    """fromListDenseSM m ll (SpMatrix.hs:239-241): column-major, entry k -> (k mod m, k div m)."""
    ll = list(ll)
    n = len(ll) // m
    return fromListSM((m, n), [(k % m, k // m, ll[k]) for k in range(m * n)], ctx)


def sparsifySM(A):
    """sparsifySM: drop entries with |x| <= 1e-12 (Eps.hs:41-42)."""
    t = [(i, j, x) for (i, j, x) in A.toListSM()[::-1] if abs(x) > 1e-12]
    return fromListSM(A.dims, t, A.ctx)


def jacobiPre(A):
    """jacobiPre x = recip <$> extractDiag x (Sparse.hs:689-690)."""
    h = C.c_void_p()
    check(lib().sla_jacobi_pre(A.h, C.byref(h)))
    return SpMatrix(A.dims, h, A.ctx)


def _tri_solve(T, b, upper):
    ctx = T.ctx
    bv = b.device() if isinstance(b, SpVector) else b
    out = DeviceVector(ctx, T.nrows)
    bad = C.c_int64(-1)
    check(lib().sla_tri_solve(T.h, 1 if upper else 0, bv.h, out.h, C.byref(bad)))
    if not isinstance(b, SpVector):
        return out
    x = out.to_host()                       # sparsifySV: the near-zero entries are structurally absent
    ix = np.nonzero(x)[0]
    return fromListSV(T.nrows, list(zip(ix.tolist(), x[ix].tolist())), ctx) if len(ix) < len(x) else fromVector(x, ctx)


def triLowerSolve(ll, b):
    """Forward substitution (Sparse.hs:750-776); raises NeedsPivoting when l_ii is missing or |l_ii| <= 1e-12.
    SpVector in -> SpVector out (sparsifySV applied), DeviceVector in -> DeviceVector out."""
    return _tri_solve(ll, b, False)


def triUpperSolve(uu, w):
    """Backward substitution (Sparse.hs:784-811)."""
    return _tri_solve(uu, w, True)


def triSolveLevels(T, upper=False):
    """(dependency levels, rows in the widest level) of the schedule the triangular solve uses."""
    lv, wd = C.c_int64(0), C.c_int64(0)
    check(lib().sla_tri_solve_info(T.h, 1 if upper else 0, C.byref(lv), C.byref(wd)))
    return lv.value, wd.value


def mSsorPre(aa, omega):
    """mSsorPre aa omega = (l, r) (Sparse.hs:712-720): l = (eye n ^-^ scale omega e) ## reciprocal d, r = d ^-^ scale omega f."""
    hl, hr = C.c_void_p(), C.c_void_p()
    check(lib().sla_ssor_pre(aa.h, float(omega), C.byref(hl), C.byref(hr)))
    return SpMatrix(aa.dims, hl, aa.ctx), SpMatrix(aa.dims, hr, aa.ctx)


def ilu0Pre(aa, exact=True):
    """ilu0Pre aa = (l, u) (Sparse.hs:696-706): the complete `lu aa` filtered to aa's stored positions (exact=True, the
    reference's definition, <= 4096 rows) or the incomplete factorisation on aa's pattern proper (exact=False: an extension,
    any size, identical whenever lu creates no fill outside the pattern).  Raises NeedsPivoting like `lu`."""
    hl, hu, bad = C.c_void_p(), C.c_void_p(), C.c_int64(-1)
    check(lib().sla_ilu0_pre(aa.h, 1 if exact else 0, C.byref(hl), C.byref(hu), C.byref(bad)))
    return SpMatrix(aa.dims, hl, aa.ctx), SpMatrix(aa.dims, hu, aa.ctx)


def diagMatMatSparsified(D, A):
    """D #~# A for a diagonal D (matMatSparsified, SpMatrix.hs:816-824): the left-preconditioned operator
    `jacobiPre aa #~# aa` without a general SpGEMM."""
    h = C.c_void_p()
    check(lib().sla_csr_diag_mul(D.h, A.h, C.byref(h)))
    return SpMatrix((D.nrows, A.ncols), h, A.ctx)


def transpose(A):
    """transposeSM (SpMatrix.hs:717): on a single-device context a device sort by (column, row) (sla_csr_transpose); sharded contexts go
    through the triplets."""
    h = C.c_void_p()
    if lib().sla_csr_transpose(A.h, C.byref(h)) == 0:      # (refused on sharded / multi-device contexts: a row block's transpose is a column block)
        return SpMatrix((A.ncols, A.nrows), h, A.ctx)
    return fromListSM((A.ncols, A.nrows), [(j, i, x) for (i, j, x) in A.toListSM()[::-1]], A.ctx)


def _matmat(A, B, transpose_b):
    if A.ncols != (B.ncols if transpose_b else B.nrows):
        raise MatVecSizeMismatchException(_lib.ERR_DIM_MISMATCH, f"matMat : incompatible matrix sizes{(A.dims, B.dims[::-1] if transpose_b else B.dims)}")
    h = C.c_void_p()
    check(lib().sla_csr_matmat(A.h, B.h, 1 if transpose_b else 0, C.byref(h)))
    m, n = C.c_int64(), C.c_int64()
    check(lib().sla_csr_dims(h, C.byref(m), C.byref(n), None, None))
    return SpMatrix((m.value, n.value), h, A.ctx)


def matMat(A, B):   # also SpMatrix.__matmul__ style helpers below
    """A ## B (matMat_ AB, SpMatrix.hs:768-811) on the device: structurally dense over rows(A) x cols(B), explicit zeros kept."""
    return _matmat(A, B, False)


def matMatT(A, B):
    """A ##^ B = A ## transpose B (matMat_ ABt)."""
    return _matmat(A, B, True)


# ---- (#>) (<#) (<.>) norms ---------------------------------------------------------------------------

def matVec(A, x):
    """A #> x (Common.hs:242-250).  The result has a key for every row present in A (and no others)."""
    if A.ncols != x.dim:
        raise MatVecSizeMismatchException(_lib.ERR_DIM_MISMATCH, f"matVec : mismatched dimensions {(A.ncols, x.dim)}")
    y = DeviceVector(A.ctx, A.nrows)
    check(lib().sla_spmv(A.h, x.device().h, y.h))
    yd = y.to_host()
    rp = A.csr()[0]
    keys = np.nonzero(np.diff(rp) > 0)[0]
    return SpVector(A.nrows, keys, yd[keys], A.ctx)


def vecMat(x, A):
    """x <# A (Common.hs:253-256)."""
    if A.nrows != x.dim:
        raise MatVecSizeMismatchException(_lib.ERR_DIM_MISMATCH, f"vecMat : mismatching dimensions {(x.dim, A.nrows)}")
    y = DeviceVector(A.ctx, A.ncols)
    check(lib().sla_spmv_t(A.h, x.device().h, y.h))
    yd = y.to_host()
    keys = np.unique(A.csr()[1])
    return SpVector(A.ncols, keys, yd[keys], A.ctx)


def dot(x, y):
    """x <.> y (SpVector.hs:116-117), evaluated on the device."""
    n = max(x.dim, y.dim)
    a = x if x.dim == n else SpVector(n, x.ix, x.vals, x.ctx)
    b = y if y.dim == n else SpVector(n, y.ix, y.vals, y.ctx)
    out = C.c_double()
    check(lib().sla_dot(a.device().h, b.device().h, C.byref(out)))
    return out.value


def norm2(x):
    """norm2 (SpVector.hs:119-129)."""
    out = C.c_double()
    check(lib().sla_nrm2(x.device().h, C.byref(out)))
    return out.value


def norm2Sq(x):
    return dot(x, x)


def normalize2(x):
    """normalize2 v = (recip (norm2 v)) .* v (Class.hs:94-95, SpVector.hs:126)."""
    return (1.0 / norm2(x)) * x


def nearZero(a):
    return abs(a) <= 1e-12                       # Eps.hs:41-42


# ---- solver state records -----------------------------------------------------------------------------

class _SolverState:
    fields = ()

    def __init__(self, method, A, b, x0):
        self.A, self.method = A, method
        self.h = C.c_void_p()
        bh = b.h if isinstance(b, DeviceVector) else b.device().h      # DeviceVector: sharded callers
        xh = x0.h if isinstance(x0, DeviceVector) else x0.device().h
        check(lib().sla_solver_init(int(method), A.h, bh, xh, C.byref(self.h)))

    def _get(self, field, dim):
        v = DeviceVector(self.A.ctx, dim)
        check(lib().sla_solver_get(self.h, field, v.h))
        return fromVector(v.to_host(), self.A.ctx)

    def step(self, k=1):
        """k steps IN PLACE on the device (the fast path: `iterate step s !! k` without materialising the k records)."""
        check(lib().sla_solver_step(self.h, int(k)))
        return self

    def clone(self):
        """A deep copy of the state record (sla_solver_clone): stepping the copy leaves this record untouched."""
        other = object.__new__(type(self))
        other.A, other.method = self.A, self.method
        other.h = C.c_void_p()
        check(lib().sla_solver_clone(self.h, C.byref(other.h)))
        return other

    def set_shadow(self, r0hat):
        """Replace the shadow residual (the explicit r0hat / rhat argument of bicgstabStep / cgsStep)."""
        rh = r0hat.h if isinstance(r0hat, DeviceVector) else r0hat.device().h
        check(lib().sla_solver_set_shadow(self.h, rh))
        return self

    def __del__(self):
        try:
            if self.h and self.A.ctx.h:
                lib().sla_solver_destroy(self.h)
        except Exception:
            pass


class BICGSTAB(_SolverState):
    """data BICGSTAB = BICGSTAB {_xBicgstab, _rBicgstab, _pBicgstab} (Sparse.hs:959-960)."""
    _xBicgstab = property(lambda s: s._get(0, s.A.ncols))
    _rBicgstab = property(lambda s: s._get(1, s.A.nrows))
    _pBicgstab = property(lambda s: s._get(2, s.A.ncols))


class CGS(_SolverState):
    """data CGS = CGS {_x, _r, _p, _u} (Sparse.hs:919)."""
    _x = property(lambda s: s._get(0, s.A.ncols))
    _r = property(lambda s: s._get(1, s.A.nrows))
    _p = property(lambda s: s._get(2, s.A.ncols))
    _u = property(lambda s: s._get(3, s.A.nrows))


class CGNE(_SolverState):
    """data CGNE = CGNE {_xCgne, _rCgne, _pCgne} (Sparse.hs:855-856)."""
    _xCgne = property(lambda s: s._get(0, s.A.ncols))
    _rCgne = property(lambda s: s._get(1, s.A.nrows))
    _pCgne = property(lambda s: s._get(2, s.A.ncols))


class BCG(_SolverState):
    """data BCG = BCG {_xBcg, _rBcg, _rHatBcg, _pBcg, _pHatBcg} (Sparse.hs:886-887).  An extension: the reference's bcgInit / bcgStep are
    commented out (:889-909) and `linSolve0 BCG_` throws there and here."""
    _xBcg = property(lambda s: s._get(0, s.A.ncols))
    _rBcg = property(lambda s: s._get(1, s.A.nrows))
    _pBcg = property(lambda s: s._get(2, s.A.ncols))
    _rHatBcg = property(lambda s: s._get(4, s.A.nrows))
    _pHatBcg = property(lambda s: s._get(5, s.A.nrows))


def bicgsInit(aa, b, x0):
    return BICGSTAB(BICGSTAB_, aa, b, x0)         # Sparse.hs:962-965


def _pure_step(args, k):
    """The reference's calling conventions: step(state [, k]) -- the in-place device fast path this package started with --
    or step(aa, r0hat, state): PURE like the Haskell (Sparse.hs:928, :972) -- a new record is returned, `state` keeps its
    value, so `iterate (bicgstabStep aa r0hat) s0 !! 20` (README.md:222-226) does not alias its elements."""
    if len(args) == 1:
        return args[0].step(k)
    aa, r0hat, state = args
    if aa is not state.A:
        raise ValueError("the state record was initialised with a different matrix")
    out = state.clone()
    if r0hat is not None:
        out.set_shadow(r0hat)
    return out.step(k)


def bicgstabStep(*args, k=1):
    """bicgstabStep aa r0hat state (Sparse.hs:972-981) -> new state; bicgstabStep(state, k=..) steps in place."""
    return _pure_step(args, k)


def cgsInit(aa, b, x0):
    return CGS(CGS_, aa, b, x0)                   # Sparse.hs:921-924


def cgsStep(*args, k=1):
    """cgsStep aa rhat state (Sparse.hs:928-939) -> new state; cgsStep(state, k=..) steps in place."""
    return _pure_step(args, k)


def cgneInit(aa, b, x0):
    return CGNE(CGNE_, aa, b, x0)                 # Sparse.hs:864-868


def cgneStep(state, k=1):
    return state.step(k)                          # Sparse.hs:870-878


def bcgInit(aa, b, x0):
    return BCG(BCG_, aa, b, x0)                   # the commented bcgInit, Sparse.hs:889-897 (p0 = r0, p0hat = r0hat = r0)


def bcgStep(*args, k=1):
    """bcgStep aa state (the commented code of Sparse.hs:899-909) -> new state; bcgStep(state, k=..) steps in place."""
    if len(args) == 1:
        return args[0].step(k)
    aa, state = args
    if aa is not state.A:
        raise ValueError("the state record was initialised with a different matrix")
    return state.clone().step(k)


# ---- linSolve0 / arnoldi / gmres / (<\>) ---------------------------------------------------------------

def _opts(kw):
    if not kw:
        return None
    o = SolveOpts(kw.get("max_iters", 200), kw.get("tol_abs", 1e-6), kw.get("tol_rel", 1e-4),
                  kw.get("check_every", 16), kw.get("true_residual", 1))
    return C.byref(o)


def linSolve0(method, aa, b, x0, return_info=False, **opts):
    """linSolve0 method aa b x0 (Sparse.hs:1016-1072).  Raises MatVecSizeMismatchException /
    IterationException like the reference; never raises on non-convergence."""
    out = DeviceVector(aa.ctx, aa.ncols)
    info = SolveInfo()
    hist = None
    if opts.pop("history", False):     # residual trace: the true residual norm after every iteration (cgsStepDebug, Sparse.hs:942-948)
        hist = np.zeros(max(int(opts.get("max_iters", 200)), 1), dtype=np.float64)
        o = SolveOpts(opts.get("max_iters", 200), opts.get("tol_abs", 1e-6), opts.get("tol_rel", 1e-4), opts.get("check_every", 16),
                      opts.get("true_residual", 1), hist.ctypes.data, len(hist))
        po = C.byref(o)
    else:
        po = _opts(opts)
    check(lib().sla_linsolve0(int(method), aa.h, b.device().h, x0.device().h, po, out.h, C.byref(info)))
    x = fromVector(out.to_host(), aa.ctx)
    d = info.as_dict()
    if hist is not None:
        d["history"] = hist[:info.history_len].copy()
    return (x, d) if return_info else x


def arnoldi(aa, b, kn):
    """arnoldi aa b kn (Sparse.hs:630-667) -> (Q, H) as dense arrays n x (k+1), (k+1) x k."""
    n = aa.ncols
    if n != b.dim:
        raise MatVecSizeMismatchException(_lib.ERR_DIM_MISMATCH, f"arnoldi {aa.dims} {b.dim}")
    Q = np.zeros((kn + 1) * n, dtype=np.float64)
    H = np.zeros((kn + 1) * kn, dtype=np.float64)
    kd = C.c_int()
    check(lib().sla_arnoldi(aa.h, b.device().h, int(kn), _p(Q), _p(H), C.byref(kd)))
    k = kd.value
    return Q.reshape(kn + 1, n).T[:, :k + 1].copy(), H.reshape(kn, kn + 1).T[:k + 1, :k].copy()


def gmres(aa, b, x0, restart=30, return_info=False, **opts):
    """Restarted GMRES(m) on the device Arnoldi (extension: the reference's gmres is commented out)."""
    out = DeviceVector(aa.ctx, aa.ncols)
    info = SolveInfo()
    check(lib().sla_gmres(aa.h, b.device().h, x0.device().h, int(restart), _opts(opts), out.h, C.byref(info)))
    x = fromVector(out.to_host(), aa.ctx)
    return (x, info.as_dict()) if return_info else x


def linSolve(aa, b, return_info=False):
    """aa <\\> b (Class.hs:244-249; dead instance Sparse.hs:1080-1084: GMRES from x0 = 0.1 * ones)."""
    out = DeviceVector(aa.ctx, aa.ncols)
    info = SolveInfo()
    check(lib().sla_linsolve(aa.h, b.device().h, out.h, C.byref(info)))
    x = fromVector(out.to_host(), aa.ctx)
    return (x, info.as_dict()) if return_info else x
