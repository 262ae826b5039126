"""Synthetic inputs of BASELINE.json's configs.  Host-side numpy generators that emit
canonical CSR (ascending columns) for a contiguous row range, so a rank can build only its own slab."""
import numpy as np


def _stencil_rows(row_begin, row_end, offsets, valid_fn, value_fn):
    """CSR for rows [row_begin, row_end) of a stencil matrix.  `offsets` ascending column offsets,
    valid_fn(rows, t) -> bool mask, value_fn(rows, t) -> values for stencil leg t."""
    rows = np.arange(row_begin, row_end, dtype=np.int64)
    nr, no = len(rows), len(offsets)
    valid = np.empty((nr, no), dtype=bool)
    cols = np.empty((nr, no), dtype=np.int64)
    vals = np.empty((nr, no), dtype=np.float64)
    for t, off in enumerate(offsets):
        valid[:, t] = valid_fn(rows, t)
        cols[:, t] = rows + off
        vals[:, t] = value_fn(rows, t)
    rowptr = np.zeros(nr + 1, dtype=np.int64)
    np.cumsum(valid.sum(axis=1), out=rowptr[1:])
    return rowptr, cols[valid], vals[valid]


def poisson2d(nx, ny, row_begin=0, row_end=None):
    """Config 2: 5-point Poisson on an nx x ny grid, row = j*nx + i, diag 4, off-diag -1, Dirichlet."""
    n = nx * ny
    row_end = n if row_end is None else row_end
    offsets = [-nx, -1, 0, 1, nx]

    def valid(rows, t):
        i, j = rows % nx, rows // nx
        return [j > 0, i > 0, np.ones(len(rows), bool), i < nx - 1, j < ny - 1][t]

    def value(rows, t):
        return np.full(len(rows), 4.0 if t == 2 else -1.0)

    return (n, n), _stencil_rows(row_begin, row_end, offsets, valid, value)


def laplace3d(nx, ny, nz, row_begin=0, row_end=None):
    """Config 4: 7-point Laplacian on nx x ny x nz, row = (k*ny + j)*nx + i, diag 6, off-diag -1,
    Dirichlet.  Contiguous row blocks are slabs in k (the slowest index)."""
    n = nx * ny * nz
    row_end = n if row_end is None else row_end
    offsets = [-nx * ny, -nx, -1, 0, 1, nx, nx * ny]

    def valid(rows, t):
        i = rows % nx
        j = (rows // nx) % ny
        k = rows // (nx * ny)
        return [k > 0, j > 0, i > 0, np.ones(len(rows), bool), i < nx - 1, j < ny - 1, k < nz - 1][t]

    def value(rows, t):
        return np.full(len(rows), 6.0 if t == 3 else -1.0)

    return (n, n), _stencil_rows(row_begin, row_end, offsets, valid, value)


def banded_nonsym(n, seed=99, row_begin=0, row_end=None):
    """Config 5: bands at offsets {-2,-1,0,+1,+3} with values {-1,-1.5,4.2,-0.5,-1} x (1 +- 5 % noise):
    diagonally dominant, non-symmetric."""
    row_end = n if row_end is None else row_end
    offsets = [-2, -1, 0, 1, 3]
    base = [-1.0, -1.5, 4.2, -0.5, -1.0]

    def valid(rows, t):
        c = rows + offsets[t]
        return (c >= 0) & (c < n)

    def value(rows, t):
        # counter-based noise so that any row range reproduces the same matrix
        h = (rows * 5 + t).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)
        h ^= h >> np.uint64(31)
        h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(29)
        u = (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)      # [0,1)
        return base[t] * (1.0 + 0.05 * (2.0 * u - 1.0))

    return (n, n), _stencil_rows(row_begin, row_end, offsets, valid, value)


_WLGEN = None


def _wlgen():
    """libsla_wlgen.so (csrc/sla_wlgen.c): counting-sort assembly of the same matrix in seconds instead of the
    minutes two global numpy argsorts take at n = 10 M.  None when it has not been built."""
    global _WLGEN
    if _WLGEN is None:
        import ctypes as C
        import os
        so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib", "libsla_wlgen.so")
        if not os.path.exists(so):
            _WLGEN = False
        else:
            L = C.CDLL(so)
            L.sla_wl_random_spd.restype = C.c_int64
            L.sla_wl_random_spd.argtypes = [C.c_int64, C.c_int64] + [C.c_void_p] * 5
            L.sla_wl_random_spd_rows.restype = C.c_int64
            L.sla_wl_random_spd_rows.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                                 C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int]
            L.sla_wl_free.restype = None
            L.sla_wl_free.argtypes = [C.c_void_p]
            _WLGEN = L
    return _WLGEN or None


def random_spd(n, k=16, seed=42):
    """Config 3a: symmetric, strictly diagonally dominant random matrix: k off-diagonal picks per row
    (value U(-1,1)), symmetrised over the union pattern, diagonal = 1 + sum |offdiag|  => SPD,
    ~2k+1 entries per row.  Returns the full CSR (not row-range aware)."""
    L = _wlgen()
    if L is None:
        return random_spd_numpy(n, k, seed)
    rng = np.random.Generator(np.random.PCG64(seed))
    c = rng.integers(0, n, size=n * k, dtype=np.int64)
    v = rng.uniform(-1.0, 1.0, size=n * k)
    cap = 2 * n * k + n
    rowptr = np.empty(n + 1, dtype=np.int64)
    col = np.empty(cap, dtype=np.int64)
    val = np.empty(cap, dtype=np.float64)
    nnz = L.sla_wl_random_spd(n, k, c.ctypes.data, v.ctypes.data, rowptr.ctypes.data, col.ctypes.data, val.ctypes.data)
    if nnz < 0:
        raise MemoryError("sla_wl_random_spd")
    return (n, n), (rowptr, col[:nnz], val[:nnz])


def random_spd_rows(n, k, seed, row_begin, row_end, threads=0):
    """Rows [row_begin, row_end) of random_spd(n, k, seed) (rowptr rebased to 0, global column ids) without assembling the
    other rows: a rank of the row-sharded bench draws the full pick list (2 x n k numbers) and builds its own slab only.
    threads > 0: the OpenMP team of the assembly (the ranks of one node share the host's cores)."""
    import ctypes as C
    L = _wlgen()
    if L is None:
        from .partition import local_rows_of
        dims, (rp, ci, va) = random_spd_numpy(n, k, seed)
        return dims, local_rows_of(rp, ci, va, row_begin, row_end)
    rng = np.random.Generator(np.random.PCG64(seed))
    c = rng.integers(0, n, size=n * k, dtype=np.int64)
    v = rng.uniform(-1.0, 1.0, size=n * k)
    rows = row_end - row_begin
    rowptr = np.empty(rows + 1, dtype=np.int64)
    pc, pv = C.c_void_p(), C.c_void_p()
    nnz = L.sla_wl_random_spd_rows(n, k, c.ctypes.data, v.ctypes.data, row_begin, row_end, rowptr.ctypes.data, C.byref(pc), C.byref(pv), int(threads))
    if nnz < 0:
        raise MemoryError("sla_wl_random_spd_rows")
    try:
        col = np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_int64)), shape=(max(nnz, 1),))[:nnz].copy()
        val = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_double)), shape=(max(nnz, 1),))[:nnz].copy()
    finally:
        L.sla_wl_free(pc)
        L.sla_wl_free(pv)
    return (n, n), (rowptr, col, val)


def random_spd_numpy(n, k=16, seed=42):
    """The same matrix assembled with numpy only (the definition; random_spd must reproduce it bit for bit)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    r = np.repeat(np.arange(n, dtype=np.int64), k)
    c = rng.integers(0, n, size=n * k, dtype=np.int64)
    v = rng.uniform(-1.0, 1.0, size=n * k)
    keep = r != c
    r, c, v = r[keep], c[keep], 0.5 * v[keep]
    rr = np.concatenate([r, c])
    cc = np.concatenate([c, r])
    vv = np.concatenate([v, v])
    key = rr * n + cc
    order = np.argsort(key, kind="stable")
    key, vv = key[order], vv[order]
    uniq, start = np.unique(key, return_index=True)
    # (R + R^T)/2 on the union pattern; duplicates summed LEFT TO RIGHT in list order (np.add.reduceat would
    # add a group's first element to the sum of the rest, which differs in the last bit for groups of >= 3)
    glen = np.diff(np.append(start, len(vv)))
    vsum = vv[start].copy()
    for j in range(1, int(glen.max()) if len(glen) else 0):
        g = np.nonzero(glen > j)[0]
        vsum[g] += vv[start[g] + j]
    ur, uc = uniq // n, uniq % n
    absrow = np.zeros(n)
    np.add.at(absrow, ur, np.abs(vsum))
    dr = np.arange(n, dtype=np.int64)
    key2 = np.concatenate([uniq, dr * n + dr])
    val2 = np.concatenate([vsum, 1.0 + absrow])
    order = np.argsort(key2, kind="stable")
    key2, val2 = key2[order], val2[order]
    rows = key2 // n
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=rowptr[1:])
    return (n, n), (rowptr, key2 % n, val2)


def dense_row_spd(n, nnz_per_row, seed=42):
    """Config 3b: density honoured at small n (e.g. n = 200 000, 1 % => 2000 per row)."""
    return random_spd(n, k=nnz_per_row // 2, seed=seed)


def spmv_bytes(nnz, n):
    """B_spmv = 12 nnz + 20 n (f64 values, i32 columns, i32 row pointers)."""
    return 12 * nnz + 20 * n


def bicgstab_step_bytes(nnz, n, true_residual=False):
    """B_bicgstab_step = 24 nnz + 160 n (+ 12 nnz + 20 n with the reference's per-iteration residual)."""
    return 24 * nnz + 160 * n + (12 * nnz + 20 * n if true_residual else 0)
