"""1-D contiguous row-block partition used by the row-sharded path.
Must agree with sla_ctx_row_range in csrc/sla_api.cpp (checked by a GPU test)."""


def shard_size(m, nranks):
    return (m + nranks - 1) // nranks


def row_block(m, rank, nranks):
    """Rows [begin, end) of an m-row matrix / m-vector owned by `rank`."""
    s = shard_size(m, nranks)
    return min(m, s * rank), min(m, s * (rank + 1))


def pad_shard(x_local, shard):
    """The send buffer of the per-SpMV all-gather: every rank contributes exactly `shard` entries
    (zero padded), so the gathered buffer is nranks * shard long and global index g lives at g."""
    import numpy as np
    out = np.zeros(shard, dtype=np.float64)
    out[: len(x_local)] = x_local
    return out


def local_rows_of(rowptr, colidx, val, begin, end):
    """Slice rows [begin, end) out of a full canonical CSR (rowptr rebased to 0, global columns)."""
    lo, hi = rowptr[begin], rowptr[end]
    return rowptr[begin:end + 1] - lo, colidx[lo:hi], val[lo:hi]


def plan_window_exchange(nranks, rank, n, windows):
    """The library's own exchange plan (sla_plan_window_exchange, pure host arithmetic): returns
    (send_begin, send_len, recv_begin, recv_len, use_window) for `rank`."""
    import ctypes as C
    import numpy as np
    from . import _lib
    w = np.ascontiguousarray(windows, dtype=np.int64).reshape(-1)
    outs = [np.zeros(nranks, dtype=np.int64) for _ in range(4)]
    use = C.c_int()
    _lib.check(_lib.lib().sla_plan_window_exchange(nranks, rank, int(n), C.c_void_p(w.ctypes.data),
                                                   *[C.c_void_p(o.ctypes.data) for o in outs], C.byref(use)))
    return (*outs, bool(use.value))
