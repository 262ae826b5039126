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


def plan_allgather_passes(nranks, rank, n, shift, groups=4, order=0):
    """The library's plan of the overlapped all-gather for all-gather-mode tile matrices (sla_plan_allgather_passes, pure host
    arithmetic): returns (visit, pass_ptr, pass_need, ngroups) for `rank` -- the panel visiting order (a row is folded over the
    panels in that order), the pass boundaries into it and the number of exchange groups each pass waits for."""
    import ctypes as C
    import numpy as np
    from . import _lib
    P = (int(n) + (1 << shift) - 1) >> shift
    visit, pp, pn = np.zeros(P, dtype=np.int32), np.zeros(P + 1, dtype=np.int32), np.zeros(P, dtype=np.int32)
    npass, ng = C.c_int(), C.c_int()
    _lib.check(_lib.lib().sla_plan_allgather_passes(nranks, rank, int(n), int(shift), int(groups), int(order), C.c_void_p(visit.ctypes.data),
                                                    C.c_void_p(pp.ctypes.data), C.c_void_p(pn.ctypes.data), C.byref(npass), C.byref(ng)))
    return visit, pp[:npass.value + 1].copy(), pn[:npass.value].copy(), ng.value


def plan_allgather_groups(nranks, n, shift, groups=4, order=0):
    """The exchange groups of that plan (sla_plan_allgather_groups; the same on every rank): a list per group of (source rank, first
    column, end column) pieces in posting order -- in group g every source sends its pieces to every peer as one grouped launch."""
    import ctypes as C
    import numpy as np
    from . import _lib
    cap = (3 * max(groups, 1) + 3) * nranks + nranks
    buf = np.zeros(4 * cap, dtype=np.int64)
    cnt = C.c_int()
    _lib.check(_lib.lib().sla_plan_allgather_groups(nranks, int(n), int(shift), int(groups), int(order), C.c_void_p(buf.ctypes.data), cap, C.byref(cnt)))
    q = buf[:4 * cnt.value].reshape(-1, 4)
    out = [[] for _ in range(nranks if order == 1 else max(groups, 1))]      # (a group may be empty: shards narrower than a panel travel whole in group 0)
    for g, src, b, e in q.tolist():
        out[g].append((src, b, e))
    return out
