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
