#!/usr/bin/env python
"""Timeline of the block-local triangular solve (option tri_syncfree = 2; SLA_TRI_TRACE): when each block was taken, finished its last
row and was left, on the lower triangle of the 216^3 Laplacian (or `poisson`).
    SLA_TRI_TRACE=/tmp/t.txt python tools/tri_trace.py [poisson] [block_rows]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-linear-algebra_amd"))
path = os.environ.setdefault("SLA_TRI_TRACE", "/tmp/sla_tri_trace.txt")
import sla_amd as sla  # noqa: E402
from sla_amd import _lib, workloads as wl  # noqa: E402

args = sys.argv[1:]
poisson = "poisson" in args
nums = [int(a) for a in args if a.isdigit()]
dims, (rp, ci, va) = wl.poisson2d(1000, 1000) if poisson else wl.laplace3d(216, 216, 216)
ctx = sla.default_context()
n = dims[0]
T = sla.fromCSR(dims, rp, ci, va, ctx)
b = sla.DeviceVector(ctx, n, np.ones(n))
x = sla.DeviceVector(ctx, n)
ctx.set_options(tri_syncfree=2, tri_block_rows=nums[0] if nums else 16384)
lib = _lib.lib()
for _ in range(3):
    _lib.check(lib.sla_tri_solve(T.h, 0, b.h, x.h, None))
    ctx.sync()
t = np.loadtxt(path)
t = t.reshape(-1, 8)
start, last, left = t[:, 4], t[:, 6], t[:, 7]
print("blocks %d, rows per block %d..%d; kernel span %.1f us" % (len(t), t[:, 3].min(), t[:, 3].max(), left.max()))
print("time a block is held (left - start): median %.1f us, p10 %.1f, p90 %.1f, max %.1f; sum / 256 CUs = %.1f us"
      % (np.median(left - start), *np.percentile(left - start, [10, 90]), (left - start).max(), (left - start).sum() / 256))
print("taken  wg  first_pos   rows   start   last_row   left")
step = max(1, len(t) // 60)
for r in t[::step]:
    print("%5d %4d %10d %6d %8.1f %8.1f %8.1f" % (r[0], r[1], r[2], r[3], r[4], r[6], r[7]))
