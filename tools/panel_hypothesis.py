#!/usr/bin/env python
"""Would column-panel blocking pay for random matrices?  Same random matrix, columns folded into a 3 MB
window (col % 384k): every gather then hits the XCD-private L2 instead of the Infinity Cache / HBM."""
import sys, time, ctypes as C
sys.path.insert(0, "sparse-linear-algebra_amd")
import numpy as np
import sla_amd as sla
from sla_amd import workloads as wl, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
dims, (rp, ci, va) = wl.random_spd(n, 16, 42)
rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
ctx = sla.default_context()
lib = _lib.lib()
for label, cols in (("original columns", ci), ("columns folded into 384k (3 MB of x)", ci % 384000)):
    A = sla.fromCOO(dims, rows, cols, va)
    x, y = sla.DeviceVector(ctx, n, np.ones(n)), sla.DeviceVector(ctx, n)
    for _ in range(3):
        _lib.check(lib.sla_spmv(A.h, x.h, y.h))
    ctx.sync(); t = time.perf_counter()
    for _ in range(10):
        _lib.check(lib.sla_spmv(A.h, x.h, y.h))
    ctx.sync(); dt = (time.perf_counter() - t) / 10
    nnz = A.nnz()
    print(f"{label}: nnz={nnz} spmv {dt*1e3:.3f} ms = {(12*nnz+20*n)/dt/1e9:.0f} GB/s, {nnz/dt/1e9:.1f} G gathers/s  [{A.kernel_info()[:40]}]")
    del A
