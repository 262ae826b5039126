#!/usr/bin/env python
"""Per-wave timeline of spmv_wdia_kernel<EPI_DOT> inside a BiCGSTAB step (trace build: tools/build_variant.sh trace
-DSLA_WD_TRACE=1; run with SLA_HIP_LIB=.../libsla_hip_trace.so).  Prints, for wave 0 of every 32nd workgroup, the
cycles spent per slice in: issue (records + epilogue operands + gathers) and wait + fold + store."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-linear-algebra_amd"))
import sla_amd as sla  # noqa: E402
from sla_amd import _lib, workloads as wl  # noqa: E402

lib = _lib.lib()
ctx = sla.default_context()
dims, csr = wl.laplace3d(216, 216, 216)
n = dims[0]
A = sla.fromCSR(dims, *csr, ctx)
b = sla.DeviceVector(ctx, n, np.ones(n))
x0 = sla.DeviceVector(ctx, n)
st = C.c_void_p()
_lib.check(lib.sla_bicgstab_init(A.h, b.h, x0.h, C.byref(st)))
_lib.check(lib.sla_bicgstab_step(st, 5))
ctx.sync()
buf = np.zeros(64 * 16 * 4, dtype=np.uint64)
rc = lib.sla_debug_wd_trace(buf.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
t = buf.reshape(64, 16, 4).astype(np.int64)
for g in range(0, 64, 8):
    rows = []
    for i in range(14):
        if t[g, i, 3] == 0:
            break
        rows.append("%5d/%5d" % (t[g, i, 1] - t[g, i, 0], t[g, i, 3] - t[g, i, 1]))
    if rows:
        print("wg %4d issue / wait+fold cycles per slice:" % (g * 32), " ".join(rows),
              " total", t[g, len(rows) - 1, 3] - t[g, 0, 0])
