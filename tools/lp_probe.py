"""LDS-panel SpMV (spmv_lpanel_kernel) on random matrices with dense rows: time per (#>) of the stream kernel, of the
default lowering and -- with a third argument -- of every lane-group shape.  python tools/lp_probe.py [rows] [half_nnz_per_row] [sweep]"""
import os, sys, time
sys.path.insert(0, "sparse-linear-algebra_amd"); sys.path.insert(0, ".")
import numpy as np
import sla_amd as sla
from sla_amd import workloads as wl

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dims, (rp, ci, va) = wl.random_spd(rows, k, 42)
n, nnz = dims[0], len(ci)
bytes_ = nnz * 12 + (n + 1) * 4 + 2 * n * 8
x = np.random.default_rng(0).standard_normal(n)
SWEEP = ({"SLA_LP_MINSEG": "1", "SLA_LP_CFG": "0"}, {"SLA_LP_MINSEG": "1", "SLA_LP_CFG": "1"}, {"SLA_LP_MINSEG": "1", "SLA_LP_CFG": "2"},
         {"SLA_LP_MINSEG": "1", "SLA_LP_CFG": "3"}) if len(sys.argv) > 3 else ()
for env in ({"SLA_LPANEL": "0"}, {}) + SWEEP:
    for kk in ("SLA_LPANEL", "SLA_LP_TASKS", "SLA_LP_ROWCOST", "SLA_LP_MINSEG", "SLA_LP_CFG"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    ctx = sla.Context(0)
    A = sla.fromCSR(dims, rp, ci, va, ctx)
    xv = sla.DeviceVector(ctx, n, x)
    yv = sla.DeviceVector(ctx, n)
    L = sla._lib.lib()
    for _ in range(3):
        L.sla_spmv(A.h, xv.h, yv.h)
    ctx.sync()
    t = time.perf_counter()
    R = 30
    for _ in range(R):
        L.sla_spmv(A.h, xv.h, yv.h)
    ctx.sync()
    dt = (time.perf_counter() - t) / R
    print(env, A.kernel_info().split()[0], f"{dt*1e3:.3f} ms  {bytes_/dt/1e9:.0f} GB/s")
    del A, xv, yv
    ctx.close()
