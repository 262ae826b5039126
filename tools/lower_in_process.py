#!/usr/bin/env python
"""Why does config 3a lower in 0.14-0.21 s stand-alone and in 0.26-0.29 s as the last block of bench.py?  One process: the random matrix is
lowered (fresh context, validating entry point -- what bench.py's side_block does) at the start, and again after each piece of the bench
line has run in the same process: the headline workload (generate + lower + 60 steps), the CPU baseline (oracle, 1 thread + OpenMP),
the general_csr block.   python tools/lower_in_process.py  (phases on stderr with SLA_DEBUG_LOWER=1)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "sparse-linear-algebra_amd"))
import numpy as np  # noqa: E402
import sla_amd as sla  # noqa: E402


def lower(tag, dims, rp, ci, va, reps=2):
    for i in range(reps):
        ctx = sla.Context(0)
        t0 = time.perf_counter()
        A = sla.fromCSRRows(dims, 0, rp, ci, va, ctx)
        ctx.sync()
        dt = time.perf_counter() - t0
        ph = A.lower_info()
        print(f"{tag:58s} #{i}: from_csr {dt:.3f} s | x-window {ph.get('x-window statistics', 0):5.1f}  offsets {ph.get('offset dictionary', 0):5.1f}  "
              f"upload(rest) {ph.get('canonical CSR upload (rest)', 0):6.1f}  tile form {ph.get('tile form', 0):5.1f} ms", flush=True)
        del A
        ctx.close()


d3, (dm3, (rp3, ci3, va3)) = bench.workload("random_spd_10m")
lower("fresh process", dm3, rp3, ci3, va3, 3)
desc, (dims, (rp, ci, va)) = bench.workload("laplace3d_10m")
ctx = sla.Context(0)
A = sla.fromCSRRows(dims, 0, rp, ci, va, ctx)
b = np.add.reduceat(va, rp[:-1])
st = sla.bicgsInit(A, sla.DeviceVector(ctx, dims[0], b, local=True), sla.DeviceVector(ctx, dims[0]))
st.step(60)
ctx.sync()
lower("after the headline workload (context alive)", dm3, rp3, ci3, va3)
cb = bench.cpu_baseline(dims, rp, ci, va, b, 4.0)
lower("after the CPU baseline (oracle, 1 thread + OpenMP)", dm3, rp3, ci3, va3)
g = bench.side_block(desc, dims, rp, ci, va, {"wdia": 0, "vdict": 0, "diag": 0}, 20, 5)
lower("after the general_csr block", dm3, rp3, ci3, va3)
e2e = bench.end_to_end_block(ctx, A, dims, rp, ci, va, b, 0.05, 4500.0, True)
lower("after the end_to_end block (cold linSolve0 + from_coo)", dm3, rp3, ci3, va3)
probes = [ctx.stream_probe(r_, w_, dims[0], 20) for r_, w_ in ((8, 0), (5, 3), (2, 1))]
lower("after the stream probes", dm3, rp3, ci3, va3)
del st, A
ctx.close()
rp = ci = va = None
lower("after the headline context is gone and its arrays freed", dm3, rp3, ci3, va3)
d3b, (dm3b, (rp3b, ci3b, va3b)) = bench.workload("random_spd_10m")
lower("a SECOND copy of the matrix, generated now", dm3b, rp3b, ci3b, va3b)
