"""Phase times of sla_csr_from_csr on one workload (SLA_DEBUG_LOWER prints them, sub-phases included), a few repeats in one process.
    SLA_HOST_THREADS=32 python tools/lower_phases.py laplace3d_10m [repeats]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SLA_DEBUG_LOWER", "1")
import bench  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "sparse-linear-algebra_amd"))
import sla_amd as sla  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "laplace3d_10m"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
desc, (dims, (rp, ci, va)) = bench.workload(name)
for i in range(reps):
    # driver-shaped since round 5: what bench.py's blocks time -- a FRESH context per matrix and the validating entry point
    # (sla_csr_from_csr_rows), so that this tool and `lowered_once.from_csr_s` of the bench line measure the same thing
    ctx = sla.Context(0)
    t0 = time.time()
    A = sla.fromCSRRows(dims, 0, rp, ci, va, ctx)
    ctx.sync()
    print(f"## {name} threads={os.environ.get('SLA_HOST_THREADS', 'default')} repeat {i}: from_csr {time.time() - t0:.3f} s  {A.kernel_info().split()[0]}", file=sys.stderr, flush=True)
    del A
    ctx.close()
