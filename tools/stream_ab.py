"""Same-box A/B of the general CSR-stream kernels (SLA_WDIA=0 SLA_VDICT=0 SLA_DIAG=0) on the 216^3 Laplacian and on a random
1 M-row matrix under typed options (round 3 used it for the pipelined kernel, which left the library in round 5: tools/experiments/r03_spmv_pipe_kernel.hip).  Prints K1 / K3 / step timings.
    python tools/stream_ab.py [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
desc, (dims, (rp, ci, va)) = bench.workload("laplace3d_10m")
base = {"wdia": 0, "vdict": 0, "diag": 0}
for name, extra in (("g2048", {}), ("g1024", {"spmv_grid": 1024}), ("g1536", {"spmv_grid": 1536}), ("g1280", {"spmv_grid": 1280}), ("g2048", {}), ("g1024", {"spmv_grid": 1024}),
                    ("g1536", {"spmv_grid": 1536})):
    r = bench.side_block(desc, dims, rp, ci, va, dict(base, **extra), steps, 5)
    print(f"laplace3d_10m {name:7s} {r['value']:8.1f} it/s  K1 {r['k1_ms'] * 1e3:7.1f} us ({r['k1_frac']:.3f})  "
          + "  ".join(f"{k} {v['ms'] * 1e3:.1f}" for k, v in r["kernels"].items()) + "  " + r["spmv_kernel"].split()[0], flush=True)
del rp, ci, va
desc, (dims, (rp, ci, va)) = bench.workload("random_spd_1m")
for name, extra in (("wide", {"stream_wide": 1, "tiles": 0, "panels": 0}), ("narrow", {"stream_wide": 0, "tiles": 0, "panels": 0}),
                    ("wide", {"stream_wide": 1, "tiles": 0, "panels": 0}), ("tiles", {})):
    r = bench.side_block(desc, dims, rp, ci, va, extra, steps, 5, rhs="A.x*")
    print(f"random_spd_1m {name:7s} {r['value']:8.1f} it/s  K1 {r['k1_ms'] * 1e3:7.1f} us ({r['k1_frac']:.3f})  "
          + "  ".join(f"{k} {v['ms'] * 1e3:.1f}" for k, v in r["kernels"].items()) + "  " + r["spmv_kernel"].split()[0], flush=True)
