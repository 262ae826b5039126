"""Eleven lower and eleven upper triangular solves of the 216^3 Laplacian's triangles (the block-local persistent launch), for
`rocprofv3 --kernel-trace --stats -- python tools/tri_prof.py` (profiles/r05_tri_kernel_stats.csv via tools/rocpd_top_kernels.py)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "sparse-linear-algebra_amd"))
import sla_amd as sla
from sla_amd import _lib, workloads as wl
dims, (rp, ci, va) = wl.laplace3d(216, 216, 216)
ctx = sla.default_context(); n = dims[0]
T = sla.fromCSR(dims, rp, ci, va, ctx)
b = sla.DeviceVector(ctx, n, np.ones(n)); x = sla.DeviceVector(ctx, n)
lib = _lib.lib()
for upper in (0, 1):
    for _ in range(11):
        _lib.check(lib.sla_tri_solve(T.h, upper, b.h, x.h, None))
ctx.sync()
print("form", ctx.get_option("tri_mode_used"), ctx.get_option("tri_plan"))
