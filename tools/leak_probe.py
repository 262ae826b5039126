import os, sys
import numpy as np
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "sparse-linear-algebra_amd")):
    sys.path.insert(0, p)
import torch
import sla_amd as sla
from sla_amd import workloads as wl
def free():
    torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 1e6
dims, (rp, ci, va) = wl.laplace3d(60, 60, 60)
dimsb, (rpb, cib, vab) = wl.banded_nonsym(200000)
dimsd, (rpd, cid, vad) = wl.random_spd(20000, 60, 3)      # ~120 entries per row: LDS-panel form (table, partials, task runs)
n = dims[0]
b = np.ones(n)
f0 = None
for it in range(60):
    ctx = sla.Context(0)
    A = sla.fromCSR(dims, rp, ci, va, ctx)
    B = sla.fromCSR(dimsb, rpb, cib, vab, ctx)
    x = sla.linSolve0(sla.BICGSTAB_, A, sla.fromVector(b, ctx), sla.fromVector(np.zeros(n), ctx))
    y = sla.vecMat(sla.fromVector(b, ctx), A)
    t = sla.triLowerSolve(A, sla.DeviceVector(ctx, n, b))
    g = sla.gmres(B, sla.fromVector(np.ones(dimsb[0]), ctx), sla.fromVector(np.zeros(dimsb[0]), ctx), restart=10)
    L, R = sla.mSsorPre(A, 1.0)
    D = sla.fromCSR(dimsd, rpd, cid, vad, ctx)
    assert "ldspanels" in D.kernel_info()
    xd = sla.linSolve0(sla.CGNE_, D, sla.fromVector(np.ones(dimsd[0]), ctx), sla.fromVector(np.zeros(dimsd[0]), ctx))
    del A, B, x, y, t, g, L, R, D, xd
    ctx.close()
    if it == 5: f0 = free()
print("free MB after warm-up:", f0, "at end:", free(), "delta:", f0 - free())
