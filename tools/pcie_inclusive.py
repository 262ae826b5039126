#!/usr/bin/env python
"""PCIe-inclusive timing of the boundary: lower the 10M-row Laplacian once (host arrays -> device CSR), then
linSolve0 with host b / x0 uploaded and x downloaded inside the timed call (what the Haskell shim would do)."""
import sys, time
sys.path.insert(0, "sparse-linear-algebra_amd")
import numpy as np
import sla_amd as sla
from sla_amd import workloads as wl

dims, (rp, ci, va) = wl.laplace3d(216, 216, 216)
n = dims[0]
t = time.perf_counter(); A = sla.fromCSR(dims, rp, ci, va); sla.default_context().sync(); t_lower = time.perf_counter() - t
b = np.add.reduceat(va, rp[:-1]); x0 = np.zeros(n)
for label, kw in (("200 iterations (tol 0)", dict(max_iters=200, tol_abs=0.0, tol_rel=0.0)), ("reference defaults", {})):
    t = time.perf_counter()
    x, info = sla.linSolve0(sla.BICGSTAB_, A, sla.fromVector(b), sla.fromVector(x0), return_info=True, **kw)
    dt = time.perf_counter() - t
    print(f"{label}: {info['iters']} iterations in {dt*1e3:.1f} ms incl. upload of b, x0 and download of x "
          f"= {info['iters']/dt:.1f} it/s PCIe-inclusive; resnorm {info['resnorm']:.3e} tol {info['tol']:.3e}")
print(f"lowering (host CSR arrays -> device, incl. validation, row blocks, column dictionary): {t_lower*1e3:.0f} ms for {len(ci)} entries")
