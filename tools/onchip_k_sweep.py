"""sla_solver_step(k) on chip against the launch flow as a function of k (config 2): a persistent launch loads and stores the state once per CALL, so a caller
that steps one at a time (the reference's pure `iterate (bicgstabStep aa r0hat)`) pays that per step.  us per step, median of 5 windows of 240 steps."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sparse-linear-algebra_amd"))
import numpy as np
import sla_amd as sla
from sla_amd import workloads as wl
for name, (dims, (rp, ci, va)) in (("poisson2d 1000^2", wl.poisson2d(1000, 1000)), ("laplace3d 64^3", wl.laplace3d(64, 64, 64))):
    n = dims[0]
    for method in ("bicgstab", "cgs"):
        for k in (1, 2, 3, 4, 8, 24, 240):
            row = []
            for onchip in (1, 0):
                ctx = sla.Context(0).set_options(onchip=onchip)
                A = sla.fromCSR(dims, rp, ci, va, ctx)
                b = sla.DeviceVector(ctx, n, np.add.reduceat(va, rp[:-1]), local=True)
                st = (sla.bicgsInit if method == "bicgstab" else sla.cgsInit)(A, b, sla.DeviceVector(ctx, n))
                for _ in range(8): st.step(k)
                ctx.sync()
                dts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    for _ in range(240 // k): st.step(k)
                    ctx.sync()
                    dts.append(time.perf_counter() - t0)
                row.append(sorted(dts)[2] / (240 // k * k) * 1e6)
                del st, A
                ctx.close()
            print(f"{name:18s} {method:8s} k={k:4d}  on chip {row[0]:7.2f} us/step   launch flow {row[1]:7.2f} us/step", flush=True)
