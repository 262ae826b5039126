"""cgneStep rate (Sparse.hs:855-878) on a workload: python tools/cgne_rate.py [workload] [steps]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "sparse-linear-algebra_amd")
import numpy as np
import bench
import sla_amd as sla
name = sys.argv[1] if len(sys.argv) > 1 else "laplace3d_10m"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
desc, (dims, (rp, ci, va)) = bench.workload(name)
ctx = sla.Context(0)
A = sla.fromCSR(dims, rp, ci, va, ctx)
n = dims[0]
b = sla.DeviceVector(ctx, n, np.add.reduceat(va, rp[:-1]))
x0 = sla.DeviceVector(ctx, n)
t0 = time.perf_counter()
st = sla.cgneInit(A, b, x0)
st.step(4)
ctx.sync()
t1 = time.perf_counter()
st.step(steps)
ctx.sync()
t2 = time.perf_counter()
print(f"{name}: cgneInit + 4 steps {t1 - t0:.3f} s (builds the transpose); {steps / (t2 - t1):.1f} cgneStep/s = {1e6 * (t2 - t1) / steps:.1f} us per step; {A.kernel_info().split()[0]}")
