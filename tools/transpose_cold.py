import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/sparse-linear-algebra_amd")
import numpy as np
import bench
import sla_amd as sla
for name in ("laplace3d_10m", "random_spd_1m"):
    desc, (dims, (rp, ci, va)) = bench.workload(name)
    ctx = sla.Context(0)
    t0 = time.time(); A = sla.fromCSR(dims, rp, ci, va, ctx); ctx.sync(); t1 = time.time()
    n = dims[0]
    b = sla.DeviceVector(ctx, n, np.ones(n)); x = sla.DeviceVector(ctx, n, np.zeros(n))
    t2 = time.time(); y = sla.DeviceVector(ctx, n); sla._lib.check(sla._lib.lib().sla_spmv_t(A.h, b.h, y.h)); ctx.sync(); t3 = time.time()
    sla._lib.check(sla._lib.lib().sla_spmv_t(A.h, b.h, y.h)); ctx.sync(); t4 = time.time()
    print(f"{name}: from_csr {t1 - t0:.3f} s; first (<#) {t3 - t2:.3f} s (builds the transpose); second {1e3 * (t4 - t3):.3f} ms", flush=True)
    del A, b, x, y
    ctx.close()
