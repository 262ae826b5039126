import sys, os, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "sparse-linear-algebra_amd"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import sla_amd as sla
from sla_amd import workloads as wl
for name, (dims, csr) in (("poisson2d 1000x1000", wl.poisson2d(1000, 1000)), ("laplace3d 100^3", wl.laplace3d(100, 100, 100)), ("poisson2d 300x300", wl.poisson2d(300, 300))):
    n = dims[0]
    b = np.ones(n); x0 = np.zeros(n)
    for mode in (1, 0, 1, 0):
        ctx = sla.Context(0).set_options(onchip=mode)
        t0 = time.perf_counter()
        A = sla.fromCSR(dims, *csr, ctx)
        ctx.sync()
        t1 = time.perf_counter()
        x, info = sla.linSolve0(sla.BICGSTAB_, A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx), return_info=True)
        t2 = time.perf_counter()
        x, info = sla.linSolve0(sla.BICGSTAB_, A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx), return_info=True)
        t3 = time.perf_counter()
        print(f"{name:22s} onchip={mode}: lower {1e3*(t1-t0):7.1f} ms | first linSolve0 {1e3*(t2-t1):7.1f} ms | second {1e3*(t3-t2):7.1f} ms | iters {info['iters']} launches {ctx.get_option('onchip_launches')} plan {float(ctx.get_option('onchip_plan_ms')):.1f} ms", flush=True)
        del A
        ctx.close()
