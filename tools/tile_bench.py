#!/usr/bin/env python
"""One process, one matrix, many knob settings: (#>) timings per configuration (config 3a unless a 3-D grid is named).
usage: python tools/tile_bench.py [n | laplace3d:G] "SLA_TILE_SLACK=2 SLA_TILE_SHIFT=17" "SLA_TILES=0" ...   (env knobs are read per Context)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sparse-linear-algebra_amd")):
    sys.path.insert(0, p)
import numpy as np
import sla_amd as sla
from sla_amd import _lib, workloads as wl

args = sys.argv[1:]
t0 = time.time()
if args and args[0].startswith("laplace3d:"):
    g = int(args.pop(0).split(":")[1])
    dims, (rp, ci, va) = wl.laplace3d(g, g, g)
    n = dims[0]
    print(f"# laplace3d {g}^3 n={n} nnz={rp[-1]} generated in {time.time() - t0:.1f} s", flush=True)
else:
    n = int(args.pop(0)) if args and args[0].isdigit() else 10000000
    dims, (rp, ci, va) = wl.random_spd(n, 16, 42)
    print(f"# random_spd n={n} nnz={rp[-1]} generated in {time.time() - t0:.1f} s", flush=True)
alg = 12 * int(rp[-1]) + 20 * n
for cfg in args or ["DEFAULT=1"]:
    saved = {}
    for kv in cfg.split():
        k, v = kv.split("=")
        saved[k] = os.environ.get(k)
        os.environ[k] = v
    t0 = time.time()
    ctx = sla.Context(0)
    A = sla.fromCSR(dims, rp, ci, va, ctx)
    tl = time.time() - t0
    lib = _lib.lib()
    pairs = 4
    xs = [sla.DeviceVector(ctx, n, np.full(n, 1.0 + 0.125 * i)) for i in range(pairs)]
    ys = [sla.DeviceVector(ctx, n) for _ in range(pairs)]
    for i in range(4):
        _lib.check(lib.sla_spmv(A.h, xs[i % pairs].h, ys[i % pairs].h))
    ctx.sync()
    reps = 16
    ctx.prof_start(_lib.KERNEL_SPMV, reps)
    for i in range(reps):
        _lib.check(lib.sla_spmv(A.h, xs[i % pairs].h, ys[i % pairs].h))
    _, rot_ms, rot_min = ctx.prof_stop()
    ctx.prof_start(_lib.KERNEL_SPMV, reps)
    for i in range(reps):
        _lib.check(lib.sla_spmv(A.h, xs[0].h, ys[0].h))
    _, one_ms, one_min = ctx.prof_stop()
    info = A.kernel_info()
    print(f"{cfg:60s} spmv {rot_ms:.3f} ms (min {rot_min:.3f}; one pair {one_ms:.3f})  {alg / rot_ms / 1e6:7.1f} GB/s = {alg / rot_ms / 8e9:.3f} of peak | lower {tl:.1f} s | {info.split()[0]} {info[info.find('slices'):] if 'slices' in info else ''}", flush=True)
    del xs, ys, A, ctx
    for k, v in saved.items():
        if v is None:
            del os.environ[k]
        else:
            os.environ[k] = v
