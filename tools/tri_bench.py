#!/usr/bin/env python
"""Times sla_tri_solve (level-scheduled substitution, HIP-graph replay) with the lower / upper triangle of the
216^3 7-pt Laplacian (10 M rows, 646 levels) and of the 1000^2 5-pt Poisson matrix (1999 levels)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-linear-algebra_amd"))
import sla_amd as sla  # noqa: E402
from sla_amd import _lib, workloads as wl  # noqa: E402

lib = _lib.lib()
ctx = sla.default_context()
for name, (dims, (rp, ci, va)) in (("laplace3d 216^3", wl.laplace3d(216, 216, 216)), ("poisson2d 1000^2", wl.poisson2d(1000, 1000))):
    n = dims[0]
    T = sla.fromCSR(dims, rp, ci, va, ctx)
    rows = np.repeat(np.arange(n), np.diff(rp))
    b = sla.DeviceVector(ctx, n, np.ones(n))
    x = sla.DeviceVector(ctx, n)
    for upper in (0, 1):
        t0 = time.perf_counter()
        lv, wd = sla.triSolveLevels(T, bool(upper))
        t_plan = time.perf_counter() - t0
        nnz_tri = int(((ci >= rows) if upper else (ci <= rows)).sum())
        _lib.check(lib.sla_tri_solve(T.h, upper, b.h, x.h, None))      # capture
        ctx.sync()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            _lib.check(lib.sla_tri_solve(T.h, upper, b.h, x.h, None))
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        bytes_ = 12 * nnz_tri + 28 * n                                   # val + col of the triangle, rowptr, order, b, x
        print("%-18s %s: %5d levels (widest %7d rows), schedule built in %.2f s; solve %.3f ms = %.2f us/level, %.0f GB/s of %d MB"
              % (name, "upper" if upper else "lower", lv, wd, t_plan, dt * 1e3, dt * 1e6 / lv, bytes_ / dt / 1e9, bytes_ // 10**6), flush=True)
        # round 5: the same solve as one persistent launch (option tri_syncfree), by grid size; must give the same bits
        ref = x.to_host()
        for grid in (64, 128, 256, 512, 1024, 2048):
            ctx.set_options(tri_syncfree=1, tri_grid=grid)
            _lib.check(lib.sla_tri_solve(T.h, upper, b.h, x.h, None))
            ctx.sync()
            same = bool(np.array_equal(x.to_host(), ref))
            t0 = time.perf_counter()
            for _ in range(5):
                _lib.check(lib.sla_tri_solve(T.h, upper, b.h, x.h, None))
            ctx.sync()
            dts = (time.perf_counter() - t0) / 5
            print("    persistent launch, %4d workgroups: %.3f ms (bit-identical: %s; fallbacks so far: %s)" % (grid, dts * 1e3, same, ctx.get_option("tri_fallbacks")), flush=True)
        ctx.set_options(tri_syncfree=0)
