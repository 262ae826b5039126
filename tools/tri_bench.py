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
def cases():
    yield "laplace3d 216^3", wl.laplace3d(216, 216, 216)
    yield "poisson2d 1000^2", wl.poisson2d(1000, 1000)
    if os.environ.get("TRI_BENCH_ZOO") == "1":
        yield "banded 2 M", wl.banded_nonsym(2_000_000)
        yield "random 1 M x 33", wl.random_spd(1_000_000, 16, 3)
        yield "laplace3d 100^3", wl.laplace3d(100, 100, 100)
        yield "poisson2d 3000^2", wl.poisson2d(3000, 3000)


for name, (dims, (rp, ci, va)) in cases():
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
        ctx.set_options(tri_syncfree=2 if lv > 100000 else 0)          # (that many dependent launches take seconds: the reference bits from the block form then)
        _lib.check(lib.sla_tri_solve(T.h, upper, b.h, x.h, None))      # capture
        ctx.sync()
        reps = 20 if lv <= 100000 else 1
        t0 = time.perf_counter()
        for _ in range(reps):
            _lib.check(lib.sla_tri_solve(T.h, upper, b.h, x.h, None))
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        bytes_ = 12 * nnz_tri + 28 * n                                   # val + col of the triangle, rowptr, order, b, x
        print("%-18s %s: %5d levels (widest %7d rows), schedule built in %.2f s; %s %.3f ms = %.2f us/level, %.0f GB/s of %d MB"
              % (name, "upper" if upper else "lower", lv, wd, t_plan, "level schedule" if lv <= 100000 else "(level schedule = that many dependent launches: not timed) block-local form",
                 dt * 1e3, dt * 1e6 / lv, bytes_ / dt / 1e9, bytes_ // 10**6), flush=True)
        # round 5: the same solve as one persistent launch (option tri_syncfree), by grid size; must give the same bits
        ref = x.to_host()
        fast = os.environ.get("TRI_BENCH_FAST") == "1"
        for mode, grid, brows in [(1, g, 0) for g in ((256,) if fast else (64, 128, 256, 512, 1024))] + [(2, 0, r) for r in ((16384,) if fast else (4096, 8192, 16384))]:
            if mode == 1 and lv > 100000:
                continue
            ctx.set_options(tri_syncfree=mode, tri_grid=grid)
            if brows:
                ctx.set_options(tri_block_rows=brows)
                t0 = time.perf_counter()
            _lib.check(lib.sla_tri_solve(T.h, upper, b.h, x.h, None))
            ctx.sync()
            t_first = time.perf_counter() - t0 if brows else 0.0
            same = bool(np.array_equal(x.to_host(), ref))
            t0 = time.perf_counter()
            for _ in range(5):
                _lib.check(lib.sla_tri_solve(T.h, upper, b.h, x.h, None))
            ctx.sync()
            dts = (time.perf_counter() - t0) / 5
            what = "rows poll x in memory" if mode == 1 else "blocks of %5d rows in LDS (plan + first solve %.2f s)" % (brows, t_first)
            print("    persistent launch, %4d workgroups, %s: %.3f ms (bit-identical: %s; fallbacks so far: %s)" % (grid, what, dts * 1e3, same, ctx.get_option("tri_fallbacks")), flush=True)
        ctx.set_options(tri_syncfree=3, tri_grid=0, tri_block_rows=16384)
        _lib.check(lib.sla_tri_solve(T.h, upper, b.h, x.h, None))
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5 if lv <= 100000 else 1):
            _lib.check(lib.sla_tri_solve(T.h, upper, b.h, x.h, None))
        ctx.sync()
        dta = (time.perf_counter() - t0) / (5 if lv <= 100000 else 1)
        print("    default (picked form %s): %.3f ms (bit-identical: %s)" % (ctx.get_option("tri_mode_used"), dta * 1e3, bool(np.array_equal(x.to_host(), ref))), flush=True)
