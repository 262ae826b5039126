#!/usr/bin/env python
"""Probe: plain SpMV (y_i = A x_i) over R rotating vector pairs, so that the vectors do NOT stay in the
256 MB MALL between launches, for stencils with and without far diagonals.  Prints us per launch."""
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


def stencil(n, offsets):
    offsets = sorted(offsets)

    def valid(rows, t):
        c = rows + offsets[t]
        return (c >= 0) & (c < n)

    def value(rows, t):
        return np.full(len(rows), 6.0 if offsets[t] == 0 else -1.0)

    return (n, n), wl._stencil_rows(0, n, offsets, valid, value)


def run(name, dims, csr, R, reps=40):
    A = sla.fromCSR(dims, *csr, ctx)
    n = dims[0]
    xs = [sla.DeviceVector(ctx, n, np.full(n, 1.0 + i)) for i in range(R)]
    ys = [sla.DeviceVector(ctx, n) for i in range(R)]
    for i in range(R):
        _lib.check(lib.sla_spmv(A.h, xs[i].h, ys[i].h))
    ctx.sync()
    t0 = time.perf_counter()
    for k in range(reps):
        i = k % R
        _lib.check(lib.sla_spmv(A.h, xs[i].h, ys[i].h))
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print("%-28s R=%d  %7.1f us/launch  %s" % (name, R, dt * 1e6, A.kernel_info().split()[0]), flush=True)


n = 216 ** 3
cases = {
    "tridiag(-1,0,1)": [-1, 0, 1],
    "diag only": [0],
    "5 near (-2..2)": [-2, -1, 0, 1, 2],
    "7pt (1,216,46656)": [-46656, -216, -1, 0, 1, 216, 46656],
    "3 far (0,+-46656)": [-46656, 0, 46656],
    "3 (0,+-216)": [-216, 0, 216],
    "7 aligned (16,208,46656)": [-46656, -208, -16, 0, 16, 208, 46656],
}
sel = [a for a in sys.argv[1:] if a in cases] or ([] if sys.argv[1:] else list(cases))
for name in sel:
    dims, csr = stencil(n, cases[name])
    for R in (1, 6):
        run(name, dims, csr, R)


def run_axpby(R, reps=60):
    xs = [sla.DeviceVector(ctx, n, np.full(n, 1.0 + i)) for i in range(R)]
    ys = [sla.DeviceVector(ctx, n, np.full(n, 2.0 + i)) for i in range(R)]
    for i in range(R):
        _lib.check(lib.sla_axpby(C.c_double(0.5), xs[i].h, C.c_double(0.25), ys[i].h))
    ctx.sync()
    t0 = time.perf_counter()
    for k in range(reps):
        _lib.check(lib.sla_axpby(C.c_double(0.5), xs[k % R].h, C.c_double(0.25), ys[k % R].h))
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print("axpby (24 B/row = %.0f MB)        R=%d  %7.1f us/launch  %.2f TB/s" % (24 * n / 1e6, R, dt * 1e6, 24 * n / dt / 1e12), flush=True)


if not sys.argv[1:] or "axpby" in sys.argv[1:]:
    for R in (1, 6):
        run_axpby(R)


def run_scal(R, reps=60):
    xs = [sla.DeviceVector(ctx, n, np.full(n, 1.0 + i)) for i in range(R)]
    for i in range(R):
        _lib.check(lib.sla_scal(C.c_double(1.0000001), xs[i].h))
    ctx.sync()
    t0 = time.perf_counter()
    for k in range(reps):
        _lib.check(lib.sla_scal(C.c_double(1.0000001), xs[k % R].h))
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print("scal  (16 B/row = %.0f MB)        R=%d  %7.1f us/launch  %.2f TB/s" % (16 * n / 1e6, R, dt * 1e6, 16 * n / dt / 1e12), flush=True)


def run_copy(R, reps=60):
    xs = [sla.DeviceVector(ctx, n, np.full(n, 1.0 + i)) for i in range(R)]
    ys = [sla.DeviceVector(ctx, n) for i in range(R)]
    for i in range(R):
        _lib.check(lib.sla_vec_copy(xs[i].h, ys[i].h))
    ctx.sync()
    t0 = time.perf_counter()
    for k in range(reps):
        _lib.check(lib.sla_vec_copy(xs[k % R].h, ys[k % R].h))
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print("copy  (16 B/row = %.0f MB)        R=%d  %7.1f us/launch  %.2f TB/s" % (16 * n / 1e6, R, dt * 1e6, 16 * n / dt / 1e12), flush=True)


if not sys.argv[1:] or "scal" in sys.argv[1:]:
    for R in (1, 6, 12):
        run_scal(R)
    for R in (1, 6):
        run_copy(R)
