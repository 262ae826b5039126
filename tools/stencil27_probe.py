#!/usr/bin/env python
"""27-point stencil (constant coefficients) on N^3: which SpMV form wins when a slice has 27 records (8 pipelined + 19 in the tail)?"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-linear-algebra_amd"))
import sla_amd as sla  # noqa: E402
from sla_amd import _lib, workloads as wl  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = N ** 3
offs, dxyz = [], []
for dz in (-1, 0, 1):
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            offs.append(dz * N * N + dy * N + dx)
            dxyz.append((dx, dy, dz))
order = np.argsort(offs)
offs = [offs[i] for i in order]
dxyz = [dxyz[i] for i in order]


def valid(rows, t):
    dx, dy, dz = dxyz[t]
    i, j, k = rows % N, (rows // N) % N, rows // (N * N)
    return (i + dx >= 0) & (i + dx < N) & (j + dy >= 0) & (j + dy < N) & (k + dz >= 0) & (k + dz < N)


def value(rows, t):
    return np.full(len(rows), 26.0 if offs[t] == 0 else -1.0)


rp, ci, va = wl._stencil_rows(0, n, offs, valid, value)
lib = _lib.lib()
for env in ({}, {"SLA_WDIA": "0"}, {"SLA_WDIA": "0", "SLA_VDICT": "0"}):
    for k in ("SLA_WDIA", "SLA_VDICT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx = sla.Context(0)
    A = sla.fromCSR((n, n), rp, ci, va, ctx)
    xs = [sla.DeviceVector(ctx, n, np.full(n, 1.0 + i)) for i in range(4)]
    ys = [sla.DeviceVector(ctx, n) for _ in range(4)]
    for i in range(4):
        _lib.check(lib.sla_spmv(A.h, xs[i].h, ys[i].h))
    ctx.sync()
    t0 = time.perf_counter()
    for k in range(40):
        _lib.check(lib.sla_spmv(A.h, xs[k % 4].h, ys[k % 4].h))
    ctx.sync()
    dt = (time.perf_counter() - t0) / 40
    print("%-24s %-28s %8.1f us  (%d rows, %d nnz)" % (env, A.kernel_info().split()[0], dt * 1e6, n, len(ci)), flush=True)
    del A
    ctx.close()
