#!/usr/bin/env python
"""Do the LDS floating-point atomics of the CU-wide tile kernel keep denormal products?  A 300 000-row matrix of 3 random columns per
row with values 1e-310 (denormal), x = 1: every row's sum is 3e-310 exactly in the reference's fold.
    python tools/denormal_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sparse-linear-algebra_amd")):
    sys.path.insert(0, p)
import numpy as np
import sla_amd as sla

n, k = 300000, 3
rng = np.random.default_rng(1)
ci = np.sort(rng.integers(0, n, (n, k)), axis=1)
ci[:, 1] = np.where(ci[:, 1] == ci[:, 0], (ci[:, 1] + 1) % n, ci[:, 1])
ci[:, 2] = np.where(ci[:, 2] <= ci[:, 1], np.minimum(ci[:, 1] + 1 + np.arange(n) % 7, n - 1), ci[:, 2])
ci = np.sort(ci, axis=1)
ok = (ci[:, 0] < ci[:, 1]) & (ci[:, 1] < ci[:, 2])
ci, m = ci[ok], int(ok.sum())
rp = np.arange(0, 3 * m + 1, 3, dtype=np.int64)
for val, label in ((1e-310, "denormal products (1e-310)"), (3e-308, "products just above the normal threshold (3e-308)"), (1e-300, "normal products")):
    va = np.full(3 * m, val)
    for relaxed in (1, 0):
        ctx = sla.Context(0).set_options(tile_relaxed=relaxed)
        A = sla.fromCSR((m, n), rp, ci.ravel().astype(np.int64), va, ctx)
        y = sla.matVec(A, sla.fromVector(np.ones(n), ctx)).toDenseListSV()
        want = (val + val) + val
        print(f"{label:52s} tile_relaxed={relaxed} {A.kernel_info().split()[0]} exact_fold={'exact_fold=1' in A.kernel_info()}: "
              f"rows equal to the fold {int(np.count_nonzero(y == want))} of {m}; zeros {int(np.count_nonzero(y == 0.0))}; min {y.min():.3e} max {y.max():.3e}")
        del A
        ctx.close()
