#!/usr/bin/env python
"""Fuzz sla_tri_solve against the oracle: random sparse matrices (random patterns, both triangles present), both sweeps,
alternating right-hand-side buffers (graph re-capture), missing / tiny diagonal entries."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sparse-linear-algebra_amd")):
    sys.path.insert(0, p)
import sla_amd as sla  # noqa: E402
from oracle import oracle as orc  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
for case in range(cases):
    n = int(rng.integers(1, 1200))
    k = int(rng.integers(0, 8))
    rows = np.repeat(np.arange(n), k)
    cols = rng.integers(0, n, n * k) if case % 2 else np.clip(rows + rng.integers(-30, 31, n * k), 0, n - 1)
    vals = rng.uniform(-1, 1, n * k)
    diag = rng.uniform(1.0, 3.0, n) * rng.choice([-1, 1], n)
    bad = -1
    if case % 5 == 0 and n > 3:
        bad = int(rng.integers(0, n))
        diag[bad] = 1e-13 if case % 10 == 0 else 0.0          # tiny / missing
    keep = np.ones(n, bool)
    if bad >= 0 and case % 10 != 0:
        keep[bad] = False
    r = np.concatenate([rows, np.arange(n)[keep]])
    c = np.concatenate([cols, np.arange(n)[keep]])
    v = np.concatenate([vals, diag[keep]])
    off = r != c
    # keep one entry per (row, col): fromListSM keeps the last; give the diagonal entries the last word
    rc, Ao = orc.coo_to_csr(n, n, np.concatenate([r[off], r[~off]]), np.concatenate([c[off], c[~off]]), np.concatenate([v[off], v[~off]]))
    T = sla.fromCSR((n, n), Ao.rowptr, Ao.colidx, Ao.val)
    b1, b2 = rng.standard_normal(n), rng.standard_normal(n)
    d1, d2 = sla.DeviceVector(T.ctx, n, b1), sla.DeviceVector(T.ctx, n, b2)
    for upper in (False, True):
        f_o = orc.tri_upper_solve if upper else orc.tri_lower_solve
        f_d = sla.triUpperSolve if upper else sla.triLowerSolve
        rc1, w1, bad_o = f_o(Ao, b1)
        if rc1 != orc.OK:
            try:
                f_d(T, d1)
                raise AssertionError(("expected NeedsPivoting", case, upper))
            except sla.NeedsPivoting as ex:
                assert "(%d,%d)" % (bad_o, bad_o) in str(ex), (str(ex), bad_o)
            continue
        rc2, w2, _ = f_o(Ao, b2)
        for d, w in ((d1, w1), (d2, w2), (d1, w1), (d1, w1)):
            x = f_d(T, d).to_host()
            assert np.array_equal(x.view(np.uint64), w.view(np.uint64)) or (np.isnan(w).any() and np.array_equal(np.isnan(x), np.isnan(w))), (case, upper, n, k)
    del T
print("tri fuzz ok:", cases)
