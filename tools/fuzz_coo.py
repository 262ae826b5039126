#!/usr/bin/env python
"""Fuzz fromListSM lowering (host builder and the rocPRIM device sort) against the oracle: duplicates (last wins),
unsorted input, empty rows, rectangular shapes.  Run with SLA_DEVICE_COO_MIN=1 to force the device path."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sparse-linear-algebra_amd")):
    sys.path.insert(0, p)
import sla_amd as sla  # noqa: E402
from oracle import oracle as orc  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
for case in range(cases):
    m, n = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    nnz = int(rng.integers(0, 4000))
    r, c = rng.integers(0, m, nnz), rng.integers(0, n, nnz)
    if case % 3 == 0 and nnz:
        c = c % max(1, n // 8)                 # many duplicates
    v = rng.standard_normal(nnz)
    rc, Ao = orc.coo_to_csr(m, n, r.astype(np.int64), c.astype(np.int64), v)
    A = sla.fromCOO((m, n), r, c, v) if hasattr(sla, "fromCOO") else sla.fromListSM((m, n), list(zip(r.tolist(), c.tolist(), v.tolist())))
    rp, ci, va = A.csr()
    assert np.array_equal(rp, Ao.rowptr) and np.array_equal(ci, Ao.colidx) and np.array_equal(va.view(np.uint64), Ao.val.view(np.uint64)), case
    x = rng.standard_normal(n)
    y = sla.matVec(A, sla.fromVector(x)).toDenseListSV()
    assert np.allclose(y, orc.spmv(Ao, x), rtol=1e-12, atol=1e-12), case
print("coo fuzz ok:", cases, "device path forced" if os.environ.get("SLA_DEVICE_COO_MIN") else "")
