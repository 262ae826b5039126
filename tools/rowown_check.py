#!/usr/bin/env python
"""Round 6 experiment: the CU-wide tile kernel on a layout where every row of a slice is owned by ONE wavefront
(option tile_rowown = 1, host builder).  Are the row sums reproducible from run to run, and are they the reference's ascending left fold?
(timing of the same layout on config 3a: tools/tile_bench.py "SLA_TILES_DEVICE=0 SLA_TILE_ROWOWN=1")"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sparse-linear-algebra_amd")):
    sys.path.insert(0, p)
import numpy as np
import sla_amd as sla
from sla_amd import workloads as wl
from oracle import oracle as orc

for n, k in ((700000, 4), (1000000, 16)):
    dims, (rp, ci, va) = wl.random_spd(n, k, 1)
    Ao = orc.Csr(n, n, rp, ci, va)
    x = np.random.default_rng(2).standard_normal(n)
    yo = orc.spmv(Ao, x)
    for opts in ({"tile_rowown": 0}, {"tile_rowown": 1, "tiles_device": 0}):
        ctx = sla.Context(0).set_options(**opts)
        A = sla.fromCSR(dims, rp, ci, va, ctx)
        ys = [sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV() for _ in range(6)]
        same = all(np.array_equal(ys[0], y) for y in ys[1:])
        exact = int((ys[0] == yo).sum())
        print(f"n={n} k={k} {opts}: {A.kernel_info().split()[0]} reruns bit-identical: {same}; rows equal to the oracle's left fold: {exact} of {n} "
              f"(max |dy| {np.abs(ys[0] - yo).max():.2e})", flush=True)
        del A
        ctx.close()
