#!/usr/bin/env python
"""Cost of the split launches of the overlapped all-gather, measured on ONE GPU.

One rank's N = 8 slab of BASELINE config 3a -- rows [r n/8, (r+1) n/8) of the 10 M-row random matrix against all 10 M columns -- is
lowered on a single-rank context that REHEARSES the pass structure of rank r of 8 (options ag_sim_ranks / ag_sim_rank: the tile launch
runs as the plan's panel passes, running row sums carried through HBM between them, no exchange and no waiting -- x is whole and
local), and (#>) is timed against the single launch (overlap = -1) on the same box.  The difference is what the overlap has to win
back: per extra pass a launch, a pipeline fill / drain, one pacing restart and 16 B per row of running sums.
usage: python tools/ag_split_bench.py [--ranks 8] [--rank 3] [--n 10000000] [--groups 1,2,4,8]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sparse-linear-algebra_amd")):
    sys.path.insert(0, p)
import numpy as np
import sla_amd as sla
from sla_amd import _lib, workloads as wl
from sla_amd.partition import row_block

ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--rank", type=int, default=3)
ap.add_argument("--n", type=int, default=10000000)
ap.add_argument("--groups", default="1,2,4,8")
ap.add_argument("--reps", type=int, default=24)
args = ap.parse_args()
n = args.n
rb, re_ = row_block(n, args.rank, args.ranks)
t0 = time.time()
dims, (rp, ci, va) = wl.random_spd_rows(n, 16, 42, rb, re_, threads=16)
rows = re_ - rb
print(f"# slab of rank {args.rank}/{args.ranks}: rows {rb}..{re_} ({rows}) x {n} columns, {rp[-1]} entries, assembled in {time.time() - t0:.1f} s", flush=True)
alg = 12 * int(rp[-1]) + 12 * rows + 8 * n      # entries + rowptr / y per row + x once
lib = _lib.lib()


def run(label, **opts):
    ctx = sla.Context(0).set_options(**opts)
    A = sla.fromCSR((rows, n), rp, ci, va, ctx)
    pairs = 3
    xs = [sla.DeviceVector(ctx, n, np.full(n, 1.0 + 0.125 * i)) for i in range(pairs)]
    ys = [sla.DeviceVector(ctx, rows) for _ in range(pairs)]
    for i in range(4):
        _lib.check(lib.sla_spmv(A.h, xs[i % pairs].h, ys[i % pairs].h))
    ctx.sync()
    ctx.prof_start(_lib.KERNEL_SPMV, args.reps)
    for i in range(args.reps):
        _lib.check(lib.sla_spmv(A.h, xs[i % pairs].h, ys[i % pairs].h))
    _, ms, mn = ctx.prof_stop()
    info = A.kernel_info()
    y = ys[(args.reps - 1) % pairs].to_host()
    print(f"{label:34s} (#>) {ms:.4f} ms (min {mn:.4f})  {alg / ms / 1e6:7.1f} GB/s | {info[info.find('slices'):]}", flush=True)
    del xs, ys, A
    ctx.close()
    return ms, y


base, y0 = run("single launch (overlap=-1)", overlap=-1)
for g in [int(t) for t in args.groups.split(",")]:
    ms, y = run(f"arrival passes, {g} group(s)", ag_sim_ranks=args.ranks, ag_sim_rank=args.rank, ag_groups=g)
    print(f"    -> +{(ms - base) * 1e3:.1f} us = {100 * (ms / base - 1):+.1f} % ; rows differing from the ascending fold: {int(np.count_nonzero(y != y0))} (max {np.abs(y - y0).max():.2e})", flush=True)
ms, y = run("ascending passes (source-ordered)", ag_sim_ranks=args.ranks, ag_sim_rank=args.rank, ag_order=1)
print(f"    -> +{(ms - base) * 1e3:.1f} us = {100 * (ms / base - 1):+.1f} % ; bit-identical to the single launch: {bool(np.array_equal(y, y0))}", flush=True)
