#!/usr/bin/env python
"""Seeded fuzzer of the on-chip solver path (csrc/sla_onchip.hip, round 6): random constant-coefficient banded / stencil matrices (1 ... 8 (offset, value)
pairs, random sizes, ragged rows, 2-D / 3-D grids of odd extents), random plan options (workgroups, rows per workgroup, bricks), random step counts --
bicgstabStep and cgsStep on chip against the launch flow (2 steps: 1e-10) and the oracle (2 steps: 1e-9), step(k) against k x step(1) bit for bit, and
linSolve0 on chip against the launch flow's verdict / iteration count / trace head.   python tools/fuzz_onchip.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sparse-linear-algebra_amd")):
    sys.path.insert(0, p)
import numpy as np
import sla_amd as sla
from sla_amd import workloads as wl
from oracle import oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 606)
rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)   # noqa: E731
taken = {"bicgstab": 0, "cgs": 0, "linsolve0": 0, "declined": 0}
for case in range(cases):
    kind = rng.integers(0, 3)
    opts = {}
    if kind == 0:      # band: random offsets around a dominant diagonal
        n = int(rng.integers(200, 40000))
        k = int(rng.integers(1, 8))
        offs = sorted(set([0] + [int(o) for o in rng.integers(-min(n // 3, 400), min(n // 3, 400) + 1, size=k)]))
        vals = [(-1.0 - 0.25 * rng.random()) if o else 0.0 for o in offs]
        vals[offs.index(0)] = 1.0 + sum(abs(v) for v in vals) + rng.random()
        drop = rng.random((n, len(offs))) < (0.15 if rng.random() < 0.4 else 0.0)

        def valid(rows, t, offs=offs, n=n, drop=drop):
            c = rows + offs[t]
            return (c >= 0) & (c < n) & (~drop[rows, t] | (offs[t] == 0))
        dims, csr = (n, n), wl._stencil_rows(0, n, offs, valid, lambda rows, t, vals=vals: np.full(len(rows), vals[t]))
    elif kind == 1:
        nx, ny = int(rng.integers(5, 160)), int(rng.integers(5, 160))
        dims, csr = wl.poisson2d(nx, ny)
        if rng.random() < 0.5:
            opts["onchip_bricks"] = 2
    else:
        nx, ny, nz = int(rng.integers(4, 40)), int(rng.integers(4, 40)), int(rng.integers(3, 30))
        dims, csr = wl.laplace3d(nx, ny, nz)
        if rng.random() < 0.6:
            opts["onchip_bricks"] = 2
    if rng.random() < 0.5:
        opts["onchip_grid"] = int(rng.integers(1, 257))
    if rng.random() < 0.3:
        opts["onchip_rows"] = int(rng.integers(64, 4000))
    n = dims[0]
    rp, ci, va = csr
    Ao = orc.Csr(n, n, rp, ci, va)
    b = orc.spmv(Ao, np.ones(n)) + 0.05 * rng.standard_normal(n)
    x0 = 0.1 * rng.standard_normal(n)
    k2 = int(rng.integers(1, 6))
    for meth in ("bicgstab", "cgs"):
        init = sla.bicgsInit if meth == "bicgstab" else sla.cgsInit
        xf = "_xBicgstab" if meth == "bicgstab" else "_x"
        got = {}
        for mode in (1, 0):
            ctx = sla.Context(0).set_options(onchip=mode, **(opts if mode else {}))
            A = sla.fromCSR(dims, rp, ci, va, ctx)
            s = init(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx)).step(2)
            x2 = getattr(s, xf).toDenseListSV()
            on = int(ctx.get_option("onchip_launches"))
            s.step(k2)
            xk = getattr(s, xf).toDenseListSV()
            if mode and on:
                s1 = init(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
                for _ in range(2 + k2):
                    s1.step(1)
                assert np.array_equal(getattr(s1, xf).toDenseListSV(), xk), (case, meth, "step(k) != k x step(1)", ctx.get_option("onchip_plan"))
            got[mode] = (x2, xk, on, ctx.get_option("onchip_plan"))
            del s, A
            ctx.close()
        so = orc.BicgstabState(Ao, b, x0) if meth == "bicgstab" else orc.CgsState(Ao, b, x0)
        so.step(b - orc.spmv(Ao, x0), 2)
        if got[1][2]:
            taken[meth] += 1
            assert rel(got[1][0], so.x) <= 1e-9, (case, meth, "oracle", rel(got[1][0], so.x), got[1][3])
            assert rel(got[1][0], got[0][0]) <= 1e-10, (case, meth, "launch flow", rel(got[1][0], got[0][0]), got[1][3])
        else:
            taken["declined"] += 1
    # linSolve0 on chip against the launch flow
    res = {}
    for mode in (1, 0):
        ctx = sla.Context(0).set_options(onchip=mode, **(opts if mode else {}))
        A = sla.fromCSR(dims, rp, ci, va, ctx)
        x, info = sla.linSolve0(sla.BICGSTAB_, A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx), return_info=True, history=True)
        res[mode] = (x.toDenseListSV(), info, int(ctx.get_option("onchip_launches")))
        del A
        ctx.close()
    if res[1][2]:
        taken["linsolve0"] += 1
        i1, i0 = res[1][1], res[0][1]
        assert i1["converged"] == i0["converged"], (case, i1, i0)
        slack = max(3, (15 * i0["iters"]) // 100)
        if abs(i1["iters"] - i0["iters"]) > slack:
            # a long, erratic BiCGSTAB run: is the spread the method's own?  The launch flow's OTHER formulation (the reference's split K4 / K5 with
            # its literal beta, bicg_fuse45 = 0) is a third evaluation of the same recurrences: the on-chip count must lie inside the launch flows' spread
            ctx = sla.Context(0).set_options(onchip=0, bicg_fuse45=0)
            A = sla.fromCSR(dims, rp, ci, va, ctx)
            _, i2 = sla.linSolve0(sla.BICGSTAB_, A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx), return_info=True)
            del A
            ctx.close()
            lo, hi = min(i0["iters"], i2["iters"]), max(i0["iters"], i2["iters"])
            print(f"# case {case}: iterations on chip {i1['iters']}, launch flow fused {i0['iters']}, launch flow split {i2['iters']}", flush=True)
            assert lo - slack <= i1["iters"] <= hi + slack, (case, i1["iters"], i0["iters"], i2["iters"])
        m = min(len(i1["history"]), len(i0["history"]), 8)
        assert np.allclose(i1["history"][:m], i0["history"][:m], rtol=1e-8), (case, i1["history"][:m], i0["history"][:m])
        if i1["converged"]:
            assert np.linalg.norm(orc.spmv(Ao, res[1][0]) - b) <= i1["tol"] * (1 + 1e-9), case
print(f"on-chip fuzz ok: {cases} cases; on chip: {taken}")
