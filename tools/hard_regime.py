#!/usr/bin/env python3
"""Parity in the hard regime: the reference's only real-world fixture (test/data/e05r0000.mtx, test/Perf.hs)
and the ill-conditioned user system of issues/issue_denjoh.hs (1000 x 1000 beam stiffness matrix, entries ~1e9 .. 7e12).

For every (fixture, x0) the residual trace ||A x_j - b|| of linSolve0 (sla_solve_opts.history) is compared iteration by iteration with

  ref      the oracle's bicgstabStep / cgsStep sequence (the reference's formulas, left-fold sums)
  ident    the oracle with beta's numerator through (s . r0hat) - omega (aas . r0hat)   (BiCGSTAB only: the product's fused sweep)

for the device's fused (bicg_fuse45 = 1, default) and split (= 0, the reference's K4 / K5) flows.  J(a, b) = the first iteration where
traces a and b differ by more than 1e-6 relative.  Prints one table per case and a summary; tests/test_gpu_hard_regime.py asserts
what this measures.  Usage (GPU box):  python tools/hard_regime.py [> profiles/r04_hard_regime.txt]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sparse-linear-algebra_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import oracle as orc  # noqa: E402
from refdata import GOLDEN, denjoh_beam, read_mtx_array, read_mtx_coordinate  # noqa: E402


def fixtures():
    dims, r, c, v = read_mtx_coordinate(f"{GOLDEN}/e05r0000.mtx")
    b = read_mtx_array(f"{GOLDEN}/e05r0000_rhs1.mtx")
    yield "e05r0000", dims, r, c, v, b
    dims, r, c, v, b = denjoh_beam()
    yield "denjoh_beam", dims, r, c, v, b


def oracle_trace(method, Ao, b, x0, nit, rho_identity=False):
    """(residual norms after each step, iterations linSolve0 would take, flags) with the reference's stopping rule (Sparse.hs:1034-1052)."""
    r0 = b - orc.spmv(Ao, x0)
    tol = max(1e-6, 1e-4 * np.linalg.norm(r0))
    st = (orc.BicgstabState if method == "BICGSTAB_" else orc.CgsState)(Ao, b, x0)
    seq = []
    with np.errstate(all="ignore"):
        for _ in range(nit):
            if method == "BICGSTAB_":
                st.step(r0, 1, rho_identity=rho_identity)
            else:
                st.step(r0, 1)
            seq.append(np.linalg.norm(orc.spmv(Ao, st.x) - b))
            if not np.isfinite(seq[-1]) or seq[-1] <= tol:
                break
    seq = np.array(seq)
    flag = "nonfinite" if not np.isfinite(seq[-1]) else "converged" if seq[-1] <= tol else "max_iters"
    return seq, tol, flag


def first_split(a, b, thr=1e-6):
    n = min(len(a), len(b))
    with np.errstate(all="ignore"):
        rel = np.abs(a[:n] - b[:n]) / np.abs(b[:n])
    bad = ~(rel <= thr)
    return int(np.argmax(bad)) if bad.any() else n


def device_trace(sla, method, dims, r, c, v, b, x0, fuse):
    ctx = sla.Context(0).set_option("bicg_fuse45", fuse)
    A = sla.fromCOO(dims, r, c, v, ctx)
    x, info = sla.linSolve0(getattr(sla, method), A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx), return_info=True, history=True)
    hist = info["history"]
    flag = "nonfinite" if info["flags"] & 16 else "converged" if info["converged"] else "max_iters"
    xd = x.toDenseListSV()
    del A
    ctx.close()
    return hist, flag, xd, info


def flag_of(info):
    return "nonfinite" if info["flags"] & 16 else "converged" if info["converged"] else "max_iters"


def run(sla, out=sys.stdout):
    summary = []
    for name, dims, r, c, v, b in fixtures():
        n = dims[0]
        rc, Ao = orc.coo_to_csr(n, n, r, c, v)
        for x0v in (0.0, 0.1):
            x0 = np.full(n, x0v)
            for method in ("BICGSTAB_", "CGS_"):
                ref, tol, fref = oracle_trace(method, Ao, b, x0, 200)
                row = {"case": f"{name} x0={x0v} {method}", "tol": tol, "ref_iters": len(ref), "ref_flag": fref}
                traces = {"ref": ref}
                if method == "BICGSTAB_":
                    ident, _, fid = oracle_trace(method, Ao, b, x0, 200, rho_identity=True)
                    traces["ident"] = ident
                    row["ident_iters"], row["ident_flag"] = len(ident), fid
                    row["J(ident,ref)"] = first_split(ident, ref)
                for fuse in ((1, 0) if method == "BICGSTAB_" else (1,)):
                    h, fl, xd, info = device_trace(sla, method, dims, r, c, v, b, x0, fuse)
                    key = "dev_fused" if fuse else "dev_split"
                    if method == "CGS_":
                        key = "dev"
                    traces[key] = h
                    row[key + "_iters"], row[key + "_flag"] = len(h), fl
                    row[f"J({key},ref)"] = first_split(h, ref)
                    if method == "BICGSTAB_":
                        row[f"J({key},ident)"] = first_split(h, traces["ident"])
                    with np.errstate(all="ignore"):
                        row[key + "_true_res"] = float(np.linalg.norm(orc.spmv(Ao, xd) - b))
                summary.append(row)
                print(f"\n== {row['case']}  tol={tol:.6e}", file=out)
                keys = list(traces)
                print("   j  " + "  ".join(f"{k:>22s}" for k in keys) + "   rel-to-ref: " + " ".join(f"{k:>9s}" for k in keys[1:]), file=out)
                L = max(len(t) for t in traces.values())
                for j in list(range(min(L, 48))) + [j for j in range(48, L) if j % 10 == 9 or j == L - 1]:
                    vals = [traces[k][j] if j < len(traces[k]) else float("nan") for k in keys]
                    with np.errstate(all="ignore"):
                        rels = [abs(x - vals[0]) / abs(vals[0]) for x in vals[1:]]
                    print(f"{j + 1:4d}  " + "  ".join(f"{x:22.15e}" for x in vals) + "                " + " ".join(f"{x:9.2e}" for x in rels), file=out)
        # (<\>) = GMRES from x0 = 0.1 * ones (dead instance Sparse.hs:1080-1084) and explicit GMRES(60): x against the oracle's x
        ctx = sla.Context(0)
        A = sla.fromCOO(dims, r, c, v, ctx)
        for label, restart, cycles in (("<\\>", 30, 7), ("gmres(60)", 60, 34)):
            if label == "<\\>":
                x, info = sla.linSolve(A, sla.fromVector(b, ctx), return_info=True)
            else:
                x, info = sla.gmres(A, sla.fromVector(b, ctx), sla.fromVector(np.full(n, 0.1), ctx), restart=restart, return_info=True, max_iters=2000)
            rco, xo, it_o, res_o, r0_o = orc.gmres(Ao, b, np.full(n, 0.1), restart=restart, max_restarts=cycles)
            xd = x.toDenseListSV()
            row = {"case": f"{name} {label}", "dev_iters": info["iters"], "orc_iters": it_o, "dev_res": info["resnorm"], "orc_res": res_o,
                   "dev_flag": flag_of(info), "|x-xo|/|xo|": float(np.linalg.norm(xd - xo) / np.linalg.norm(xo)),
                   "dev_true_res": float(np.linalg.norm(orc.spmv(Ao, xd) - b)), "orc_true_res": float(np.linalg.norm(orc.spmv(Ao, xo) - b))}
            summary.append(row)
        del A
        ctx.close()
    print("\n== summary", file=out)
    for row in summary:
        print("  " + "  ".join(f"{k}={v:.6e}" if isinstance(v, float) else f"{k}={v}" for k, v in row.items()), file=out)
    return summary


if __name__ == "__main__":
    import sla_amd
    run(sla_amd)
