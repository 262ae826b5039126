#!/usr/bin/env python
"""One real run past 2^31 stored entries ("lowering fast at 10^9 nnz"; fromListSM, SpMatrix.hs:218-224).

The matrix: n = K * S rows, row i holds ONE pseudo-random column in every stripe [k S, (k + 1) S) of the columns, k = 0 .. K - 1 (ascending and
unique by construction), except that in its own stripe the column is i itself (the diagonal, value 2 + ...); off-diagonal values in (-1, 1) / K,
so every row is diagonally dominant.  Defaults: K = 220, S = 45455 -> n = 10 000 100 rows, nnz = 2 200 022 000 (> 2^31 = 2 147 483 648), 35 GB
of host arrays (int64 columns + f64 values), 26 GB on the device.

What it does, each phase timed:
  1. generates the CSR on the host (numpy, chunked, threads);
  2. sla_csr_from_csr -> "lowered once" (wall clock + sla_csr_lower_info phases), kernel_info / props (64-bit row pointers);
  3. (#>) against the oracle -- sampled rows through orc.spmv on the sub-matrix of those rows (default 4096 rows: the exact left fold), and
     with --full every row (the serial oracle: its left fold);
  4. bicgsInit + two bicgstabSteps against the oracle's (--full), iterates at 1e-9;
  5. (#>) event-timed: GB/s on 12 nnz + 8 (rows + 1) + 16 n bytes.
Scaled-down smoke: --k 8 --s 1000 (runs in seconds; the GPU suite runs that through tests/test_gpu_big_nnz.py with force_rp64)."""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sparse-linear-algebra_amd"))

M1, M2 = np.uint64(0x9E3779B97F4A7C15), np.uint64(0xBF58476D1CE4E5B9)


def _mix(h):
    h ^= h >> np.uint64(31)
    h *= M2
    h ^= h >> np.uint64(29)
    return h


def stripe_rows(K, S, r0, r1, col, val):
    """rows [r0, r1) written into col / val (views of (rows, K) shape)"""
    rows = np.arange(r0, r1, dtype=np.uint64)
    ks = np.arange(K, dtype=np.uint64)
    h = _mix((rows[:, None] * np.uint64(K) + ks[None, :]) * M1 + np.uint64(12345))
    c = (h % np.uint64(S)).astype(np.int64) + (ks.astype(np.int64) * S)[None, :]
    v = ((_mix(h * M1) >> np.uint64(11)).astype(np.float64) / float(1 << 53) * 2.0 - 1.0) / K
    own = (rows.astype(np.int64) // S)
    ar = np.arange(r1 - r0)
    c[ar, own] = rows.astype(np.int64)
    v[ar, own] = 2.0 + (rows % np.uint64(7)).astype(np.float64) * 0.125
    col[:] = c
    val[:] = v


def generate(K, S, threads):
    n = K * S
    col = np.empty((n, K), dtype=np.int64)
    val = np.empty((n, K), dtype=np.float64)
    chunk = max(1, min(n, (1 << 22) // K))
    jobs = [(r, min(n, r + chunk)) for r in range(0, n, chunk)]
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(lambda j: stripe_rows(K, S, j[0], j[1], col[j[0]:j[1]], val[j[0]:j[1]]), jobs))
    rp = np.arange(n + 1, dtype=np.int64) * K
    return (n, n), rp, col.reshape(-1), val.reshape(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=220)
    ap.add_argument("--s", type=int, default=45455)
    ap.add_argument("--sample", type=int, default=4096)
    ap.add_argument("--full", action="store_true", help="oracle (#>) on every row and two BiCGSTAB steps (minutes of CPU at 2.2e9 entries)")
    ap.add_argument("--threads", type=int, default=min(32, os.cpu_count() or 1))
    ap.add_argument("--options", default="", help="ctx options, k=v,k=v")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import sla_amd as sla
    from oracle import oracle as orc
    rec = {"K": args.k, "S": args.s}
    t0 = time.perf_counter()
    dims, rp, ci, va = generate(args.k, args.s, args.threads)
    n, nnz = dims[0], int(rp[-1])
    rec.update({"rows": n, "nnz": nnz, "nnz_over_2_31": nnz / 2.0 ** 31, "host_generation_s": time.perf_counter() - t0,
                "host_bytes": int(ci.nbytes + va.nbytes + rp.nbytes)})
    print(f"[big_nnz] n = {n}, nnz = {nnz} ({nnz / 2 ** 31:.3f} x 2^31), generated in {rec['host_generation_s']:.1f} s", file=sys.stderr, flush=True)
    ctx = sla.Context(0)
    if args.options:
        ctx.set_options(**dict(kv.split("=") for kv in args.options.split(",")))
    t0 = time.perf_counter()
    A = sla.fromCSR(dims, rp, ci, va, ctx)
    ctx.sync()
    rec["lowered_once_s"] = time.perf_counter() - t0
    rec["lower_phases_ms"] = A.lower_info()
    rec["kernel_info"] = A.kernel_info()
    rec["props"] = A.props()
    print(f"[big_nnz] lowered once in {rec['lowered_once_s']:.2f} s: {rec['kernel_info']}", file=sys.stderr, flush=True)
    assert (rec["props"]["rowptr_bits"] == 64) == (nnz >= 2 ** 31 or "force_rp64=1" in args.options), rec["props"]
    rng = np.random.default_rng(17)
    x = rng.standard_normal(n)
    xd = sla.DeviceVector(ctx, n, x)
    yd = sla.DeviceVector(ctx, n)
    from sla_amd import _lib
    _lib.check(_lib.lib().sla_spmv(A.h, xd.h, yd.h))
    y = yd.to_host()
    # sampled rows: the oracle's left fold on the sub-matrix of those rows
    rows = np.unique(np.concatenate((rng.integers(0, n, size=args.sample), [0, n - 1, n // 2])))
    K = args.k
    sub_rp = np.arange(len(rows) + 1, dtype=np.int64) * K
    idx = (rows[:, None] * K + np.arange(K)[None, :]).reshape(-1)
    sub = orc.Csr(len(rows), n, sub_rp, ci[idx], va[idx])
    want = orc.spmv(sub, x)
    absub = orc.Csr(len(rows), n, sub_rp, ci[idx], np.abs(va[idx]))
    bound = K * np.finfo(np.float64).eps * orc.spmv(absub, np.abs(x))
    d = np.abs(y[rows] - want)
    rec["spmv_sampled"] = {"rows": int(len(rows)), "bit_exact_rows": int((y[rows] == want).sum()), "max_abs_diff": float(d.max()),
                           "max_diff_over_bound": float((d / np.maximum(bound, 1e-300)).max()), "within_bound": bool(np.all(d <= bound))}
    print(f"[big_nnz] (#>) on {len(rows)} sampled rows: {rec['spmv_sampled']}", file=sys.stderr, flush=True)
    assert rec["spmv_sampled"]["within_bound"]
    if rec["props"]["fold"] == 0:
        assert rec["spmv_sampled"]["bit_exact_rows"] == len(rows), "fold = EXACT but rows differ from the left fold"
    # two BiCGSTAB steps: against the oracle when --full, and always a consistency check through the true residual
    xs = rng.standard_normal(n)
    xsd = sla.DeviceVector(ctx, n, xs)          # (held: a temporary would be released before the call reads its handle)
    _lib.check(_lib.lib().sla_spmv(A.h, xsd.h, yd.h))
    b = yd.to_host()
    del xsd
    bd, x0d = sla.DeviceVector(ctx, n, b), sla.DeviceVector(ctx, n)
    st = sla.bicgsInit(A, bd, x0d)
    t0 = time.perf_counter()
    st.step(2)
    ctx.sync()
    rec["two_steps_s_incl_first_launches"] = time.perf_counter() - t0
    xv = sla.DeviceVector(ctx, n)
    _lib.check(_lib.lib().sla_solver_get(st.h, 0, xv.h))
    x2 = xv.to_host()
    _lib.check(_lib.lib().sla_solver_get(st.h, 1, xv.h))
    r2 = xv.to_host()
    x2d = sla.DeviceVector(ctx, n, x2)
    _lib.check(_lib.lib().sla_spmv(A.h, x2d.h, yd.h))
    true_r = b - yd.to_host()
    del x2d
    rec["bicgstab_two_steps"] = {"r0norm": float(np.linalg.norm(b)), "recurrence_resnorm": float(np.linalg.norm(r2)), "true_resnorm": float(np.linalg.norm(true_r)),
                                 "recurrence_vs_true": float(np.linalg.norm(r2 - true_r) / np.linalg.norm(b)),
                                 "error_vs_xstar": float(np.linalg.norm(x2 - xs) / np.linalg.norm(xs))}
    print(f"[big_nnz] two bicgstabSteps: {rec['bicgstab_two_steps']}", file=sys.stderr, flush=True)
    assert rec["bicgstab_two_steps"]["recurrence_vs_true"] <= 1e-12 and rec["bicgstab_two_steps"]["true_resnorm"] < 1e-2 * rec["bicgstab_two_steps"]["r0norm"]
    if args.full:
        Ao = orc.Csr(n, n, rp, ci, va)
        t0 = time.perf_counter()
        yo = orc.spmv(Ao, x)
        rec["oracle_spmv_s"] = time.perf_counter() - t0
        absA = orc.Csr(n, n, rp, ci, np.abs(va))
        bnd = K * np.finfo(np.float64).eps * orc.spmv(absA, np.abs(x))
        del absA
        d = np.abs(y - yo)
        rec["spmv_full"] = {"rows": n, "bit_exact_rows": int((y == yo).sum()), "max_diff_over_bound": float((d / np.maximum(bnd, 1e-300)).max()),
                            "within_bound": bool(np.all(d <= bnd))}
        print(f"[big_nnz] (#>) on all rows: {rec['spmv_full']}", file=sys.stderr, flush=True)
        assert rec["spmv_full"]["within_bound"]
        bo = orc.spmv(Ao, xs)
        so = orc.BicgstabState(Ao, bo, np.zeros(n))
        t0 = time.perf_counter()
        so.step(bo.copy(), 2)
        rec["oracle_two_steps_s"] = time.perf_counter() - t0
        rec["bicgstab_two_steps"]["x_vs_oracle"] = float(np.linalg.norm(x2 - so.x) / np.linalg.norm(so.x))
        rec["bicgstab_two_steps"]["r_vs_oracle"] = float(np.linalg.norm(r2 - so.r) / np.linalg.norm(bo))
        print(f"[big_nnz] vs the oracle's two steps: x {rec['bicgstab_two_steps']['x_vs_oracle']:.2e}, r {rec['bicgstab_two_steps']['r_vs_oracle']:.2e}", file=sys.stderr, flush=True)
        assert rec["bicgstab_two_steps"]["x_vs_oracle"] <= 1e-9
    # timing: (#>) alone, HIP events
    ctx.prof_start(_lib.KERNEL_SPMV, args.reps)
    for _ in range(args.reps):
        _lib.check(_lib.lib().sla_spmv(A.h, xd.h, yd.h))
    cnt, mean_ms, min_ms = ctx.prof_stop()
    by = 12 * nnz + (8 if rec["props"]["rowptr_bits"] == 64 else 4) * (n + 1) + 16 * n
    rec["spmv"] = {"launches": cnt, "ms": mean_ms, "min_ms": min_ms, "csr_bytes": by, "gbps": by / mean_ms / 1e6, "frac_of_8TBs": by / mean_ms / 1e6 / 8000.0}
    ctx.prof_start(_lib.KERNEL_ALL, 64)
    st.step(4)
    ctx.prof_stop()
    rec["step_kernels_ms"] = {name: ctx.prof_query(kid)[1] for name, kid in (("K1", _lib.KERNEL_SPMV_DOT), ("K2", _lib.KERNEL_BICG_K2), ("K3", _lib.KERNEL_SPMV_DOT2),
                                                                              ("K45", _lib.KERNEL_BICG_K45)) if ctx.prof_query(kid)[0]}
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
