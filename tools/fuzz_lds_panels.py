"""Randomised check of the LDS-panel SpMV (spmv_lpanel_kernel) against the oracle's left fold: random shapes, row-length
distributions (uniform, skewed, empty stretches), every lane-group shape forced onto every segment length (so that the
multi-round loop of narrow groups meets long segments and wide groups meet tiny ones), 32- and 64-bit row pointers.
python tools/fuzz_lds_panels.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, "sparse-linear-algebra_amd"); sys.path.insert(0, ".")
import numpy as np
import sla_amd as sla
from oracle import oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2025)
seen = {}
for case in range(cases):
    m = int(rng.integers(1, 6000))
    n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 20000), rng.integers(20000, 90000)]))
    base = int(rng.integers(1, 400))
    kind = case % 4
    if kind == 0:
        lens = rng.integers(0, 2 * base + 1, m)
    elif kind == 1:
        lens = np.where(rng.random(m) < 0.9, rng.integers(0, 6, m), rng.integers(base, 6 * base + 1, m))
    elif kind == 2:
        lens = np.full(m, base)
        lens[rng.random(m) < 0.3] = 0
    else:
        lens = rng.integers(base // 2, base + 1, m)
        lens[: m // 3] = 0
    lens = np.minimum(lens, n).astype(np.int64)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ci = np.concatenate([np.sort(rng.choice(n, size=int(k), replace=False)) for k in lens] + [np.zeros(0, np.int64)]).astype(np.int64)
    va = rng.standard_normal(len(ci)) * 10.0 ** rng.integers(-3, 4, len(ci))
    x = rng.standard_normal(n)
    if case % 7 == 0 and n > 3:
        x[rng.integers(0, n, 3)] = [np.inf, -np.inf, np.nan]
    Ao = orc.Csr(m, n, rp, ci, va)
    with np.errstate(all="ignore"):
        want = orc.spmv(Ao, x)
        absum = orc.spmv(orc.Csr(m, n, rp, ci, np.abs(va)), np.abs(np.where(np.isfinite(x), x, 0.0)))
    os.environ["SLA_LP_MINSEG"] = "1"
    os.environ["SLA_LP_CFG"] = str(case % 5 - 1)            # -1: the lowering's own choice
    os.environ["SLA_FORCE_RP64"] = "1" if case % 3 == 0 else "0"
    os.environ["SLA_LP_TASKS"] = str(int(rng.choice([1, 4, 32, 200])))
    ctx = sla.Context(0)
    A = sla.fromCSR((m, n), rp, ci, va, ctx)
    info = A.kernel_info()
    key = (info.split()[0], "lanes=" + (info.split("lanes_per_segment=")[1].split()[0] if "lanes_per_segment" in info else "-"), "i64" if "rowptr=i64" in info else "i32")
    seen[key] = seen.get(key, 0) + 1
    y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
    fin = np.isfinite(want)
    assert np.array_equal(np.isnan(y), np.isnan(want)), (case, info)
    assert np.array_equal(y[np.isinf(want)], want[np.isinf(want)]), (case, info)
    bound = (lens + 64) * 1.2e-16 * absum + 1e-300
    # rows that reference a non-finite x are exempt from the magnitude bound (they are checked for class above)
    assert np.all(np.abs(y[fin] - want[fin]) <= bound[fin]), (case, info, np.abs(y[fin] - want[fin]).max())
    del A
    ctx.close()
print("lds-panel fuzz ok:", cases, "cases;", dict(sorted(seen.items())))
