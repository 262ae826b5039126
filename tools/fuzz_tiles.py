"""Randomised check of the tile form (spmv_tile_kernel, both builders) against the oracle's left fold, BIT FOR BIT: random shapes, row-length
distributions (uniform, skewed, empty stretches, a few rows of hundreds of entries: deep layers), random panel widths, 32- / 64-bit row
pointers, the device builder against the host builder, every pacing / poll variant, non-finite x.  (Round 4: layers sorted by column, layer
starts flagged in bit 31 of the index dword.)   python tools/fuzz_tiles.py [cases] [seed]"""
import sys

sys.path.insert(0, "sparse-linear-algebra_amd"); sys.path.insert(0, ".")
import numpy as np
import sla_amd as sla
from oracle import oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 404)
taken = 0
for case in range(cases):
    m = int(rng.integers(1, 9000))
    n = int(rng.choice([rng.integers(3000, 20000), rng.integers(20000, 200000)]))
    base = int(rng.integers(1, 40))
    kind = case % 4
    if kind == 0:
        lens = rng.integers(0, 2 * base + 1, m)
    elif kind == 1:
        lens = np.where(rng.random(m) < 0.98, rng.integers(0, 6, m), rng.integers(100, 700, m))      # a few long rows: deep layers
    elif kind == 2:
        lens = np.full(m, base)
        lens[rng.random(m) < 0.3] = 0
    else:
        lens = rng.integers(base // 2, base + 1, m)
        lens[: m // 3] = 0
    lens = np.minimum(lens, n).astype(np.int64)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    if rp[-1] == 0:
        continue
    ci = np.concatenate([np.sort(rng.choice(n, size=int(k), replace=False)) for k in lens] + [np.zeros(0, np.int64)]).astype(np.int64)
    va = rng.standard_normal(len(ci)) * 10.0 ** rng.integers(-3, 4, len(ci))
    x = rng.standard_normal(n)
    if case % 7 == 0:
        x[rng.integers(0, n, 3)] = [np.inf, -np.inf, np.nan]
    with np.errstate(all="ignore"):
        want = orc.spmv(orc.Csr(m, n, rp, ci, va), x)
    shift = int(rng.integers(10, 14))
    # round 5: the CU-wide relaxed-order kernel (tile_relaxed = 1) on the same case -- rows of finite data within nnz_i eps sum |a_ij x_j|,
    # rows of <= 2 entries and NaN / Inf placement exact, device and host builders alike
    with np.errstate(all="ignore"):
        bound = lens * np.finfo(np.float64).eps * orc.spmv(orc.Csr(m, n, rp, ci, np.abs(va)), np.abs(x))
    for dev in (2, 0):
        ctx = sla.Context(0).set_options(tile_shift=shift, lpanel=0, lflat=0, tiles_device=dev, force_rp64=1 if case % 3 == 0 else 0, tile_relaxed=1)
        A = sla.fromCSR((m, n), rp, ci, va, ctx)
        if "cu_slices=1" in A.kernel_info():
            relaxed_taken = globals().get("relaxed_taken", 0) + 1
            for opts in ({}, {"tile_slack": 0}):
                ctx.set_options(**opts)
                with np.errstate(all="ignore"):
                    y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
                fin = np.isfinite(want) & np.isfinite(bound)
                assert np.array_equal(np.isnan(y), np.isnan(want)) and np.array_equal(y[np.isinf(want)], want[np.isinf(want)]), (case, dev, "relaxed: non-finite rows")
                assert np.all(np.abs(y[fin] - want[fin]) <= bound[fin]), (case, dev, opts, shift, "relaxed: bound")
                assert np.array_equal(y[fin & (lens <= 2)], want[fin & (lens <= 2)]), (case, dev, "relaxed: short rows")
        del A
        ctx.close()
    got = {}
    # the exact forms: rows owned by wavefronts inside CU-wide slices (round 6, tile_rowown = -1: what tile_relaxed = 0 selects) and the
    # wavefront-private slices (tile_rowown = 0); both builders each
    for dev, ro in ((2, -1), (0, -1), (2, 0), (0, 0)):
        ctx = sla.Context(0).set_options(tile_shift=shift, lpanel=0, lflat=0, tiles_device=dev, force_rp64=1 if case % 3 == 0 else 0, tile_relaxed=0, tile_rowown=ro)
        A = sla.fromCSR((m, n), rp, ci, va, ctx)
        info = A.kernel_info()
        if "algo=tiles" not in info:
            del A
            ctx.close()
            break
        if dev == 2 and ro == 0:
            taken += 1
        for opts in ({}, {"tile_poll": 0}, {"tile_slack": 0}, {"tile_prefetch": 2}):
            ctx.set_options(**opts)
            with np.errstate(all="ignore"):
                y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
            keys = np.nonzero(lens > 0)[0]
            assert np.array_equal(y.view(np.uint64)[keys], want.view(np.uint64)[keys]) or \
                (np.array_equal(np.isnan(y), np.isnan(want)) and np.array_equal(y[~np.isnan(want)], want[~np.isnan(want)])), (case, dev, opts, shift, info)
        got[(dev, ro)] = y
        del A
        ctx.close()
    for ro in (-1, 0):
        if (0, ro) in got and (2, ro) in got:
            a, b = got[(0, ro)], got[(2, ro)]
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]), (case, ro, "builders differ")
print(f"tile fuzz ok: {cases} cases, exact tile form taken in {taken}, CU-wide relaxed form in {globals().get('relaxed_taken', 0)}")
