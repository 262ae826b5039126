import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sparse-linear-algebra_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
os.environ["SLA_TILE_SHIFT"] = "10"
import sla_amd as sla
from oracle import oracle as orc
from test_gpu_tiles import _rand_rows
for name, build in (("6/row", lambda: _rand_rows(5000, 5000, lambda i, r: 6, 1)), ("1/row", lambda: _rand_rows(150000, 150000, lambda i, r: 1, 4))):
    dims, (rp, ci, va) = build()
    m, n = dims
    x = np.random.default_rng(11).standard_normal(n)
    want = orc.spmv(orc.Csr(m, n, rp, ci, va), x)
    for slack in ("0", "1"):
        os.environ["SLA_TILE_SLACK"] = slack
        ctx = sla.Context(0)
        A = sla.fromCSR(dims, rp, ci, va, ctx)
        y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
        bad = np.nonzero(y != want)[0]
        print(name, "slack", slack, A.kernel_info()[-90:], "mismatches", len(bad), bad[:40])
        for i in bad[:6]:
            terms = va[rp[i]:rp[i+1]] * x[ci[rp[i]:rp[i+1]]]
            print("  row", i, "cols", ci[rp[i]:rp[i+1]], "got", y[i], "want", want[i], "partial sums", np.cumsum(terms), "terms", terms)
    if name == "6/row":
        import itertools
        shift = 10
        srow = [0]
        # same slicing as the lowering: 79 slices by entry target
        S0 = (m + 63) // 64; target = (rp[-1] + S0 - 1) // S0; r = 0
        while r < m:
            rcap = min(m, r + 1024); e = int(np.searchsorted(rp[r + 1:rcap + 1], rp[r] + target, side="right")) + r + 1
            e = max(r + 1, min(e, rcap)); srow.append(e); r = e
        srow = np.array(srow)
        for i in bad[:12]:
            s = int(np.searchsorted(srow, i, side="right")) - 1
            r0, r1 = srow[s], srow[s + 1]
            terms = va[rp[i]:rp[i+1]] * x[ci[rp[i]:rp[i+1]]]
            found = None
            for k in range(len(terms) + 1):
                for sub in itertools.combinations(range(len(terms)), k):
                    if abs(sum(terms[list(sub)]) - y[i]) < 1e-12: found = sub
            # lane position of each entry inside its tile
            pos = []
            for k in range(rp[i], rp[i + 1]):
                j = ci[k] >> shift
                before = sum(int(np.count_nonzero((ci[rp[q]:rp[q+1]] >> shift) == j)) for q in range(r0, i)) + int(np.count_nonzero((ci[rp[i]:k] >> shift) == j))
                tile_n = sum(int(np.count_nonzero((ci[rp[q]:rp[q+1]] >> shift) == j)) for q in range(r0, r1))
                pos.append((int(j), before, tile_n))
            print("  row", i, "slice rows", r0, r1, "kept terms", found, "(panel, lane-in-tile, tile entries)", pos)
