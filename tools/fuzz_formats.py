"""Seeded fuzzer of the array-layout entries (round 6): random rectangular matrices (empty rows / columns, single rows, ragged last CSB blocks, repeated
coordinates) through sla_csr_from_csc / sla_csr_export_csc / sla_csr_transpose / sla_csr_from_csb, each against the oracle's restatement -- index arrays and
values bit for bit -- and (#>) / (<#) of the result against the oracle's folds (bit for bit where the lowered form's fold is exact).
usage: python tools/fuzz_formats.py [cases] [seed]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sparse-linear-algebra_amd"))
import numpy as np
import sla_amd as sla
from oracle import oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 606)
forms = {}


def same(A, Ao):
    rp, ci, va = A.csr()
    return np.array_equal(rp, Ao.rowptr) and np.array_equal(ci, Ao.colidx) and np.array_equal(va, Ao.val)


def mv_ok(A, got, want):
    return np.array_equal(got, want) if A.props()["fold"] == 0 else np.allclose(got, want, rtol=1e-12, atol=1e-12)


for case in range(cases):
    m = int(rng.choice([1, 2, 7, 130, 1000, 5000, 40000, 150000]) * rng.uniform(0.5, 1.5)) + 1
    n = m if rng.random() < 0.4 else int(rng.choice([1, 3, 100, 3000, 60000]) * rng.uniform(0.5, 1.5)) + 1
    per = float(rng.choice([0.3, 2, 7, 25, 90]))
    k = int(min(m * per, 0.5 * m * n, 3e6))
    key = np.sort(rng.choice(m * n, size=k, replace=False)) if k else np.zeros(0, np.int64)
    r, c, v = key // n, key % n, rng.standard_normal(k)
    rc, Ao = orc.coo_to_csr(m, n, r, c, v)
    To = orc.transpose(Ao)
    cp, ri, va = orc.to_csc(m, n, r, c, v)
    assert np.array_equal(cp, To.rowptr) and np.array_equal(ri, To.colidx)
    A = sla.fromCSC((m, n), cp, ri, va)
    assert same(A, Ao), ("from_csc", case, m, n, k)
    cp2, ri2, va2 = A.csc()
    assert np.array_equal(cp2, cp) and np.array_equal(ri2, ri) and np.array_equal(va2, va), ("export_csc", case)
    T = sla.transpose(A)
    assert T.dims == (n, m) and same(T, To), ("transpose", case, m, n, k)
    B = sla.fromCSR((m, n), Ao.rowptr, Ao.colidx, Ao.val)
    cp3, ri3, va3 = B.csc()                                    # export of a matrix that arrived as CSR: the device sort
    assert np.array_equal(cp3, cp) and np.array_equal(ri3, ri) and np.array_equal(va3, va), ("export_csc of a CSR arrival", case)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    if k:
        assert mv_ok(A, sla.matVec(A, sla.fromVector(x)).toDenseListSV(), orc.spmv(Ao, x)), ("(#>)", case, A.kernel_info())
        assert mv_ok(T, sla.matVec(T, sla.fromVector(y)).toDenseListSV(), orc.spmv(To, y)), ("(#>) of the transpose", case, T.kernel_info())
        assert np.allclose(sla.vecMat(sla.fromVector(y), A).toDenseListSV(), orc.spmv(To, y), rtol=1e-12, atol=1e-12), ("(<#)", case)
    forms[A.kernel_info().split()[0]] = forms.get(A.kernel_info().split()[0], 0) + 1
    # CSB: the same triplets plus repeats, in a random order, through the reference's binning
    beta = int(rng.choice([1, 3, 16, 64, 512, 4096, 1 << 20]))
    if (-(-m // beta)) * (-(-n // beta)) <= 4000000:
        extra = rng.integers(0, max(k, 1), min(k, 50)) if k else np.zeros(0, np.int64)
        rr, cc, vv = np.concatenate((r, r[extra])), np.concatenate((c, c[extra])), np.concatenate((v, rng.standard_normal(len(extra))))
        p = rng.permutation(len(rr))
        bp, rx, cx, vx = orc.to_csb((m, n), beta, rr[p], cc[p], vv[p])
        i, j, xx = orc.csb_to_coo((m, n), beta, bp, rx, cx, vx)
        rc, Co = orc.coo_to_csr(m, n, i, j, xx)
        C = sla.fromCSB((m, n), beta, bp, rx, cx, vx)
        assert rc == orc.OK and same(C, Co), ("from_csb", case, m, n, beta)
    del A, T, B
print(f"formats fuzz ok: {cases} cases; forms of the CSC arrivals {forms}")
