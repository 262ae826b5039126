"""Helpers that rebuild the reference's literal test matrices from tests/golden/ (data only)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden():
    with open(os.path.join(GOLDEN, "reference_vectors.json")) as f:
        return json.load(f)


def coo_from_dense_colmajor(m, ll, sparsify=False):
    """fromListDenseSM m ll (SpMatrix.hs:239-241, indexed2 Utils.hs:85-90): entry k -> (k % m, k // m).
    sparsify=True mirrors sparsifySM (drops |x| <= 1e-12)."""
    n = len(ll) // m
    rows, cols, vals = [], [], []
    for k, x in enumerate(ll[: m * n]):
        if sparsify and abs(x) <= 1e-12:
            continue
        rows.append(k % m)
        cols.append(k // m)
        vals.append(float(x))
    return (m, n), np.array(rows, dtype=np.int64), np.array(cols, dtype=np.int64), np.array(vals)


def coo_from_triples(dims, triples):
    t = np.array(triples, dtype=np.float64).reshape(-1, 3)
    return tuple(dims), t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2].copy()


def coo_of(entry):
    if "coo" in entry:
        return coo_from_triples(entry["dims"], entry["coo"])
    return coo_from_dense_colmajor(entry["m"], entry["dense_colmajor"], entry.get("sparsify", False))


def tridiag_coo(n, lo, d, up):
    """tm7 = mkSubDiagonal n 1 .. ^+^ mkSubDiagonal n 0 .. ^+^ mkSubDiagonal n (-1) .. (LibSpec.hs:1334-1339)"""
    rows, cols, vals = [], [], []
    for i in range(n):
        if i > 0:
            rows.append(i); cols.append(i - 1); vals.append(lo)
        rows.append(i); cols.append(i); vals.append(d)
        if i < n - 1:
            rows.append(i); cols.append(i + 1); vals.append(up)
    return (n, n), np.array(rows, dtype=np.int64), np.array(cols, dtype=np.int64), np.array(vals, dtype=np.float64)


def read_mtx_coordinate(path):
    """Loader semantics of test/Perf.hs:20-45: 1-based -> 0-based, entries in file order."""
    rows, cols, vals = [], [], []
    with open(path) as f:
        header = f.readline()
        assert header.startswith("%%MatrixMarket matrix coordinate real general")
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        m, n, nz = (int(t) for t in line.split())
        for line in f:
            t = line.split()
            if not t:
                continue
            rows.append(int(t[0]) - 1); cols.append(int(t[1]) - 1); vals.append(float(t[2]))
    assert len(rows) == nz
    return (m, n), np.array(rows, dtype=np.int64), np.array(cols, dtype=np.int64), np.array(vals)


def read_mtx_array(path):
    with open(path) as f:
        f.readline()
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        m, n = (int(t) for t in line.split())
        vals = [float(l) for l in f if l.strip()]
    assert len(vals) == m * n
    return np.array(vals)


def dense_of(A):
    """Dense ndarray of an oracle Csr (tests only)."""
    D = np.zeros((A.m, A.n))
    for i in range(A.m):
        for k in range(A.rowptr[i], A.rowptr[i + 1]):
            D[i, A.colidx[k]] = A.val[k]
    return D


def denjoh_beam(nel=500, length=100.0, inertia=400000000.0, emod=210000.0, spring=2000.0):
    """The user system of issues/issue_denjoh.hs:60-71 re-created as a generator (no values copied: the element formulas of :7-23
    evaluated here): `nel` Euler-Bernoulli beam elements of equal length / inertia, 2 degrees of freedom per node (deflection,
    rotation), assembled the way `createStiffnessMatrix` does -- every (row, col) appears ONCE in the triple list, the shared
    2 x 2 node blocks as `k (a+2) (b+2) + k a b` (so their off-diagonal entries are explicit 0.0 entries, kept by fromListSM) --
    then `insertSprings` adds `spring` to the diagonal entries 200, 400 .. 1000 (the supports), and the solved system is the
    submatrix of rows / columns 2 .. 2 nel + 1 shifted by -2 (the clamped first node removed) with the right-hand side
    `dropSV 2 fVec`: 5000 at the last deflection, 8.19e8 at the last rotation.  Entries are ~1e9 .. 7e12: the ill-conditioned case.
    Returns (dims, rows, cols, vals, b) of the 2 nel x 2 nel system; triples in assembly order (order is irrelevant: no duplicates).
    Operation order of every entry follows the Haskell expression (`12 * eMod * i / (l ** 3)` = ((12 * eMod) * i) / pow l 3)."""
    E, i, l = float(emod), float(inertia), float(length)
    k00 = 12.0 * E * i / (l ** 3)
    k01 = 6.0 * E * i / (l ** 2)
    k11 = 4.0 * E * i / l
    k13 = 2.0 * E * i / l
    ke = [[k00, k01, -k00, k01],
          [k01, k11, -k01, k13],
          [-k00, -k01, k00, -k01],
          [k01, k13, -k01, k11]]
    ndof = 2 * nel + 2
    ent = {}

    def put(r, c, v):
        assert (r, c) not in ent        # createStiffnessMatrix never emits a position twice
        ent[(r, c)] = v

    for a in range(4):                  # first element: everything but its lower-right node block
        for b_ in range(4):
            if not (a > 1 and b_ > 1):
                put(a, b_, ke[a][b_])
    for e in range(1, nel):             # elements 1 .. nel-1 at rn = 2 e
        rn = 2 * e
        for a in range(2):
            for b_ in range(2):
                put(rn + a, rn + b_, ke[a + 2][b_ + 2] + ke[a][b_])
        last = e == nel - 1
        for a in range(4):
            for b_ in range(4):
                if a < 2 and b_ < 2:
                    continue
                if a > 1 and b_ > 1 and not last:
                    continue
                put(rn + a, rn + b_, ke[a][b_])
    for x in range(200, 1001, 200):     # insertSprings: k @@! (x, x) + 2000
        if (x, x) in ent:
            ent[(x, x)] = ent[(x, x)] + spring
    rows, cols, vals = [], [], []
    for (r, c), v in ent.items():       # extractSubmatrixSM (subtract 2) (subtract 2) knew (2, ndof - 1) (2, ndof - 1)
        if r >= 2 and c >= 2:
            rows.append(r - 2); cols.append(c - 2); vals.append(v)
    n = ndof - 2
    b = np.zeros(n)
    b[n - 2] = 5000.0
    b[n - 1] = 819000000.0
    return (n, n), np.array(rows, dtype=np.int64), np.array(cols, dtype=np.int64), np.array(vals, dtype=np.float64), b
