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
