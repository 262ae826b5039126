"""CSC / CSB array layouts of the reference's `vector/` package: the oracle's restatements against the literals the reference holds --
the triplet example at the foot of vector/src/Data/Sparse/Internal/CSC.hs:121-125 and csPtrV's documented value (Vector/Utils.hs:10-11) -- and against
each other.  (The package has no test suite: beyond these two literals the layouts are pinned by their definitions.)"""
import numpy as np

from oracle import oracle as orc


def test_to_csc_on_the_reference_example():
    # CSC.hs:121-125 (comment): row = [0, 0, 1, 2, 2, 2], col = [0, 2, 2, 0, 1, 2], data = [1 .. 6]
    cp, ri, va = orc.to_csc(3, 3, [0, 0, 1, 2, 2, 2], [0, 2, 2, 0, 1, 2], [1, 2, 3, 4, 5, 6])
    assert cp.tolist() == [0, 2, 3, 6] and ri.tolist() == [0, 2, 2, 0, 1, 2] and va.tolist() == [1, 4, 5, 2, 3, 6]
    assert orc.cs_ptr(4, [1, 1, 2, 3]).tolist() == [0, 0, 2, 3, 4]               # Vector/Utils.hs:10-11
    A = orc.csc_to_csr(3, 3, cp, ri, va)
    assert A.rowptr.tolist() == [0, 2, 3, 6] and A.colidx.tolist() == [0, 2, 2, 0, 1, 2] and A.val.tolist() == [1, 2, 3, 4, 5, 6]


def test_csc_of_rectangular_matrices_is_the_csr_of_the_transpose():
    rng = np.random.default_rng(5)
    for m, n, k in ((7, 19, 40), (19, 7, 60), (1, 5, 3), (6, 1, 4), (4, 4, 0)):
        key = rng.choice(m * n, size=k, replace=False)
        r, c, v = np.sort(key) // n, np.sort(key) % n, rng.standard_normal(k)   # listed by rows: toCSC's stable sort leaves rows ascending inside a column
        rc, A = orc.coo_to_csr(m, n, r, c, v)
        cp, ri, va = orc.to_csc(m, n, r, c, v)
        T = orc.transpose(A)
        assert np.array_equal(cp, T.rowptr) and np.array_equal(ri, T.colidx) and np.array_equal(va, T.val)
        B = orc.csc_to_csr(m, n, cp, ri, va)
        assert np.array_equal(B.rowptr, A.rowptr) and np.array_equal(B.colidx, A.colidx) and np.array_equal(B.val, A.val)


def test_csb_layout_round_trip_and_block_order():
    rng = np.random.default_rng(6)
    for (m, n), beta in (((11, 7), 4), ((8, 8), 4), ((5, 13), 16), ((9, 9), 1)):
        k = 3 * max(m, n)
        r, c, v = rng.integers(0, m, k), rng.integers(0, n, k), rng.standard_normal(k)
        bp, rx, cx, vv = orc.to_csb((m, n), beta, r, c, v)
        nbx, nby = -(-m // beta), -(-n // beta)
        assert len(bp) == nbx * nby + 1 and bp[0] == 0 and bp[-1] == k and np.all(np.diff(bp) >= 0)
        assert rx.min() >= 0 and rx.max() < beta and cx.min() >= 0 and cx.max() < beta
        i, j, x = orc.csb_to_coo((m, n), beta, bp, rx, cx, vv)
        assert sorted(zip(i.tolist(), j.tolist(), x.tolist())) == sorted(zip(r.tolist(), c.tolist(), v.tolist()))
        f = orc.csb_block_index((m, n), beta, i, j)
        assert np.all(np.diff(f) >= 0)                                          # blocks ascending in blockIx order (block row fastest)
        assert np.array_equal(f, np.repeat(np.arange(nbx * nby), np.diff(bp)))
        b = int(f[0])                                                           # consBlocks conses: reverse input order inside a block
        mine = np.where(orc.csb_block_index((m, n), beta, r, c) == b)[0][::-1]
        assert np.array_equal(i[bp[b]:bp[b + 1]], r[mine]) and np.array_equal(j[bp[b]:bp[b + 1]], c[mine])
