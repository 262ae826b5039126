"""sla_csr_from_csc / sla_csr_export_csc / sla_csr_transpose / sla_csr_from_csb (the array layouts of the reference's `vector/`
package either side of the lowered matrix) against the oracle's restatements: index arrays bit for bit, values bit for bit."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sla():
    import sla_amd
    return sla_amd


def _same(A, Ao):
    rp, ci, va = A.csr()
    return np.array_equal(rp, Ao.rowptr) and np.array_equal(ci, Ao.colidx) and np.array_equal(va, Ao.val)


def _mv_ok(A, got, want):
    """(#>) against the oracle's left fold: bit for bit where the lowered form folds a row in the reference's order (sla_csr_get_props: fold = exact),
    within the rounding bound of a regrouped fold otherwise (long rows, SURVEY A1's contract)."""
    if A.props()["fold"] == 0:
        return np.array_equal(got, want)
    return np.allclose(got, want, rtol=1e-13, atol=1e-13)


def _random(rng, m, n, k):
    key = np.sort(rng.choice(m * n, size=k, replace=False))
    return key // n, key % n, rng.standard_normal(k)


def test_reference_example_through_csc(sla):
    # vector/src/Data/Sparse/Internal/CSC.hs:121-125: row = [0,0,1,2,2,2], col = [0,2,2,0,1,2], data = [1..6]
    A = sla.fromCSC((3, 3), [0, 2, 3, 6], [0, 2, 2, 0, 1, 2], [1, 4, 5, 2, 3, 6])
    assert sorted(A.toListSM()) == [(0, 0, 1.0), (0, 2, 2.0), (1, 2, 3.0), (2, 0, 4.0), (2, 1, 5.0), (2, 2, 6.0)]
    cp, ri, va = A.csc()
    assert cp.tolist() == [0, 2, 3, 6] and ri.tolist() == [0, 2, 2, 0, 1, 2] and va.tolist() == [1, 4, 5, 2, 3, 6]


@pytest.mark.parametrize("m,n,k", [(700, 1900, 9000), (1900, 700, 15000), (1, 50, 20), (60, 1, 30), (40, 40, 0), (300000, 300000, 2400000)])
def test_csc_in_and_out(sla, m, n, k):
    rng = np.random.default_rng(m + n)
    r, c, v = _random(rng, m, n, k)
    rc, Ao = orc.coo_to_csr(m, n, r, c, v)
    cp, ri, va = orc.to_csc(m, n, r, c, v)
    A = sla.fromCSC((m, n), cp, ri, va)
    assert A.dims == (m, n) and _same(A, Ao)                              # the matrix sla_csr_from_csr gives for the CSR arrays, bit for bit
    assert A.lower_info().get("from_csc") == 1.0
    cp2, ri2, va2 = A.csc()
    assert np.array_equal(cp2, cp) and np.array_equal(ri2, ri) and np.array_equal(va2, va)
    if k:
        x, y = rng.standard_normal(n), rng.standard_normal(m)
        assert _mv_ok(A, sla.matVec(A, sla.fromVector(x)).toDenseListSV(), orc.spmv(Ao, x))
        # (<#) runs on the attached CSC side (whatever form that side was lowered to: the suite's tolerance for (<#))
        assert np.allclose(sla.vecMat(sla.fromVector(y), A).toDenseListSV(), orc.spmv(orc.transpose(Ao), y), rtol=1e-13, atol=1e-13)
    # the CSR arrays of the same matrix lower to the same thing
    B = sla.fromCSR((m, n), Ao.rowptr, Ao.colidx, Ao.val)
    assert _same(B, Ao) and A.kernel_info().split(" grid")[0] == B.kernel_info().split(" grid")[0]


def test_transpose_is_an_owned_handle(sla):
    from sla_amd import workloads as wl
    rng = np.random.default_rng(11)
    m, n, k = 5000, 3000, 60000
    r, c, v = _random(rng, m, n, k)
    A = sla.fromListSM((m, n), list(zip(r.tolist(), c.tolist(), v.tolist())))
    rc, Ao = orc.coo_to_csr(m, n, r, c, v)
    T = sla.transpose(A)
    assert T.dims == (n, m) and _same(T, orc.transpose(Ao))
    assert _same(sla.transpose(T), Ao)                                    # transposeSM . transposeSM = id, on the arrays
    y = rng.standard_normal(m)
    want = orc.spmv(orc.transpose(Ao), y)
    assert _mv_ok(T, sla.matVec(T, sla.fromVector(y)).toDenseListSV(), want)
    del T                                                                 # A keeps nothing of the handle it gave away: (<#) builds its own
    assert np.allclose(sla.vecMat(sla.fromVector(y), A).toDenseListSV(), want, rtol=1e-13, atol=1e-13)
    # CGNE_ (transpose aa #> r every step) on a matrix that arrived as CSC: no second sort, same iterates as the CSR arrival
    dims, (rp, ci, va) = wl.banded_nonsym(20000, 3)
    Bo = orc.Csr(dims[0], dims[1], rp, ci, va)
    To = orc.transpose(Bo)
    b = orc.spmv(Bo, np.ones(dims[0]))
    A1 = sla.fromCSR(dims, rp, ci, va)
    A2 = sla.fromCSC(dims, To.rowptr, To.colidx, To.val)
    s1 = sla.cgneInit(A1, sla.fromVector(b), sla.fromVector(np.zeros(dims[0]))).step(5)
    s2 = sla.cgneInit(A2, sla.fromVector(b), sla.fromVector(np.zeros(dims[0]))).step(5)
    assert np.array_equal(s1._xCgne.toDenseListSV(), s2._xCgne.toDenseListSV())


@pytest.mark.parametrize("dims,beta", [((1100, 700), 64), ((800, 800), 128), ((50, 1300), 4096), ((900, 900), 1), ((200000, 200000), 512)])
def test_csb_ingestion(sla, dims, beta):
    rng = np.random.default_rng(dims[0] + beta)
    m, n = dims
    k = 8 * max(m, n)
    r, c, v = rng.integers(0, m, k), rng.integers(0, n, k), rng.standard_normal(k)     # duplicates included
    bp, rx, cx, vv = orc.to_csb(dims, beta, r, c, v)
    A = sla.fromCSB(dims, beta, bp, rx, cx, vv)
    i, j, x = orc.csb_to_coo(dims, beta, bp, rx, cx, vv)
    rc, Ao = orc.coo_to_csr(m, n, i, j, x)                                # storage order into fromListSM: the later element wins
    assert rc == orc.OK and _same(A, Ao)
    xv = rng.standard_normal(n)
    assert _mv_ok(A, sla.matVec(A, sla.fromVector(xv)).toDenseListSV(), orc.spmv(Ao, xv))


def test_bad_layouts_are_refused(sla):
    with pytest.raises(sla.SlaError):                                     # rows not ascending inside a column
        sla.fromCSC((3, 2), [0, 2, 3], [2, 0, 1], [1.0, 2.0, 3.0])
    with pytest.raises(sla.SlaError):                                     # row index outside the matrix
        sla.fromCSC((3, 2), [0, 2, 3], [0, 3, 1], [1.0, 2.0, 3.0])
    with pytest.raises(sla.SlaError):                                     # column pointer not monotone
        sla.fromCSC((3, 2), [0, 2, 1], [0, 1], [1.0, 2.0])
    with pytest.raises(sla.SlaError):                                     # block-relative index outside the block
        sla.fromCSB((8, 8), 4, [0, 1, 1, 1, 1], [4], [0], [1.0])
    with pytest.raises(sla.SlaError):                                     # inside the block, outside the matrix (ragged last block)
        sla.fromCSB((6, 6), 4, [0, 0, 0, 0, 1], [2], [0], [1.0])
    with pytest.raises(sla.SlaError):
        sla.fromCSB((8, 8), 4, [0, 2, 1, 1, 1], [0], [0], [1.0])
    with pytest.raises(ValueError):
        sla.fromCSB((8, 8), 4, [0, 1], [0], [0], [1.0])
    many = sla.Context.multi([0, 0])
    with pytest.raises(sla.SlaError) as e:
        sla.fromCSC((3, 3), [0, 2, 3, 6], [0, 2, 2, 0, 1, 2], [1, 4, 5, 2, 3, 6], many)
    assert "multi-device" in str(e.value)
    A = sla.fromCSB((3, 3), 2, [0, 1, 1, 1, 2], [0, 0], [0, 0], [1.0, 2.0], many)   # CSB goes through fromListSM's path: every kind of context
    assert sorted(A.toListSM()) == [(0, 0, 1.0), (2, 2, 2.0)]
    At = sla.transpose(A)                                                 # (triplet path on a multi-device context)
    assert sorted(At.toListSM()) == [(0, 0, 1.0), (2, 2, 2.0)]
    del A, At
    many.close()
