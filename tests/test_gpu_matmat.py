"""(##) / (##^) through the C ABI (sla_csr_matmat, csrc/sla_matmat.hip) against the reference's literals and the oracle
(reference matMat_, SpMatrix.hs:768-811).  Index structure and values are compared BIT-EXACT: the
product is structurally dense over (present rows of m1) x (present columns of m2), explicit zeros included, every entry
the ascending left fold of separately rounded products."""
import numpy as np
import pytest

from oracle import oracle as orc
from refdata import coo_of, golden, tridiag_coo

pytestmark = pytest.mark.gpu
G = golden()


@pytest.fixture(scope="module")
def sla():
    import sla_amd
    return sla_amd


def _both(sla, entry):
    (m, n), r, c, v = coo_of(entry) if isinstance(entry, dict) else entry
    rc, Ao = orc.coo_to_csr(m, n, r, c, v)
    assert rc == orc.OK
    return sla.fromCOO((m, n), r, c, v), Ao


def _same(C, Co):
    rp, ci, va = C.csr()
    assert C.dims == (Co.m, Co.n)
    assert np.array_equal(rp, Co.rowptr) and np.array_equal(ci, Co.colidx)
    assert np.array_equal(va, Co.val) and np.array_equal(np.signbit(va), np.signbit(Co.val))


def test_reference_literals(sla):
    M = G["matmat"]
    m1, o1 = _both(sla, M["m1"]); m2, o2 = _both(sla, M["m2"])
    m1p, o1p = _both(sla, M["m1p"]); m2p, o2p = _both(sla, M["m2p"])
    want = {k: _both(sla, M[k])[0].toDense() for k in ("m1m2", "m1m2p", "m2m1p")}
    assert np.array_equal(sla.matMat(m1, m2).toDense(), want["m1m2"])          # m1 ## m2 == m1m2        (LibSpec.hs:61-62)
    assert np.array_equal(sla.matMat(m1p, m2p).toDense(), want["m1m2p"])       # m1' ## m2' == m1m2'     (:63)
    assert np.array_equal(sla.matMat(m2p, m1p).toDense(), want["m2m1p"])       # m2' ## m1' == m2m1'     (:64-65)
    for a, b, ao, bo in ((m1, m2, o1, o2), (m1p, m2p, o1p, o2p), (m2p, m1p, o2p, o1p)):
        _same(sla.matMat(a, b), orc.matmat(ao, bo)[1])                         # structure incl. explicit zeros
    with pytest.raises(sla.MatVecSizeMismatchException) as e:                  # error "matMat : incompatible matrix sizes" (:795)
        sla.matMat(m1, m2p)
    assert "matMat : incompatible matrix sizes" in str(e.value)
    assert np.array_equal(sla.matMatT(m1, sla.transpose(m2)).toDense(), want["m1m2"])   # m1 ##^ transpose m2 == m1 ## m2


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_rectangular_products_vs_oracle(sla, seed):
    rng = np.random.default_rng(seed)
    m, k, n = int(rng.integers(1, 60)), int(rng.integers(1, 80)), int(rng.integers(1, 70))

    def rnd(r, c, fill):
        nz = max(1, int(fill * r * c))
        return (r, c), rng.integers(0, r, nz), rng.integers(0, c, nz), np.where(rng.random(nz) < 0.1, 0.0, rng.standard_normal(nz))
    A, Ao = _both(sla, rnd(m, k, 0.15))            # empty rows / columns, explicit zeros, duplicate triples (last wins)
    B, Bo = _both(sla, rnd(k, n, 0.2))
    _same(sla.matMat(A, B), orc.matmat(Ao, Bo)[1])
    Bt, Bto = _both(sla, rnd(n, k, 0.2))
    _same(sla.matMatT(A, Bt), orc.matmat(Ao, orc.transpose(Bto))[1])          # A ##^ B = A ## transpose B


def test_check_arnoldi_with_the_product_matmat(sla):
    """checkArnoldi (LibSpec.hs:642-653): nearZero (normFrobenius (aa ## q' ^-^ q ## h)) with both products on the device."""
    for A, kn in ((_both(sla, G["arnoldi"]["aa4"])[0], 3),
                  (_both(sla, tridiag_coo(G["arnoldi"]["tm7"]["n"], *G["arnoldi"]["tm7"]["tridiag"]))[0], G["arnoldi"]["tm7"]["kn"])):
        Q, H = sla.arnoldi(A, sla.onesSV(A.nrows), kn)

        def dense_sm(D):
            ii, jj = np.meshgrid(np.arange(D.shape[0]), np.arange(D.shape[1]), indexing="ij")
            return sla.fromCOO(D.shape, ii.ravel(), jj.ravel(), D.ravel())
        q, qp, h = dense_sm(Q), dense_sm(Q[:, :-1]), dense_sm(H)
        lhs, rhs = sla.matMat(A, qp).toDense(), sla.matMat(q, h).toDense()
        assert np.linalg.norm(lhs - rhs, "fro") <= 1e-12
