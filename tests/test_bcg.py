"""bcgInit / bcgStep -- an EXTENSION: the reference declares `data BCG` and keeps the two functions commented out
(Numeric/LinearAlgebra/Sparse.hs:886-909); `linSolve0 BCG_` throws there (:1031) and keeps throwing here.

Parity is UNPINNED BY THE REFERENCE for this row (no test, literal or call site of the commented code exists there).  What pins it here:
  * CPU: the oracle's restatement against a second, independent numpy transcription of the commented formulas, and the textbook identity that
    BCG with rhat0 = r0 on a symmetric matrix IS conjugate gradients (rhat_k = r_k, phat_k = p_k; x after k steps = CG's);
  * GPU (-m gpu): the device record after k steps against the oracle's at 1e-9 on the SPD, the non-symmetric banded and a random non-symmetric
    problem, single steps and k-step launches, clone purity, and that sla_linsolve0(BCG_) still returns the reference's IterE."""
import numpy as np
import pytest

from oracle import oracle as orc


def _numpy_bcg(A, b, x0, k):
    """the commented bcgInit / bcgStep, transcribed on dense numpy arrays (Sparse.hs:889-909)"""
    x = x0.copy(); r = b - A @ x0; rhat = r.copy(); p = r.copy(); phat = r.copy()
    for _ in range(k):
        aap = A @ p
        alpha = (r @ rhat) / (aap @ phat)
        x1 = x + alpha * p
        r1 = r - alpha * aap
        rhat1 = rhat - alpha * (A.T @ phat)
        beta = (r1 @ rhat1) / (r @ rhat)
        p, phat = r1 + beta * p, rhat1 + beta * phat
        x, r, rhat = x1, r1, rhat1
    return x, r, rhat, p, phat


def _problems():
    from sla_amd.workloads import random_spd, banded_nonsym   # (host generators; the library loads without a GPU, only contexts need one)
    out = []
    n = 300
    dims, (rp, ci, va) = random_spd(n, k=3, seed=5)
    out.append(("spd", n, rp, ci, va))
    n = 400
    dims, (rp, ci, va) = banded_nonsym(n)
    out.append(("banded", n, rp, ci, va))
    rng = np.random.default_rng(11)
    n = 257   # odd: the sweeps' tail element
    rows, cols, vals = [], [], []
    for i in range(n):
        cj = np.unique(np.concatenate(([i], rng.choice(n, size=4, replace=False))))
        rows.append(np.full(len(cj), i)); cols.append(cj)
        v = rng.standard_normal(len(cj)) * 0.3
        v[cj == i] = 4.0 + rng.random()
        vals.append(v)
    rc, A = orc.coo_to_csr(n, n, np.concatenate(rows), np.concatenate(cols), np.concatenate(vals))
    assert rc == orc.OK
    out.append(("random_nonsym", n, A.rowptr.copy(), A.colidx.copy(), A.val.copy()))
    return out


@pytest.fixture(scope="module")
def problems():
    return _problems()


def _dense(n, rp, ci, va):
    A = np.zeros((n, n))
    for i in range(n):
        A[i, ci[rp[i]:rp[i + 1]]] = va[rp[i]:rp[i + 1]]
    return A


def test_oracle_bcg_vs_independent_transcription(problems):
    for name, n, rp, ci, va in problems:
        Ao, Ad = orc.Csr(n, n, rp, ci, va), _dense(n, rp, ci, va)
        xs = np.random.default_rng(3).standard_normal(n)
        b, x0 = orc.spmv(Ao, xs), np.full(n, 0.1)
        so = orc.BcgState(Ao, b, x0)
        assert np.array_equal(so.rhat, so.r) and np.array_equal(so.p, so.r) and np.array_equal(so.phat, so.r)
        assert np.array_equal(so.r, b - orc.spmv(Ao, x0))
        for k in (1, 2, 5):
            so = orc.BcgState(Ao, b, x0).step(k)
            ref = _numpy_bcg(Ad, b, x0, k)
            for got, want, f in zip((so.x, so.r, so.rhat, so.p, so.phat), ref, "x r rhat p phat".split()):
                assert np.linalg.norm(got - want) <= 1e-9 * np.linalg.norm(want) + 1e-13 * np.linalg.norm(b), (name, k, f)


def test_oracle_bcg_is_cg_on_a_symmetric_matrix(problems):
    name, n, rp, ci, va = problems[0]
    Ao, Ad = orc.Csr(n, n, rp, ci, va), _dense(n, rp, ci, va)
    assert np.array_equal(Ad, Ad.T)
    xs = np.random.default_rng(4).standard_normal(n)
    b, x0 = orc.spmv(Ao, xs), np.zeros(n)
    so = orc.BcgState(Ao, b, x0).step(12)
    x, r = x0.copy(), b - Ad @ x0            # textbook CG
    p = r.copy()
    for _ in range(12):
        ap = Ad @ p
        a = (r @ r) / (p @ ap)
        x, r1 = x + a * p, r - a * ap
        p = r1 + (r1 @ r1) / (r @ r) * p
        r = r1
    assert np.linalg.norm(so.x - x) <= 1e-9 * np.linalg.norm(x)
    assert np.linalg.norm(so.rhat - so.r) <= 1e-9 * np.linalg.norm(b) and np.linalg.norm(so.phat - so.p) <= 1e-9 * np.linalg.norm(b)
    assert np.linalg.norm(b - Ad @ so.x) < 1e-3 * np.linalg.norm(b)      # and it converges


def test_oracle_linsolve0_bcg_still_throws(problems):
    name, n, rp, ci, va = problems[0]
    Ao = orc.Csr(n, n, rp, ci, va)
    rc, *_ = orc.linsolve0(orc.BCG_, Ao, np.ones(n), np.zeros(n))
    assert rc == orc.ERR_UNSUPPORTED                                        # Sparse.hs:1031


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def sla():
    import sla_amd
    return sla_amd


@pytest.mark.gpu
def test_bcg_steps_vs_oracle(sla, problems):
    for name, n, rp, ci, va in problems:
        A, Ao = sla.fromCSR((n, n), rp, ci, va), orc.Csr(n, n, rp, ci, va)
        xs = np.random.default_rng(3).standard_normal(n)
        b, x0 = orc.spmv(Ao, xs), np.full(n, 0.1)
        so, sd = orc.BcgState(Ao, b, x0), sla.bcgInit(A, sla.fromVector(b), sla.fromVector(x0))
        for f, want in ((sd._rBcg, so.r), (sd._rHatBcg, so.rhat), (sd._pBcg, so.p), (sd._pHatBcg, so.phat)):
            assert np.allclose(f.toDenseListSV(), want, rtol=1e-13, atol=1e-13), name
        assert np.array_equal(sd._xBcg.toDenseListSV(), x0)
        for k in (1, 1, 3, 4):            # (4: the step-graph replay path of small problems)
            so.step(k); sd.step(k)
            for f, want, tag in ((sd._xBcg, so.x, "x"), (sd._rBcg, so.r, "r"), (sd._rHatBcg, so.rhat, "rhat"), (sd._pBcg, so.p, "p"),
                                 (sd._pHatBcg, so.phat, "phat")):
                got = f.toDenseListSV()
                assert np.linalg.norm(got - want) <= 1e-9 * np.linalg.norm(want) + 1e-12 * np.linalg.norm(b), (name, k, tag)


@pytest.mark.gpu
def test_bcg_pure_step_and_clone(sla, problems):
    name, n, rp, ci, va = problems[1]
    A = sla.fromCSR((n, n), rp, ci, va)
    b, x0 = sla.fromVector(np.ones(n)), sla.fromVector(np.zeros(n))
    s0 = sla.bcgInit(A, b, x0)
    s1 = sla.bcgStep(A, s0)                      # pure: a new record
    assert np.array_equal(s0._xBcg.toDenseListSV(), np.zeros(n))
    s2 = sla.bcgStep(A, s1)
    t = sla.bcgInit(A, b, x0).step(2)            # in place
    for f in ("_xBcg", "_rBcg", "_rHatBcg", "_pBcg", "_pHatBcg"):
        assert np.array_equal(getattr(s2, f).toDenseListSV(), getattr(t, f).toDenseListSV()), f
    with pytest.raises(sla.SlaError):
        s2._get(3, n)                            # a BCG record has no _u


@pytest.mark.gpu
def test_bcg_converges_and_linsolve0_still_throws(sla, problems):
    name, n, rp, ci, va = problems[0]
    A, Ao = sla.fromCSR((n, n), rp, ci, va), orc.Csr(n, n, rp, ci, va)
    xs = np.random.default_rng(8).standard_normal(n)
    b = orc.spmv(Ao, xs)
    s = sla.bcgInit(A, sla.fromVector(b), sla.fromVector(np.zeros(n))).step(40)
    assert np.linalg.norm(s._xBcg.toDenseListSV() - xs) <= 1e-8 * np.linalg.norm(xs)
    with pytest.raises(sla.IterationException):                             # Sparse.hs:1031: the drop-in behaviour stays
        sla.linSolve0(sla.BCG_, A, sla.fromVector(b), sla.fromVector(np.zeros(n)))
    with pytest.raises(sla.SlaError):                                       # rectangular: BCG needs a square matrix
        R = sla.fromCOO((3, 4), [0, 1, 2], [0, 1, 3], [1.0, 2.0, 3.0])
        sla.bcgInit(R, sla.fromVector(np.ones(3)), sla.fromVector(np.zeros(4)))
