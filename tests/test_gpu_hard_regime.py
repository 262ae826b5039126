"""Parity in the hard regime: the reference's own hard inputs -- test/data/e05r0000.mtx (test/Perf.hs) and the
ill-conditioned beam system of issues/issue_denjoh.hs:60-71 (entries 1e9 .. 7e12, condition number 5e11) -- through linSolve0
BICGSTAB_ / CGS_ (residual traces, both BiCGSTAB flows), GMRES and (<\\>), against the oracle ITERATION BY ITERATION.

A Krylov iteration on such a system amplifies last-bit differences until two evaluations of the same recurrence part ways.  The
oracle itself shows where: its BiCGSTAB is run with the reference's beta (r' . r0hat, Sparse.hs:980) and with the numerator
through (s . r0hat) - omega (aas . r0hat) (what the product's fused K4+K5 sweep evaluates); J = the first iteration where those
two traces differ by 1e-6 relative (measured: 21 .. 30 on these systems, tools/hard_regime.py -> profiles/r04_hard_regime.txt).
Up to there the device trace has to follow the oracle's; from there on only the OUTCOME can be compared (the flags, the returned
residual against the stopping rule).  The bounds below are the measured ones with headroom; see the profile for the traces."""
import numpy as np
import pytest

from oracle import oracle as orc
from refdata import GOLDEN, denjoh_beam, read_mtx_array, read_mtx_coordinate

import importlib.util
import os

_spec = importlib.util.spec_from_file_location("hard_regime", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "hard_regime.py"))
hr = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(hr)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sla():
    import sla_amd
    return sla_amd


def _fixture(name):
    for f in hr.fixtures():
        if f[0] == name:
            return f
    raise KeyError(name)


@pytest.mark.parametrize("name", ["e05r0000", "denjoh_beam"])
@pytest.mark.parametrize("x0v", [0.0, 0.1])
def test_bicgstab_traces_follow_the_oracle_until_its_own_formulas_part(sla, name, x0v):
    _, dims, r, c, v, b = _fixture(name)
    n = dims[0]
    rc, Ao = orc.coo_to_csr(n, n, r, c, v)
    x0 = np.full(n, x0v)
    ref, tol, fref = hr.oracle_trace("BICGSTAB_", Ao, b, x0, 200)
    ident, _, fid = hr.oracle_trace("BICGSTAB_", Ao, b, x0, 200, rho_identity=True)
    J = hr.first_split(ident, ref)            # where the oracle's two formulas part
    L = min(J, len(ref))
    with np.errstate(all="ignore"):
        env = np.maximum.accumulate(np.abs(ident[:L] - ref[:L]) / ref[:L])   # the oracle's own sensitivity envelope
    for fuse in (1, 0):
        h, fl, xd, info = hr.device_trace(sla, "BICGSTAB_", dims, r, c, v, b, x0, fuse)
        m = min(L, len(h))
        assert m >= min(4, len(ref)), (name, x0v, fuse, m)
        rel = np.abs(h[:m] - ref[:m]) / ref[:m]
        # iteration by iteration: within 1000 x the envelope of the oracle's own two formulas (floor 1e-10): the device regroups
        # every inner product, which moves a trace like the formula change does
        bound = np.maximum(1e3 * env[:m], 1e-10)
        assert np.all(rel <= bound), (name, x0v, fuse, int(np.argmax(rel > bound)), rel[:m].tolist())
        # the fused (identity) flow must not leave the oracle's trace earlier than the reference's split does by more than a few steps
        Jd = hr.first_split(h, ref)
        assert Jd >= J - 8, (name, x0v, fuse, Jd, J)
        # outcome: same verdict as the oracle, and a verdict that is TRUE of the returned iterate
        assert fl == fref, (name, x0v, fuse, fl, fref, len(h), len(ref))
        res = np.linalg.norm(orc.spmv(Ao, xd) - b)
        if fl == "converged":
            assert res <= tol * (1 + 1e-6) and abs(len(h) - len(ref)) <= 3
        else:
            assert len(h) == 200 and res > tol


@pytest.mark.parametrize("name", ["e05r0000", "denjoh_beam"])
@pytest.mark.parametrize("x0v", [0.0, 0.1])
def test_cgs_traces_follow_the_oracle(sla, name, x0v):
    _, dims, r, c, v, b = _fixture(name)
    n = dims[0]
    rc, Ao = orc.coo_to_csr(n, n, r, c, v)
    x0 = np.full(n, x0v)
    ref, tol, fref = hr.oracle_trace("CGS_", Ao, b, x0, 200)
    h, fl, xd, info = hr.device_trace(sla, "CGS_", dims, r, c, v, b, x0, 1)
    m = min(len(h), len(ref), 20)
    rel = np.abs(h[:m] - ref[:m]) / ref[:m]
    assert np.all(rel[:min(m, 10)] <= 1e-9) and np.all(rel <= 1e-5), (name, x0v, rel.tolist())
    assert fl == fref, (name, x0v, fl, fref, len(h), len(ref))
    res = np.linalg.norm(orc.spmv(Ao, xd) - b)
    if fl == "converged":
        assert res <= tol * (1 + 1e-6)
    else:
        assert len(h) == 200


def _gmres_exact_lsq(Ao, b, x0, restart, cycles):
    """GMRES(restart) over whole cycles built from the ORACLE's arnoldi (Sparse.hs:630-667 restated) and an exact dense least-squares
    solve of min ||beta e1 - H y|| (numpy lstsq): the comparator for systems on which the reference's own qr + triUpperSolve route
    (orc.gmres) is not usable -- see the test below."""
    x, tol, it = x0.copy(), None, 0
    for cyc in range(cycles + 1):
        r = b - orc.spmv(Ao, x)
        beta = np.linalg.norm(r)
        if tol is None:
            tol = max(1e-6, 1e-4 * beta)
        if beta <= tol or cyc == cycles:
            break
        rc, Q, H, k = orc.arnoldi(Ao, r, restart)
        e1 = np.zeros(k + 1)
        e1[0] = beta
        y = np.linalg.lstsq(H, e1, rcond=None)[0]
        x = x + Q[:, :k] @ y
        it += k
    return x, it, beta


@pytest.mark.parametrize("name", ["e05r0000", "denjoh_beam"])
def test_gmres_and_backslash_against_the_oracles_x(sla, name):
    """GMRES x against the ORACLE's x (not against the product's own residual).  GMRES(30) and GMRES(60) from 0.1 * ones over whole
    restart cycles: same iteration counts, x within 1e-6 relative of
      (a) the oracle's arnoldi + an exact dense least-squares solve per cycle (both fixtures), and
      (b) orc.gmres -- the reference's commented sketch, qr + triUpperSolve (Sparse.hs:837-848) -- on e05r0000.
    On the beam system (b) is NOT a usable comparator, measured in round 4: the restated reference `qr` leaves a lower part of 3e12
    (max|R| = 1e13) in the R of this 31 x 30 Hessenberg matrix (Q R = H and Q^T Q = I still hold to 4e-16), so its least-squares step
    returns ||b - A x|| = 7.3e8 where the exact minimiser over the same Krylov basis gives 1.78e7 -- which is what the product
    returns (its Givens sweep is an independent formulation).  GMRES is dead code in the reference (parity unpinned).
    Then (<\\>) (dead instance Sparse.hs:1080-1084: GMRES(30) from 0.1 * ones, 200 iterations) = that GMRES call, bit for bit."""
    _, dims, r, c, v, b = _fixture(name)
    n = dims[0]
    rc, Ao = orc.coo_to_csr(n, n, r, c, v)
    A = sla.fromCOO(dims, r, c, v)
    bv, x0 = sla.fromVector(b), np.full(n, 0.1)
    for restart, cycles in ((30, 6), (60, 20)):
        x, info = sla.gmres(A, bv, sla.fromVector(x0), restart=restart, return_info=True, max_iters=restart * cycles)
        xd = x.toDenseListSV()
        xe, it_e, res_e = _gmres_exact_lsq(Ao, b, x0, restart, cycles)
        r0_o = np.linalg.norm(b - orc.spmv(Ao, x0))
        assert abs(info["r0norm"] - r0_o) <= 1e-10 * r0_o
        assert info["iters"] == it_e, (name, restart, info["iters"], it_e)
        assert np.linalg.norm(xd - xe) <= 1e-6 * np.linalg.norm(xe), (name, restart, np.linalg.norm(xd - xe) / np.linalg.norm(xe))
        assert abs(info["resnorm"] - res_e) <= 1e-5 * res_e + 1e-12 * r0_o, (name, restart, info["resnorm"], res_e)
        res = np.linalg.norm(orc.spmv(Ao, xd) - b)
        assert abs(res - info["resnorm"]) <= 1e-6 * max(res, info["resnorm"]) + 1e-9 * r0_o
        assert info["converged"] == (res_e <= max(1e-6, 1e-4 * r0_o))
        rco, xo, it_o, res_o, r0o = orc.gmres(Ao, b, x0, restart=restart, max_restarts=cycles)
        if name == "e05r0000":
            assert info["iters"] == it_o and np.linalg.norm(xd - xo) <= 1e-6 * np.linalg.norm(xo), (restart, np.linalg.norm(xd - xo) / np.linalg.norm(xo))
        else:   # the reference-sketch route loses to the exact least-squares step on this system (see above): never the other way round
            assert res <= np.linalg.norm(orc.spmv(Ao, xo) - b) * (1 + 1e-9)
    xb, ib = sla.linSolve(A, bv, return_info=True)
    xg, ig = sla.gmres(A, bv, sla.fromVector(x0), restart=30, return_info=True, max_iters=200)
    assert np.array_equal(xb.toDenseListSV(), xg.toDenseListSV()) and ib["iters"] == ig["iters"] and ib["flags"] == ig["flags"]
