"""spmv_wave_kernel (csrc/sla_spmv_wave.hip, round 4): the plain CSR (#>) with wavefront-private 128-row blocks and row-pair
stores, against the oracle.  One lane folds a row from the wavefront's LDS stage in ascending order with separately rounded multiply
and add: BIT-EXACT with the reference's left fold (Common.hs:247-260) for every row it takes (<= 128 entries), and bit-identical to
spmv_stream_kernel's lane-per-row fold.  Every instantiation (2 / 4 / 8 entry pairs per lane and chunk, with and without the next chunk prefetched), rows spanning chunk boundaries, empty rows,
odd row counts (a last pair with one row), blocks past the end of a short matrix, rectangular shapes, odd first entries, every fused
epilogue through the solvers."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
BASE = dict(wdia=0, vdict=0, diag=0, tiles=0, panels=0, lpanel=0, lflat=0)


@pytest.fixture(scope="module")
def sla():
    import sla_amd
    return sla_amd


def _rows(m, n, row_len, seed):
    rng = np.random.default_rng(seed)
    rp, cols, vals = [0], [], []
    for i in range(m):
        k = min(int(row_len(i, rng)), n)
        c = np.sort(rng.choice(n, size=k, replace=False)) if k else np.zeros(0, np.int64)
        cols.append(c.astype(np.int64))
        vals.append(rng.uniform(-1.0, 1.0, k))
        rp.append(rp[-1] + k)
    return (m, n), (np.array(rp, np.int64), np.concatenate(cols), np.concatenate(vals))


CASES = {
    "7 per row, 10 001 rows (odd: last pair has one row)": lambda: _rows(10001, 10001, lambda i, r: 7, 1),
    "ragged 0 / 1 / 3 / 33 / 128": lambda: _rows(7003, 9000, lambda i, r: (0, 1, 3, 33, 128)[i % 5], 2),
    "100 per row (every block spans chunks)": lambda: _rows(1500, 4000, lambda i, r: 100, 3),
    "one row": lambda: _rows(1, 300, lambda i, r: 77, 4),
    "127 rows (less than a block)": lambda: _rows(127, 500, lambda i, r: r.integers(0, 20), 5),
    "129 rows": lambda: _rows(129, 500, lambda i, r: r.integers(0, 20), 6),
    "all rows empty but one": lambda: _rows(900, 900, lambda i, r: 5 if i == 450 else 0, 7),
    "wide 300 x 50000": lambda: _rows(300, 50000, lambda i, r: r.integers(0, 64), 8),
    "odd first entries: 1 then 6 per row": lambda: _rows(5000, 5000, lambda i, r: 1 if i == 0 else 6, 9),
}


@pytest.mark.parametrize("name", list(CASES))
def test_wave_kernel_matches_the_oracle_bit_for_bit(sla, name):
    dims, (rp, ci, va) = CASES[name]()
    m, n = dims
    Ao = orc.Csr(m, n, rp, ci, va)
    x = np.random.default_rng(11).standard_normal(n)
    want = orc.spmv(Ao, x)
    got = {}
    for wave in (0, 1, 802, 604, 408, 1308):
        ctx = sla.Context(0).set_options(stream_wave=wave, **BASE)
        A = sla.fromCSR(dims, rp, ci, va, ctx)
        info = A.kernel_info()
        assert info.startswith("algo=stream+wave " if wave else "algo=stream "), info
        y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
        if wave:
            assert np.array_equal(y, want), (name, wave, int(np.count_nonzero(y != want)))
        else:   # spmv_stream_kernel reduces blocks of few long rows by wavefront segments: the rounding bound of SURVEY 8(a) A1
            bound = np.diff(rp) * np.finfo(float).eps * orc.spmv(orc.Csr(m, n, rp, ci, np.abs(va)), np.abs(x)) + 1e-300
            assert np.all(np.abs(y - want) <= bound), (name, float(np.abs(y - want).max()))
        got[wave] = y
        del A
        ctx.close()


def test_wave_kernel_steps_aside_for_long_rows(sla):
    dims, (rp, ci, va) = _rows(600, 5000, lambda i, r: 129 if i == 77 else 5, 12)
    ctx = sla.Context(0).set_options(**BASE)
    A = sla.fromCSR(dims, rp, ci, va, ctx)
    assert A.kernel_info().startswith("algo=stream "), A.kernel_info()
    x = np.random.default_rng(1).standard_normal(5000)
    y, yo = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV(), orc.spmv(orc.Csr(600, 5000, rp, ci, va), x)
    assert np.all(np.abs(y - yo) <= 129 * np.finfo(float).eps * np.abs(x).max() * 129)


@pytest.mark.parametrize("wave", [1, 604, 408])
def test_wave_kernel_solver_epilogues(sla, wave):
    """bicgsInit / bicgstabStep fused and split (EPI_SUB, EPI_DOT, EPI_DOT2, EPI_DOT4), cgsStep (EPI_AXPY_DOT), cgneStep (EPI_AXPY_DOT on A,
    EPI_XPBY_NRM on the transpose), linSolve0's residual sweep (EPI_RES): two steps against the oracle, and the iterates bit-identical
    to spmv_stream_kernel's (same row sums, same epilogue arithmetic; only the grouping of the fused partial sums differs -> 1e-12)."""
    from sla_amd import workloads as wl
    n = 30001
    dims, (rp, ci, va) = wl.random_spd(n, 6, 3)
    Ao = orc.Csr(n, n, rp, ci, va)
    b = orc.spmv(Ao, np.linspace(-1.0, 2.0, n))
    x0 = np.full(n, 0.25)
    for fuse in (1, 0):
        ctx = sla.Context(0).set_options(stream_wave=wave, bicg_fuse45=fuse, **BASE)
        A = sla.fromCSR(dims, rp, ci, va, ctx)
        assert "stream+wave" in A.kernel_info()
        so, sd = orc.BicgstabState(Ao, b, x0), sla.bicgsInit(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
        so.step(b - orc.spmv(Ao, x0), 2)
        sd.step(2)
        for nm, dev, ref in (("x", sd._xBicgstab, so.x), ("r", sd._rBicgstab, so.r), ("p", sd._pBicgstab, so.p)):
            assert np.linalg.norm(dev.toDenseListSV() - ref) <= 1e-11 * np.linalg.norm(ref), (fuse, nm)
        del sd
        if fuse:
            sc, sdc = orc.CgsState(Ao, b, x0), sla.cgsInit(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
            sc.step(b - orc.spmv(Ao, x0), 2)
            sdc.step(2)
            for nm, dev, ref in (("x", sdc._x, sc.x), ("r", sdc._r, sc.r), ("p", sdc._p, sc.p), ("u", sdc._u, sc.u)):
                assert np.linalg.norm(dev.toDenseListSV() - ref) <= 1e-11 * np.linalg.norm(ref), ("cgs", nm)
            sn, sdn = orc.CgneState(Ao, b, x0), sla.cgneInit(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
            sn.step(2)
            sdn.step(2)
            assert np.linalg.norm(sdn._xCgne.toDenseListSV() - sn.x) <= 1e-11 * np.linalg.norm(sn.x)
            del sdc, sdn
            for meth, ometh in ((sla.BICGSTAB_, orc.BICGSTAB_), (sla.CGS_, orc.CGS_)):
                xs, inf = sla.linSolve0(meth, A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx), return_info=True)
                rc, xo, it_o, res_o, r0_o = orc.linsolve0(ometh, Ao, b, x0)
                assert inf["converged"] and abs(inf["iters"] - it_o) <= 2 and np.linalg.norm(orc.spmv(Ao, xs.toDenseListSV()) - b) <= inf["tol"] * (1 + 1e-9)
        del A
        ctx.close()
