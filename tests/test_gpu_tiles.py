"""The row-slice x column-panel tile form of (#>) (spmv_tile_kernel, csrc/sla_spmv_tiles.hip; lowering
csrc/sla_lower_tiles.cpp) against the oracle.  The form is meant for irregular matrices whose x exceeds an L2-sized
panel (2^17 columns by default); SLA_TILE_SHIFT=10 (1024-column panels) forces it onto test-sized matrices so that
every structural case runs in seconds: many panels (> 64: the tile-offset block reload), empty rows and empty tiles,
rectangular shapes, 64-bit row pointers, (row, panel) segments of several entries (layer boundaries inside a 64-entry
group: multi-pass groups), rows with hundreds of entries per panel (the lowering must step aside) and every fused
epilogue through the solvers.

Parity: inside a tile the entries are ordered by (layer, column; round 4 -- by row before) and every product is added to its row sum on its own, in
ascending column order, with separately rounded multiply and add -- BIT-EXACT with the reference's left fold
(Common.hs:247-260) whatever the row length.  Matrices the lowering leaves to the older forms are checked against
|dy_i| <= nnz_i * eps * sum_j |a_ij x_j|."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


@pytest.fixture(scope="module")
def sla():
    import sla_amd
    return sla_amd


def _rand_rows(m, n, row_len, seed):
    """m x n CSR, row i holds row_len(i) distinct uniformly random columns (ascending), values in [-1, 1)."""
    rng = np.random.default_rng(seed)
    rp = [0]
    cols, vals = [], []
    for i in range(m):
        k = min(int(row_len(i, rng)), n)
        c = np.sort(rng.choice(n, size=k, replace=False)) if k else np.zeros(0, np.int64)
        cols.append(c.astype(np.int64))
        vals.append(rng.uniform(-1.0, 1.0, k))
        rp.append(rp[-1] + k)
    return (m, n), (np.array(rp, np.int64), np.concatenate(cols) if cols else np.zeros(0, np.int64),
                    np.concatenate(vals) if vals else np.zeros(0))


CASES = {
    # name: (builder, tile form expected)
    "5 panels, 6 per row": (lambda: _rand_rows(5000, 5000, lambda i, r: 6, 1), True),
    "98 panels (> 64: offset block reload), 40 per row": (lambda: _rand_rows(3000, 100000, lambda i, r: 40, 2), True),
    "ragged: empty / 1 / 3 / 33 / 200 entries": (lambda: _rand_rows(7001, 30000, lambda i, r: (0, 1, 3, 33, 200)[i % 5], 3), True),
    "2 entries per row, 147 panels (mostly empty tiles)": (lambda: _rand_rows(150000, 150000, lambda i, r: 2, 4), True),
    "wide 900 x 70000": (lambda: _rand_rows(900, 70000, lambda i, r: r.integers(0, 60), 5), True),
    "tall 60000 x 2500, segments of ~2 (layer boundaries in most groups)": (lambda: _rand_rows(60000, 2500, lambda i, r: 5, 6), True),
    "segments of ~6: 2500 x 8000, 48 per row": (lambda: _rand_rows(2500, 8000, lambda i, r: 48, 10), True),
    "one row": (lambda: _rand_rows(1, 9000, lambda i, r: 77, 7), False),
    "dense rows among sparse ones (hundreds of layers)": (lambda: _rand_rows(4000, 12000, lambda i, r: 9000 if i % 1000 == 7 else 4, 8), False),
    "all rows dense-ish: 300 per row, 3 panels (100-layer tiles)": (lambda: _rand_rows(2500, 3000, lambda i, r: 300, 9), True),
}


@pytest.mark.parametrize("name", list(CASES))
def test_cu_wide_tiles_relaxed_order_within_the_bound(sla, name):
    """The OPT-IN relaxed order of the CU-wide tile form (option tile_relaxed = 1, round 5's default; csrc/sla_spmv_ctiles.hip): slices shared by a workgroup's four wavefronts,
    every tile one column-sorted run, products added into the row sums by LDS atomics in whatever order the wavefronts get there.
    Contract: |y_i - fold_i| <= nnz_i eps sum_j |a_ij x_j| -- every product is rounded separately, only the order
    of the additions differs from the reference's left fold; a row with ONE or TWO entries is still exact (no order to differ).
    Device builder and host builder produce the same layout: same kernel, so compared through y within the bound and through
    rows of <= 2 entries bit for bit."""
    build, _ = CASES[name]
    dims, csr = build()
    m, n = dims
    rp, ci, va = csr
    Ao = orc.Csr(m, n, rp, ci, va)
    x = np.random.default_rng(11).standard_normal(n)
    want = orc.spmv(Ao, x)
    lens = np.diff(rp)
    bound = lens * EPS * orc.spmv(orc.Csr(m, n, rp, ci, np.abs(va)), np.abs(x)) + 1e-300
    for rp64, dev in (("0", 2), ("1", 2), ("0", 0), ("1", 0)):
        ctx = sla.Context(0).set_options(tile_shift=10, lpanel=0, lflat=0, force_rp64=rp64, tiles_device=dev, tile_relaxed=1)
        A = sla.fromCSR(dims, rp, ci, va, ctx)
        info = A.kernel_info()
        if m > 1:   # (relaxed order has no layers to step aside for: every case but the single row takes the form)
            assert "algo=tiles" in info and "cu_slices=1" in info and "exact_fold=0" in info, (name, info)
            assert ("tile builder on device" in A.lower_info()) == (dev == 2), (name, dev, A.lower_info())
        for rep in range(3):
            y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
            assert np.all(np.abs(y - want) <= bound), (name, rp64, dev, float((np.abs(y - want) / bound).max()))
            assert np.array_equal(y[lens <= 2], want[lens <= 2]), (name, rp64, dev)
        if rp64 == "0" and dev == 2:   # pacing variants only change WHEN a tile is walked
            for opts in ({"tile_slack": 1}, {"tile_slack": 0}):
                ctx.set_options(**opts)
                y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
                assert np.all(np.abs(y - want) <= bound), (name, opts)
        del A
        ctx.close()


def test_cu_wide_tile_builder_in_chunks_gives_the_same_layout(sla):
    """The device builder sorts chunks of whole slices (bounded scratch: csrc/sla_tiles_build.hip); SLA_TILE_BUILD_CHUNK (entries per
    chunk, a test hook read at every build) cuts a test-sized matrix into many chunks -- one slice each, a few slices each, all in one --
    and the layout must not depend on it: rows of <= 2 entries are bit-exact in the relaxed kernel, all rows are within the bound, and
    the host builder (no chunks) agrees."""
    import os
    dims, (rp, ci, va) = _rand_rows(40000, 300000, lambda i, r: (1, 2, 2, 9, 30)[i % 5], 12)
    m, n = dims
    x = np.random.default_rng(4).standard_normal(n)
    want = orc.spmv(orc.Csr(m, n, rp, ci, va), x)
    lens = np.diff(rp)
    bound = lens * EPS * orc.spmv(orc.Csr(m, n, rp, ci, np.abs(va)), np.abs(x)) + 1e-300
    old = os.environ.get("SLA_TILE_BUILD_CHUNK")
    try:
        for chunk, dev in (("1", 2), ("3000", 2), ("50000", 2), (None, 2), (None, 0)):
            if chunk is None:
                os.environ.pop("SLA_TILE_BUILD_CHUNK", None)
            else:
                os.environ["SLA_TILE_BUILD_CHUNK"] = chunk
            ctx = sla.Context(0).set_options(tile_shift=12, lpanel=0, lflat=0, tiles_device=dev, tile_relaxed=1)
            A = sla.fromCSR(dims, rp, ci, va, ctx)
            assert "cu_slices=1" in A.kernel_info(), A.kernel_info()
            y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
            assert np.all(np.abs(y - want) <= bound), (chunk, dev)
            assert np.array_equal(y[lens <= 2], want[lens <= 2]), (chunk, dev)
            del A
            ctx.close()
    finally:
        if old is None:
            os.environ.pop("SLA_TILE_BUILD_CHUNK", None)
        else:
            os.environ["SLA_TILE_BUILD_CHUNK"] = old


@pytest.mark.parametrize("rowown", [-1, 0])
@pytest.mark.parametrize("name", list(CASES))
def test_tiles_match_the_oracle(sla, name, rowown):
    """The EXACT tile forms (tile_relaxed = 0, the default since the end of round 6): rowown = -1 (default) -- CU-wide slices with every row owned by one wavefront, all
    LDS adds of a row from that wavefront in program order (ascending panels, ascending columns); rowown = 0 -- the wavefront-private slices of
    rounds 2-4.  Both: every row bit for bit the reference's left fold, reruns bit-identical, device builder == host builder."""
    build, expect_tiles = CASES[name]
    dims, csr = build()
    m, n = dims
    rp, ci, va = csr
    Ao = orc.Csr(m, n, rp, ci, va)
    rng = np.random.default_rng(11)
    x = rng.standard_normal(n)
    want = orc.spmv(Ao, x)
    bound = np.diff(rp) * EPS * orc.spmv(orc.Csr(m, n, rp, ci, np.abs(va)), np.abs(x)) + 1e-300
    # (lpanel=0: the dense-row cases would otherwise take the LDS-panel form; tiles_device: the re-ordering as a device sort -- round 4,
    # sla_tiles_build.hip -- and by the host builder: the same decision and the same bits from both)
    for rp64, dev in (("0", 2), ("1", 2), ("0", 0), ("1", 0)):
        ctx = sla.Context(0).set_options(tile_shift=10, lpanel=0, lflat=0, force_rp64=rp64, tiles_device=dev, tile_relaxed=0, tile_rowown=rowown)
        A = sla.fromCSR(dims, rp, ci, va, ctx)
        info = A.kernel_info()
        if rowown != 0:
            expect_tiles = m > 1         # (CU-wide slices have no layers: dense rows among sparse ones do not turn the form down; only the single row does)
        assert ("algo=tiles" in info) == expect_tiles, (name, info)
        assert ("tile builder on device" in A.lower_info()) == (expect_tiles and dev == 2), (name, dev, A.lower_info())
        y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
        if expect_tiles:
            assert "exact_fold=1" in info and ("cu_slices=0" if rowown == 0 else "cu_slices=1 row_owned=1") in info, info
            assert A.props()["fold"] == 0                                     # SLA_FOLD_EXACT
            assert np.array_equal(y, want), (name, rp64, dev, int(np.count_nonzero(y != want)))
        else:
            assert np.all(np.abs(y - want) <= bound), (name, rp64, float(np.abs(y - want).max()))
        y2 = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
        assert np.array_equal(y, y2)                             # deterministic
        if expect_tiles and rp64 == "0" and dev == 2:
            # the kernel's pacing / prefetch variants (round 4: look-ahead poll by LDS-DMA, x-panel prefetch -- off by default, measured
            # slower -- and no pacing at all) only change WHEN a tile is walked, never what is added to a row sum
            for opts in ({"tile_poll": 0}, {"tile_prefetch": 1}, {"tile_prefetch": 3, "tile_slack": 1}, {"tile_slack": 0}):
                ctx.set_options(**opts)
                assert np.array_equal(sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV(), want), (name, opts)
            ctx.set_options(tile_poll=1, tile_prefetch=0, tile_slack=3)
    # the same matrix on the forms the tile form replaces
    ctx = sla.Context(0).set_options(tile_shift=10, lpanel=0, lflat=0, tiles=0, tile_relaxed=0)
    A = sla.fromCSR(dims, rp, ci, va, ctx)
    assert "tiles" not in A.kernel_info()
    y0 = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
    assert np.all(np.abs(y0 - want) <= bound)


@pytest.mark.parametrize("relaxed", [1, 0])
def test_tiles_transpose_and_solver_epilogues(sla, relaxed):
    """(<#), bicgsInit / bicgstabStep (EPI_SUB, EPI_DOT, EPI_DOT2), cgsStep (EPI_AXPY_DOT), cgneStep (EPI_AXPY_DOT on A,
    EPI_XPBY_NRM on the transpose) and linSolve0's residual sweep (EPI_RES) on the tile form -- the CU-wide relaxed-order kernel
    (default) and the bit-exact wavefront-private one."""
    from sla_amd import workloads as wl
    ctx = sla.Context(0).set_options(tile_shift=10, tile_relaxed=relaxed)
    n = 6000
    dims, (rp, ci, va) = wl.random_spd(n, 5, 3)
    A, Ao = sla.fromCSR(dims, rp, ci, va, ctx), orc.Csr(n, n, rp, ci, va)
    assert "algo=tiles" in A.kernel_info() and ("cu_slices=1 pacing" if relaxed else "cu_slices=1 row_owned=1") in A.kernel_info()
    rng = np.random.default_rng(5)
    u = rng.standard_normal(n)
    absA = orc.Csr(n, n, rp, ci, np.abs(va))
    same = (lambda got, ref, bnd: np.array_equal(got, ref)) if not relaxed else (lambda got, ref, bnd: bool(np.all(np.abs(got - ref) <= bnd)))
    assert same(sla.vecMat(sla.fromVector(u, ctx), A).toDenseListSV(), orc.spmv(orc.transpose(Ao), u),
                np.diff(orc.transpose(Ao).rowptr) * EPS * orc.spmv(orc.transpose(absA), np.abs(u)))
    xs = rng.standard_normal(n)
    b, x0 = orc.spmv(Ao, xs), np.full(n, 0.1)
    r0hat = b - orc.spmv(Ao, x0)
    so, sd = orc.BicgstabState(Ao, b, x0), sla.bicgsInit(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
    assert same(sd._rBicgstab.toDenseListSV(), so.r, np.diff(rp) * EPS * orc.spmv(absA, np.abs(x0)) + EPS * np.abs(so.r))   # r0 = b - A x0: same subtraction (relaxed: of a row sum within the bound)
    for k in (1, 2):
        so.step(r0hat, k); sd.step(k)
        assert np.linalg.norm(sd._xBicgstab.toDenseListSV() - so.x) <= 1e-9 * np.linalg.norm(so.x)
        assert np.linalg.norm(sd._pBicgstab.toDenseListSV() - so.p) <= 1e-8 * np.linalg.norm(so.p)
    so, sd = orc.CgsState(Ao, b, x0), sla.cgsInit(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
    so.step(r0hat, 3); sd.step(3)
    assert np.linalg.norm(sd._x.toDenseListSV() - so.x) <= 1e-9 * np.linalg.norm(so.x)
    assert np.linalg.norm(sd._u.toDenseListSV() - so.u) <= 1e-8 * np.linalg.norm(so.u)
    so, sd = orc.CgneState(Ao, b, x0), sla.cgneInit(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
    so.step(3); sd.step(3)
    assert np.linalg.norm(sd._xCgne.toDenseListSV() - so.x) <= 1e-9 * np.linalg.norm(so.x)
    for meth in (sla.BICGSTAB_, sla.CGS_, sla.CGNE_):
        x, info = sla.linSolve0(meth, A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx), return_info=True)
        rc, xo, it_o, res_o, r0_o = orc.linsolve0(meth, Ao, b, x0)
        assert info["converged"] and abs(info["iters"] - it_o) <= 2, (meth, info, it_o)
        assert np.linalg.norm(orc.spmv(Ao, x.toDenseListSV()) - b) <= info["tol"] * (1 + 1e-9)
        assert abs(info["r0norm"] - r0_o) <= 1e-12 * r0_o


def test_default_panel_width_picks_tiles_only_beyond_the_l2(sla):
    from sla_amd import workloads as wl
    ctx = sla.Context(0)
    dims, (rp, ci, va) = wl.random_spd(250000, 4, 1)          # x = 2 MB: fits the L2, no panels of any kind
    assert "tiles" not in sla.fromCSR(dims, rp, ci, va, ctx).kernel_info()
    dims, (rp, ci, va) = wl.random_spd(700000, 4, 1)          # 5.6 MB of x: six 1 MiB panels (CU-wide slices: 2^17 columns at every size)
    A = sla.fromCSR(dims, rp, ci, va, ctx)
    assert "algo=tiles" in A.kernel_info() and "panels=6 panel_cols=131072" in A.kernel_info() and "cu_slices=1" in A.kernel_info(), A.kernel_info()
    x = np.random.default_rng(2).standard_normal(dims[0])
    yo = orc.spmv(orc.Csr(*dims, rp, ci, va), x)
    bound = np.diff(rp) * EPS * orc.spmv(orc.Csr(*dims, rp, ci, np.abs(va)), np.abs(x))
    assert np.all(np.abs(sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV() - yo) <= bound)
    ctx0 = sla.Context(0).set_options(tile_relaxed=0, tile_rowown=0)   # the wavefront-private exact form: eleven 512 KiB panels (2^16 columns below 6 M columns), the reference's fold
    A0 = sla.fromCSR(dims, rp, ci, va, ctx0)
    assert "panels=11 panel_cols=65536" in A0.kernel_info() and "exact_fold=1 cu_slices=0" in A0.kernel_info(), A0.kernel_info()
    assert np.array_equal(sla.matVec(A0, sla.fromVector(x, ctx0)).toDenseListSV(), yo)
    ctx1 = sla.Context(0).set_option("tile_relaxed", 0)        # the exact form since round 6: the CU-wide slices, rows owned by wavefronts
    A1 = sla.fromCSR(dims, rp, ci, va, ctx1)
    assert "panels=6 panel_cols=131072" in A1.kernel_info() and "exact_fold=1 cu_slices=1 row_owned=1" in A1.kernel_info(), A1.kernel_info()
    assert np.array_equal(sla.matVec(A1, sla.fromVector(x, ctx1)).toDenseListSV(), yo)


@pytest.mark.parametrize("relaxed", [0, 1])
@pytest.mark.parametrize("ranks,rank,groups", [(4, 1, 4), (8, 7, 2), (3, 0, 3), (2, 1, 1)])
def test_panel_passes_of_the_overlapped_allgather_on_one_rank(sla, ranks, rank, groups, relaxed):
    """The pass structure of the overlapped all-gather (round 4) REHEARSED on a single-rank context (options
    ag_sim_ranks / ag_sim_rank: the tile launch of rank `rank` of `ranks` as the plan's panel passes, running row sums carried from
    pass to pass, no exchange): every cut of the visiting order -- many panels per pass (> 64: the offset block reload inside a
    pass), ragged rows, empty tiles, rectangular shapes.
      arrival order:   rows == the oracle's left fold over the panels in the plan's visiting order, bit for bit;
      ascending order: rows == the reference's ascending left fold (orc.spmv), bit for bit -- the carry through yinit is exact;
    (tile_relaxed = 1, the CU-wide kernel: the same passes, every row within nnz_i eps sum |a_ij x_j| of the ascending fold)
    and two BiCGSTAB steps + two CGS steps through the fused epilogues of the LAST pass against the oracle."""
    from sla_amd.partition import plan_allgather_passes
    shift = 10
    for label, (dims, (rp, ci, va)) in (("98 panels", _rand_rows(3000, 100000, lambda i, r: 40, 2)),
                                         ("ragged", _rand_rows(7001, 30000, lambda i, r: (0, 1, 3, 33, 200)[i % 5], 3)),
                                         ("square", _rand_rows(20000, 20000, lambda i, r: 17, 21))):
        m, n = dims
        Ao = orc.Csr(m, n, rp, ci, va)
        x = np.random.default_rng(9).standard_normal(n)
        yo = orc.spmv(Ao, x)
        for order in (0, 1):
            ctx = sla.Context(0).set_options(tile_shift=shift, lpanel=0, lflat=0, ag_sim_ranks=ranks, ag_sim_rank=rank, ag_groups=groups, ag_order=order, tile_relaxed=relaxed)
            A = sla.fromCSR(dims, rp, ci, va, ctx)
            info = A.kernel_info()
            visit, pptr, pneed, ng = plan_allgather_passes(ranks, rank, n, shift, groups, order)
            assert "algo=tiles" in info and f"allgather={'ascending' if order else 'arrival'} groups={ng} passes={len(pneed)} (rehearsal)" in info, info
            y = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
            if relaxed:
                assert f"cu_slices=1" in info
                assert np.all(np.abs(y - yo) <= np.diff(rp) * EPS * orc.spmv(orc.Csr(m, n, rp, ci, np.abs(va)), np.abs(x))), (label, info)
            elif order == 1:
                assert np.array_equal(y, yo), (label, info)
            else:
                assert np.array_equal(y, orc.spmv_panel_order(Ao, x, shift, visit)), (label, info)
                assert np.all(np.abs(y - yo) <= np.diff(rp) * EPS * orc.spmv(orc.Csr(m, n, rp, ci, np.abs(va)), np.abs(x)))
            if label == "square":
                # diagonally dominant variant: the fused epilogues (K1, K3 with four sums, CGS's C3) run in the last pass only
                d = np.zeros(n)
                np.add.at(d, np.repeat(np.arange(n), np.diff(rp)), np.abs(va))
                rows = np.repeat(np.arange(n), np.diff(rp))
                rr, cc, vv = np.append(rows, np.arange(n)), np.append(ci, np.arange(n)), np.append(va, d + 1.0)
                rc, Do = orc.coo_to_csr(n, n, rr, cc, vv)
                D = sla.fromCSR((n, n), Do.rowptr, Do.colidx, Do.val, ctx)
                assert "allgather=" in D.kernel_info()
                b = orc.spmv(Do, np.ones(n))
                x0 = np.zeros(n)
                so, sd = orc.BicgstabState(Do, b, x0), sla.bicgsInit(D, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
                so.step(b, 2)
                sd.step(2)
                for name, dev, ref in (("x", sd._xBicgstab, so.x), ("r", sd._rBicgstab, so.r), ("p", sd._pBicgstab, so.p)):
                    assert np.linalg.norm(dev.toDenseListSV() - ref) <= 1e-11 * np.linalg.norm(ref), (label, order, name)
                sc, sdc = orc.CgsState(Do, b, x0), sla.cgsInit(D, sla.fromVector(b, ctx), sla.fromVector(x0, ctx))
                sc.step(b, 2)
                sdc.step(2)
                assert np.linalg.norm(sdc._x.toDenseListSV() - sc.x) <= 1e-11 * np.linalg.norm(sc.x)
                xs, inf = sla.linSolve0(sla.BICGSTAB_, D, sla.fromVector(b, ctx), sla.fromVector(x0, ctx), return_info=True)
                assert inf["converged"] and np.linalg.norm(xs.toDenseListSV() - 1.0) <= 1e-4 * np.sqrt(n)
                del sd, sdc, D
            del A
            ctx.close()


def test_typed_fold_kind_flag_and_bit_identical_rerun_of_the_exact_tile_form(sla):
    """VERDICT r05 weak 4 / ADVICE r05: a caller must be able to LEARN, without parsing a string, that a matrix's (#>) is order-relaxed --
    sla_csr_get_props().fold and the SLA_FLAG_RELAXED_ORDER bit of sla_solve_info.flags -- and tile_relaxed = 0 must restore reruns that are
    bit-identical on a TILE-form matrix, not only on the banded one of test_rerun_is_bit_identical."""
    from sla_amd import _lib, workloads as wl
    n = 6000
    dims, (rp, ci, va) = wl.random_spd(n, 5, 3)
    rng = np.random.default_rng(9)
    b, x0 = orc.spmv(orc.Csr(n, n, rp, ci, va), rng.standard_normal(n)), np.zeros(n)
    runs = {}
    for relaxed in (1, 0):
        ctx = sla.Context(0).set_options(tile_shift=10, tile_relaxed=relaxed)
        A = sla.fromCSR(dims, rp, ci, va, ctx)
        assert "algo=tiles" in A.kernel_info()
        p = A.props()
        assert p["fold"] == (_lib.FOLD_RELAXED if relaxed else _lib.FOLD_EXACT) and p["x_exchange"] == 0 and p["nranks"] == 1
        assert p["rows_local"] == n and p["nnz_local"] == int(rp[-1]) and p["rowptr_bits"] == 32
        assert (f"exact_fold={0 if relaxed else 1}" in A.kernel_info())                     # the string and the typed field agree
        out = []
        for _ in range(3):
            x, info = sla.linSolve0(sla.BICGSTAB_, A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx), return_info=True)
            assert info["converged"] and info["relaxed_order"] is bool(relaxed) and bool(info["flags"] & _lib.FLAG_RELAXED_ORDER) is bool(relaxed)
            out.append((x.toDenseListSV(), info["iters"], info["resnorm"]))
        runs[relaxed] = out
        _, ig = sla.gmres(A, sla.fromVector(b, ctx), sla.fromVector(x0, ctx), restart=20, return_info=True)
        assert bool(ig["flags"] & _lib.FLAG_RELAXED_ORDER) is bool(relaxed)
        with pytest.raises(sla.SlaError):
            A.exchange_plan()                                                                # single-rank matrix: no plan
        del A
        ctx.close()
    (xa, ia, ra), (xb, ib, rb), (xc, ic, rc) = runs[0]
    assert np.array_equal(xa, xb) and np.array_equal(xa, xc) and ia == ib == ic and ra == rb == rc      # exact form: reruns bit-identical
    # the relaxed form: every run within the solver's tolerance of the exact form's answer (its last bits may differ from run to run)
    for xr, ir, rr in runs[1]:
        assert abs(ir - ia) <= 2 and np.linalg.norm(xr - xa) <= 1e-6 * np.linalg.norm(xa)


def test_fold_kind_of_the_other_forms(sla):
    from sla_amd import _lib, workloads as wl
    ctx = sla.Context(0)
    dims, (rp, ci, va) = wl.laplace3d(24, 24, 24)
    assert sla.fromCSR(dims, rp, ci, va, ctx).props()["fold"] == _lib.FOLD_EXACT                 # value-indexed: one lane per row
    dims, (rp, ci, va) = wl.banded_nonsym(30000)
    assert sla.fromCSR(dims, rp, ci, va, ctx).props()["fold"] == _lib.FOLD_EXACT
    dims, (rp, ci, va) = wl.random_spd(20000, 400, 7)                                            # dense rows: LDS panels, lane groups per segment
    A = sla.fromCSR(dims, rp, ci, va, ctx)
    assert A.props()["fold"] == _lib.FOLD_REGROUPED, A.kernel_info()
    x = np.random.default_rng(2).standard_normal(dims[0])
    y1 = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
    y2 = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
    assert np.array_equal(y1, y2)                                                                # regrouped, but fixed: reruns bit-identical


def test_row_owned_tiles_fold_in_ascending_order_when_lanes_collide(sla):
    """The exact CU-wide form relies on ONE wavefront adding all products of a row in program order -- and, where several lanes of one
    ds_add_f64 instruction hold the SAME row, on the LDS resolving them in ascending lane (= ascending column) order.  Provoked here: rows whose
    entries sit in clusters of consecutive columns that few other rows share (so a 64-entry group of a tile's column-sorted run is full of
    repeats of a row), with values whose sum depends on the order of the additions (+-2^60, +-1 and small fractions in one row).  Every row must
    still be the reference's left fold bit for bit, on both builders, and twice the same."""
    rng = np.random.default_rng(31)
    m, n, per = 6000, 40000, 24
    rows, cols, vals = [], [], []
    for i in range(m):
        starts = rng.choice(n - per, size=3, replace=False)
        cj = np.unique(np.concatenate([np.arange(s0, s0 + per // 3) for s0 in starts]))
        v = rng.standard_normal(len(cj))
        big = rng.choice(len(cj), size=4, replace=False)
        v[big] = np.array([2.0 ** 60, -2.0 ** 60, 2.0 ** 40, -2.0 ** 40]) * (1.0 + rng.random(4))
        rows.append(np.full(len(cj), i)); cols.append(cj); vals.append(v)
    rc, Ao = orc.coo_to_csr(m, n, np.concatenate(rows), np.concatenate(cols), np.concatenate(vals))
    assert rc == orc.OK
    x = 1.0 + rng.random(n)
    want = orc.spmv(Ao, x)
    other = orc.spmv(Ao, x)   # (sanity of the construction: a reversed fold differs on most rows)
    rev = np.array([np.sum((Ao.val[Ao.rowptr[i]:Ao.rowptr[i + 1]] * x[Ao.colidx[Ao.rowptr[i]:Ao.rowptr[i + 1]]])[::-1].cumsum()[-1:]) for i in range(200)])
    assert np.count_nonzero(rev != want[:200]) > 50 and np.array_equal(other, want)
    for dev in (2, 0):
        for shift in (10, 13):
            ctx = sla.Context(0).set_options(tile_shift=shift, lpanel=0, lflat=0, tiles_device=dev, tile_relaxed=0)
            A = sla.fromCSR((m, n), Ao.rowptr, Ao.colidx, Ao.val, ctx)
            assert "row_owned=1" in A.kernel_info(), A.kernel_info()
            y1 = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
            y2 = sla.matVec(A, sla.fromVector(x, ctx)).toDenseListSV()
            assert np.array_equal(y1, y2)
            assert np.array_equal(y1, want), (dev, shift, int(np.count_nonzero(y1 != want)))
            del A
            ctx.close()
