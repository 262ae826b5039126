"""The row-sharded code path with P > 1 ranks on ONE GPU: ranks are host threads, the communicator is the
library's in-process loopback test backend (RCCL refuses two ranks per GPU)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("kind", ["laplace", "banded", "random", "dense", "denseband"])
@pytest.mark.parametrize("nranks", [2, 3, 4])
def test_sharded_path_on_loopback_ranks(nranks, kind):
    out = subprocess.run([sys.executable, os.path.join(HERE, "_loopback_worker.py"), str(nranks), kind],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0 and f"LOOPBACK_OK {nranks} {kind}" in out.stdout, out.stdout[-3000:]
    assert f"BCG_OK {nranks}" in out.stdout      # (round 6: the sharded bcgStep -- (#>) with exchange, (<#) with reduce-scatter -- against the oracle)


def test_split_launch_partials_stay_inside_their_slot_array():
    """the interior and the boundary launch of an overlapped (#>) write their fused partial sums into consecutive
    slots of one 2048-slot array.  With SLA_WD_GRID=2048 on slabs of >= 2048 interior steps (128^3 on 2 ranks) an unclamped
    interior grid pushed the boundary partials into the NEXT array (omega silently wrong on the plain sharded flow): three
    BiCGSTAB steps must match the oracle."""
    env = dict(os.environ, SLA_WD_GRID="2048", SLA_BICG_GHOST="0")
    out = subprocess.run([sys.executable, os.path.join(HERE, "_loopback_worker.py"), "2", "laplace_2m"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0 and "LOOPBACK_OK 2 laplace_2m" in out.stdout, out.stdout[-3000:]
    assert "overlap=streams" in out.stdout and "grid=2048" in out.stdout, out.stdout[-2000:]


def _randtile(nranks, **env_extra):
    env = dict(os.environ, SLA_TILE_SHIFT="10", **{"SLA_TILE_RELAXED": "0", **env_extra})   # (the bit-for-bit claims below are the exact form's)
    out = subprocess.run([sys.executable, os.path.join(HERE, "_loopback_worker.py"), str(nranks), "randtile"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0 and f"LOOPBACK_OK {nranks} randtile" in out.stdout, out.stdout[-3000:]
    assert "algo=tiles" in out.stdout
    return out.stdout, sorted(l for l in out.stdout.splitlines() if l.startswith("XHASH"))


@pytest.mark.parametrize("nranks", [2, 3, 4, 8])
def test_tile_form_on_row_slabs(nranks):
    """The row-slice x column-panel tile form (BASELINE config 3a's SpMV) on ROW SLABS of a random matrix: SLA_TILE_SHIFT=10 makes
    a 20 000-row matrix take it (20 panels), every rank lowers its own slab (row_begin > 0, global column ids, x all-gathered into
    the full-length buffer); BiCGSTAB, CGS, CGNE, Arnoldi and GMRES run through the fused epilogues of the tile kernel.
    Round 4: the all-gather of x is OVERLAPPED with the SpMV -- G grouped send/recv exchanges on the comm
    stream, the tile launch as panel passes behind them, the running row sums carried from pass to pass.
      * default (arrival order, overlap on): every row = the left fold over the panels in the plan's visiting order, bit for bit
        against the oracle's restatement of that order (and the reference's ascending fold to rounding);
      * overlap = 0: the same groups serialised on the compute stream in front of the same passes: the SAME bits, all solvers;
      * overlap = -1 (plain ncclAllGather, one launch) and ag_order = 1 (source-ordered groups, ascending panel passes): the
        reference's ascending left fold bit for bit (Common.hs:247-260) -- and the same solver iterates as each other."""
    o1, h1 = _randtile(nranks)
    assert "allgather=arrival groups=4" in o1 and "PANEL_ORDER_ROWS_DIFFERING_FROM_ASCENDING" in o1, o1[-2000:]
    o0, h0 = _randtile(nranks, SLA_OVERLAP="0")
    assert "allgather=arrival" in o0 and h0 == h1, (h0, h1)
    om, hm = _randtile(nranks, SLA_OVERLAP="-1")
    assert "allgather=" not in om
    oa, ha = _randtile(nranks, SLA_AG_ORDER="1")
    assert "allgather=ascending" in oa and f"groups={nranks}" in oa and ha == hm, (ha, hm)
    o2, h2 = _randtile(nranks, SLA_AG_GROUPS="2")
    assert "allgather=arrival groups=2" in o2


@pytest.mark.parametrize("nranks", [2, 5])
def test_cu_wide_tile_form_on_row_slabs(nranks):
    """The relaxed-order tile form (opt-in tile_relaxed = 1: CU-wide slices, LDS atomics, csrc/sla_spmv_ctiles.hip) on row slabs:
    overlapped all-gather passes (running row sums carried through yinit), serialised groups, the plain all-gather and ascending
    source-ordered groups -- every row within nnz_i eps sum |a_ij x_j| of the reference's fold (the worker checks it), all solvers
    converge like the oracle's."""
    for extra in ({}, {"SLA_OVERLAP": "0"}, {"SLA_OVERLAP": "-1"}, {"SLA_AG_ORDER": "1"}):
        out, _ = _randtile(nranks, SLA_TILE_RELAXED="1", **extra)
        assert "cu_slices=1" in out and "RELAXED_ORDER_ROWS_WITHIN_BOUND" in out, out[-2000:]


@pytest.mark.parametrize("seed,nranks", [(1, 2), (2, 3), (3, 4), (4, 5), (6, 3)])
def test_sharded_path_on_random_banded_matrices(seed, nranks):
    """Random banded matrices (constant / arbitrary values, ragged rows, odd slab boundaries) through the sharded path:
    in-place halo exchange, the wave-sliced kernels on slabs with row_begin > 0, all solvers."""
    out = subprocess.run([sys.executable, os.path.join(HERE, "_loopback_worker.py"), str(nranks), "fuzz%d" % seed],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0 and f"LOOPBACK_OK {nranks} fuzz{seed}" in out.stdout, out.stdout[-3000:]


@pytest.mark.parametrize("kind,nranks", [("laplace", 3), ("fuzz1", 2), ("fuzz2", 3), ("fuzz4", 5), ("tinyband", 16), ("laplace_big", 2)])
def test_sharded_path_with_the_lds_window_kernel(kind, nranks):
    """SLA_WD_LDS=2 takes stencil slabs of <= 8 pairs through spmv_wdia_lds_kernel at any size: slabs with row_begin > 0 (odd
    first rows: the staged windows start one element early), ghost-extended x, interior / boundary step lists of the overlapped
    exchange.  The worker compares every (#>) with the oracle bit for bit and the solvers with its iterates."""
    env = dict(os.environ, SLA_WD_LDS="2")
    out = subprocess.run([sys.executable, os.path.join(HERE, "_loopback_worker.py"), str(nranks), kind], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0 and f"LOOPBACK_OK {nranks} {kind}" in out.stdout, out.stdout[-3000:]
    if kind.startswith("laplace"):
        assert "ldswin" in out.stdout, out.stdout[-3000:]


@pytest.mark.parametrize("kind,nranks", [("laplace_big", 2), ("laplace_big", 3), ("laplace_big", 5)])
def test_sharded_path_with_the_plane_march(kind, nranks):
    """wd_march=2 walks a row slab like a matrix of its own (planes counted from its first row; slabs whose first row is odd keep the
    LDS-window kernel): the whole-slab launches of the ghost-row solver flows take spmv_wdia_march_kernel with x addressed by global
    column, the interior / boundary launches of an overlapped exchange stay on the step-based kernels -- same rows either way.  The
    worker compares every (#>) with the oracle bit for bit and the solvers with its iterates."""
    env = dict(os.environ, SLA_WD_LDS="2", SLA_WD_MARCH="2")
    out = subprocess.run([sys.executable, os.path.join(HERE, "_loopback_worker.py"), str(nranks), kind], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0 and f"LOOPBACK_OK {nranks} {kind}" in out.stdout, out.stdout[-3000:]
    assert "wdia+march" in out.stdout, out.stdout[-3000:]


@pytest.mark.parametrize("nranks", [2, 3])
def test_k2_folded_into_k3_on_row_slabs_same_bits(nranks):
    """bicg_fuse23 on the ghost-row flow: where the whole-slab K3 runs the plane-march kernel, s = r - alpha Ap is built in the staged
    windows (ghost planes included: r and Ap are valid there) and the fused K4+K5 sweep rebuilds it on own + ghost rows -- the solution
    must be BIT-identical to the flow with K2 as a launch of its own, same iteration count."""
    got = {}
    for f23 in ("1", "0"):
        env = dict(os.environ, SLA_WD_LDS="2", SLA_WD_MARCH="2", SLA_BICG_FUSE23=f23, SLA_DEBUG_EXCHANGE="1")
        out = subprocess.run([sys.executable, os.path.join(HERE, "_loopback_worker.py"), str(nranks), "laplace_big"], env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert out.returncode == 0 and f"LOOPBACK_OK {nranks} laplace_big" in out.stdout, out.stdout[-3000:]
        assert "wdia+march" in out.stdout and "ghost-row BiCGSTAB" in out.stdout, out.stdout[-3000:]
        got[f23] = [l for l in out.stdout.splitlines() if l.startswith("XHASH")]
    assert got["1"] and got["1"] == got["0"], got


@pytest.mark.parametrize("kind,nranks", [("laplace", 3), ("laplace", 8), ("tiny", 4), ("tinyband", 16), ("banded", 2), ("denseband", 4), ("fuzz2", 3), ("fuzz5", 2), ("random", 2)])
def test_ghost_row_bicgstab_and_cgs_equal_the_plain_sharded_flow(kind, nranks):
    """Sharded BiCGSTAB keeps r, p, Ap and s valid on the ghost rows and needs 3 grouped exchanges per step instead of 5
    (enqueue_bicgstab_ghost); CGS likewise 2 instead of 4 (enqueue_cgs_ghost).  Every ghost value is computed from the same bits by the same kernel as on its owner, so
    the solution must be BIT-identical to the plain flow (SLA_BICG_GHOST=0), with the same iteration count.  ("random" and the
    5-row "tiny" matrix -- whose last rank owns no row at all -- use the all-gather exchange: the ghost flow must step aside
    there, on every rank alike.  "tinyband": 17 tridiagonal rows on 16 ranks, seven of them without rows, window exchange
    with ONE ghost row per side -- the odd count that the extended kernels round up to keep their 16-byte pairs aligned.)"""
    got = {}
    for ghost in ("1", "0"):
        # (SLA_OVERLAP=-1: the plain flow's exchange-preceded SpMVs would otherwise run as interior + boundary launches, whose fused
        # partial sums are grouped differently from the ghost flow's single launches -- same mathematics, other last bits)
        env = dict(os.environ, SLA_BICG_GHOST=ghost, SLA_DEBUG_EXCHANGE="1", SLA_OVERLAP="-1")
        out = subprocess.run([sys.executable, os.path.join(HERE, "_loopback_worker.py"), str(nranks), kind], env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        assert out.returncode == 0 and f"LOOPBACK_OK {nranks} {kind}" in out.stdout, out.stdout[-3000:]
        got[ghost] = [l for l in out.stdout.splitlines() if l.startswith("XHASH")]
        for name in ("ghost-row BiCGSTAB", "ghost-row CGS"):
            assert (name in out.stdout) == (ghost == "1" and kind not in ("random", "tiny")), out.stdout[-2000:]
    assert len(got["1"]) == 2
    assert got["1"] and got["1"] == got["0"], got


@pytest.mark.parametrize("nranks", [2, 4])
def test_overlapped_halo_exchange_equals_the_serial_flow_bit_for_bit(nranks):
    """the sharded (#>) runs the interior 512-row steps while the halo exchange is in flight on a second
    stream and the boundary steps after it (spmv_exchanged).  SLA_OVERLAP=0 issues the very same two launches with the
    exchange serialised on the compute stream: every solver iterate must be BIT-identical (plain sharded flows,
    SLA_BICG_GHOST=0: an exchange precedes each SpMV).  SLA_OVERLAP=-1 (one unsplit launch) groups the fused partial sums
    differently: same (#>) bits -- the worker compares those with the oracle in every mode -- and iteration counts within 4."""
    got, kern = {}, {}
    for ov in ("1", "0", "-1"):
        env = dict(os.environ, SLA_OVERLAP=ov, SLA_BICG_GHOST="0")
        out = subprocess.run([sys.executable, os.path.join(HERE, "_loopback_worker.py"), str(nranks), "laplace_big"], env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert out.returncode == 0 and f"LOOPBACK_OK {nranks} laplace_big" in out.stdout, out.stdout[-3000:]
        got[ov] = [l for l in out.stdout.splitlines() if l.startswith("XHASH")]
        kern[ov] = [l for l in out.stdout.splitlines() if l.startswith("KERNEL")][0]
    assert "overlap=streams" in kern["1"] and "overlap=serial" in kern["0"] and "overlap=" not in kern["-1"], kern
    assert "interior_steps=" in kern["1"] and "x_exchange=window" in kern["1"]
    assert len(got["1"]) == 2 and got["1"] == got["0"], got
    for a, b in zip(got["1"], got["-1"]):                                                    # iteration counts: Krylov-sensitive (50 vs 53 on 2 ranks)
        assert abs(int(a.split()[-1]) - int(b.split()[-1])) <= 4, (a, b)


def test_overlap_also_under_the_ghost_row_flows():
    """With the ghost-row BiCGSTAB / CGS (default) only the residual sweeps, the initial r0 = b - A x0, CGNE, GMRES and plain
    (#>) still exchange before a SpMV: those take the overlapped path, the rest is untouched."""
    out = subprocess.run([sys.executable, os.path.join(HERE, "_loopback_worker.py"), "3", "laplace_big"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0 and "LOOPBACK_OK 3 laplace_big" in out.stdout and "overlap=streams" in out.stdout, out.stdout[-3000:]
