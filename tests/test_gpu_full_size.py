"""BASELINE.json configurations at (or near) full size: size-independent properties and the bench contract."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sla():
    import sla_amd
    return sla_amd


def test_config5_banded_2m_gmres30(sla):
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.banded_nonsym(2000000)                       # config 5
    n = dims[0]
    A = sla.fromCSR(dims, rp, ci, va)
    assert "wdia-vv" in A.kernel_info()                                  # 5 noisy diagonals -> variable-coefficient wave-sliced form
    b = np.add.reduceat(va, rp[:-1])                                     # b = A 1
    x, info = sla.gmres(A, sla.fromVector(b), sla.fromVector(np.zeros(n)), restart=30, return_info=True)
    xd = x.toDenseListSV()
    r = sla.matVec(A, x).toDenseListSV() - b
    assert info["converged"] and np.linalg.norm(r) <= info["tol"] * (1 + 1e-9)
    assert np.abs(xd - 1.0).max() <= 1e-3                                # diagonally dominant: x* = 1
    # non-symmetric: (A u).v != (A v).u in general, but (A^T) is consistent with A: u.(A v) = (u <# A).v
    rng = np.random.default_rng(0)
    u, v = rng.standard_normal(n), rng.standard_normal(n)
    av = sla.matVec(A, sla.fromVector(v)).toDenseListSV()
    ua = sla.vecMat(sla.fromVector(u), A).toDenseListSV()
    assert abs(np.dot(u, av) - np.dot(ua, v)) <= 1e-9 * abs(np.dot(u, av))


def test_config3a_random_spd_1m_cgs_vs_bicgstab(sla):
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.random_spd(300000, 16, 42)                   # config 3a construction, reduced n
    n = dims[0]
    A = sla.fromCSR(dims, rp, ci, va)
    assert "diagdict" not in A.kernel_info()                             # random columns: i32 kernels
    rng = np.random.default_rng(7)
    xs = rng.standard_normal(n)
    b = sla.matVec(A, sla.fromVector(xs)).toDenseListSV()
    u, v = rng.standard_normal(n), rng.standard_normal(n)
    au = sla.matVec(A, sla.fromVector(u)).toDenseListSV()
    avv = sla.matVec(A, sla.fromVector(v)).toDenseListSV()
    assert abs(np.dot(v, au) - np.dot(u, avv)) <= 1e-10 * abs(np.dot(v, au))     # symmetric
    assert np.dot(u, au) > 0                                                      # positive definite
    for meth in (sla.CGS_, sla.BICGSTAB_):
        x, info = sla.linSolve0(meth, A, sla.fromVector(b), sla.fromVector(np.zeros(n)), return_info=True)
        assert info["converged"] and info["iters"] <= 60
        assert np.linalg.norm(x.toDenseListSV() - xs) <= 1e-3 * np.linalg.norm(xs)


def test_bench_default_line_as_the_driver_types_it():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` at FULL size (the CPU leg shortened): BASELINE.json's metric answered at the top level -- BiCGSTAB
    iterations / s on config 4, the literal CSR SpMV fraction, and the north star's own matrix (config 3a) under BOTH folds: the default (exact, reruns
    bit-identical: rows owned by wavefronts) and the opt-in relaxed order with the ceiling of its access pattern."""
    env = {k: v for k, v in os.environ.items() if not k.startswith("SLA_")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "0.5"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"] == "bicgstab_iters_per_sec" and d["n_gpus"] == 1 and d["steps"] == 20 and d["config"]["rows"] == 10077696
    assert 0.5 < d["roofline"]["frac"] <= 1.0 and 0.5 < d["csr_spmv_frac"] <= 1.0 and d["csr_spmv_kernel"] == "algo=stream+wave"
    ns = d["north_star_target"]
    assert "exact_fold=1" in ns["spmv_kernel"] and "row_owned=1" in ns["spmv_kernel"] and ns["fold"].startswith("exact")
    rx = ns["relaxed_order_opt_in"]
    assert "exact_fold=0" in rx["spmv_kernel"] and rx["option"] == "tile_relaxed=1"
    assert rx["iters_per_s"] > ns["iters_per_s"] > 0.7 * rx["iters_per_s"]                # the price of a reproducible default: 15 - 20 %
    assert 0.6 < rx["ceiling"]["frac_of_ceiling"] <= 1.0 and rx["ceiling"]["hbm_roof_ms"] < rx["ceiling"]["k1_ms_at_ceiling"] < rx["k1_ms"]
    # configs 2 and 5 ride in the same line: the default flows (one persistent launch per window / per Gram-Schmidt step) beside the launch flows
    c2, c5 = d["baseline_configs"]["config2_poisson2d_1m"], d["baseline_configs"]["config5_gmres_banded_2m"]
    assert c2["default"]["onchip_launches"] > 0 and c2["launch_flow"]["onchip_launches"] == 0
    assert c2["default"]["value"] > 1.5 * c2["launch_flow"]["value"] > 15000
    assert c5["default"]["fused_gram_schmidt_launches"] > 0 and c5["launch_flow"]["fused_gram_schmidt_launches"] == 0
    assert c5["default"]["value"] > c5["launch_flow"]["value"] > 4000


@pytest.mark.parametrize("fuse45", ["1", "0", "onchip"])
def test_bench_contract_small(fuse45):
    """fuse45 = 1 (single rank): K4 and K5 are one sweep (K45), K3 also streams r0hat; 0: the reference's split -- both with the launch
    flow (SLA_ONCHIP=0); "onchip": the default at this size since round 6 -- the whole timed window is ONE persistent launch."""
    onchip = fuse45 == "onchip"
    env = dict(os.environ, SLA_BICG_FUSE45="1" if onchip else fuse45, SLA_BICG_FUSE23="1", SLA_ONCHIP="1" if onchip else "0")
    env.pop("SLA_BENCH_WINDOWS", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "laplace3d_small", "--steps", "8",
                          "--warmup", "2", "--cpu-seconds", "0.5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                         env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                               # ONE JSON line
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    # value = the median of five consecutive windows of exactly `steps` steps, with the spread beside it
    assert d["value_windows"]["windows"] == 5 and d["value_windows"]["steps_per_window"] == 8 and len(d["value_windows"]["iters_per_s"]) == 5
    assert d["value_min"] <= d["value"] <= d["value_max"] and sorted(d["value_windows"]["iters_per_s"])[2] == pytest.approx(d["value"], rel=1e-9)
    # a roofline statement: every printed fraction is priced on bytes the kernel really streams and stays <= 1
    assert 0.0 < d["roofline"]["frac"] <= 1.0 and d["roofline"]["achieved"] <= d["roofline"]["peak"]
    # (fuse45 = 1 on this stencil: K2 is folded into K3 as well -- option bicg_fuse23, one launch "K23" that never stores s)
    assert d["onchip"] is onchip
    if onchip:   # one launch per window: the table's entry is per LAUNCH (8 steps), the roofline says what the figure means
        assert set(d["kernels"]) == {"ONCHIP"} and d["kernels"]["ONCHIP"]["launches"] == 1 and d["kernels"]["ONCHIP"]["steps_per_launch"] == 8
        assert d["roofline"]["launches_timed"] == 5 and "ms_per_step" in d["roofline"] and "note" in d["roofline"]
        assert "onchip:" in d["kernels"]["ONCHIP"]["plan"]
    else:
        assert set(d["kernels"]) == ({"K1", "K23", "K45"} if fuse45 == "1" else {"K1", "K2", "K3", "K4", "K5"})
        # (launch-bound size: the timed windows replay a captured graph -- no stream events inside it -- and the dominant kernel's figure comes
        # from the event-timed pass of one window's length right after them; otherwise its events cover all five windows)
        graph_replay = "captured HIP graph" in d["roofline"]["timing"]
        assert d["roofline"]["launches_timed"] == (8 if graph_replay else 8 * 5)
    for k in d["kernels"].values():
        assert 0.0 < k["frac"] <= 1.0 and k["bytes"] <= k["csr_bytes"] + 64 and k["launches"] == (1 if onchip else 8)
    assert d["roofline"]["kernel"].split(":")[0] in d["kernels"]
    assert d["rccl_ranks"] is None and "general_csr" not in d          # (side blocks ride on the default workload only)
    # nothing named *_gbps without "effective" may exceed the chip's HBM peak: value-indexed forms stream less
    # than the CSR bytes they are priced on, and say so in the key
    assert "spmv_gbps" not in d and d["spmv_effective_gbps_on_csr_bytes"] > 0

    def walk(o, path=""):
        if isinstance(o, dict):
            for k, v in o.items():
                if isinstance(v, (int, float)) and k.endswith("gbps") and "effective" not in k:
                    assert v <= d["roofline"]["peak"] * 1.0001, (path + k, v)
                walk(v, path + k + ".")
    walk(d)


def test_bench_two_ranks_without_a_launcher_loopback_rehearsal():
    """`python bench.py --gpus 2` as the driver types it, on a 1-GPU box: with SLA_BENCH_LOOPBACK=1 the two ranks run as threads
    through the library's loopback communicator (the real sharded flow: slabs, halo exchange plan, rank-ordered sums)."""
    env = dict(os.environ, SLA_BENCH_LOOPBACK="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "laplace3d_small", "--steps", "8",
                          "--warmup", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    import sla_amd
    if sla_amd.Context.device_count() >= 2:
        assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and "loopback" not in d      # a real node: RCCL saw both ranks
    else:
        assert d["n_gpus"] == 2 and d["loopback"] is True and d["loopback_ranks"] == 2
    assert "x_exchange=window" in d["config"]["spmv_kernel"] and "ghost-row bicgstab: 2 grouped exchanges" in d["config"]["exchange"]
    assert d["steps"] == 8 and 0.0 < d["roofline"]["frac"] <= 1.0
    # the north star's literal target rides along on the same sharded context (test size here): the random matrix with its x
    # all-gathered per SpMV, timed like the headline, its collectives event-timed
    rb = d["random_spd_10m"]
    assert rb["value"] > 0 and "x_exchange=allgather" in rb["spmv_kernel"] and rb["rows"] == 60000
    assert rb["exchanges_rank0"]["x_exchange"]["launches"] > 0 and rb["exchanges_rank0"]["sums"]["launches"] > 0
    assert d["exchanges"]["sums"]["launches"] > 0
    assert d["hbm_measured_ceiling_gbps"] >= d["hbm_measured"]["sweep_5r3w"]["gbps"] > 0
    # round 5: the three blocks of the N > 1 line -- halo headline, the contract's literal ncclAllGather flow, the random matrix --
    # behind a pre-flight that exercised every collective once (and checked what arrived)
    assert set(d["preflight"]) == {"ncclAllGather", "ncclAllReduce(max, int32)", "grouped ncclSend/ncclRecv all-gather"}
    assert all(v["max_abs_err"] == 0.0 for v in d["preflight"].values()) and "fallback" not in d
    c = d["contract_allgather"]
    assert c["value"] > 0 and "x_exchange=allgather" in c["spmv_kernel"] and c["x_exchange"] == "ncclAllGather" and c["allgather_launches"] > 0
    # round 6, SURVEY 8(d)/(e): every N > 1 block prices its event-timed exchanges against the xGMI link peak on the PLAN's bytes and puts the
    # model's prediction beside the measured step (a rehearsal says that no link was crossed)
    two_gpus = sla_amd.Context.device_count() >= 2
    for blk, mode in ((d, "window"), (rb, "allgather"), (c, "allgather")):
        xr = blk["exchange_roofline"]
        assert xr["link_peak_gbps"] == 153.0 and xr["links_per_gpu"] == 7 and xr["mode"] == mode and xr["peers"] == 1
        assert xr["real_links"] is two_gpus and ("rehearsal" in xr["links"]) is (not two_gpus)
        assert xr["bytes_to_busiest_peer"] > 0 and xr["step"]["model_ms"] > 0 and xr["step"]["measured_ms"] == pytest.approx(blk["ms_per_step"])
        for e in xr["exchanges"].values():
            for k in ("bytes_to_busiest_peer", "ms", "per_step", "gbps_busiest_link", "frac_of_link_peak", "model_ms", "measured_over_model", "bound"):
                assert k in e, k
            assert e["per_step"] > 0 and e["model_ms"] >= 0.02
    # the halo of the 64^3 test grid: one 64 x 64 plane of doubles per neighbour; the all-gather moves the peer's whole shard
    assert d["exchange_roofline"]["bytes_to_busiest_peer"] == 8 * 64 * 64
    assert d["exchange_roofline"]["exchanges"]["sums"]["bytes_to_busiest_peer"] == 8 * 64 * 64 + 32       # ghost-row flow: the halo rides with the sums
    assert c["exchange_roofline"]["bytes_to_busiest_peer"] == 8 * (64 ** 3) // 2


@pytest.mark.parametrize("fault,word", [("p2p", "pre-flight of the grouped"), ("p2p_hang", "watchdog"), ("p2p_data_rank1", "failed on another rank")])
def test_bench_two_ranks_fallback_ladder(fault, word):
    """First contact gone wrong, rehearsed on one GPU (SLA_FAULT_INJECT, csrc/sla_dist.cpp): the grouped ncclSend / ncclRecv flow
    ERRORS in the pre-flight ("p2p"), HANGS there ("p2p_hang": the staged watchdog replaces the process after
    SLA_BENCH_PREFLIGHT_S) or fails on ONE rank only ("p2p_data_rank1": rank 1 alone sees wrong data; the ranks agree on the fallback over
    the control plane, so rank 0 -- whose own pre-flight passed -- follows) -- either way the driver still gets ONE line, on the plain ncclAllGather flow, with "fallback" saying why."""
    env = dict(os.environ, SLA_BENCH_LOOPBACK="1", SLA_FAULT_INJECT=fault, SLA_BENCH_PREFLIGHT_S="4")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "laplace3d_small", "--steps", "8",
                          "--warmup", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["value"] and d["value"] > 0 and word in d["fallback"], d.get("fallback")
    assert "x_exchange=allgather" in d["config"]["spmv_kernel"] and "grouped ncclSend/ncclRecv all-gather" not in d["preflight"]
    assert d["random_spd_10m"]["value"] > 0 and d["random_spd_10m"]["x_exchange"]["mode"] == "serial"


def test_full_size_triangular_solves_recover_ones(sla):
    """SURVEY 8(f).2 at BASELINE size: forward / backward substitution with the triangles of the 216^3 Laplacian
    (10 M rows, 646 dependency levels = hyperplanes of the grid).  With b = T . 1 every row's arithmetic is exact in
    fp64 (small integers, division by 6), so the substitution must return 1.0 in all 10 077 696 rows."""
    from sla_amd import workloads as wl
    dims, (rp, ci, va) = wl.laplace3d(216, 216, 216)
    n = dims[0]
    T = sla.fromCSR(dims, rp, ci, va)
    rows = np.repeat(np.arange(n), np.diff(rp))
    for upper in (False, True):
        keep = (ci >= rows) if upper else (ci <= rows)
        b = np.bincount(rows[keep], weights=va[keep], minlength=n)           # (triangle of A) . 1
        assert sla.triSolveLevels(T, upper) == (216 * 3 - 2, 34992)
        x = (sla.triUpperSolve if upper else sla.triLowerSolve)(T, sla.DeviceVector(T.ctx, n, b)).to_host()
        assert np.array_equal(x, np.ones(n))


def test_bench_stdout_is_one_json_line_on_the_rccl_path():
    """The driver reads ONE JSON line from rank 0's stdout.  On the multi-rank path native libraries print there too (gloo
    announces its connections, RCCL its version banner at ncclCommInitRank): exercised here with the 1-rank RCCL communicator
    (SLA_BENCH_FORCE_DIST=1) -- stdout must hold exactly the line, everything else goes to stderr."""
    env = dict(os.environ, SLA_BENCH_FORCE_DIST="1", SLA_X_EXCHANGE="window", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "laplace3d_small", "--steps", "8", "--warmup", "2",
                          "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("\n") == 1 and out.stdout.startswith("{"), out.stdout[:400]
    d = json.loads(out.stdout)
    assert d["rccl_ranks"] == 1 and d["exchanges"]["sums"]["launches"] > 0
    xr = d["exchange_roofline"]          # one rank: the plan moves nothing, the line still carries the model and says it is a rehearsal
    assert xr["real_links"] is False and xr["peers"] == 0 and xr["bytes_to_busiest_peer"] == 0 and xr["step"]["model_ms"] > 0
    # the fused sweep runs on sharded contexts too (round 3); with ghost rows K2 is folded into K3 as on one rank (round 5)
    assert set(d["kernels"]) in ({"K1", "K2", "K3", "K45"}, {"K1", "K23", "K45"})
