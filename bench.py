#!/usr/bin/env python
"""bench.py -- BiCGSTAB iterations/s (+ CSR SpMV GB/s against the HBM3E roofline) on MI355X.

One "step" = one bicgstabStep (Sparse.hs:972-981: 2 SpMV + 5 inner products + 6 vector updates, all on
the device) on the BASELINE.json workload.  Default workload = configs[3], the one the metric is quoted
on at 1/2/4/8 GPUs: the 10M-row (216^3 = 10 077 696) fp64 7-point 3-D Laplacian; at N > 1 it is
row-sharded in contiguous slabs (strong scaling: total size fixed), each SpMV preceded by an exchange of its input
over RCCL (the slab's halo planes through grouped ncclSend/ncclRecv, or the all-gather when a matrix needs most
of x).  Inputs are resident in HBM before the timed region.

The ONE JSON line rank 0 prints carries, next to the contract fields:
  kernels         every kernel of the timed step (K1..K5), HIP-event timed inside the timed region on the library's
                  stream: ms, the bytes of the storage form it streams (`bytes`), the SURVEY 8(d) CSR figure
                  (`csr_bytes`), GB/s and fraction of the 8 TB/s peak for both;
  roofline        the DOMINANT kernel of the step (largest share of the step time).  `frac` is priced on the bytes of the
                  storage form actually streamed and is <= 1; `effective_*` uses the SURVEY 8(d) CSR bytes (a losslessly
                  compressed matrix form makes that exceed the streamed figure);
  general_csr     the same matrix with the value-indexed / dictionary forms off (SLA_WDIA=0 SLA_VDICT=0 SLA_DIAG=0): the
                  plain f64 + i32 CSR-stream kernel -- "CSR SpMV achieved HBM GB/s" in the literal sense;
  random_spd_10m  BASELINE configs[2] (10 M rows, ~33 random columns per row: the north star's target matrix);
  cpu_baseline    the oracle port timed on this host's cores (bounded sample).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--method bicgstab|cgs] [--mode step|linsolve0|gmres]
                    [--workload laplace3d_10m|poisson2d_1m|banded_2m|random_spd_1m|random_spd_10m|dense_rows_200k]
                    [--no-cpu-baseline] [--no-extra-blocks]

N > 1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) or stand-alone --
`python bench.py --gpus N` then spawns the N ranks itself (one process per GPU, rendezvous on 127.0.0.1).  With fewer
than N GPUs visible and SLA_BENCH_LOOPBACK=1 the N ranks run as threads of one process on one GPU through the library's
loopback communicator (a rehearsal of the sharded flow, flagged "loopback": true -- not a scaling measurement).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

# (the host driver of this pool only supports dmabuf IPC: without this RCCL's buffer sharing between ranks fails with
# hipIpcGetMemHandle: invalid argument -- set before any HIP / HSA library is loaded, whoever launched the ranks)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "sparse-linear-algebra_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def workload(name, row_begin=0, row_end=None):
    from sla_amd import workloads as wl
    from sla_amd.partition import local_rows_of
    if name == "laplace3d_10m":
        return "10M-row fp64 7-pt 3-D Laplacian (216^3 = 10077696 rows), BiCGSTAB", wl.laplace3d(216, 216, 216, row_begin, row_end)
    if name == "poisson2d_1m":
        return "1M-row fp64 5-pt Poisson (1000x1000), BiCGSTAB", wl.poisson2d(1000, 1000, row_begin, row_end)
    if name == "banded_2m":
        return "2M-row fp64 non-symmetric banded (5 bands), BiCGSTAB", wl.banded_nonsym(2000000, 99, row_begin, row_end)
    if name == "laplace3d_1m":   # the size of one rank's slab of the 216^3 problem at 8 GPUs
        return "108^3 7-pt Laplacian (1.26M rows)", wl.laplace3d(108, 108, 108, row_begin, row_end)
    if name == "laplace3d_slab8":  # ... in its real shape: 27 planes of 216 x 216
        return "216x216x27 7-pt Laplacian (1.26M rows: one rank's slab of the 216^3 problem at 8 GPUs)", wl.laplace3d(216, 216, 27, row_begin, row_end)
    if name == "laplace3d_5m":   # ... at 2 GPUs
        return "216x216x108 7-pt Laplacian (5.04M rows)", wl.laplace3d(216, 216, 108, row_begin, row_end)
    if name == "laplace3d_2m5":  # ... at 4 GPUs
        return "216x216x54 7-pt Laplacian (2.52M rows)", wl.laplace3d(216, 216, 54, row_begin, row_end)
    if name == "laplace3d_small":
        return "64^3 7-pt Laplacian (test size)", wl.laplace3d(64, 64, 64, row_begin, row_end)
    rand = {"random_spd_10m": (10000000, 16, "10M-row fp64 random SPD (~33 nnz/row, density 3.3e-6)"),
            "dense_rows_200k": (200000, 1000, "200k-row fp64 random SPD at 1 % density (~2000 nnz/row)"),
            "random_spd_1m": (1000000, 16, "1M-row fp64 random SPD (~33 nnz/row), BiCGSTAB"),
            "random_spd_small": (60000, 16, "60k-row random SPD (test size)")}
    if name in rand:
        n, k, desc = rand[name]
        dims, (rp, ci, va) = wl.random_spd(n, k, 42)
        return desc, (dims, local_rows_of(rp, ci, va, row_begin, dims[0] if row_end is None else row_end))
    raise SystemExit(f"unknown workload {name}")


import contextlib  # noqa: E402


@contextlib.contextmanager
def stdout_to_stderr():
    """Route file descriptor 1 to stderr for the duration (native libraries that print on stdout)."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def host_threads_available():
    """Cores this process may use: the affinity mask capped by a container CPU quota (cgroup v2 cpu.max)."""
    threads = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            threads = max(1, min(threads, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return threads


def cpu_baseline(dims, rp, ci, va, b, seconds):
    """The oracle (scalar C port of the reference's algorithm, 1 thread) timed on this host, on a bounded
    sample: as many bicgstabStep's of the SAME matrix as fit in ~`seconds`."""
    from oracle import oracle as orc
    n = dims[0]
    Ao = orc.Csr(n, n, rp, ci, va)
    x0 = np.zeros(n)
    st = orc.BicgstabState(Ao, b, x0)
    r0hat = b.copy()
    t0 = time.perf_counter()
    st.step(r0hat, 1)
    t1 = time.perf_counter() - t0
    steps = max(2, min(200, int(seconds / max(t1, 1e-6))))
    t0 = time.perf_counter()
    st.step(r0hat, steps)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    reps = max(1, min(50, int(0.25 * seconds / max(t1 / 3, 1e-6))))
    for _ in range(reps):
        orc.spmv(Ao, b)
    dts = (time.perf_counter() - t0) / reps
    out = {"value": steps / dt, "unit": "iters/s", "cores": 1, "kind": "port",
           "sample": f"{steps} bicgstabStep iterations of the same {n}-row matrix, single thread, oracle/sla_oracle.c (gcc -O2 -ffp-contract=off)",
           "host_cores_available": os.cpu_count(),
           "spmv_gbps": (12 * len(ci) + 20 * n) / dts / 1e9}
    # the same port built with OpenMP (row-parallel SpMV, parallel reductions), all host cores; every thread first-touches
    # its own rows of the matrix copy and of the vectors it works on (orc_numa_copy), so a multi-socket host is not
    # starved by one NUMA node's memory
    try:
        threads = host_threads_available()
        os.environ.setdefault("OMP_NUM_THREADS", str(threads))
        os.environ.setdefault("OMP_PROC_BIND", "spread")
        os.environ.setdefault("OMP_PLACES", "cores")
        orc.use_omp(True)
        Ap, bp, r0p = orc.numa_copy(Ao), orc.numa_vec(b), orc.numa_vec(r0hat)
        st = orc.BicgstabState(Ap, bp, x0)                    # (its x / r / p are first written by the parallel init loops)
        st.step(r0p, 2)                                       # warm up the thread team
        t0 = time.perf_counter()
        st.step(r0p, 1)
        t1 = max(time.perf_counter() - t0, 1e-6)
        osteps = max(5, min(400, int(0.4 * seconds / t1)))
        t0 = time.perf_counter()
        st.step(r0p, osteps)
        odt = time.perf_counter() - t0
        out["omp"] = {"value": osteps / odt, "unit": "iters/s", "cores": threads, "kind": "port",
                      "step_effective_gbps_on_csr_bytes": (24 * len(ci) + 160 * n) * osteps / odt / 1e9,
                      "temporaries": "kept across steps (no per-step malloc / first-touch faults in the timing build)",
                      "first_touch": "matrix and vectors copied row-parallel before timing (orc_par_copy_csr)",
                      "sample": f"{osteps} bicgstabStep iterations, OpenMP build of the same port (liboracle_omp.so), {threads} threads"}
    except Exception as e:  # the OpenMP leg is informative only
        out["omp"] = {"error": repr(e)}
    finally:
        orc.use_omp(False)
    return out


def kernel_table(ctx, A, nnz_local, n_local, method, steps_per_launch=1):
    """Per-kernel HIP-event statistics of the last ctx.prof_start(KERNEL_ALL) recording, priced two ways: `bytes` = the
    compulsory bytes of the storage form the kernel streams (matrix_bytes of sla_csr_kernel_info + the vectors),
    `csr_bytes` = the algorithmic CSR figure of SURVEY 8(d)."""
    from sla_amd import _lib
    info = A.kernel_info()
    mb = int(info.split("matrix_bytes=")[1].split()[0]) if "matrix_bytes=" in info else 12 * nnz_local + 4 * n_local
    n, z = n_local, nnz_local
    if method == "bicgstab":
        # single-rank flow (SLA_BICG_FUSE45, default): K3 also reads r0hat (rho_{j+1} = s . r0hat - omega As . r0hat by linearity),
        # K4 and K5 are ONE sweep over p, s, As, x, Ap -> x, r, p; the reference's split (sharded contexts, SLA_BICG_FUSE45=0)
        # shows up as K4 + K5 instead of K45
        fused = ctx.prof_query(_lib.KERNEL_BICG_K45)[0] > 0
        # round 5 (option bicg_fuse23, plane-march form on one rank): K2 is folded into K3 -- s = r - alpha Ap is built while the x windows are
        # staged and never stored (K4+K5 rebuild it from r and Ap): ONE launch "K23" that streams r, Ap, r0hat in and As out
        k23 = fused and ctx.prof_query(_lib.KERNEL_BICG_K2)[0] == 0 and ctx.prof_query(_lib.KERNEL_SPMV_DOT2)[0] > 0
        defs = [("K1", _lib.KERNEL_SPMV_DOT, "Ap = A p ; Ap . r0hat", mb + 24 * n, 12 * z + 28 * n),
                ("K2", _lib.KERNEL_BICG_K2, "alpha ; s = r - alpha Ap", 24 * n, 24 * n),
                ("K23", _lib.KERNEL_SPMV_DOT2, "alpha ; s = r - alpha Ap (staged, never stored) ; As = A s ; As . s, As . As, As . r0hat, s . r0hat",
                 mb + 32 * n, 12 * z + 52 * n) if k23 else
                ("K3", _lib.KERNEL_SPMV_DOT2, "As = A s ; As . s, As . As" + (", As . r0hat, s . r0hat" if fused else ""),
                 mb + (24 if fused else 16) * n, 12 * z + (28 if fused else 20) * n),
                ("K4", _lib.KERNEL_BICG_K4, "omega ; x += alpha p + omega s ; r = s - omega As ; r . r0hat", 56 * n, 56 * n),
                ("K5", _lib.KERNEL_BICG_K5, "beta ; p = r + beta (p - omega Ap)", 32 * n, 32 * n),
                ("K45", _lib.KERNEL_BICG_K45, "omega, beta ; x += alpha p + omega s ; r = s - omega As ; p = r + beta (p - omega Ap)",
                 64 * n, 64 * n)]
    else:   # cgsStep (Sparse.hs:928-939); the vectors each launch really streams (sla_solvers.cpp: enqueue_cgs -- the x update
        # rides in C2, which reads u, A p, x and writes q, u + q, x; C3 reads u + q, r, rhat and writes r)
        # (round 5, plane-march form on one rank: C2 folded away -- C3 builds u + q in its staged windows ("C23": u, A p, r, rhat in; r out),
        # one sweep does C2's x update and C4's u, p ("C24": u, A p, x, r, p in; x, u, p out))
        c23 = ctx.prof_query(_lib.KERNEL_CGS_C2)[0] == 0 and ctx.prof_query(_lib.KERNEL_CGS_C4)[0] > 0
        defs = [("C1", _lib.KERNEL_SPMV_DOT, "A p ; A p . rhat", mb + 24 * n, 12 * z + 28 * n)] + ([
                ("C23", _lib.KERNEL_SPMV_DOT2, "alpha ; u + q with q = u - alpha A p (staged, never stored) ; r -= alpha A (u + q) ; r . rhat", mb + 40 * n, 12 * z + 60 * n),
                ("C24", _lib.KERNEL_CGS_C4, "x += alpha (u + q) ; beta ; u, p updates", 64 * n, 64 * n)] if c23 else [
                ("C2", _lib.KERNEL_CGS_C2, "alpha ; q = u - alpha A p ; u + q ; x += alpha (u + q)", 48 * n, 48 * n),
                ("C3", _lib.KERNEL_SPMV_DOT2, "r -= alpha A (u + q) ; r . rhat", mb + 32 * n, 12 * z + 36 * n),
                ("C4", _lib.KERNEL_CGS_C4, "beta ; u, p updates", 40 * n, 40 * n)])
    out = {}
    # round 6 (option onchip): the whole step loop as ONE persistent launch with the solver state in registers + LDS (csrc/sla_onchip.hip).
    # What crosses HBM per LAUNCH: x, r, p, r0hat in and x, r, p out + the plan's tables (12 B per row); the steps themselves move the
    # boundary rows of Ap and As through the L2 / memory-side cache and nothing else -- `csr_bytes` (the SURVEY 8(d) step bytes x steps) is
    # what the launch flow would have streamed, so effective_* is an equivalence, not a bandwidth
    cnt, mean, mn = ctx.prof_query(_lib.KERNEL_ONCHIP)
    if cnt:
        bts, csr = 68 * n, (24 * z + 160 * n) * steps_per_launch
        out["ONCHIP"] = {"id": _lib.KERNEL_ONCHIP, "what": f"{steps_per_launch} bicgstabSteps in one persistent launch, solver state on chip; two counter barriers per step "
                         "carry the sums and the boundary rows of Ap / As", "launches": cnt, "ms": mean, "min_ms": mn, "steps_per_launch": steps_per_launch,
                         "ms_per_step": mean / steps_per_launch, "bytes": bts, "csr_bytes": csr, "gbps": bts / mean / 1e6, "frac": bts / mean / 1e6 / HBM_PEAK_GBS,
                         "effective_gbps": csr / mean / 1e6, "effective_frac": csr / mean / 1e6 / HBM_PEAK_GBS, "plan": ctx.get_option("onchip_plan")}
    for name, kid, what, bts, csr in defs:
        cnt, mean, mn = ctx.prof_query(kid)
        if not cnt:
            continue
        out[name] = {"id": kid, "what": what, "launches": cnt, "ms": mean, "min_ms": mn, "bytes": bts, "csr_bytes": csr,
                     "gbps": bts / mean / 1e6, "frac": bts / mean / 1e6 / HBM_PEAK_GBS,
                     "effective_gbps": csr / mean / 1e6, "effective_frac": csr / mean / 1e6 / HBM_PEAK_GBS}
    return out


def exchange_table(ctx):
    """Row-sharded contexts: HIP-event statistics of the collectives of the last KERNEL_ALL recording (rank 0's view; events on
    the stream each one is issued on -- the second stream for a halo exchange that overlaps the interior rows)."""
    from sla_amd import _lib
    out = {}
    for name, kid, what in (("x_exchange", _lib.KERNEL_EXCHANGE, "exchange of an SpMV's input vector (ncclAllGather / grouped halo ncclSend+ncclRecv)"),
                            ("sums", _lib.KERNEL_SUMS, "per-rank partial sums made global (finalize + all-gather; ghost-row flows: the grouped exchange that also carries a halo)")):
        cnt, mean, mn = ctx.prof_query(kid)
        if cnt:
            out[name] = {"what": what, "launches": cnt, "ms": mean, "min_ms": mn}
    return out


# xGMI on MI355X: 7 links per GPU (one to each peer of an 8-GPU node), ~153 GB/s per link and direction (SURVEY section 5 / the hardware brief).
XGMI_LINK_GBS = 153.0
XGMI_LINKS = 7
RCCL_LAUNCH_US = 20.0   # LABNOTES L6: 15-25 us per grouped RCCL launch over xGMI (an ESTIMATE until an N > 1 run replaces it)


def exchange_roofline(A, ex, kt, steps_recorded, step_ms, real_links, halo_rides_with_sums):
    """the link roofline of a row-sharded block.  For every event-timed exchange of exchange_table(): the bytes the PLAN
    moves per peer in one exchange (sla_csr_exchange_plan), GB/s on the busiest link against the per-link xGMI peak, aggregate receive rate
    against the links in use -- and next to the measurement the model DESIGN.md section 6 / LABNOTES L6 predict the first N > 1 run with
    (launch latency + busiest-peer bytes / link peak, exchanges not overlapped), so that a deviation names itself on first contact.
    real_links: the ranks are distinct GPUs (an RCCL communicator of > 1 rank); otherwise the bytes cross no link (loopback ranks / one rank)
    and the fractions only say what these bytes WOULD cost."""
    try:
        send, recv = A.exchange_plan()
        props = A.props()
    except Exception as e:   # (a single-rank matrix: no plan)
        return {"available": False, "why": str(e)}
    busiest = int(8 * max(int(send.max(initial=0)), int(recv.max(initial=0))))
    recv_total, send_total = int(8 * recv.sum()), int(8 * send.sum())
    peers = int(((send > 0) | (recv > 0)).sum())
    nr = int(props["nranks"])
    sums_bytes = 32 * max(nr - 1, 0)      # per-rank partial sums: <= 4 doubles to / from every peer
    out = {"link_peak_gbps": XGMI_LINK_GBS, "links_per_gpu": XGMI_LINKS, "mode": {0: "none", 1: "allgather", 2: "window"}[int(props["x_exchange"])],
           "peers": peers, "bytes_to_busiest_peer": busiest, "bytes_received_per_exchange": recv_total, "bytes_sent_per_exchange": send_total,
           "real_links": bool(real_links),
           "links": "xGMI between distinct GPUs" if real_links else
                    "none -- rehearsal (loopback ranks on one GPU or a 1-rank communicator): no byte crosses a link; fractions say what the plan's bytes would cost",
           "model": {"launch_latency_us": RCCL_LAUNCH_US, "link_gbps": XGMI_LINK_GBS, "formula": "ms = latency + bytes_to_busiest_peer / link peak, per exchange; "
                     "step = sum of the kernels' event times + the exchanges of a step, nothing overlapped (an upper bound where halos overlap interior rows)",
                     "source": "DESIGN.md section 6 / LABNOTES L6 (estimates; no N > 1 run exists yet)"},
           "exchanges": {}}
    model_step = sum(v["ms"] * v["launches"] for v in kt.values()) / max(steps_recorded, 1) if kt else 0.0
    for name, e in ex.items():
        carries_plan = name == "x_exchange" or (name == "sums" and halo_rides_with_sums)
        b_peer = (busiest if carries_plan else 0) + (32 if name == "sums" and nr > 1 else 0)
        b_recv = (recv_total if carries_plan else 0) + (sums_bytes if name == "sums" else 0)
        ms = e["ms"]
        model_ms = RCCL_LAUNCH_US * 1e-3 + b_peer / (XGMI_LINK_GBS * 1e9) * 1e3
        per_step = e["launches"] / max(steps_recorded, 1)
        model_step += per_step * model_ms
        out["exchanges"][name] = {
            "bytes_to_busiest_peer": b_peer, "bytes_received": b_recv, "ms": ms, "per_step": per_step,
            "gbps_busiest_link": b_peer / (ms * 1e-3) / 1e9 if ms else None,
            "frac_of_link_peak": b_peer / (ms * 1e-3) / 1e9 / XGMI_LINK_GBS if ms else None,
            "gbps_received": b_recv / (ms * 1e-3) / 1e9 if ms else None,
            "frac_of_links_in_use": b_recv / (ms * 1e-3) / 1e9 / (XGMI_LINK_GBS * max(1, min(peers, XGMI_LINKS))) if ms else None,
            "model_ms": model_ms, "measured_over_model": ms / model_ms if model_ms else None,
            "bound": "latency" if b_peer / (XGMI_LINK_GBS * 1e9) * 1e6 < RCCL_LAUNCH_US else "link"}
    out["step"] = {"measured_ms": step_ms, "model_ms": model_step, "measured_over_model": step_ms / model_step if model_step else None}
    return out


import threading as _threading
_TIMED = _threading.local()


def timed_steps(ctx, st, steps, warmup, sync_all, event_free=False, conv=None):
    """Warm-up (which also finds the kernel with the largest share of a step), then EXACTLY `steps` timed steps bracketed by
    barrier + sync, with HIP events around the dominant kernel's launches only (events around all five kernels of a
    0.3 ms step cost ~8 % of it), then an untimed pass of the same length with every kernel event-timed for the table.
    Returns (seconds of the timed region, (launches, mean ms, min ms) of the dominant kernel inside it, its kernel id)."""
    from sla_amd import _lib
    ids = (_lib.KERNEL_SPMV_DOT, _lib.KERNEL_SPMV_DOT2, _lib.KERNEL_BICG_K2, _lib.KERNEL_BICG_K4, _lib.KERNEL_BICG_K5,
           _lib.KERNEL_CGS_C2, _lib.KERNEL_CGS_C4, _lib.KERNEL_BICG_K45, _lib.KERNEL_ONCHIP)
    ctx.prof_start(_lib.KERNEL_ALL, max(warmup, 1) * 6 + 8)
    st.step(max(warmup, 1))
    ctx.prof_stop()
    tot = {k: (lambda c, m, _: c * m)(*ctx.prof_query(k)) for k in ids}
    dom = max(tot, key=tot.get)
    if dom == _lib.KERNEL_ONCHIP:
        event_free = False   # (one launch for all the steps: no graph replay, and the launch carries its own pair of events)
    if conv is not None:   # residual of the state the timed region starts from (collective when sharded: every rank calls it)
        conv["res_before"] = state_residual(ctx, st, st.A.ncols)
    # median and spread.  The timed region is `windows` consecutive windows of EXACTLY `steps` steps, each bracketed by
    # barrier + sync on both sides; the line's `value` is the MEDIAN window (ms_per_step x steps = that window), value_min / value_max the
    # slowest / fastest one.  _TIMED.windows holds every window's seconds (N > 1: the caller takes the max over ranks per window).
    windows = max(1, int(os.environ.get("SLA_BENCH_WINDOWS", "5")))
    sync_all()
    dts = []
    if event_free:
        # launch-bound sizes: the library replays the steps as a captured HIP graph (sla_solver_step) -- kernels inside a graph
        # cannot be bracketed by stream events, so the timed region runs clean and the dominant kernel's duration comes from the
        # untimed event pass below
        st.step(4)                                      # (capture + instantiate outside the timed region)
        sync_all()
        for _ in range(windows):
            t0 = time.perf_counter()
            st.step(steps)
            sync_all()
            dts.append(time.perf_counter() - t0)
        dom_stats = (0, 0.0, 0.0)
    else:
        ctx.prof_start(dom, steps * windows)
        for _ in range(windows):
            t0 = time.perf_counter()
            st.step(steps)
            sync_all()
            dts.append(time.perf_counter() - t0)
        dom_stats = ctx.prof_stop()
    _TIMED.windows = dts   # (thread-local: the loopback rehearsal runs its ranks as threads of one process)
    dt = sorted(dts)[len(dts) // 2]
    ctx.prof_start(_lib.KERNEL_ALL, steps * 6 + 8)      # untimed: the per-kernel table
    st.step(steps)
    ctx.prof_stop()
    return dt, dom_stats, dom


def state_residual(ctx, st, n):
    """||r|| of a solver state record (the recurrence residual), evaluated on the device."""
    import ctypes as C
    import sla_amd as sla
    from sla_amd import _lib
    r = sla.DeviceVector(ctx, n)
    _lib.check(_lib.lib().sla_solver_get(st.h, 1, r.h))
    out = C.c_double()
    _lib.check(_lib.lib().sla_nrm2(r.h, C.byref(out)))
    return out.value


def convergence_note(ctx, st, n, r0norm, res_before):
    """SURVEY 8(d) times >= 100 iterations per solver; a well-conditioned system converges sooner.  The kernels' work does not
    depend on the data (no early exit in the step functions), so the timing stands -- but say so in the line."""
    tol = max(1e-6, 1e-4 * r0norm)
    res_after = state_residual(ctx, st, n)
    ok = np.isfinite(res_after)
    return {"r0norm": r0norm, "tol": tol, "resnorm_before_timed_region": res_before, "resnorm_after": res_after if ok else None,
            "converged_before_timed_region": bool(res_before <= tol),
            "note": "step functions have no early exit: the timed kernels do the same work on a converged state (timing is data-independent)"}


def side_block(name, dims, rp, ci, va, options, steps, warmup, rhs="A.1"):
    """One more driver-timed block in the same process: lower (dims, rp, ci, va) on a fresh context with the typed knob settings
    `options` (sla_ctx_set_option), time `steps` bicgstabSteps (barrier + sync on both sides) with every kernel event-timed.
    SURVEY 8(d) protocol: >= 5 warm-ups; rhs "A.1" (configs 2 / 4: x* = 1) or "A.x*" with x* = N(0, 1), seed 7 (config 3)."""
    import sla_amd as sla
    steps, warmup = max(steps, 20), max(warmup, 5)
    ctx = sla.Context(0).set_options(**options)
    n, nnz = dims[0], int(rp[-1])
    t_low = time.perf_counter()
    A = sla.fromCSRRows(dims, 0, rp, ci, va, ctx)
    ctx.sync()
    t_low = time.perf_counter() - t_low
    if rhs == "A.1":
        bvec = sla.DeviceVector(ctx, n, np.add.reduceat(va, rp[:-1]), local=True)
    else:
        xstar = sla.DeviceVector(ctx, n, np.random.Generator(np.random.PCG64(7)).standard_normal(n), local=True)
        bvec = sla.DeviceVector(ctx, n)
        from sla_amd import _lib
        _lib.check(_lib.lib().sla_spmv(A.h, xstar.h, bvec.h))
        del xstar
    st = sla.bicgsInit(A, bvec, sla.DeviceVector(ctx, n))
    r0norm = state_residual(ctx, st, n)
    conv = {}
    dt, _, _ = timed_steps(ctx, st, steps, warmup, ctx.sync, conv=conv)
    kt = kernel_table(ctx, A, nnz, n, "bicgstab", steps)
    k1 = kt.get("K1", {})
    rec = {"workload": name, "rows": n, "nnz": nnz, "steps": steps, "warmup": warmup, "options": options, "rhs": rhs + ", x0 = 0",
           "convergence": convergence_note(ctx, st, n, r0norm, conv.get("res_before", float("nan"))),
           "value": steps / dt, "unit": "iters/s", "ms_per_step": dt / steps * 1e3,
           "step_effective_gbps_on_csr_bytes": (24 * nnz + 160 * n) / (dt / steps) / 1e9,
           "step_effective_frac_on_csr_bytes": (24 * nnz + 160 * n) / (dt / steps) / 1e9 / HBM_PEAK_GBS,
           "k1_ms": k1.get("ms"), "k1_gbps": k1.get("gbps"), "k1_frac": k1.get("frac"),
           "k1_csr_gbps": k1.get("effective_gbps"), "k1_csr_frac": k1.get("effective_frac"),
           "kernels": {k: {"ms": v["ms"], "frac": v["frac"]} for k, v in kt.items()},
           "spmv_kernel": A.kernel_info(),
           "lowered_once": {"from_csr_s": t_low, "phases_ms": A.lower_info(), "steps_it_buys": t_low * steps / dt}}
    del st, A
    ctx.close()
    return rec


def baseline_config_blocks():
    """BASELINE configs 2 and 5 beside the headline, in the driver's own line (round 6): each in its default flow -- ONE persistent launch where
    the state fits the chip (bicgstabStep on the 1 M-row Poisson matrix: csrc/sla_onchip.hip; the Gram-Schmidt of an Arnoldi step on the 2 M-row
    banded matrix: csrc/sla_arnoldi_orth.hip) -- and in the launch flow it replaces, same process, same box.  Median of five windows each."""
    import ctypes as C
    import sla_amd as sla
    from sla_amd import _lib, workloads as wl
    lib = _lib.lib()
    out = {}

    def median_rate(fn, units):
        dts = []
        for _ in range(5):
            t0 = time.perf_counter()
            fn()
            dts.append(time.perf_counter() - t0)
        dt = sorted(dts)[2]
        return {"value": units / dt, "us_per_unit": dt / units * 1e6, "value_min": units / max(dts), "value_max": units / min(dts)}

    dims, (rp, ci, va) = wl.poisson2d(1000, 1000)
    n, steps = dims[0], 200
    blk = {"workload": "config 2: 1M-row fp64 5-pt Poisson (1000^2), bicgstabStep, windows of 200 steps", "unit": "iters/s"}
    for label, opts in (("default", {}), ("launch_flow", {"onchip": 0})):
        ctx = sla.Context(0).set_options(**opts)
        A = sla.fromCSRRows(dims, 0, rp, ci, va, ctx)
        st = sla.bicgsInit(A, sla.DeviceVector(ctx, n, np.add.reduceat(va, rp[:-1]), local=True), sla.DeviceVector(ctx, n))
        st.step(steps)                                   # (plan / graph capture outside the timed windows)
        ctx.sync()
        r = median_rate(lambda: (st.step(steps), ctx.sync()), steps)
        r["onchip_launches"] = int(ctx.get_option("onchip_launches"))
        r["flow"] = "ONE persistent on-chip launch per window" if r["onchip_launches"] else "three launches per step (K1 | K23 | K45)"
        r["spmv_kernel"] = A.kernel_info().split()[0]
        blk[label] = r
        del st, A
        ctx.close()
    out["config2_poisson2d_1m"] = blk

    dims, (rp, ci, va) = wl.banded_nonsym(2000000)
    n, steps, restart = dims[0], 120, 30
    blk = {"workload": "config 5: 2M-row fp64 non-symmetric banded, GMRES(30), windows of 120 Arnoldi steps", "unit": "arnoldi_steps/s"}
    for label, opts in (("default", {}), ("launch_flow", {"arn_orth": 0})):
        ctx = sla.Context(0).set_options(**opts)
        A = sla.fromCSRRows(dims, 0, rp, ci, va, ctx)
        bvec, x0, res = sla.DeviceVector(ctx, n, np.add.reduceat(va, rp[:-1]), local=True), sla.DeviceVector(ctx, n), sla.DeviceVector(ctx, n)
        info = _lib.SolveInfo()
        o = _lib.SolveOpts(steps, 0.0, 0.0, 16, 1)       # tol 0: exactly `steps` Arnoldi steps

        def run():
            _lib.check(lib.sla_gmres(A.h, bvec.h, x0.h, restart, C.byref(o), res.h, C.byref(info)))
            ctx.sync()
        run()
        r = median_rate(run, steps)
        r["fused_gram_schmidt_launches"] = int(ctx.get_option("arn_orth_launches"))
        r["flow"] = "(#>) + ONE persistent Gram-Schmidt launch per step" if r["fused_gram_schmidt_launches"] else "(#>) + dots | update | normalisation"
        r["spmv_kernel"] = A.kernel_info().split()[0]
        blk[label] = r
        del bvec, x0, res, A
        ctx.close()
    out["config5_gmres_banded_2m"] = blk
    return out


def end_to_end_block(ctx, A, dims, rp, ci, va, b_host, t_from_csr, steady_its, with_coo):
    """What "lowered once" costs next to the steady-state rate: seconds of sla_csr_from_csr (already spent on the
    headline matrix: `t_from_csr`, phases as the library recorded them), of sla_csr_from_coo on the same entries in toListSM's
    DESCENDING (row, col) order (SpMatrix.hs:251-253: what a Haskell caller's triple list looks like), and the wall-clock of one COLD
    reference-faithful linSolve0 BICGSTAB_ through the host-array boundary: upload b (x0 = 0 needs none), solve (true residual every iteration,
    <= 200 iterations), download x.  cold_total = lowering + that call."""
    import sla_amd as sla
    from sla_amd import _lib
    import ctypes as C
    n = dims[0]
    out = {"workload_form": A.kernel_info().split()[0], "from_csr_s": t_from_csr, "from_csr_phases_ms": A.lower_info(),
           "host_bytes": int(12 * rp[-1] + 8 * (n + 1))}
    t0 = time.perf_counter()
    bv = sla.DeviceVector(ctx, n, b_host, local=True)
    xv = sla.DeviceVector(ctx, n)                      # x0 = 0 (the reference's empty SpVector): created on the device, nothing to upload
    ctx.sync()
    t1 = time.perf_counter()
    res = sla.DeviceVector(ctx, n)
    info = _lib.SolveInfo()
    _lib.check(_lib.lib().sla_linsolve0(4, A.h, bv.h, xv.h, None, res.h, C.byref(info)))
    t2 = time.perf_counter()
    xh = res.to_host()
    t3 = time.perf_counter()
    solve_s = t2 - t1
    out["linsolve0"] = {"iters": info.iters, "converged": bool(info.flags & 1), "resnorm": info.resnorm, "tol": info.tol,
                        "h2d_vectors_s": t1 - t0, "solve_s": solve_s, "iters_per_s_incl_true_residual": info.iters / solve_s if solve_s > 0 else None,
                        "d2h_x_s": t3 - t2, "x_checksum": float(np.sum(xh))}
    out["cold_linsolve0_s"] = t_from_csr + (t3 - t0)
    out["cold_over_solve"] = out["cold_linsolve0_s"] / solve_s if solve_s > 0 else None
    out["lowering_in_steady_state_steps"] = t_from_csr * steady_its
    del bv, xv, res
    if with_coo:
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))[::-1].copy()
        cols, vals = np.ascontiguousarray(ci[::-1], dtype=np.int64), np.ascontiguousarray(va[::-1])
        t0 = time.perf_counter()
        B = sla.fromCOO(dims, rows, cols, vals, ctx)
        ctx.sync()
        out["from_coo_s"] = time.perf_counter() - t0
        out["from_coo_phases_ms"] = B.lower_info()
        out["from_coo_form"] = B.kernel_info().split()[0]
        del B, rows, cols, vals
    return out


def sharded_random_block(ctx, name, rank, world, sync_all, allreduce, steps, warmup, real_links=False):
    """BASELINE config 3a (n rows, 16 random picks per row, symmetrised: ~33 entries per row) row-sharded over the ranks of `ctx`:
    every rank assembles its own slab (workloads.random_spd_rows), b = A x* with x* = N(0, 1) seed 7, x0 = 0;
    timed like the headline (barrier + sync on both sides, max over ranks).  Every rank must call this (collectives inside)."""
    import sla_amd as sla
    from sla_amd import _lib, workloads as wl
    from sla_amd.partition import row_block
    n, k = {"random_spd_10m": (10000000, 16), "random_spd_small": (60000, 16)}[name]
    rb, re_ = row_block(n, rank, world)
    t0 = time.perf_counter()
    # (the ranks of one node share its cores: each takes its share for the assembly -- an oversubscribed OpenMP team spins)
    dims, (rp, ci, va) = wl.random_spd_rows(n, k, 42, rb, re_, threads=max(1, min(16, host_threads_available() // world)))
    t_gen = time.perf_counter() - t0
    nnz_local, n_local = int(rp[-1]), re_ - rb
    A = sla.fromCSRRows(dims, rb, rp, ci, va, ctx)
    del rp, ci, va
    xstar = sla.DeviceVector(ctx, n, np.random.Generator(np.random.PCG64(7)).standard_normal(n)[rb:re_], local=True)
    bvec = sla.DeviceVector(ctx, n)
    _lib.check(_lib.lib().sla_spmv(A.h, xstar.h, bvec.h))
    st = sla.bicgsInit(A, bvec, sla.DeviceVector(ctx, n))
    r0norm = state_residual(ctx, st, n)
    conv = {}
    dt, _, _ = timed_steps(ctx, st, steps, warmup, sync_all, conv=conv)
    dt = allreduce(dt, "max")
    nnz = int(allreduce(nnz_local))
    kt = kernel_table(ctx, A, nnz_local, n_local, "bicgstab")
    ex = exchange_table(ctx)
    xroof = exchange_roofline(A, ex, kt, steps, dt / steps * 1e3, real_links, False)
    note = convergence_note(ctx, st, n, r0norm, conv.get("res_before", float("nan")))
    k1 = kt.get("K1", {})
    rec = {"workload": f"{n}-row fp64 random SPD (~33 nnz/row), row-sharded x{world}", "rows": n, "nnz": nnz, "steps": steps, "warmup": warmup,
           "rhs": "A.x* (x* = N(0,1), seed 7), x0 = 0", "convergence": note,
           "value": steps / dt, "unit": "iters/s", "ms_per_step": dt / steps * 1e3,
           "step_effective_gbps_on_csr_bytes": (24 * nnz + 160 * n) / (dt / steps) / 1e9,
           "step_effective_frac_on_csr_bytes": (24 * nnz + 160 * n) / (dt / steps) / 1e9 / (HBM_PEAK_GBS * world),
           "k1_ms": k1.get("ms"), "k1_frac": k1.get("frac"), "k1_csr_frac": k1.get("effective_frac"),
           "kernels_rank0": {k_: {"ms": v["ms"], "frac": v["frac"]} for k_, v in kt.items()},
           "exchanges_rank0": ex, "exchange_roofline": xroof, "spmv_kernel": A.kernel_info(), "slab_assembly_s": t_gen}
    kinfo = A.kernel_info()
    if "allgather=" in kinfo:   # the x all-gather goes out as grouped send/recv launches on the comm stream, the tile launch as panel passes behind them
        tok = dict(t.split("=") for t in kinfo.split() if "=" in t)
        rec["x_exchange"] = {"mode": "overlapped", "order": tok.get("allgather"), "groups": int(tok.get("groups", 0)), "passes": int(tok.get("passes", "0").split("(")[0]),
                             "note": "grouped ncclSend/ncclRecv exchanges on a second stream, one event per group; the tile launch runs as column-panel passes, "
                                     "each waiting only for the groups its panels need (own panels first); K1 / K3 times above include any exposed wait"}
    else:
        rec["x_exchange"] = {"mode": "serial", "note": "ncclAllGather of x, then one launch"}
    del st, A, bvec, xstar
    return rec


def pmc_traffic(workload_name, mode, world, kernel_name, kinfo):
    """HBM bytes per launch from the committed PMC passes (bench.py cannot collect counters on itself)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            tr = json.load(f).get(f"{workload_name}/{mode}/n{world}/{kernel_name}")
        if tr and tr.get("kernel_algo", "") in kinfo:
            return tr
    except OSError:
        pass
    return None


# ---- N > 1: first contact ------------------------------------------------------------------------------------------------------
# What the multi-rank run is doing right now, for the staged watchdog of main(): a hang names its collective.
STAGE = {"name": "start", "deadline": None}


def stamp(rank, text):
    sys.stderr.write(f"[bench {time.strftime('%H:%M:%S')}] rank {rank}: {text}\n")
    sys.stderr.flush()


def enter_stage(rank, name, limit_s=None):
    STAGE["name"] = name
    STAGE["deadline"] = None if limit_s is None else time.monotonic() + limit_s
    stamp(rank, name)


FALLBACK_OPTIONS = {"x_exchange": "allgather", "ag_groups": 0, "bicg_ghost": 0}


def preflight(ctx, rank, world, fallback, allreduce=None):
    """One checked collective at a time across the real ranks BEFORE anything is lowered or timed (sla_dist_preflight), each phase
    stamped on stderr: ncclAllGather, the integer all-reduce, then the all-gather as one ncclSend / ncclRecv group (the pattern of
    the halo exchange and of the overlapped all-gather).  If the grouped phase fails, the job falls back to the plain all-gather
    flow (FALLBACK_OPTIONS: every (#>) input through ncclAllGather, no ghost-row flows) and says so in its line; a phase that hangs
    is the staged watchdog's business (main()).  `fallback`: set when this process IS the re-run on the fallback flow."""
    import sla_amd as sla
    from sla_amd import _lib
    limit = float(os.environ.get("SLA_BENCH_PREFLIGHT_S", "120"))
    out = {}
    phases = [(0, "ncclAllGather"), (2, "ncclAllReduce(max, int32)")] + ([] if fallback else [(1, "grouped ncclSend/ncclRecv all-gather")])
    for ph, name in phases:
        enter_stage(rank, f"preflight: {name} across {world} rank(s)", limit)
        try:
            err, ms = ctx.preflight(ph, 1 << 16)
            if err != 0.0:
                raise _lib.SlaError(-1, f"{name}: data arrived wrong (max |arrived - sent| = {err:g})")
        except _lib.SlaError as e:
            if ph != 1:
                raise
            fallback = f"pre-flight of the grouped ncclSend/ncclRecv flow failed ({e}); every exchange through plain ncclAllGather instead"
            stamp(rank, "FALLBACK: " + fallback)
            break
        out[name] = {"ms": ms, "max_abs_err": err}
        stamp(rank, f"preflight: {name} ok in {ms:.1f} ms")
    # The fallback decision must be the SAME on every rank: a grouped phase that failed on some ranks only would leave the
    # ranks issuing different collectives in the next stage -- a hang with no deadline.  Agree over the CONTROL plane (gloo / the loopback
    # barrier, not the communicator under test): anyone's failure sends everybody to the fallback flow.
    if allreduce is not None and world > 1:
        enter_stage(rank, "preflight: agreeing on the flow across ranks", limit)
        anyone = allreduce(1.0 if fallback else 0.0, "max") > 0.0
        if anyone and not fallback:
            fallback = "pre-flight of the grouped ncclSend/ncclRecv flow failed on another rank; every exchange through plain ncclAllGather instead"
            out.pop("grouped ncclSend/ncclRecv all-gather", None)
            stamp(rank, "FALLBACK: " + fallback)
    if fallback:
        ctx.set_options(**FALLBACK_OPTIONS)
    enter_stage(rank, "lowering / timing")
    return out, fallback


def contract_allgather_block(ctx, dims, rb, rp, ci, va, b_local, n, method, steps, warmup, sync_all, allreduce, real_links=False):
    """BASELINE.json config 4 to the letter: "RCCL all-gather(x) per BiCGSTAB step" -- the same slabs lowered again under
    x_exchange = allgather / ag_groups = 0 (every (#>) input through ONE ncclAllGather into the full-length buffer, no halo
    exchange, no ghost-row flow), timed with the headline's protocol, collectives event-timed.  Every rank must call this."""
    import sla_amd as sla
    saved = {k: ctx.get_option(k) for k in FALLBACK_OPTIONS}
    ctx.set_options(**FALLBACK_OPTIONS)
    try:
        A = sla.fromCSRRows(dims, rb, rp, ci, va, ctx)
        kinfo = A.kernel_info()
        bvec = sla.DeviceVector(ctx, n, b_local, local=True)
        st = sla.bicgsInit(A, bvec, sla.DeviceVector(ctx, n)) if method == "bicgstab" else sla.cgsInit(A, bvec, sla.DeviceVector(ctx, n))
        dt, _, _ = timed_steps(ctx, st, steps, warmup, sync_all)
        dt = allreduce(dt, "max")
        kt = kernel_table(ctx, A, int(rp[-1]), len(b_local), method)
        ex = exchange_table(ctx)
        xroof = exchange_roofline(A, ex, kt, steps, dt / steps * 1e3, real_links, False)
        del st, A, bvec
    finally:
        ctx.set_options(**saved)
    xb = ex.get("x_exchange", {})
    return {"what": "config 4 with the contract's collective: ncclAllGather of the SpMV input vector (2 per step) into a full-length buffer; "
                    "the headline above exchanges halos instead (DESIGN.md section 6)",
            "value": steps / dt, "unit": "iters/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
            "x_exchange": "ncclAllGather" if "x_exchange=allgather" in kinfo and "allgather=" not in kinfo else "see spmv_kernel",
            "allgather_ms": xb.get("ms"), "allgather_launches": xb.get("launches"),
            "allgather_bytes_received_per_rank": 8 * (n - len(b_local)),
            "kernels_rank0": {k_: {"ms": v["ms"], "frac": v["frac"]} for k_, v in kt.items()}, "exchanges_rank0": ex, "exchange_roofline": xroof, "spmv_kernel": kinfo}


def run_rank(args, rank, world, local_rank, dist_mode, loop=None):
    """dist_mode: None (single GPU), "rccl" (one process per GPU; torch.distributed for the control plane),
    "loopback" (threads of one process on one GPU; `loop` = (barrier, shared dict))."""
    import sla_amd as sla
    from sla_amd import _lib
    from sla_amd.partition import row_block

    dist = None
    if dist_mode == "rccl":
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        # control plane (unique-id broadcast, barriers, max-over-ranks) on gloo; the data plane is the library's own
        # RCCL communicator over xGMI
        with stdout_to_stderr():      # (gloo announces its connections on stdout: the driver reads ONE JSON line there)
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            uid = [sla.Context.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
        ctx = sla.Context(local_rank, rank, world, uid[0])
    elif dist_mode == "loopback":
        ctx = sla.Context.loopback(rank, world, 4242)
    else:
        ctx = sla.Context(0)
        sla.set_default_context(ctx)
    fallback, pre = os.environ.get("SLA_BENCH_FALLBACK"), None
    real_links = dist_mode == "rccl" and world > 1      # distinct GPUs behind the ranks: the exchanges cross xGMI (exchange_roofline)

    def allreduce(v, op="sum"):
        if dist_mode == "rccl":
            import torch
            t = torch.tensor([v], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
            return float(t.item())
        if dist_mode == "loopback":
            barrier, shared = loop
            shared.setdefault(("ar", op), {})[rank] = v
            barrier.wait()
            vals = list(shared[("ar", op)].values())
            out = max(vals) if op == "max" else sum(vals)
            barrier.wait()
            if rank == 0:
                shared[("ar", op)] = {}
            barrier.wait()
            return out
        return v

    if dist_mode:
        pre, fallback = preflight(ctx, rank, world, fallback, allreduce)

    # ---- build this rank's slab on the host, lower it once to the device CSR ---------------------------
    if world == 1:
        desc, (dims, (rp, ci, va)) = workload(args.workload, 0, None)
        n = dims[0]
        rb, re_ = 0, n
    else:
        _, (dims, _) = workload(args.workload, 0, 1)
        n = dims[0]
        rb, re_ = row_block(n, rank, world)
        desc, (dims, (rp, ci, va)) = workload(args.workload, rb, re_)
    nnz_local = int(rp[-1])
    n_local = re_ - rb
    t_lower = time.perf_counter()
    A = sla.fromCSRRows(dims, rb, rp, ci, va, ctx)
    ctx.sync()
    t_lower = time.perf_counter() - t_lower
    b_local = np.add.reduceat(va, rp[:-1]) if nnz_local else np.zeros(0)   # b = A . 1  (x* = 1), x0 = 0
    if len(b_local) != n_local:                                              # (rows without entries at a slab's end)
        b_local = np.resize(b_local, n_local)

    nnz = int(allreduce(nnz_local))
    bvec = sla.DeviceVector(ctx, n, b_local, local=True)
    x0 = sla.DeviceVector(ctx, n)

    def sync_all():
        ctx.sync()
        if dist_mode == "rccl":
            import torch
            torch.cuda.synchronize()
            dist.barrier()
        elif dist_mode == "loopback":
            loop[0].wait()

    extra = {}
    lib = _lib.lib()
    import ctypes as C
    kt = {}
    if args.mode == "step":
        st = sla.bicgsInit(A, bvec, x0) if args.method == "bicgstab" else sla.cgsInit(A, bvec, x0)
        graph = (dist_mode is None and args.method == "bicgstab" or dist_mode is None and args.method == "cgs") and n_local <= 2500000 \
            and os.environ.get("SLA_STEP_GRAPH", "-1") != "0"
        extra["step_graph"] = bool(graph)
        dt, dom_stats, dom_id = timed_steps(ctx, st, args.steps, args.warmup, sync_all, event_free=graph or os.environ.get("SLA_BENCH_EVENT_FREE") == "1")
        kt = kernel_table(ctx, A, nnz_local, n_local, args.method, args.steps)
        extra["onchip"] = "ONCHIP" in kt
        if extra["onchip"]:
            extra["step_graph"] = False
        if dist_mode:
            extra["exchanges"] = exchange_table(ctx)
            # (exchange_roofline is added below, once `dt` is the median over the ranks' windows -- the step time the line reports)
        launches, mean_ms, min_ms = dom_stats
        step_bytes = 24 * nnz + 160 * n
        mode_desc = f"{'bicgstabStep' if args.method == 'bicgstab' else 'cgsStep'} (2 SpMV, no true-residual SpMV)"
    elif args.mode == "gmres":
        # config 5: GMRES(30) on the device Arnoldi; a "step" = one Arnoldi step (SpMV + 2-pass classical GS)
        out = sla.DeviceVector(ctx, n)
        info = _lib.SolveInfo()
        restart = 30
        o = _lib.SolveOpts(restart, 0.0, 0.0, 16, 1)
        _lib.check(lib.sla_gmres(A.h, bvec.h, x0.h, restart, C.byref(o), out.h, C.byref(info)))
        sync_all()
        o = _lib.SolveOpts(args.steps, 0.0, 0.0, 16, 1)                       # tol 0: exactly K Arnoldi steps
        ctx.prof_start(_lib.KERNEL_SPMV, 0 if os.environ.get("SLA_BENCH_EVENT_FREE") == "1" else args.steps)
        t0 = time.perf_counter()
        _lib.check(lib.sla_gmres(A.h, bvec.h, x0.h, restart, C.byref(o), out.h, C.byref(info)))
        sync_all()
        dt = time.perf_counter() - t0
        launches, mean_ms, min_ms = ctx.prof_stop()
        # B_arnoldi_step(k) = 12 nnz + 20 n + 16 k n + 40 n, k = basis size; averaged over a GMRES(30) cycle
        ks = [(i % restart) + 1 for i in range(args.steps)]
        step_bytes = sum(12 * nnz + 60 * n + 16 * k * n for k in ks) / len(ks)
        mode_desc = f"GMRES({restart}) Arnoldi step (SpMV + h = Q^T w + w -= Q h + normalise), averaged over the cycle"
        extra["gmres_iters"] = info.iters
    else:
        # reference-faithful linSolve0 iteration: bicgstabStep + true residual ||A x - b|| every iteration
        out = sla.DeviceVector(ctx, n)
        info = _lib.SolveInfo()
        o = _lib.SolveOpts(args.warmup, 0.0, 0.0, 16, 1)
        _lib.check(lib.sla_linsolve0(4, A.h, bvec.h, x0.h, C.byref(o), out.h, C.byref(info)))
        sync_all()
        o = _lib.SolveOpts(args.steps, 0.0, 0.0, max(args.steps, 1), 1)      # tol 0: run exactly K iterations
        kinfo0 = A.kernel_info()
        dual = (dist_mode is None and os.environ.get("SLA_DUAL_SPMV", "1") != "0" and os.environ.get("SLA_SPMV_ALGO", "stream") == "stream"
                and "algo=wdia" not in kinfo0 and "ldspanels" not in kinfo0 and "algo=tiles" not in kinfo0 and "colpanels" not in kinfo0)
        ctx.prof_start(_lib.KERNEL_SPMV_DUAL if dual else _lib.KERNEL_SPMV_DOT, args.steps)
        t0 = time.perf_counter()
        _lib.check(lib.sla_linsolve0(4, A.h, bvec.h, x0.h, C.byref(o), out.h, C.byref(info)))
        sync_all()
        dt = time.perf_counter() - t0
        launches, mean_ms, min_ms = ctx.prof_stop()
        # dual SpMV: the residual of the previous iterate rides on K1's matrix sweep (x and b are the only
        # extra streams): 24 nnz + 176 n per iteration instead of the three-sweep 36 nnz + 180 n
        step_bytes = 24 * nnz + 176 * n if dual else 36 * nnz + 180 * n
        mode_desc = ("linSolve0 iteration (bicgstabStep + per-iteration true residual fused into K1: 2 matrix sweeps)" if dual
                     else "linSolve0 iteration (bicgstabStep + per-iteration true residual, 3 SpMV)")
        extra["dual_spmv"] = dual
        extra["linsolve0_iters"] = info.iters

    if args.mode == "step":   # (every rank timed the same windows: max over ranks per window, then the median window)
        wins = [allreduce(w, "max") for w in _TIMED.windows]
        dt = sorted(wins)[len(wins) // 2]
        extra["value_windows"] = {"windows": len(wins), "steps_per_window": args.steps, "iters_per_s": [args.steps / w for w in wins],
                                  "note": "value = the median window; each window is exactly `steps` steps between barrier + device synchronisation"}
        extra["value_min"], extra["value_max"] = args.steps / max(wins), args.steps / min(wins)
        if dist_mode:
            ghost_flow = "x_exchange=window" in A.kernel_info() and ctx.get_option("bicg_ghost") != "0"
            extra["exchange_roofline"] = exchange_roofline(A, extra["exchanges"], kt, args.steps, dt / args.steps * 1e3, real_links, ghost_flow)
    else:
        dt = allreduce(dt, "max")

    # ---- N > 1: the north star's literal target next to the Laplacian -- the 10 M-row random matrix on the SAME sharded context
    # (x all-gathered per SpMV: its rows reference all of x), so that the driver's 1/2/4/8 sweep yields a curve for it as well
    rblock = None
    rname = {"auto": {"laplace3d_10m": "random_spd_10m", "laplace3d_small": "random_spd_small"}.get(args.workload)}.get(args.random_block, args.random_block)
    if dist_mode and args.mode == "step" and args.method == "bicgstab" and rname and rname != "none":
        enter_stage(rank, f"{rname} block (row-sharded random matrix, x all-gathered)")
        rblock = sharded_random_block(ctx, rname, rank, world, sync_all, allreduce, max(20, args.steps // 4), max(5, args.warmup // 2), real_links)

    cblock = None
    if dist_mode and args.mode == "step" and args.workload.startswith("laplace3d") and os.environ.get("SLA_BENCH_CONTRACT", "1") != "0":
        enter_stage(rank, "contract_allgather block (config 4 through ncclAllGather)")
        cblock = contract_allgather_block(ctx, dims, rb, rp, ci, va, b_local, n, args.method, max(20, args.steps // 2), max(5, args.warmup // 2),
                                          sync_all, allreduce, real_links)
        enter_stage(rank, "plain (#>) timing")

    # ---- plain SpMV bandwidth (rank-local rows; includes the exchange when sharded) -----------------------
    # Over ROTATING vector pairs: with one pair the 2 x 80 MB stay in the 256 MB memory-side cache (MALL) between
    # launches and the figure flatters the kernel; inside a solver the vectors never stay there.
    pairs = max(1, min(6, int(2.0e9 // max(1, 16 * n_local))))      # <= 2 GB of extra vectors
    xs = [sla.DeviceVector(ctx, n, np.full(n_local, 1.0 + 0.125 * i), local=True) for i in range(pairs)]
    ys = [sla.DeviceVector(ctx, n) for _ in range(pairs)]
    for i in range(max(5, pairs)):
        _lib.check(lib.sla_spmv(A.h, xs[i % pairs].h, ys[i % pairs].h))
    sync_all()
    reps = 60 if n_local < 5000000 or "wdia" in A.kernel_info() else 24
    ctx.prof_start(_lib.KERNEL_SPMV, reps)
    for i in range(reps):
        _lib.check(lib.sla_spmv(A.h, xs[i % pairs].h, ys[i % pairs].h))
    sp_launch, sp_mean_ms, sp_min_ms = ctx.prof_stop()
    ctx.prof_start(_lib.KERNEL_SPMV, reps)
    for _ in range(reps):
        _lib.check(lib.sla_spmv(A.h, xs[0].h, ys[0].h))
    _, sp_cached_ms, _ = ctx.prof_stop()
    spmv_bytes_local = 12 * nnz_local + 20 * n_local

    # ---- measured streaming ceiling of this GPU: the access shape of the vector kernels themselves (sla_stream_probe: R vectors
    # read + W written per element, 16 B per lane, same grid, non-temporal loads past the memory-side cache), HIP-event timed.
    # Pure read (8 R), the K4+K5 sweep's shape (5 R + 3 W) and a triad (2 R + 1 W); the ceiling is the best of the three.
    del xs[1:], ys[1:]
    probes = {}
    for nm, (r_, w_) in (("read_8r", (8, 0)), ("sweep_5r3w", (5, 3)), ("triad_2r1w", (2, 1))):
        pm, pmin, pg = ctx.stream_probe(r_, w_, max(n_local, 2), 20)
        probes[nm] = {"ms": pm, "min_ms": pmin, "gbps": pg, "bytes": 8 * (r_ + w_) * (max(n_local, 2) & ~1)}
    triad_gbps = max(v["gbps"] for v in probes.values())
    comm_ranks = ctx.comm_ranks()
    sync_all()
    if rank != 0:
        return None

    kinfo = A.kernel_info()
    rec = {
        "metric": "bicgstab_iters_per_sec" if args.mode != "gmres" else "gmres_arnoldi_steps_per_sec",
        "value": args.steps / dt,
        "unit": "iters/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": desc, "rows": n, "nnz": nnz, "timed": mode_desc,
                   "index_types": "i32 col / i32 rowptr", "parallelism": f"row-block x{world}",
                   "spmv_kernel": kinfo},
        "rccl_ranks": comm_ranks if dist_mode == "rccl" else None,
    }
    if dist_mode == "loopback":
        rec["loopback"] = True
        rec["loopback_note"] = (f"{world} ranks as threads of one process on ONE GPU through the library's loopback communicator "
                                "(SLA_BENCH_LOOPBACK=1): a rehearsal of the sharded flow, not a multi-GPU measurement")
        rec["loopback_ranks"] = comm_ranks
    if dist_mode:   # what the sharded step exchanges
        ghost = ("x_exchange=window" in kinfo and os.environ.get("SLA_BICG_GHOST", "1") != "0" and args.mode in ("step", "linsolve0"))
        rec["config"]["exchange"] = (
            ("halo (window) send/recv received in place" if "x_exchange=window" in kinfo else "all-gather of x")
            + (f"; ghost-row {args.method}: {(2 if ctx.get_option('bicg_fuse45') == '1' else 3) if args.method == 'bicgstab' else 2} grouped exchanges per step"
               + (" + the residual sweep's own exchange and sum" if args.mode == "linsolve0" else "") if ghost
               else ("; plain flow" if args.mode != "gmres" else "; Arnoldi: one exchange per SpMV, per-column sums all-gathered")))
    rec.update({
        "step_effective_gbps_on_csr_bytes": step_bytes / (dt / args.steps) / 1e9,   # SURVEY 8(d) CSR bytes of the step / step time ("effective")
        # (#>) alone, priced on the SURVEY 8(d) CSR bytes 12 nnz + 20 n whatever the storage form streams: on a value-indexed form
        # (the 216^3 stencil is stored in 10 MB) this EXCEEDS the HBM peak -- it is not a bandwidth; the literal CSR kernel's is csr_spmv_gbps
        "spmv_effective_gbps_on_csr_bytes": spmv_bytes_local * world / (sp_mean_ms * 1e-3) / 1e9 if sp_launch else None,
        "spmv_ms": sp_mean_ms,                     # rotating over `spmv_vector_pairs` x / y pairs (HBM-resident)
        "spmv_ms_cache_resident": sp_cached_ms,    # one pair re-used: x and y stay in the memory-side cache
        "spmv_vector_pairs": pairs,
        "hbm_measured_ceiling_gbps": triad_gbps,   # best of the three probe shapes below (same run, same vector length)
        "hbm_measured": probes,
    })
    # ---- roofline: the dominant kernel of the timed step -------------------------------------------------
    if kt:
        step_ms = dt / args.steps * 1e3
        dom = next(k for k in kt if kt[k]["id"] == dom_id)
        d = dict(kt[dom])
        if launches:   # the dominant kernel's own events INSIDE the timed region
            d.update({"launches": launches, "ms": mean_ms, "min_ms": min_ms, "gbps": d["bytes"] / mean_ms / 1e6,
                      "frac": d["bytes"] / mean_ms / 1e6 / HBM_PEAK_GBS, "effective_gbps": d["csr_bytes"] / mean_ms / 1e6,
                      "effective_frac": d["csr_bytes"] / mean_ms / 1e6 / HBM_PEAK_GBS})
        rec["kernels"] = kt
        rec["step_bytes_streamed"] = sum(v["bytes"] for v in kt.values())
        rec["step_gbps"] = rec["step_bytes_streamed"] / (dt / args.steps) / 1e9
        rec["step_frac_of_hbm_peak"] = rec["step_gbps"] / (HBM_PEAK_GBS * world)
        rec["roofline"] = {
            "bound": "hbm", "kernel": f"{dom}: {d['what']}" + (f" [{kinfo.split()[0]}]" if dom in ("K1", "K3", "K23", "C1", "C3", "C23") else ""),
            "share_of_step": d["ms"] * d["launches"] / args.steps / step_ms,
            "bytes_definition": "compulsory bytes of the storage form the kernel streams (matrix_bytes of the chosen SpMV form + the "
                                "vectors; equal to the SURVEY 8(d) figure for the vector kernels and for the plain CSR forms); "
                                "effective_* prices the same time on the SURVEY 8(d) CSR bytes (f64 values + i32 columns + i32 row pointers)",
            "achieved": d["gbps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["frac"],
            "effective_achieved": d["effective_gbps"], "effective_frac": d["effective_frac"],
            "frac_of_measured_ceiling": d["gbps"] / triad_gbps if triad_gbps else None,
            "traffic": None, "hbm_achieved": None, "hbm_frac": None,
            "bytes_per_launch": d["bytes"], "csr_bytes_per_launch": d["csr_bytes"], "avg_launch_ms": d["ms"], "min_launch_ms": d["min_ms"],
            "launches_timed": d["launches"],
            "timing": ("HIP events around this kernel's launches inside the timed region (library stream); the `kernels` table comes "
                       "from an untimed pass of the same length right after it, every kernel event-timed") if launches else
                      ("the timed region replays the steps as a captured HIP graph (launch-bound size: no stream events inside a graph); "
                       "this kernel's duration is from the event-timed pass of the same length right after it")}
        if dom == "ONCHIP":
            rec["roofline"]["note"] = ("persistent on-chip launch: the solver state lives in registers + LDS, so `achieved` (the bytes one LAUNCH moves: state in and "
                                       "out) says nothing about the steps -- they are bound by two grid-wide counter barriers and the issue rate of "
                                       "the CUs, not by HBM; effective_* = the SURVEY 8(d) step bytes the launch flow would stream / this time; ms_per_step "
                                       "is the figure of merit")
            rec["roofline"]["ms_per_step"] = d["ms"] / args.steps
        tr = pmc_traffic(args.workload, args.mode, world, dom, kinfo)
        if tr:
            rec["roofline"].update({"traffic": tr["traffic_bytes"], "traffic_source": tr["source"],
                                    "hbm_achieved": tr["traffic_bytes"] / (d["ms"] * 1e-3) / 1e9,
                                    "hbm_frac": tr["traffic_bytes"] / (d["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS})
    else:   # gmres / linsolve0 modes: the SpMV kernel that was event-timed
        vec_bytes = (20 if args.mode == "gmres" else 44 if extra.get("dual_spmv") else 28) * n_local
        csr_bytes = 12 * nnz_local + vec_bytes
        mb = int(kinfo.split("matrix_bytes=")[1].split()[0]) if "matrix_bytes=" in kinfo else 12 * nnz_local + 4 * n_local
        k1_bytes = mb + vec_bytes - 4 * n_local          # matrix_bytes already holds the row pointers of the CSR forms
        achieved = k1_bytes / (mean_ms * 1e-3) / 1e9 if launches else 0.0
        eff = csr_bytes / (mean_ms * 1e-3) / 1e9 if launches else 0.0
        rec["roofline"] = {"bound": "hbm", "kernel": f"{kinfo.split()[0]} SpMV of the timed mode", "achieved": achieved, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "effective_achieved": eff, "effective_frac": eff / HBM_PEAK_GBS,
                           "traffic": None, "bytes_per_launch": k1_bytes, "csr_bytes_per_launch": csr_bytes,
                           "bytes_definition": "compulsory bytes of the storage form streamed (matrix_bytes + the vectors): frac <= 1; effective_* "
                                               "prices the same time on the SURVEY 8(d) CSR bytes",
                           "avg_launch_ms": mean_ms, "min_launch_ms": min_ms, "launches_timed": launches}
    rec.update(extra)
    if dist_mode:
        rec["preflight"] = pre
        if fallback:
            rec["fallback"] = fallback
    if cblock:
        rec["contract_allgather"] = cblock
    if rblock:
        rec[rname if rname != "random_spd_small" else "random_spd_10m"] = rblock
    # ---- the blocks the default line carries next to the headline (single GPU, default workload only) --------
    if world == 1 and args.mode == "step" and args.method == "bicgstab" and args.workload == "laplace3d_10m" and not args.no_extra_blocks:
        bs, bw = max(20, args.steps // 2), max(5, args.warmup // 2)
        try:
            rec["general_csr"] = side_block(desc, dims, rp, ci, va, {"wdia": 0, "vdict": 0, "diag": 0}, bs, bw)
        except Exception as e:
            rec["general_csr"] = {"error": repr(e)}
    if world == 1 and not args.no_extra_blocks and args.mode == "step":
        try:
            rec["end_to_end"] = end_to_end_block(ctx, A, dims, rp, ci, va, b_local, t_lower, rec["value"], args.workload == "laplace3d_10m")
        except Exception as e:
            rec["end_to_end"] = {"error": repr(e)}
    if not args.no_cpu_baseline and world == 1:
        rec["cpu_baseline"] = cpu_baseline(dims, rp, ci, va, b_local, args.cpu_seconds)
    if world == 1 and args.mode == "step" and args.method == "bicgstab" and args.workload == "laplace3d_10m" and not args.no_extra_blocks:
        del xs, ys, bvec, x0, A
        rp = ci = va = None
        try:
            d3, (dm3, (rp3, ci3, va3)) = workload("random_spd_10m")
            rec["random_spd_10m"] = side_block(d3, dm3, rp3, ci3, va3, {}, max(20, args.steps // 4), max(5, args.warmup // 2), rhs="A.x*")
        except Exception as e:
            rec["random_spd_10m"] = {"error": repr(e)}
        try:   # the OPT-IN beside the default: row sums in relaxed order (tile_relaxed = 1: faster, not reproducible bit for bit)
            rec["random_spd_10m_relaxed_order"] = side_block(d3, dm3, rp3, ci3, va3, {"tile_relaxed": 1}, max(20, args.steps // 4), max(5, args.warmup // 2), rhs="A.x*")
        except Exception as e:
            rec["random_spd_10m_relaxed_order"] = {"error": repr(e)}
        rp3 = ci3 = va3 = None
        try:   # configs 2 and 5 in the same line (default flow and the launch flow it replaces)
            rec["baseline_configs"] = baseline_config_blocks()
        except Exception as e:
            rec["baseline_configs"] = {"error": repr(e)}
    # ---- BASELINE.json's metric, answered at the top level: "BiCGSTAB iters/sec + CSR SpMV achieved HBM GB/s, 10 M x 10 M fp64" ----
    g = rec.get("general_csr") or {}
    if g.get("k1_csr_gbps"):     # the literal CSR SpMV (f64 values + i32 columns + i32 row pointers) on config 4's matrix, K1 inside BiCGSTAB
        rec["csr_spmv_gbps"], rec["csr_spmv_frac"] = g["k1_csr_gbps"], g["k1_csr_frac"]
        rec["csr_spmv_kernel"] = g["spmv_kernel"].split()[0]
    r3 = rec.get("random_spd_10m") or {}
    if r3.get("value"):          # the north star's Target sentence: the 10 M-row random matrix (config 3a)
        rec["north_star_target"] = {"workload": r3["workload"], "iters_per_s": r3["value"], "k1_ms": r3.get("k1_ms"), "k1_frac": r3.get("k1_csr_frac"),
                                    "spmv_kernel": " ".join(t for t in r3.get("spmv_kernel", "").split() if t.startswith(("algo=", "exact_fold=", "cu_slices=", "row_owned=")))}
        rec["north_star_target"]["fold"] = "exact, reruns bit-identical (the default)" if "exact_fold=1" in r3.get("spmv_kernel", "") else "relaxed order"
        rx = rec.get("random_spd_10m_relaxed_order") or {}
        if rx.get("value"):      # what the opt-in buys, and the ceiling of ITS access pattern (the probe is the relaxed dealing's)
            rec["north_star_target"]["relaxed_order_opt_in"] = {
                "option": "tile_relaxed=1", "iters_per_s": rx["value"], "k1_ms": rx.get("k1_ms"), "k1_frac": rx.get("k1_csr_frac"),
                "spmv_kernel": " ".join(t for t in rx.get("spmv_kernel", "").split() if t.startswith(("algo=", "exact_fold=", "cu_slices=", "row_owned="))),
                "note": "row sums by LDS atomics in timing order: within nnz_i eps sum |a_ij x_j| of the reference's fold, not reproducible bit for bit; "
                        "announced by sla_csr_get_props().fold and SLA_FLAG_RELAXED_ORDER"}
            c = tile_form_ceiling(rx)
            if c:
                rec["north_star_target"]["relaxed_order_opt_in"]["ceiling"] = c
        c = tile_form_ceiling(r3) if "exact_fold=0" in r3.get("spmv_kernel", "") else None
        if c:
            rec["north_star_target"]["ceiling"] = c
    return rec


# The bare gather pattern of the CU-wide tile kernel (a workgroup walks ONE column-sorted run, 64 consecutive entries per gather instruction,
# products into LDS by ds_add_f64, 12 B per entry streamed) measured by tools/gather_share_probe.cpp at d entries per 128-byte line of x:
# G gathers / s, column "lds atomic: wg256" of profiles/r05_gather_share_probe.txt.
GATHER_SHARE_PROBE = ((0.25, 159.3), (0.5, 175.8), (1.0, 283.7), (2.0, 353.8), (4.0, 443.2), (16.0, 439.7))
L2_AGGREGATE_TBS = 34.5     # L2 -> L1 fabric of the eight XCDs together (MI355X_MICROARCH.md)


def tile_form_ceiling(r3):
    """what bounds config 3a's (#>) is not HBM.  With x beyond the L2 every row slice of the tile form touches (nearly)
    every 128-byte line of x once, so slices x 8 n bytes cross the L2 -> L1 fabric whatever the kernel does; the rate at which a CU turns such
    gathers around is what the bare pattern of the probe reaches at the matrix's density d = slice rows x entries per row x 16 / columns.
    k1_ms_at_ceiling = entries / that rate; frac_of_ceiling = it / the measured K1 -- the headroom the 0.39-of-HBM figure does not mean."""
    import math
    info = dict(t.split("=", 1) for t in r3.get("spmv_kernel", "").split() if "=" in t)
    if info.get("algo") != "tiles" or not r3.get("k1_ms") or "slices" not in info:
        return None
    rows, nnz, slices = r3["rows"], r3["nnz"], int(info["slices"])
    d = (rows / slices) * (nnz / rows) * 16.0 / rows
    pts = GATHER_SHARE_PROBE
    if d <= pts[0][0]:
        rate = pts[0][1]
    elif d >= pts[-1][0]:
        rate = pts[-1][1]
    else:
        (d0, r0), (d1, r1) = next((a, b) for a, b in zip(pts, pts[1:]) if a[0] <= d <= b[0])
        rate = r0 + (r1 - r0) * (math.log(d) - math.log(d0)) / (math.log(d1) - math.log(d0))
    ms_probe = nnz / (rate * 1e9) * 1e3
    ms_fabric = slices * 8.0 * rows / (L2_AGGREGATE_TBS * 1e12) * 1e3
    return {"basis": "L2 -> L1 gather fabric, one 128-byte line per touched line of x and slice -- not HBM (x is re-read from the L2s `slices` times)",
            "entries_per_x_line_and_slice": d, "probe_gathers_per_s": rate * 1e9,
            "probe": "tools/gather_share_probe.cpp, bare pattern with LDS atomics at this density (profiles/r05_gather_share_probe.txt)",
            "k1_ms_at_ceiling": ms_probe, "frac_of_ceiling": ms_probe / r3["k1_ms"],
            "fabric_bytes": slices * 8 * rows, "fabric_ms_at_34_5_TBs": ms_fabric,
            "hbm_roof_ms": (12 * nnz + 28 * rows) / 8e12 * 1e3,
            "note": "k1_frac prices K1 on CSR bytes against 8 TB/s of HBM, a roof this matrix cannot reach with 160 KB of LDS per CU; frac_of_ceiling is the honest headroom"}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: one child process per GPU, torchrun-style environment."""
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def loopback_ranks(args):
    """N ranks as N threads on ONE GPU (library loopback communicator): rehearsal of the sharded flow on a 1-GPU box."""
    barrier = threading.Barrier(args.gpus)
    shared, out, err = {}, {}, []

    def work(r):
        try:
            out[r] = run_rank(args, r, args.gpus, 0, "loopback", (barrier, shared))
        except BaseException as e:  # noqa: BLE001 -- a dead rank must not leave the others in a barrier
            err.append(e)
            barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(args.gpus)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if err:
        raise err[0]
    return out[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=os.environ.get("SLA_BENCH_WORKLOAD", "laplace3d_10m"))
    ap.add_argument("--mode", default="step", choices=["step", "linsolve0", "gmres"])
    ap.add_argument("--method", default="bicgstab", choices=["bicgstab", "cgs"], help="step mode: which solver step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-blocks", action="store_true", help="skip the general_csr / random_spd_10m blocks of the default line")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--random-block", default="auto",
                    help="N > 1: also time the row-sharded random SPD matrix on the same context (auto: random_spd_10m next to the default workload; none)")
    args = ap.parse_args()

    # The driver reads ONE JSON line from rank 0's stdout.  Native libraries print there too (gloo announces its connections, RCCL its
    # version banner at ncclCommInitRank): everything but the line goes to stderr, the line itself to the saved descriptor (handed
    # on through SLA_BENCH_JSON_FD when the staged watchdog below re-runs the process on the fallback flow).
    def claim_stdout():
        sys.stdout.flush()
        fd = int(os.environ["SLA_BENCH_JSON_FD"]) if os.environ.get("SLA_BENCH_JSON_FD") else os.dup(1)
        os.dup2(2, 1)
        return fd

    def start_watchdog(json_fd, rank, world):
        """A hung collective on the first multi-GPU contact must not take the run with it.  Two stages:
          1. a stage with a deadline of its own (the pre-flight phases: SLA_BENCH_PREFLIGHT_S, default 120 s each) that runs out -- or a
             peer rank saying so through a flag file (one node: /tmp is shared) -- makes EVERY rank replace itself (execve: a rank stuck
             inside RCCL cannot be unwound) by the same command on the fallback flow: SLA_BENCH_FALLBACK set, rendezvous one port up,
             grouped send / recv skipped, every exchange through plain ncclAllGather; the line then carries "fallback";
          2. past SLA_BENCH_WATCHDOG_S (default 1500 s; 0 = off) -- or a stage deadline missed on the fallback flow itself -- every rank
             dumps its Python stacks, rank 0 writes an error line naming the stage where the driver reads the JSON, and the process exits."""
        limit = float(os.environ.get("SLA_BENCH_WATCHDOG_S", "1500"))
        if limit <= 0:
            return
        import faulthandler
        t_end = time.monotonic() + limit
        flag = f"/tmp/sla_bench_fallback_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
        in_fallback = bool(os.environ.get("SLA_BENCH_FALLBACK"))
        if rank == 0 and not in_fallback and os.path.exists(flag):
            os.unlink(flag)                      # (stale: a flag is at least one stage deadline younger than its job)

        def bark(why):
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
            if rank == 0:
                os.write(json_fd, (json.dumps({"metric": "bicgstab_iters_per_sec", "value": None, "unit": "iters/s", "n_gpus": world,
                                               "error": f"watchdog: {why}; stage: {STAGE['name']} (stacks on stderr)"}) + "\n").encode())
            os._exit(3)

        def rerun(reason):
            stamp(rank, f"WATCHDOG: {reason} -> re-running on the fallback flow (plain ncclAllGather)")
            try:
                with open(flag, "w") as f:
                    f.write(reason)
            except OSError:
                pass
            os.set_inheritable(json_fd, True)
            env = dict(os.environ, SLA_BENCH_FALLBACK=f"watchdog: {reason}; re-run with every exchange through plain ncclAllGather",
                       SLA_BENCH_JSON_FD=str(json_fd))
            if env.get("MASTER_PORT"):
                env["MASTER_PORT"] = str(int(env["MASTER_PORT"]) + 1)
            os.execve(sys.executable, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env)

        def watch():
            t_start = time.monotonic()
            while True:
                time.sleep(0.5)
                now = time.monotonic()
                dl = STAGE["deadline"]
                if dl is not None and now > dl:
                    if in_fallback:
                        bark(f"stage deadline missed on the fallback flow after {now - t_start:.0f} s")
                    rerun(f"stage '{STAGE['name']}' did not finish within its deadline")
                if not in_fallback and now - t_start > 5 and os.path.exists(flag):
                    rerun("a peer rank asked for the fallback flow")
                if now > t_end:
                    bark(f"rank 0 still running after {limit:.0f} s")

        th = threading.Thread(target=watch, daemon=True)
        th.start()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        import sla_amd as sla
        have = sla.Context.device_count()
        if have >= args.gpus:
            spawn_ranks(args)                      # does not return
        if os.environ.get("SLA_BENCH_LOOPBACK") == "1" and have >= 1:
            json_fd = claim_stdout()
            start_watchdog(json_fd, 0, args.gpus)
            rec = loopback_ranks(args)
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(rec) + "\n").encode())
            return
        raise SystemExit(f"--gpus {args.gpus}: only {have} GPU(s) visible (SLA_BENCH_LOOPBACK=1 rehearses the sharded flow on one GPU)")
    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    json_fd = claim_stdout()
    # SLA_BENCH_FORCE_DIST=1 drives the multi-rank code path (gloo control plane, RCCL communicator, forced collectives)
    # with a 1-rank communicator on a single GPU
    use_dist = world > 1 or os.environ.get("SLA_BENCH_FORCE_DIST") == "1"
    if use_dist and world == 1:
        os.environ["SLA_FORCE_COLLECTIVES"] = "1"
    if use_dist:
        start_watchdog(json_fd, rank, world)
    rec = run_rank(args, rank, world, local_rank, "rccl" if use_dist else None)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(rec) + "\n").encode())
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
