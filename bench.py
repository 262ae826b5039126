#!/usr/bin/env python
"""bench.py -- BiCGSTAB iterations/s (+ CSR SpMV GB/s against the HBM3E roofline) on MI355X.

One "step" = one bicgstabStep (Sparse.hs:972-981: 2 SpMV + 5 inner products + 6 vector updates, all on
the device) on the BASELINE.json workload.  Default workload = configs[3], the one the metric is quoted
on at 1/2/4/8 GPUs: the 10M-row (216^3 = 10 077 696) fp64 7-point 3-D Laplacian (the "~1 % density" of
configs[2] would be 10^12 entries; its 33-entries-per-row reading is --workload random_spd_10m); at N > 1 it is
row-sharded in contiguous slabs (strong scaling: total size fixed), each SpMV preceded by an exchange of its input
over RCCL (the slab's halo planes through grouped ncclSend/ncclRecv, or the all-gather when a matrix needs most
of x).  Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--method bicgstab|cgs] [--mode step|linsolve0|gmres] [--no-cpu-baseline]
                    [--workload laplace3d_10m|poisson2d_1m|banded_2m|random_spd_1m|random_spd_10m|dense_rows_200k]

For N > 1 launch through torch.distributed.run (one rank per GPU); rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "sparse-linear-algebra_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def workload(name, row_begin=0, row_end=None):
    from sla_amd import workloads as wl
    if name == "laplace3d_10m":
        return "10M-row fp64 7-pt 3-D Laplacian (216^3 = 10077696 rows), BiCGSTAB", wl.laplace3d(216, 216, 216, row_begin, row_end)
    if name == "poisson2d_1m":
        return "1M-row fp64 5-pt Poisson (1000x1000), BiCGSTAB", wl.poisson2d(1000, 1000, row_begin, row_end)
    if name == "banded_2m":
        return "2M-row fp64 non-symmetric banded (5 bands), BiCGSTAB", wl.banded_nonsym(2000000, 99, row_begin, row_end)
    if name == "laplace3d_1m":   # the size of one rank's slab of the 216^3 problem at 8 GPUs
        return "108^3 7-pt Laplacian (1.26M rows)", wl.laplace3d(108, 108, 108, row_begin, row_end)
    if name == "laplace3d_small":
        return "64^3 7-pt Laplacian (test size)", wl.laplace3d(64, 64, 64, row_begin, row_end)
    if name == "random_spd_10m":
        dims, (rp, ci, va) = wl.random_spd(10000000, 16, 42)
        row_end = dims[0] if row_end is None else row_end
        from sla_amd.partition import local_rows_of
        return "10M-row fp64 random SPD (~33 nnz/row, density 3.3e-6)", (dims, local_rows_of(rp, ci, va, row_begin, row_end))
    if name == "dense_rows_200k":
        dims, (rp, ci, va) = wl.random_spd(200000, 1000, 42)
        row_end = dims[0] if row_end is None else row_end
        from sla_amd.partition import local_rows_of
        return "200k-row fp64 random SPD at 1 % density (~2000 nnz/row)", (dims, local_rows_of(rp, ci, va, row_begin, row_end))
    if name == "random_spd_1m":
        dims, (rp, ci, va) = wl.random_spd(1000000, 16, 42)
        row_end = dims[0] if row_end is None else row_end
        from sla_amd.partition import local_rows_of
        return "1M-row fp64 random SPD (~33 nnz/row), BiCGSTAB", (dims, local_rows_of(rp, ci, va, row_begin, row_end))
    raise SystemExit(f"unknown workload {name}")


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
    # "fair CPU": the same port built with OpenMP (row-parallel SpMV, parallel reductions), all host cores
    try:
        threads = len(os.sched_getaffinity(0))
        try:  # a container CPU quota (cgroup v2 cpu.max) caps the useful thread count below the visible cores
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if quota != "max":
                threads = max(1, min(threads, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
        os.environ.setdefault("OMP_NUM_THREADS", str(threads))
        os.environ.setdefault("OMP_PROC_BIND", "spread")
        orc.use_omp(True)
        st = orc.BicgstabState(Ao, b, x0)
        st.step(r0hat, 2)                                     # warm up the thread team / first touch
        t0 = time.perf_counter()
        st.step(r0hat, 1)
        t1 = max(time.perf_counter() - t0, 1e-6)
        osteps = max(5, min(400, int(0.4 * seconds / t1)))
        t0 = time.perf_counter()
        st.step(r0hat, osteps)
        odt = time.perf_counter() - t0
        out["omp"] = {"value": osteps / odt, "unit": "iters/s", "cores": threads, "kind": "port",
                      "sample": f"{osteps} bicgstabStep iterations, OpenMP build of the same port (liboracle_omp.so), {threads} threads"}
    except Exception as e:  # the OpenMP leg is informative only
        out["omp"] = {"error": repr(e)}
    finally:
        orc.use_omp(False)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=os.environ.get("SLA_BENCH_WORKLOAD", "laplace3d_10m"))
    ap.add_argument("--mode", default="step", choices=["step", "linsolve0", "gmres"])
    ap.add_argument("--method", default="bicgstab", choices=["bicgstab", "cgs"], help="step mode: which solver step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    dist = None
    # SLA_BENCH_FORCE_DIST=1 drives the multi-rank code path (torch first, gloo control plane, RCCL
    # communicator, forced collectives) on a single GPU: the only way to rehearse it on a 1-GPU box
    use_dist = world > 1 or os.environ.get("SLA_BENCH_FORCE_DIST") == "1"
    if use_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ["SLA_FORCE_COLLECTIVES"] = "1"
    if use_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        # control plane (unique-id broadcast, barriers, max-over-ranks) on gloo; the data plane is the
        # library's own RCCL communicator over xGMI
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    import sla_amd as sla
    from sla_amd import _lib
    from sla_amd.partition import row_block

    if use_dist:
        import torch
        uid = [sla.Context.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx = sla.Context(local_rank, rank, world, uid[0])
    else:
        ctx = sla.Context(0)
    sla.set_default_context(ctx)

    # ---- build this rank's slab on the host, lower it once to the device CSR ---------------------------
    if world == 1:
        desc, (dims, (rp, ci, va)) = workload(args.workload, 0, None)
        n = dims[0]
        rb, re_ = 0, n
    else:
        desc0, (dims, _) = workload(args.workload, 0, 1)
        n = dims[0]
        rb, re_ = row_block(n, rank, world)
        desc, (dims, (rp, ci, va)) = workload(args.workload, rb, re_)
    nnz_local = int(rp[-1])
    A = sla.fromCSRRows(dims, rb, rp, ci, va, ctx)
    b_local = np.add.reduceat(va, rp[:-1]) if nnz_local else np.zeros(0)   # b = A . 1  (x* = 1), x0 = 0
    nnz = nnz_local
    if use_dist:
        import torch
        t = torch.tensor([nnz_local], dtype=torch.int64)
        dist.all_reduce(t)
        nnz = int(t.item())
    bvec = sla.DeviceVector(ctx, n, b_local, local=True)
    x0 = sla.DeviceVector(ctx, n)

    def sync_all():
        ctx.sync()
        if use_dist:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    extra = {}
    if args.mode == "step":
        st = sla.bicgsInit(A, bvec, x0) if args.method == "bicgstab" else sla.cgsInit(A, bvec, x0)
        st.step(args.warmup)
        sync_all()
        ctx.prof_start(_lib.KERNEL_SPMV_DOT, args.steps)
        t0 = time.perf_counter()
        st.step(args.steps)
        sync_all()
        dt = time.perf_counter() - t0
        launches, mean_ms, min_ms = ctx.prof_stop()
        step_bytes = 24 * nnz + 160 * n
        mode_desc = f"{'bicgstabStep' if args.method == 'bicgstab' else 'cgsStep'} (2 SpMV, no true-residual SpMV)"
    elif args.mode == "gmres":
        # config 5: GMRES(30) on the device Arnoldi; a "step" = one Arnoldi step (SpMV + 2-pass classical GS)
        lib = _lib.lib()
        import ctypes as C
        out = sla.DeviceVector(ctx, n)
        info = _lib.SolveInfo()
        restart = 30
        o = _lib.SolveOpts(restart, 0.0, 0.0, 16, 1)
        _lib.check(lib.sla_gmres(A.h, bvec.h, x0.h, restart, C.byref(o), out.h, C.byref(info)))
        sync_all()
        o = _lib.SolveOpts(args.steps, 0.0, 0.0, 16, 1)                       # tol 0: exactly K Arnoldi steps
        ctx.prof_start(_lib.KERNEL_SPMV, args.steps)
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
        lib = _lib.lib()
        import ctypes as C
        out = sla.DeviceVector(ctx, n)
        info = _lib.SolveInfo()
        o = _lib.SolveOpts(args.warmup, 0.0, 0.0, 16, 1)
        _lib.check(lib.sla_linsolve0(4, A.h, bvec.h, x0.h, C.byref(o), out.h, C.byref(info)))
        sync_all()
        o = _lib.SolveOpts(args.steps, 0.0, 0.0, max(args.steps, 1), 1)      # tol 0: run exactly K iterations
        # (the wave-sliced form streams ~2 B of matrix per row: nothing to fuse, the library keeps the sweeps apart)
        dual = (not use_dist and os.environ.get("SLA_DUAL_SPMV", "1") != "0" and os.environ.get("SLA_SPMV_ALGO", "stream") == "stream"
                and "algo=wdia" not in A.kernel_info() and "ldspanels" not in A.kernel_info())
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

    if use_dist:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- plain SpMV bandwidth (rank-local rows; includes the all-gather when sharded) -------------------
    # Over ROTATING vector pairs: with one pair the 2 x 80 MB stay in the 256 MB memory-side cache (MALL) between
    # launches and the figure flatters the kernel; inside a solver the vectors never stay there.  The single-pair
    # (cache-resident) time is reported next to it.
    lib = _lib.lib()
    n_local = re_ - rb
    pairs = max(1, min(6, int(2.0e9 // max(1, 16 * n_local))))      # <= 2 GB of extra vectors
    xs = [sla.DeviceVector(ctx, n, np.full(n_local, 1.0 + 0.125 * i), local=True) for i in range(pairs)]
    ys = [sla.DeviceVector(ctx, n) for _ in range(pairs)]
    xv, yv = xs[0], ys[0]
    for i in range(max(5, pairs)):
        _lib.check(lib.sla_spmv(A.h, xs[i % pairs].h, ys[i % pairs].h))
    sync_all()
    reps = 60
    ctx.prof_start(_lib.KERNEL_SPMV, reps)
    for i in range(reps):
        _lib.check(lib.sla_spmv(A.h, xs[i % pairs].h, ys[i % pairs].h))
    sp_launch, sp_mean_ms, sp_min_ms = ctx.prof_stop()
    ctx.prof_start(_lib.KERNEL_SPMV, reps)
    for _ in range(reps):
        _lib.check(lib.sla_spmv(A.h, xv.h, yv.h))
    _, sp_cached_ms, _ = ctx.prof_stop()
    spmv_bytes_local = 12 * nnz_local + 20 * n_local

    # ---- measured streaming ceiling of this GPU (triad-like y = a x + b y: 24 B per element), same rotation ----
    triad_reps = 36
    for i in range(pairs):
        _lib.check(lib.sla_axpby(1.0, xs[i].h, 0.5, ys[i].h))
    ctx.sync()
    t0 = time.perf_counter()
    for i in range(triad_reps):
        _lib.check(lib.sla_axpby(1.0, xs[i % pairs].h, 0.5, ys[i % pairs].h))
    ctx.sync()
    triad_gbps = 24.0 * n_local * triad_reps / (time.perf_counter() - t0) / 1e9

    if rank == 0:
        kinfo = A.kernel_info()
        k1_bytes = 12 * nnz_local + 28 * n_local      # K1 = SpMV (12 nnz + 20 n) + r0hat read for the fused dot (8 n)
        if args.mode == "gmres":
            k1_bytes = 12 * nnz_local + 20 * n_local  # the Arnoldi SpMV is the plain kernel
        if args.mode == "linsolve0" and extra.get("dual_spmv"):
            k1_bytes = 12 * nnz_local + 44 * n_local  # K1 + gather of x (8 n) + b (8 n) for the fused residual
        achieved = k1_bytes / (mean_ms * 1e-3) / 1e9 if launches else 0.0
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
                       "spmv_kernel": A.kernel_info()},
        }
        if use_dist:   # what the sharded step exchanges
            kinfo_x = A.kernel_info()
            ghost = ("x_exchange=window" in kinfo_x and os.environ.get("SLA_BICG_GHOST", "1") != "0" and args.mode in ("step", "linsolve0"))
            rec["config"]["exchange"] = (
                ("halo (window) send/recv received in place" if "x_exchange=window" in kinfo_x else "all-gather of x")
                + (f"; ghost-row {args.method}: {3 if args.method == 'bicgstab' else 2} grouped exchanges per step"
                   + (" + the residual sweep's own exchange and sum" if args.mode == "linsolve0" else "") if ghost
                   else ("; plain flow" if args.mode != "gmres" else "; Arnoldi: one exchange per SpMV, per-column sums all-gathered")))
        rec.update({
            "step_gbps": step_bytes / (dt / args.steps) / 1e9 / 1.0,
            "step_frac_of_hbm_peak": step_bytes / (dt / args.steps) / 1e9 / (HBM_PEAK_GBS * world),
            "spmv_gbps": spmv_bytes_local * world / (sp_mean_ms * 1e-3) / 1e9 if sp_launch else None,
            "spmv_ms": sp_mean_ms,                     # rotating over `spmv_vector_pairs` x / y pairs (HBM-resident)
            "spmv_ms_cache_resident": sp_cached_ms,    # one pair re-used: x and y stay in the memory-side cache
            "spmv_vector_pairs": pairs,
            "hbm_measured_ceiling_gbps": triad_gbps,   # axpby triad over the same rotating vectors, same run
            "step_frac_of_measured_ceiling": step_bytes / (dt / args.steps) / 1e9 / (triad_gbps * world) if triad_gbps else None,
            "roofline": {"bound": "hbm",
                         "kernel": f"{kinfo.split()[0]} {'K1D (K1 + true residual, one sweep)' if extra.get('dual_spmv') else ('plain SpMV' if args.mode == 'gmres' else 'K1: Ap = A p fused with Ap . r0hat')}",
                         "bytes_definition": "algorithmic CSR bytes of SURVEY 8(d): f64 values + i32 column indices + i32 row "
                                             "pointers + the vectors.  The value-indexed kernels (wdia / vdict: constant-coefficient "
                                             "stencils) and the dictionary-index kernel stream a losslessly compressed matrix, so "
                                             "`traffic` (PMC) is BELOW this figure and `frac` can exceed 1; `hbm_achieved` = traffic "
                                             "/ launch time is the bandwidth the DRAM side actually delivered",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_ceiling": achieved / triad_gbps if triad_gbps else None,
                         "traffic": None, "hbm_achieved": None, "hbm_frac": None,
                         "bytes_per_launch": k1_bytes, "avg_launch_ms": mean_ms, "min_launch_ms": min_ms,
                         "launches_timed": launches},
        })
        # HBM traffic of the dominant kernel: measured with PMC counters in separate rocprofv3 passes
        # (bench.py cannot collect counters on itself) and committed under profiles/
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                tr = json.load(f).get(f"{args.workload}/{args.mode}/n{world}")
            if tr:
                rec["roofline"]["traffic"] = tr["traffic_bytes"]
                rec["roofline"]["traffic_source"] = tr["source"]
                if tr.get("kernel_algo", "") in kinfo and launches:   # the counters were taken on this kernel
                    rec["roofline"]["hbm_achieved"] = tr["traffic_bytes"] / (mean_ms * 1e-3) / 1e9
                    rec["roofline"]["hbm_frac"] = rec["roofline"]["hbm_achieved"] / HBM_PEAK_GBS
        except OSError:
            pass
        rec.update(extra)
        if not args.no_cpu_baseline and world == 1:
            rec["cpu_baseline"] = cpu_baseline(dims, rp, ci, va, b_local, args.cpu_seconds)
        print(json.dumps(rec), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
