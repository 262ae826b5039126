#!/bin/bash
# Regenerates the measurement evidence of profiles/ on the GPU box (run through gpurun); everything lands in
# gpurun_out/refresh/, tools/install_profiles.py then copies it into profiles/ and rebuilds pmc_traffic.json.
#   gpurun --timeout 3000 -- 'bash tools/refresh_profiles.sh r03'
set -u
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
S=gpurun_out/refresh; rm -rf $S; mkdir -p $S
B="python bench.py --steps 200 --warmup 20"
run() { name=$1; shift; timeout 900 "$@" 2>/dev/null | tail -1 > $S/${tag}_bench_$name.json; echo "$name: $(cut -c1-120 $S/${tag}_bench_$name.json)"; }
run default python bench.py
run linsolve0 $B --mode linsolve0 --no-cpu-baseline
run cgs $B --method cgs --no-cpu-baseline --no-extra-blocks
run poisson2d_1m $B --workload poisson2d_1m --no-cpu-baseline
run banded_2m $B --workload banded_2m --no-cpu-baseline
run gmres_banded_2m python bench.py --mode gmres --workload banded_2m --steps 120 --warmup 0 --no-cpu-baseline
SLA_ARN_ORTH=0 run gmres_banded_2m_launchflow python bench.py --mode gmres --workload banded_2m --steps 120 --warmup 0 --no-cpu-baseline
run random_spd_1m $B --workload random_spd_1m --no-cpu-baseline
run random_spd_10m_bicgstab python bench.py --workload random_spd_10m --steps 40 --warmup 5 --no-cpu-baseline
run random_spd_10m_cgs python bench.py --workload random_spd_10m --method cgs --steps 40 --warmup 5 --no-cpu-baseline
run dense_rows_200k python bench.py --workload dense_rows_200k --steps 40 --warmup 5 --no-cpu-baseline
# (end of round 6: the exact fold is the default; the relaxed order of round 5 beside it as the opt-in it is now)
SLA_TILE_RELAXED=1 run random_spd_1m_relaxed $B --workload random_spd_1m --no-cpu-baseline
SLA_TILE_RELAXED=1 run random_spd_10m_bicgstab_relaxed python bench.py --workload random_spd_10m --steps 40 --warmup 5 --no-cpu-baseline
SLA_TILE_RELAXED=1 run random_spd_10m_cgs_relaxed python bench.py --workload random_spd_10m --method cgs --steps 40 --warmup 5 --no-cpu-baseline
# the same default command under the kernel tracer (per-kernel durations must agree with bench.py's HIP events)
rocprofv3 --kernel-trace --stats --output-format csv -d $S/ks -o ks -- python bench.py --no-cpu-baseline --no-extra-blocks 2>/dev/null | tail -1 > $S/${tag}_bench_traced_run.json
cp "$(find $S/ks -name '*kernel_stats.csv' | head -1)" $S/${tag}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $S/ks3a -o ks -- python bench.py --workload random_spd_10m --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $S/${tag}_bench_traced_run_random_spd_10m.json
cp "$(find $S/ks3a -name '*kernel_stats.csv' | head -1)" $S/${tag}_bench_kernel_stats_random_spd_10m.csv
# round 4: the same for config 5 (GMRES(30) Arnoldi steps on the 2 M-row banded matrix)
rocprofv3 --kernel-trace --stats --output-format csv -d $S/ksg -o ks -- python bench.py --mode gmres --workload banded_2m --steps 120 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 > $S/${tag}_bench_traced_run_gmres_banded_2m.json
cp "$(find $S/ksg -name '*kernel_stats.csv' | head -1)" $S/${tag}_bench_kernel_stats_gmres_banded_2m.csv
# PMC passes (separate --pmc runs, kernel-trace only): FETCH/WRITE size, L2 hit/miss, EA requests
pmc() { out=$1; shift
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    d=$S/pmc_tmp; rm -rf $d; mkdir -p $d
    timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o p -- "$@" > /dev/null 2>&1
    python - "$d" <<'PY' >> $out
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # (a kernel name can cover launches of different sizes -- e.g. spmv_tile_kernel<1> on a second, smaller matrix of the same run: the
    # figure per launch is the mean over the launches within 10 % of the largest, i.e. the full-size ones; n = how many those were)
    for k, cs in agg.items():
        out = {}
        for c, v in cs.items():
            big = [x for x in v if x >= 0.9 * max(v)] if max(v) > 0 else v
            out[c] = (len(big), sum(big) / len(big))
        print(k, out)
PY
  done; }
pmc $S/${tag}_bench_pmc_counters.txt python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-blocks
pmc $S/${tag}_bench_pmc_counters_random_spd_10m.txt python bench.py --workload random_spd_10m --steps 8 --warmup 2 --no-cpu-baseline
pmc $S/${tag}_bench_pmc_counters_poisson2d_1m.txt python bench.py --workload poisson2d_1m --steps 20 --warmup 3 --no-cpu-baseline
pmc $S/${tag}_bench_pmc_counters_dense_rows_200k.txt python bench.py --workload dense_rows_200k --steps 10 --warmup 2 --no-cpu-baseline
pmc $S/${tag}_bench_pmc_counters_gmres_banded_2m.txt python bench.py --mode gmres --workload banded_2m --steps 60 --warmup 0 --no-cpu-baseline
# round 4: the plain CSR kernels (general_csr: spmv_wave_kernel) on the headline matrix -- HBM bytes per launch against 12 nnz + 28 n
SLA_WDIA=0 SLA_VDICT=0 SLA_DIAG=0 pmc $S/${tag}_bench_pmc_counters_general_csr.txt python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-blocks
# round 3: one rank's 8-GPU slab (108^3 rows) through a 1-rank RCCL communicator: ghost-row flow with the fused K4+K5 sweep on own + ghost rows
# (2 grouped exchanges per step) against the reference's split (3), window and all-gather exchange
for f in 1 0; do for xe in window allgather; do
  SLA_BICG_FUSE45=$f SLA_BENCH_FORCE_DIST=1 SLA_X_EXCHANGE=$xe timeout 600 python bench.py --workload laplace3d_1m --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep '^{' | tail -1 > $S/${tag}_bench_slab_1rank_rccl_${xe}_fuse$f.json
done; done
# the sharded flow rehearsed with two loopback ranks on this one GPU (both workloads in one line; not a scaling number)
SLA_BENCH_LOOPBACK=1 timeout 900 python bench.py --gpus 2 --steps 40 --warmup 5 2>/dev/null | grep '^{' | tail -1 > $S/${tag}_bench_loopback_2ranks.json
# the CSR-stream skeleton: load widths, structure variants, the y store (tools/stream_width_probe.cpp)
[ -x tools/stream_width_probe ] && timeout 120 tools/stream_width_probe 216 > $S/${tag}_stream_width_probe.txt 2>&1
# L1 <-> L2 request counters of the tile kernel (config 3a) and of the CSR-stream kernel (general_csr): the fabric-side statement of their bounds
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  d=$S/pmc_tmp; rm -rf $d; mkdir -p $d
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o p -- python bench.py --workload random_spd_10m --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python - "$d" tile_kernel <<'PY' >> $S/${tag}_pmc_l1_l2_tile_kernel_10m.txt
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if sys.argv[2] in k: print(k, {c: (len(v), sum(v) / len(v)) for c, v in cs.items()})
PY
  d=$S/pmc_tmp; rm -rf $d; mkdir -p $d
  SLA_WDIA=0 SLA_VDICT=0 SLA_DIAG=0 timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o p -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-blocks > /dev/null 2>&1
  python - "$d" spmv_wave <<'PY' >> $S/${tag}_pmc_l1_l2_stream_kernel.txt
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if sys.argv[2] in k: print(k, {c: (len(v), sum(v) / len(v)) for c, v in cs.items()})
PY
done
# the bare 7-pt access pattern: gathers vs LDS-staged windows, visiting orders (tools/stencil_probe.cpp)
[ -x tools/stencil_probe ] && { timeout 120 tools/stencil_probe 216; timeout 120 tools/stencil_probe 256; } > $S/${tag}_stencil_probe.txt 2>&1
# ... and the plane march: tile sizes, planes in flight (PROBE_MARCH), the fused K2 + K3 variants are the last lines of the file above
[ -x tools/stencil_probe ] && { PROBE_MARCH=1 timeout 120 tools/stencil_probe 216; PROBE_MARCH=1 timeout 120 tools/stencil_probe 256; } > $S/${tag}_stencil_march_probe.txt 2>&1
# the K4+K5 sweep's shape on rotating vector sets (nothing cached): loads / stores plain or non-temporal, in place, pipelined
[ -x tools/sweep_probe ] && timeout 120 tools/sweep_probe > $S/${tag}_sweep_probe.txt 2>&1
# same-box A/B of the stencil forms on the headline (gather / LDS windows / plane march) and on slab-sized problems
bash tools/ab_march2.sh > $S/${tag}_ab_march_knobs.txt 2>&1
bash tools/ab_slab.sh > $S/${tag}_ab_slab_forms.txt 2>&1
# L1 <-> L2 request counters of the plane-march kernel (default workload)
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  d=$S/pmc_tmp; rm -rf $d; mkdir -p $d
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o p -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-blocks > /dev/null 2>&1
  python - "$d" wdia_march <<'PY' >> $S/${tag}_pmc_l1_l2_march_kernel.txt
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if sys.argv[2] in k: print(k, {c: (len(v), sum(v) / len(v)) for c, v in cs.items()})
PY
done
# round 4: parity in the hard regime (traces on e05r0000 / the beam system), the split cost of the overlapped all-gather, plain-CSR kernel A/B
timeout 600 python tools/hard_regime.py > $S/${tag}_hard_regime.txt 2>&1
timeout 900 python tools/ag_split_bench.py > $S/${tag}_ag_split_cost.txt 2>&1
timeout 900 python tools/wave_ab.py 30 2 > $S/${tag}_ab_wave_kernel.txt 2>&1
# round 5: kernel trace of the plain-CSR block (spmv_wave_kernel on the 216^3 Laplacian: the metric's "CSR SpMV achieved HBM GB/s")
SLA_WDIA=0 SLA_VDICT=0 SLA_DIAG=0 rocprofv3 --kernel-trace --stats --output-format csv -d $S/ksc -o ks -- python bench.py --no-cpu-baseline --no-extra-blocks 2>/dev/null | tail -1 > $S/${tag}_bench_traced_run_general_csr.json
cp "$(find $S/ksc -name '*kernel_stats.csv' | head -1)" $S/${tag}_bench_kernel_stats_general_csr.csv
# round 5: first contact of a multi-rank job rehearsed on one GPU -- the three blocks of the N > 1 line behind the pre-flight (1-rank RCCL
# communicator and two loopback ranks), and the fallback ladder under an injected grouped-send/recv failure and an injected hang
SLA_BENCH_FORCE_DIST=1 timeout 900 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>$S/${tag}_bench_1rank_rccl_stderr.txt | grep '^{' | tail -1 > $S/${tag}_bench_1rank_rccl.json
SLA_BENCH_LOOPBACK=1 SLA_FAULT_INJECT=p2p timeout 900 python bench.py --gpus 2 --workload laplace3d_small --steps 20 --warmup 5 2>$S/${tag}_bench_loopback_fault_p2p_stderr.txt | grep '^{' | tail -1 > $S/${tag}_bench_loopback_fault_p2p.json
SLA_BENCH_LOOPBACK=1 SLA_FAULT_INJECT=p2p_hang SLA_BENCH_PREFLIGHT_S=5 timeout 900 python bench.py --gpus 2 --workload laplace3d_small --steps 20 --warmup 5 2>$S/${tag}_bench_loopback_fault_hang_stderr.txt | grep '^{' | tail -1 > $S/${tag}_bench_loopback_fault_hang.json
# round 5: the form tournament over the matrix zoo (the pick against every forced form) and the lowering phases of configs 3a / 4
for w in laplace3d_10m laplace3d_1m banded_2m poisson2d_1m e05_tiled e05_tiled_10m varcoef7 random_spd_1m rand100 rand200 rand500 powerlaw; do timeout 900 python tools/form_tournament.py $w 40 2>/dev/null | grep -v "^#"; done > $S/${tag}_form_tournament.txt 2>&1
{ timeout 600 python tools/lower_phases.py random_spd_10m 3; timeout 600 python tools/lower_phases.py laplace3d_10m 3; } > $S/${tag}_lowering_phases.txt 2>&1
timeout 900 python tools/tile_bench.py 10000000 "DEFAULT=1" "SLA_TILE_RELAXED=1" "SLA_TILE_RELAXED=1 SLA_TILE_DEPTH=1" "SLA_TILE_SLACK=2" "SLA_TILE_SLACK=4" "SLA_TILE_SLACK=0" "SLA_TILE_SHIFT=16" > $S/${tag}_ab_tile_knobs.txt 2>&1
# round 6: the on-chip solver step (one persistent launch) against the launch flow on the sizes it takes -- config 2 and config 4's per-rank slab at N = 8
for w in poisson2d_1m laplace3d_slab8 laplace3d_1m; do
  run onchip_$w $B --workload $w --no-cpu-baseline --no-extra-blocks
  SLA_ONCHIP=0 run launchflow_$w $B --workload $w --no-cpu-baseline --no-extra-blocks
done
run onchip_poisson2d_1m_20steps python bench.py --workload poisson2d_1m --steps 20 --warmup 5 --no-cpu-baseline --no-extra-blocks
# ... cgsStep on chip, and linSolve0 (step + true residual + test in the launch) for both methods, each against the launch flow
run onchip_cgs_poisson2d_1m $B --workload poisson2d_1m --method cgs --no-cpu-baseline --no-extra-blocks
SLA_ONCHIP=0 run launchflow_cgs_poisson2d_1m $B --workload poisson2d_1m --method cgs --no-cpu-baseline --no-extra-blocks
run linsolve0_onchip_poisson2d_1m $B --workload poisson2d_1m --mode linsolve0 --no-cpu-baseline --no-extra-blocks
SLA_ONCHIP=0 run linsolve0_launchflow_poisson2d_1m $B --workload poisson2d_1m --mode linsolve0 --no-cpu-baseline --no-extra-blocks
rocprofv3 --kernel-trace --stats --output-format csv -d $S/kso -o ks -- python bench.py --workload poisson2d_1m --steps 200 --warmup 20 --no-cpu-baseline --no-extra-blocks 2>/dev/null | tail -1 > $S/${tag}_bench_traced_run_poisson2d_1m.json
cp "$(find $S/kso -name '*kernel_stats.csv' | head -1)" $S/${tag}_bench_kernel_stats_poisson2d_1m.csv
# round 6: the exact tile forms (rows owned by wavefronts = the default / wavefront-private slices) beside the relaxed order (opt-in) on config 3a; the asymmetric pre-flight failure
timeout 900 python tools/tile_bench.py 10000000 "DEFAULT=1" "SLA_TILE_RELAXED=1" "SLA_TILE_ROWOWN=0" > $S/${tag}_tile_exact_forms.txt 2>&1
timeout 1200 python tools/tile_default_ab.py 2>/dev/null | grep -v "^\[" > $S/${tag}_tile_default_ab.txt
SLA_BENCH_LOOPBACK=1 SLA_FAULT_INJECT=p2p_data_rank1 timeout 900 python bench.py --gpus 2 --workload laplace3d_small --steps 20 --warmup 5 2>$S/${tag}_bench_loopback_fault_rank1_stderr.txt | grep '^{' | tail -1 > $S/${tag}_bench_loopback_fault_rank1.json
SLA_BENCH_LOOPBACK=1 timeout 900 python bench.py --gpus 2 --steps 40 --warmup 5 --no-cpu-baseline 2>$S/${tag}_bench_loopback_2ranks_stderr.txt | grep '^{' | tail -1 > $S/${tag}_bench_loopback_2ranks.json
# round 6: one real run past 2^31 stored entries (lowering, (#>) on all rows and two bicgstabSteps against the oracle)
timeout 1500 python tools/big_nnz.py --full > $S/${tag}_big_nnz.json 2> $S/${tag}_big_nnz_log.txt
ls -la $S | head -120
