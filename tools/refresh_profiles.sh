#!/bin/bash
# Regenerates the measurement evidence of profiles/ on the GPU box (run through gpurun); everything lands in
# gpurun_out/refresh/, tools/install_profiles.py then copies it into profiles/ and rebuilds pmc_traffic.json.
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh r01'
set -u
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
S=gpurun_out/refresh; rm -rf $S; mkdir -p $S
B="python bench.py --steps 200 --warmup 20"
run() { name=$1; shift; timeout 600 "$@" 2>/dev/null | tail -1 > $S/${tag}_bench_$name.json; echo "$name: $(cut -c1-120 $S/${tag}_bench_$name.json)"; }
run default $B
run linsolve0 $B --mode linsolve0 --no-cpu-baseline
run cgs $B --method cgs --no-cpu-baseline
run poisson2d_1m $B --workload poisson2d_1m --no-cpu-baseline
run banded_2m $B --workload banded_2m --no-cpu-baseline
run gmres_banded_2m python bench.py --mode gmres --workload banded_2m --steps 120 --warmup 0 --no-cpu-baseline
run random_spd_1m $B --workload random_spd_1m --no-cpu-baseline
run random_spd_10m_bicgstab python bench.py --workload random_spd_10m --steps 40 --warmup 5 --no-cpu-baseline
run random_spd_10m_cgs python bench.py --workload random_spd_10m --method cgs --steps 40 --warmup 5 --no-cpu-baseline
run dense_rows_200k python bench.py --workload dense_rows_200k --steps 40 --warmup 5 --no-cpu-baseline
# the same default command under the kernel tracer (per-kernel durations must agree with bench.py's HIP events)
rocprofv3 --kernel-trace --stats --output-format csv -d $S/ks -o ks -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > $S/${tag}_bench_traced_run.json
cp "$(find $S/ks -name '*kernel_stats.csv' | head -1)" $S/${tag}_bench_kernel_stats.csv
# PMC passes (separate --pmc runs, kernel-trace only): FETCH/WRITE size, L2 hit/miss, EA requests, SQ busy/wait
for w in laplace3d_10m poisson2d_1m random_spd_1m; do
  bash tools/pmc_kbench.sh ${tag}_$w python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $S/${tag}_bench_pmc_counters_$w.txt 2>&1
done
bash tools/pmc_kbench.sh ${tag}_dense_rows_200k python bench.py --workload dense_rows_200k --steps 10 --warmup 2 --no-cpu-baseline > $S/${tag}_bench_pmc_counters_dense_rows_200k.txt 2>&1
timeout 300 tools/kbench > $S/${tag}_kbench_spmv_variants.txt 2>&1
bash tools/pmc_kbench.sh ${tag}_kbench tools/kbench x pmc 2>&1 | grep -E "axpby_kernel|dot_kernel|fill_kernel" > $S/${tag}_kbench_pmc_calibration.txt
# same-box ablation of the knobs behind the default line (BiCGSTAB it/s, K1 ms, rotating SpMV ms)
tools/ab.sh DEFAULT=1 SLA_VEC_NT=0 SLA_WD_TILE=0 SLA_XCD_REMAP=0 SLA_WDIA=0 "SLA_WDIA=0 SLA_VDICT=0" "SLA_WDIA=0 SLA_VDICT=0 SLA_VEC_NT=0" \
  "SLA_WDIA=0 SLA_VDICT=0 SLA_DIAG=0 SLA_VEC_NT=0" "SLA_WDIA=0 SLA_VDICT=0 SLA_DIAG=0 SLA_XWIN=0 SLA_VEC_NT=0" DEFAULT=2 > $S/${tag}_ablation.txt 2>&1
timeout 120 tools/xcc_probe > $S/${tag}_xcc_probe.txt 2>&1
timeout 120 tools/l1_probe > $S/${tag}_l1_probe.txt 2>&1
ls -la $S | head -40
