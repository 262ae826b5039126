cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
S=gpurun_out/gm; rm -rf $S; mkdir -p $S
python bench.py --mode gmres --workload banded_2m --steps 120 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 > $S/r06_bench_gmres_banded_2m.json
SLA_ARN_ORTH=0 python bench.py --mode gmres --workload banded_2m --steps 120 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 > $S/r06_bench_gmres_banded_2m_launchflow.json
rocprofv3 --kernel-trace --stats --output-format csv -d $S/ksg -o ks -- python bench.py --mode gmres --workload banded_2m --steps 120 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 > $S/r06_bench_traced_run_gmres_banded_2m.json
cp "$(find $S/ksg -name '*kernel_stats.csv' | head -1)" $S/r06_bench_kernel_stats_gmres_banded_2m.csv; rm -rf $S/ksg
cut -c1-120 $S/r06_bench_gmres_banded_2m.json $S/r06_bench_gmres_banded_2m_launchflow.json
