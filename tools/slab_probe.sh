cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks4 -o ks -- python bench.py --workload laplace3d_1m --no-cpu-baseline --steps 400 --warmup 20 > gpurun_out/ks4.log 2>&1
tail -1 gpurun_out/ks4.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
f=$(find gpurun_out/ks4 -name "*kernel_stats.csv" | head -1)
python tools/ks_print.py "$f"
SLA_BENCH_FORCE_DIST=1 python bench.py --workload laplace3d_1m --no-cpu-baseline --steps 400 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced collectives (1 rank):', d['value'], d['ms_per_step'])"
