cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gpu_check; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python bench.py --no-cpu-baseline --no-extra-blocks 2>/dev/null | tail -1 > $O/bench_traced_run.json
cp "$(find $O/ks -name '*kernel_stats.csv' | head -1)" $O/bench_kernel_stats.csv; rm -rf $O/ks
tail -3 $O/pytest.log; cut -c1-400 $O/bench_default.json
