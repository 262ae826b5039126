cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06k; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_onchip.py -m gpu -q > $O/pytest_onchip.log 2>&1; echo "rc=$?" >> $O/pytest_onchip.log
for w in poisson2d_1m laplace3d_1m; do
  python bench.py --workload $w --method cgs --steps 200 --warmup 20 --no-cpu-baseline --no-extra-blocks 2>/dev/null | tail -1 > $O/bench_onchip_cgs_$w.json
  SLA_ONCHIP=0 python bench.py --workload $w --method cgs --steps 200 --warmup 20 --no-cpu-baseline --no-extra-blocks 2>/dev/null | tail -1 > $O/bench_launchflow_cgs_$w.json
done
tail -12 $O/pytest_onchip.log; for f in $O/bench_*cgs*.json; do echo $f; cut -c1-200 $f; done
