cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06l; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_onchip.py tests/test_gpu_parity.py -m gpu -q > $O/pytest_onchip.log 2>&1; echo "rc=$?" >> $O/pytest_onchip.log
python bench.py --workload poisson2d_1m --mode linsolve0 --steps 200 --warmup 20 --no-cpu-baseline --no-extra-blocks 2>$O/ls_on.err | tail -1 > $O/bench_linsolve0_onchip_poisson2d_1m.json
SLA_ONCHIP=0 python bench.py --workload poisson2d_1m --mode linsolve0 --steps 200 --warmup 20 --no-cpu-baseline --no-extra-blocks 2>/dev/null | tail -1 > $O/bench_linsolve0_launchflow_poisson2d_1m.json
tail -15 $O/pytest_onchip.log; for f in $O/bench_linsolve0*.json; do echo $f; cut -c1-260 $f; done; tail -3 $O/ls_on.err
