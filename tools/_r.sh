for w in laplace3d_10m laplace3d_1m poisson2d_1m banded_2m; do
  python bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value'],1), 'it/s', round(d['ms_per_step']*1e3,2), 'us/step')" $w
done
python bench.py --mode gmres --workload banded_2m --steps 120 --warmup 0 --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('gmres', round(d['value'],1))"
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -3
