L=$GRAFT_REPO_ROOT/sparse-linear-algebra_amd/lib
run() { # name lib env...
  name=$1; lib=$2; shift 2
  env "$@" SLA_HIP_LIB=$lib python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-blocks 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-14s it/s %8.1f  ' % (sys.argv[1], d['value']) + '  '.join('%s %.1f' % (k, v['ms'] * 1e3) for k, v in d['kernels'].items()) + '  ' + d['config']['spmv_kernel'][:40])" "$name"
}
for rep in 1 2; do
  run late0 $L/libsla_hip_late0.so SLA_WD_MARCH=0
  run late1 $L/libsla_hip.so SLA_WD_MARCH=0
  run march_occ3 $L/libsla_hip.so SLA_WD_MARCH=1
  run march_occ4 $L/libsla_hip.so SLA_WD_MARCH=1 SLA_WD_MARCH_OCC=4
done
