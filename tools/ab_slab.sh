#!/bin/bash
# one rank's slab of the 216^3 problem at 8 / 4 / 2 GPUs on the sharded flow (1-rank RCCL communicator): which stencil form?
run() { name=$1; wl=$2; shift 2
  env "$@" SLA_BENCH_FORCE_DIST=1 SLA_X_EXCHANGE=window MASTER_ADDR=127.0.0.1 MASTER_PORT=29581 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-extra-blocks 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-10s %-14s it/s %8.1f  ' % (sys.argv[2], sys.argv[1], d['value']) + '  '.join('%s %.1f' % (k, v['ms'] * 1e3) for k, v in d['kernels'].items()) + '  ' + d['config']['spmv_kernel'][:44])" "$name" "$wl"
}
for wl in laplace3d_1m laplace3d_2m5 laplace3d_5m; do for rep in 1 2; do
  run gather $wl SLA_WD_LDS=0
  run ldswin $wl SLA_WD_LDS=2 SLA_WD_MARCH=0
  run march $wl SLA_WD_LDS=2 SLA_WD_MARCH=2
done; done
