#!/bin/bash
# same-box A/B of library builds: tools/ab_lib.sh <name> ... runs the default bench (no side blocks) with sparse-linear-algebra_amd/lib/libsla_hip_<name>.so
# ("-" = the product library), interleaved twice; prints it/s and every kernel's HIP-event time
L=$GRAFT_REPO_ROOT/sparse-linear-algebra_amd/lib
for rep in 1 2; do for n in "$@"; do
  lib=$L/libsla_hip_$n.so; [ "$n" = "-" ] && lib=$L/libsla_hip.so
  SLA_HIP_LIB=$lib python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-blocks ${AB_ARGS:-} 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-8s it/s %8.1f  ' % (sys.argv[1], d['value']) + '  '.join('%s %.1f' % (k, v['ms'] * 1e3) for k, v in d['kernels'].items()))" "$n"
done; done
