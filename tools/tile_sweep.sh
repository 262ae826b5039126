#!/bin/bash
# usage: tools/tile_sweep.sh "<env assignments>" ... ; BiCGSTAB step on random_spd_10m per configuration
for cfg in "$@"; do
  env $cfg python bench.py --workload ${WL:-random_spd_10m} --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-40s it/s %7.1f  K1 %.3f ms  spmv %.3f ms | %s' % (sys.argv[1], d['value'], d['roofline']['avg_launch_ms'], d['spmv_ms'], d['config']['spmv_kernel'][-70:]))" "$cfg"
done
