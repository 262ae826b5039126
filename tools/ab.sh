#!/bin/bash
# usage: tools/ab.sh "<env assignments>" ... ; runs bench.py once per configuration, prints it/s, K1 ms, plain SpMV ms
for cfg in "$@"; do
  env $cfg python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-70s it/s %8.1f  K1 %.4f ms  spmv %.4f ms' % (sys.argv[1], d['value'], d['roofline']['avg_launch_ms'], d['spmv_ms']))" "$cfg"
done
