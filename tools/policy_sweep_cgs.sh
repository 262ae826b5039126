#!/bin/bash
# the same for CGS's stores (bits 11..14 of vec_policy: C2's q, uq; C4's u, p)
run() { name=$1; pol=$2
  SLA_VEC_POLICY=$pol python bench.py --method cgs --steps 200 --warmup 20 --no-cpu-baseline --no-extra-blocks 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-22s it/s %8.1f  ' % (sys.argv[1], d['value']) + '  '.join('%s %.1f' % (k, v['ms'] * 1e3) for k, v in d['kernels'].items()))" "$name"
}
base=${1:-1023}
for rep in 1 2; do
run "base" $base
run "C2.q!" $((base ^ (1 << 11)))
run "C2.uq!" $((base ^ (1 << 12)))
run "C4.u!" $((base ^ (1 << 13)))
run "C4.p!" $((base ^ (1 << 14)))
run "q! u!" $((base ^ (1 << 11) ^ (1 << 13)))
run "q! uq! u!" $((base ^ (1 << 11) ^ (1 << 12) ^ (1 << 13)))
done
