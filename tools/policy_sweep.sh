#!/bin/bash
# which streams of K2 / the K4+K5 sweep should go past the caches?  One bit of ctx option vec_policy flipped at a time on the headline.
names=(K2.r K2.Ap K2.s! K45.s K45.As K45.Ap K45.p K45.x K45.x! K45.r! K45.p!)
run() { name=$1; pol=$2
  SLA_VEC_POLICY=$pol python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-blocks 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-22s it/s %8.1f  ' % (sys.argv[1], d['value']) + '  '.join('%s %.1f' % (k, v['ms'] * 1e3) for k, v in d['kernels'].items()))" "$name"
}
base=${1:-1019}
run "base $base" $base
for b in 0 1 2 3 4 5 6 7 8 9 10; do run "flip ${names[$b]}" $((base ^ (1 << b))); done
run "base $base" $base
