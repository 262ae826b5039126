run() { name=$1; pol=$2
  SLA_VEC_POLICY=$pol python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extra-blocks 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-22s it/s %8.1f  ' % (sys.argv[1], d['value']) + '  '.join('%s %.1f' % (k, v['ms'] * 1e3) for k, v in d['kernels'].items()))" "$name"
}
for rep in 1 2 3; do
run "base" 1019
run "s! nt" $((1019 ^ 4))
run "s! nt + K45.s plain" $((1019 ^ 4 ^ 8))
done
