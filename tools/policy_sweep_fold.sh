names=(K2.r K2.Ap K2.s! K45.r K45.As K45.Ap K45.p K45.x K45.x! K45.r! K45.p!)
run() { name=$1; pol=$2; shift 2
  env "$@" SLA_VEC_POLICY=$pol python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-blocks 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-26s it/s %8.1f  ' % (sys.argv[1], d['value']) + '  '.join('%s %.1f' % (k, v['ms'] * 1e3) for k, v in d['kernels'].items()))" "$name"
}
base=11263
run "base $base" $base A=1
for b in 3 4 5 6 7 8 9 10; do run "flip ${names[$b]}" $((base ^ (1 << b))) A=1; done
run "base $base" $base A=1
run "As store nt (wd_nt_store=1)" $base SLA_WD_NT_STORE=1
run "As store nt + flip K45.As" $((base ^ 16)) SLA_WD_NT_STORE=1
run "base $base" $base A=1
