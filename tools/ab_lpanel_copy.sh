run() { name=$1; shift
  env "$@" python bench.py --workload dense_rows_200k --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-10s it/s %8.1f  ' % (sys.argv[1], d['value']) + '  '.join('%s %.1f' % (k, v['ms'] * 1e3) for k, v in d['kernels'].items()) + '  ' + d['config']['spmv_kernel'][-40:])" "$name"
}
for rep in 1 2; do run rowmajor SLA_LP_COPY=0; run copy SLA_LP_COPY=1; done
