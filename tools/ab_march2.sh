#!/bin/bash
# same-box A/B of the stencil SpMV forms on the headline (216^3 Laplacian, one BiCGSTAB step): the gather kernel, the LDS-window
# kernel, the plane march (default) and its knobs; interleaved twice
run() { name=$1; shift
  env "$@" python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-blocks 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-14s it/s %8.1f  ' % (sys.argv[1], d['value']) + '  '.join('%s %.1f' % (k, v['ms'] * 1e3) for k, v in d['kernels'].items()) + '  ' + d['config']['spmv_kernel'][:40])" "$name"
}
for rep in 1 2; do
  run gather SLA_WD_LDS=0
  run ldswin SLA_WD_MARCH=0
  run march X=1
  run march_occ3 SLA_WD_MARCH_OCC=3
  run march_nt1 SLA_WD_NT_STORE=1
  run march_noxcd SLA_XCD_REMAP=0
done
