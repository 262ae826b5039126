#!/bin/bash
# same-box A/B of library builds on the plain CSR-stream kernel (216^3 Laplacian, wdia=0 vdict=0 diag=0):
#   tools/csr_ab_lib.sh <variant> ...   ("-" = the product library); interleaved twice
L=$GRAFT_REPO_ROOT/sparse-linear-algebra_amd/lib
for rep in 1 2; do for n in "$@"; do
  lib=$L/libsla_hip_$n.so; [ "$n" = "-" ] && lib=$L/libsla_hip.so
  SLA_HIP_LIB=$lib python - "$n" 2>/dev/null <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench
desc, (dims, (rp, ci, va)) = bench.workload("laplace3d_10m")
r = bench.side_block(desc, dims, rp, ci, va, {"wdia": 0, "vdict": 0, "diag": 0}, 40, 5)
print(f"{sys.argv[1]:8s} {r['value']:8.1f} it/s  K1 {r['k1_ms'] * 1e3:7.1f} us ({r['k1_frac']:.3f})  " + "  ".join(f"{k} {v['ms'] * 1e3:.1f}" for k, v in r["kernels"].items()) + "  " + r["spmv_kernel"].split()[0], flush=True)
PY
done; done
