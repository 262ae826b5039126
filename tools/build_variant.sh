#!/bin/bash
# Kernel A/B experiments: build libsla_hip_<name>.so with extra -D flags next to the product library.
#   tools/build_variant.sh nt_rec -DSLA_WD_NT_REC=1
# then on the GPU box:  SLA_HIP_LIB=sparse-linear-algebra_amd/lib/libsla_hip_nt_rec.so python bench.py ...
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/sparse-linear-algebra_amd/csrc; out=$root/sparse-linear-algebra_amd/lib
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -I$root/include -I$src --offload-arch=gfx950 -munsafe-fp-atomics "$@" -c $src/sla_kernels.hip -o $out/sla_kernels_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -o $out/libsla_hip_$name.so $out/sla_kernels_$name.o \
  $out/sla_coo_sort.o $out/sla_api.o $out/sla_solvers.o $out/sla_csr_build.o $out/sla_dist.o $out/sla_mmio.o -ldl -Wl,-rpath,/opt/rocm/lib
echo built $out/libsla_hip_$name.so
