#!/bin/bash
# Kernel A/B experiments: build libsla_hip_<name>.so with extra -D flags next to the product library.
#   tools/build_variant.sh occ6 -DSLA_TILE_ROWS=768 -DSLA_TILE_OCC=6
# then on the GPU box:  SLA_HIP_LIB=sparse-linear-algebra_amd/lib/libsla_hip_occ6.so python bench.py ...
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/sparse-linear-algebra_amd/csrc; out=$root/sparse-linear-algebra_amd/lib; obj=$out/variant_$name
mkdir -p $obj
flags="-O3 -std=c++17 -fPIC -Wno-unused-function -I$root/include -I$src --offload-arch=gfx950 -munsafe-fp-atomics"
pids=()
for f in $(cd $src && ls *.hip); do
  /opt/rocm/bin/hipcc $flags "$@" -c $src/$f -o $obj/${f%.*}.o & pids+=($!)
done
for f in $(cd $src && ls *.cpp); do
  /opt/rocm/bin/hipcc $flags "$@" -x hip -c $src/$f -o $obj/${f%.*}.o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -o $out/libsla_hip_$name.so $obj/*.o -ldl -Wl,-rpath,/opt/rocm/lib
rm -rf $obj
echo built $out/libsla_hip_$name.so
