#!/bin/bash
# Kernel A/B experiments on ONE translation unit: libsla_hip_<name>.so = the product's objects with <file> rebuilt under extra -D flags.
#   tools/build_variant_one.sh u12 sla_spmv_ctiles.hip -DSLA_CT_U=12      (run `make -C sparse-linear-algebra_amd/csrc` first)
# then on the GPU box:  SLA_HIP_LIB=sparse-linear-algebra_amd/lib/libsla_hip_u12.so python tools/tile_bench.py ...
set -e
name=$1; file=$2; shift; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/sparse-linear-algebra_amd/csrc; out=$root/sparse-linear-algebra_amd/lib
flags="-O3 -std=c++17 -fPIC -Wno-unused-function -I$root/include -I$src --offload-arch=gfx950 -munsafe-fp-atomics"
obj=$out/variant_${name}_${file%.*}.o
x=""; case $file in *.cpp) x="-x hip";; esac
/opt/rocm/bin/hipcc $flags "$@" $x -c $src/$file -o $obj
others=$(ls $out/sla_*.o | grep -v "/${file%.*}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -o $out/libsla_hip_$name.so $obj $others -ldl -Wl,-rpath,/opt/rocm/lib
rm -f $obj
echo built $out/libsla_hip_$name.so
