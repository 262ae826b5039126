#!/bin/bash
# builds the development harness tools/kbench against the in-tree libsla_hip.so
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -Iinclude -Isparse-linear-algebra_amd/csrc tools/kbench.cpp \
  -Lsparse-linear-algebra_amd/lib -lsla_hip -Wl,-rpath,'$ORIGIN/../sparse-linear-algebra_amd/lib' -o tools/kbench
