#!/bin/bash
# tools/build_here.sh <name> [extra hipcc flags]: builds the working tree's engine as mvskit_amd/lib/variant_<name>.so
# (A/B timing on one GPU box: MVS_ENGINE_LIB=... python bench.py ..., see tools/ab.sh)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-implicit-const-int-float-conversion "$@" \
  -I "$root/include" -I "$root/mvskit_amd/csrc" -x hip "$root/mvskit_amd/csrc/mvs_kernels.hip" "$root/mvskit_amd/csrc/mvs_engine.cpp" -o "$root/mvskit_amd/lib/variant_$name.so" -ldl
echo "$root/mvskit_amd/lib/variant_$name.so"
