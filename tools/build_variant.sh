#!/bin/bash
# tools/build_variant.sh <git-rev> <name> [extra hipcc flags]: builds the engine of another revision as
# mvskit_amd/lib/variant_<name>.so, for A/B timing on one GPU box (MVS_ENGINE_LIB=... python bench.py ...)
set -e
rev=$1; name=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" mvskit_amd/csrc include | tar -x -C "$tmp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-implicit-const-int-float-conversion "$@" \
  -I "$tmp/include" -I "$tmp/mvskit_amd/csrc" -x hip "$tmp/mvskit_amd/csrc/mvs_kernels.hip" "$tmp/mvskit_amd/csrc/mvs_engine.cpp" -o "$root/mvskit_amd/lib/variant_$name.so"
rm -rf "$tmp"
echo "$root/mvskit_amd/lib/variant_$name.so"
