#!/bin/bash
# Builds the library once more with extra compiler flags into fluidsolver_amd/lib/variants/<name>.so (for A/B timing
# through FSA_LIB_PATH); only the listed sources are recompiled, the other objects are taken from the normal build.
#   tools/build_variant.sh <name> "<flags>" <source.hip> [...]
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../fluidsolver_amd/csrc"
mkdir -p ../lib/variants/$name
objs=""
for o in ../lib/*.o; do objs="$objs $o"; done
for src in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $flags -c $src -o ../lib/variants/$name/${src%.hip}.o
  objs=$(echo $objs | sed "s#../lib/${src%.hip}.o#../lib/variants/$name/${src%.hip}.o#")
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o ../lib/variants/$name.so
rm -rf ../lib/variants/$name
echo built ../lib/variants/$name.so
