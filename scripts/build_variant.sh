#!/bin/bash
# Builds dynamo-depth_amd/csrc/variants/<name>.so = the library with ONE source compiled under extra flags (A/B runs on the GPU box:
# DYNAMO_HIP_LIB=<that .so> python scripts/<timing script>).   usage: build_variant.sh <source without .hip> <name> [hipcc flags...]
set -e
src=$1; name=$2; shift 2
cd "$(dirname "$0")/../dynamo-depth_amd/csrc"
mkdir -p variants
make -s >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c $src.hip -o variants/$name.o
objs=$(ls dd_*.o | grep -v "^$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs variants/$name.o -o variants/$name.so
rm variants/$name.o
echo built variants/$name.so
