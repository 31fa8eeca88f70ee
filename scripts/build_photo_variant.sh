#!/bin/bash
# Builds dynamo-depth_amd/csrc/variants/<name>.so = the library with dd_photo.hip compiled under extra flags (A/B runs of the
# photometric kernel on the GPU box: DYNAMO_HIP_LIB=<that .so> python scripts/time_photo.py).
# usage: build_photo_variant.sh <name> [hipcc flags...]
set -e
name=$1; shift
cd "$(dirname "$0")/../dynamo-depth_amd/csrc"
mkdir -p variants
make -s >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c dd_photo.hip -o variants/$name.o
objs=$(ls dd_*.o | grep -v dd_photo.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs variants/$name.o -o variants/$name.so
rm variants/$name.o
echo built variants/$name.so
