#!/bin/bash
# build + run scripts/microbench/mfma_valu.hip on the GPU box
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w mfma_valu.hip -o /tmp/mfma_valu
/tmp/mfma_valu
