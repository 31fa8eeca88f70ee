#!/bin/bash
# on the GPU box: bash scripts/microbench/run_pk_rate.sh
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o /tmp/pk_rate && /tmp/pk_rate
