#!/bin/bash
# What does each task of dd::fused_post_kernel cost inside the launch?  Variant build (-DDD_REG_DEBUG_SKIP), DD_POST_SKIP bit mask:
# 1 footprint sums + smoothness, 2 tile-record fold, 4 candidate scoring, 8 disparity sums; rocprofv3 kernel trace of the loss alone
# per mask (results are wrong with a task removed -- timing only).  GPU box:  bash scripts/post_task_costs.sh <outdir>
cd "$(dirname "$0")/.."
root=$PWD; out=${1:-$root/gpurun_out/post_tasks}; case $out in /*) ;; *) out=$root/$out ;; esac; mkdir -p $out
cd dynamo-depth_amd/csrc && mkdir -p variants && make -s >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DDD_REG_DEBUG_SKIP -c dd_reg.hip -o variants/regskip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls dd_*.o | grep -v dd_reg.o) variants/regskip.o -o variants/regskip.so && rm variants/regskip.o
cd $root
export TMPDIR=/tmp
for mask in 0 1 2 4 8 14 11 7; do
  ( cd /tmp && DD_POST_SKIP=$mask DYNAMO_HIP_LIB=$root/dynamo-depth_amd/csrc/variants/regskip.so timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/t_$mask -- python $root/scripts/loss_path_workload.py fine_tune 12 20 > $out/log_$mask.txt 2>&1 )
  tr=$(find $out/t_$mask -name '*kernel_trace.csv' | head -1)
  echo "DD_POST_SKIP=$mask: $(python scripts/loss_kernels.py "$tr" | grep fused_post_kernel)"
  rm -rf $out/t_$mask
done
