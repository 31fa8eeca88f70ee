#!/bin/bash
# round 4, fourth GPU call: whole GPU suite on the new regulariser kernels / fp16 replay / RCCL fix; cache=0 NaN statistics; loss-path
# kernel trace; stand-alone photometric timings at the other configs' shapes; config-5 fp16 replayed bench
mkdir -p gpurun_out/r4d
cd /root/repo
root=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -80 > gpurun_out/r4d/pytest.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/r4d/losstrace -- python $root/scripts/loss_path_workload.py fine_tune 12 30 > $root/gpurun_out/r4d/loss_workload.log 2>&1 )
tr=$(find gpurun_out/r4d/losstrace -name '*kernel_trace.csv' | head -1)
python scripts/loss_kernels.py "$tr" 20 > gpurun_out/r4d/r04_loss_path_kernels.txt 2>&1
rm -rf gpurun_out/r4d/losstrace
{
DD_SMOOTH=1 DD_PHASES=disp_init,fine_tune timeout 200 python scripts/time_photo.py
DD_SMOOTH=1 DD_PHASES=fine_tune DD_B=8 DD_H=320 DD_W=480 timeout 200 python scripts/time_photo.py
DD_SMOOTH=1 DD_PHASES=fine_tune DD_B=16 DD_H=288 DD_W=512 DD_SCALES=0,1,2,3 timeout 200 python scripts/time_photo.py
} > gpurun_out/r4d/r04_time_photo.txt 2>&1
DD_AMP_CACHE=0 timeout 700 python scripts/probe_amp_nan.py --steps 8 --runs 14 > gpurun_out/r4d/probe_cache0.log 2>&1
for i in 1 2; do timeout 400 python bench.py --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp fp16 --no_cpu_baseline > gpurun_out/r4d/bench_c5_fp16_$i.json 2> gpurun_out/r4d/bench_c5_fp16_$i.err; done
tail -12 gpurun_out/r4d/pytest.log; cat gpurun_out/r4d/r04_loss_path_kernels.txt gpurun_out/r4d/r04_time_photo.txt; grep -E "^run|post-mortem" gpurun_out/r4d/probe_cache0.log | cut -c1-300; for i in 1 2; do cut -c1-900 gpurun_out/r4d/bench_c5_fp16_$i.json; tail -3 gpurun_out/r4d/bench_c5_fp16_$i.err; done
