#!/bin/bash
# round 4, second GPU call: MFMA/VALU microbenchmark, per-module fp16 NaN probe, stream placement A/B of the replayed step
mkdir -p gpurun_out/r4b
cd /root/repo
bash scripts/microbench/run_mfma_valu.sh > gpurun_out/r4b/mfma_valu.txt 2>&1
timeout 120 python scripts/probe_stream_queues.py 8 > gpurun_out/r4b/stream_queues.txt 2>&1
for c in 1 0; do
  DD_PROBE_LEVEL=1 DD_AMP_CACHE=$c timeout 600 python scripts/probe_amp_nan.py --steps 10 --runs 5 > gpurun_out/r4b/probe_l1_cache$c.log 2>&1
done
DD_SEG_TIMING=1 timeout 300 python bench.py --mode graph --no_cpu_baseline > gpurun_out/r4b/bench_pick.json 2> gpurun_out/r4b/bench_pick.err
DD_SEG_TIMING=1 DD_STREAM_PICK=0 DD_SEG_DEC_STREAM=own timeout 300 python bench.py --mode graph --no_cpu_baseline > gpurun_out/r4b/bench_r3streams.json 2> gpurun_out/r4b/bench_r3streams.err
DD_SEG_TIMING=1 DD_STREAM_PICK=0 timeout 300 python bench.py --mode graph --no_cpu_baseline > gpurun_out/r4b/bench_nopick_shared.json 2> gpurun_out/r4b/bench_nopick_shared.err
cat gpurun_out/r4b/mfma_valu.txt gpurun_out/r4b/stream_queues.txt; grep "^run\|first non-finite module\|largest" gpurun_out/r4b/probe_l1_cache*.log | cut -c1-400
for f in pick r3streams nopick_shared; do python -c "
import json,sys
d=json.load(open('gpurun_out/r4b/bench_$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline'].get('loss_path_replayed_us'))"; grep segment gpurun_out/r4b/bench_$f.err; done
