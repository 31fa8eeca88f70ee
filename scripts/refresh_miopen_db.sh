#!/bin/bash
# Re-records the shipped MIOpen find-db (dynamo-depth_amd/miopen_db/) on a GPU box: runs the bench workloads whose convolution
# problems the shipped records lack with MIOpen Find on -- the user db (a persistent per-rank copy of the shipped files,
# miopen_env._private_copy) collects what Find learns -- and copies the grown files to gpurun_out/<tag>/miopen_db/ for
# `cp gpurun_out/<tag>/miopen_db/* dynamo-depth_amd/miopen_db/`.      bash scripts/refresh_miopen_db.sh <tag> [per-run timeout s]
set -u
tag=${1:-r4db}
lim=${2:-900}
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/$tag/miopen_db
run() {  # label, bench args...
  label=$1; shift
  t0=$(date +%s)
  timeout $lim python bench.py --no_cpu_baseline --mode graph --steps 10 --warmup 3 "$@" > gpurun_out/$tag/$label.json 2> gpurun_out/$tag/$label.err
  echo "$label rc=$? $(( $(date +%s) - t0 )) s: $(cut -c1-160 gpurun_out/$tag/$label.json)"
}
run kitti_bf16 --amp bf16
run kitti_fp16 --amp fp16
run c5_fp16 --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp fp16
run c5_bf16 --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp bf16
run c5_fp32 --dataset nuscenes --depth_model monodepthv2 --batch 16
run c4_fp32 --dataset waymo --batch 8
run kitti_md2 --depth_model monodepthv2
db=$(ls -d /tmp/dd_miopen_db_*/*_rank0 | head -1)
cp $db/* gpurun_out/$tag/miopen_db/
ls -la gpurun_out/$tag/miopen_db/; wc -l gpurun_out/$tag/miopen_db/*
# second pass with the grown db: how long does the warm-up take now?
run kitti_bf16_again --amp bf16
run c5_fp16_again --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp fp16
grep "warm-up done\|trainer built" gpurun_out/$tag/kitti_bf16_again.err gpurun_out/$tag/c5_fp16_again.err
