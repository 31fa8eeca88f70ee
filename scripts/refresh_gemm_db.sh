#!/bin/bash
# Re-records dynamo-depth_amd/gemm_db/tunableop_gfx950.csv on a GPU box: eager bench steps of the workloads below with PyTorch's TunableOp
# tuning every GEMM shape it meets (a few seconds per shape), all into ONE file.   bash scripts/refresh_gemm_db.sh <tag>
# then: cp gpurun_out/<tag>/tunableop0.csv dynamo-depth_amd/gemm_db/tunableop_gfx950.csv
set -u
tag=${1:-gemmdb}
cd "$(dirname "$0")/.." || exit 1
out=$PWD/gpurun_out/$tag; mkdir -p $out
[ -f dynamo-depth_amd/gemm_db/tunableop_gfx950.csv ] && cp dynamo-depth_amd/gemm_db/tunableop_gfx950.csv $out/tunableop0.csv     # grow the shipped table
export TMPDIR=/tmp PYTHONUNBUFFERED=1
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$out/tunableop.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=20 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
run() {  # label, bench args...
  label=$1; shift
  t0=$(date +%s)
  timeout ${DD_JOB_TIMEOUT:-600} python bench.py --no_cpu_baseline --mode eager --steps 2 --warmup 2 "$@" > $out/$label.json 2> $out/$label.err
  echo "$label rc=$? $(( $(date +%s) - t0 )) s, $(wc -l < $out/tunableop0.csv) lines"
}
run kitti_fp32
run kitti_disp_init --phase disp_init
run waymo_fp32 --dataset waymo --batch 8
run kitti_bf16 --amp bf16
run kitti_fp16 --amp fp16
