set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
bash scripts/gpu_job.sh r6d tests tests/test_conv_half_gpu.py
bash scripts/gpu_job.sh r6d py scripts/time_half_convs.py
for v in 1 0 1 0; do
  DD_HALF_MFMA_CONV=$v bash scripts/gpu_job.sh r6d bench --no_cpu_baseline --mode graph --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp fp16
done
DD_HALF_MFMA_CONV=1 bash scripts/gpu_job.sh r6d bench --no_cpu_baseline --mode graph --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp bf16
DD_HALF_MFMA_CONV=0 bash scripts/gpu_job.sh r6d bench --no_cpu_baseline --mode graph --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp bf16
