set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
bash scripts/gpu_job.sh r6g tests tests/test_conv_half_gpu.py
bash scripts/gpu_job.sh r6g py scripts/time_half_convs.py
for v in 0 1 0 1; do
  DD_STOCK_HALF_WGRAD=$v bash scripts/gpu_job.sh r6g bench --no_cpu_baseline --mode graph --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp fp16
done
