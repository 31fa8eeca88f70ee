set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
bash scripts/gpu_job.sh r6a bench
for at in start loss pose; do
  DD_SEG_TIMING=1 DD_SEG_SIDE_AT=$at bash scripts/gpu_job.sh r6a bench --no_cpu_baseline --mode graph
done
bash scripts/gpu_job.sh r6a py scripts/time_half_convs.py
bash scripts/gpu_job.sh r6a photo
bash scripts/gpu_job.sh r6a loss
bash scripts/gpu_job.sh r6a suite --durations=70
