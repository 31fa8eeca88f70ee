set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
bash scripts/gpu_job.sh r6r suite --durations=8
bash scripts/gpu_job.sh r6r smoke
bash scripts/gpu_job.sh r6r bench
