set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
bash scripts/gpu_job.sh r6m smoke
bash scripts/gpu_job.sh r6m bench
bash scripts/gpu_job.sh r6m suite --durations=12
