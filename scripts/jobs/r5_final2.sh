set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
bash scripts/gpu_job.sh r5f bench
bash scripts/gpu_job.sh r5f bench
bash scripts/profile_round.sh r05 > gpurun_out/r5f_profile_round.log 2>&1; tail -28 gpurun_out/r5f_profile_round.log | cut -c1-200
bash scripts/gpu_job.sh r5f smoke
bash scripts/gpu_job.sh r5f suite
