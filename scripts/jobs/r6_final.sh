set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
bash scripts/gpu_job.sh r6z bench
bash scripts/gpu_job.sh r6z bench --no_cpu_baseline
bash scripts/profile_round.sh r06 > gpurun_out/r6z_profile_round.log 2>&1; tail -40 gpurun_out/r6z_profile_round.log | cut -c1-200
bash scripts/gpu_job.sh r6z smoke
bash scripts/gpu_job.sh r6z loss
bash scripts/gpu_job.sh r6z loss disp_init
bash scripts/gpu_job.sh r6z photo
bash scripts/sweep_bench.sh r06 > gpurun_out/r6z_sweep.txt 2>&1; cat gpurun_out/r6z_sweep.txt | cut -c1-200
bash scripts/gpu_job.sh r6z suite --durations=40
