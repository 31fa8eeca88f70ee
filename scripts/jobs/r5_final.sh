set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
bash scripts/gpu_job.sh r5z bench
bash scripts/profile_round.sh r05 > gpurun_out/r5z_profile_round.log 2>&1; tail -30 gpurun_out/r5z_profile_round.log
bash scripts/gpu_job.sh r5z loss
bash scripts/gpu_job.sh r5z loss disp_init
DD_PACKED=0 bash scripts/gpu_job.sh r5z loss
DD_PACKED=0 bash scripts/gpu_job.sh r5z loss disp_init
bash scripts/gpu_job.sh r5z photo
bash scripts/gpu_job.sh r5z tests tests/test_trainer_gpu.py tests/test_photo_gpu.py -k "packed or pack_rgb"
