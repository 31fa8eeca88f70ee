set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
DD_BENCH_DEBUG_LEGS=1 bash scripts/gpu_job.sh r6h bench --no_cpu_baseline --mode graph --depth_model monodepthv2
grep "leg:" gpurun_out/r6h/bench_0.err | cut -c1-600
bash scripts/gpu_job.sh r6h suite --durations=15
