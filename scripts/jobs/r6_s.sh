set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
bash scripts/gpu_job.sh r6s tests tests/test_pw_gemm_gpu.py -k "recomputed" -s
grep -E "own|passed|failed" gpurun_out/r6s/pytest_0.log | head -30
for rep in 1 2; do
for v in 1 0; do
  DD_MLP_RECOMPUTE=$v bash scripts/gpu_job.sh r6s bench --no_cpu_baseline --mode graph
done
done
bash scripts/gpu_job.sh r6s tests tests/test_trainer_gpu.py -k "litemono"
