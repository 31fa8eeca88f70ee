set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
bash scripts/gpu_job.sh r5m tests tests/test_conv_mfma_gpu.py
for v in 1 0 1 0; do
  DD_FLAT_MFMA_CONV=$v bash scripts/gpu_job.sh r5m bench --no_cpu_baseline | cut -c1-110
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5m/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['config'].get('final_loss'), d['config'].get('conv3x3_stride1','')[-60:])
    except Exception as e: print(f, e)
P
bash scripts/gpu_job.sh r5m tests tests/test_trainer_gpu.py -k "hooks_match_stock or matches_reference"
