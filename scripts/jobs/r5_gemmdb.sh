set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
bash scripts/refresh_gemm_db.sh r5g
cp gpurun_out/r5g/tunableop0.csv dynamo-depth_amd/gemm_db/tunableop_gfx950.csv
wc -l dynamo-depth_amd/gemm_db/tunableop_gfx950.csv
for v in 1 0 1 0; do
  DD_GEMM_TUNED=$v bash scripts/gpu_job.sh r5g bench --no_cpu_baseline | cut -c1-120
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5g/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['config'].get('library_gemms'))
    except Exception as e: print(f, e)
P
