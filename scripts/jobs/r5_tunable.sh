set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5t; mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=$PWD/$out/tunableop.csv PYTORCH_TUNABLEOP_VERBOSE=1
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=20 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
t0=$(date +%s)
PYTORCH_TUNABLEOP_TUNING=1 timeout 900 python bench.py --mode eager --steps 2 --warmup 2 --no_cpu_baseline > $out/tune.json 2> $out/tune.err
echo "tuning run rc $? in $(( $(date +%s) - t0 )) s"; ls -la $out; wc -l $out/tunableop*.csv 2>/dev/null
export PYTORCH_TUNABLEOP_VERBOSE=0
PYTORCH_TUNABLEOP_TUNING=0 timeout 500 python bench.py --no_cpu_baseline > $out/bench_tuned.json 2> $out/bench_tuned.err; echo "rc $?"; cut -c1-330 $out/bench_tuned.json
PYTORCH_TUNABLEOP_ENABLED=0 timeout 500 python bench.py --no_cpu_baseline > $out/bench_plain.json 2> $out/bench_plain.err; echo "rc $?"; cut -c1-330 $out/bench_plain.json
PYTORCH_TUNABLEOP_TUNING=0 timeout 500 python bench.py --no_cpu_baseline > $out/bench_tuned2.json 2> $out/bench_tuned2.err; echo "rc $?"; cut -c1-330 $out/bench_tuned2.json
