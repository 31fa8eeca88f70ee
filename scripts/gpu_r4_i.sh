#!/bin/bash
# round 4, GPU call i: gradient all-reduce placement with a one-rank RCCL group (overlap vs end; eager buckets), the scoring kernel with its
# points prefetched, and the bench profile again on the final code
cd /root/repo
root=$PWD
export TMPDIR=/tmp
out=$root/gpurun_out/r4i
mkdir -p $out
timeout 900 python -m pytest tests/test_fused_loss_gpu.py tests/test_ground_pin.py tests/test_ddp_gpu.py tests/test_bench_gpu.py tests/test_trainer_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -8 > $out/pytest.log; tail -3 $out/pytest.log
row() { label=$1; shift; env "$@" MASTER_ADDR=127.0.0.1 MASTER_PORT=29579 timeout 300 python bench.py --no_cpu_baseline --mode ${MODE:-graph} > $out/$label.json 2> $out/$label.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/$label.json').read().strip().splitlines()[-1]); print('%-28s'%'$label', d['value'], 'img/s', d['ms_per_step'], 'ms', d['config']['mode'], 'rccl ranks', d['config']['rccl_ranks'])
except Exception as e: print('$label failed', e)
PY
}
row no_group DD_X=0
row rccl1_end DD_BENCH_FORCE_DIST=1 DD_SEG_REDUCE=end
row rccl1_overlap DD_BENCH_FORCE_DIST=1 DD_SEG_REDUCE=overlap
MODE=eager row rccl1_eager_bucket48 DD_BENCH_FORCE_DIST=1 DD_DDP_BUCKET_MB=48
MODE=eager row rccl1_eager_bucket1024 DD_BENCH_FORCE_DIST=1 DD_DDP_BUCKET_MB=1024
MODE=eager row no_group_eager DD_X=0
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/losstrace -- python $root/scripts/loss_path_workload.py fine_tune 12 30 > $out/loss_workload.log 2>&1 )
python scripts/loss_kernels.py "$(find $out/losstrace -name '*kernel_trace.csv' | head -1)" 20 > $out/r04_loss_path_kernels.txt 2>&1; rm -rf $out/losstrace; cat $out/r04_loss_path_kernels.txt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -- python $root/bench.py --no_cpu_baseline > $out/bench.log 2>&1 )
grep "^{" $out/bench.log | tail -1 > $out/r04_bench_line_profiled.json
st=$(find $out/bench -name '*kernel_stats.csv' | head -1); tr=$(find $out/bench -name '*kernel_trace.csv' | head -1)
cp "$st" $out/r04_bench_default_rocprofv3_kernel_stats.csv
python scripts/steady_state_stats.py "$tr" 10 $out/r04_bench_fine_tune_steady_kernel_stats.csv 22
python scripts/categorise_stats.py $out/r04_bench_fine_tune_steady_kernel_stats.csv > $out/r04_bench_categories.txt 2>&1
python scripts/tile_populations.py "$tr" 20 > $out/r04_tile_kernel_populations.txt 2>&1
rm -rf $out/bench
cat $out/r04_tile_kernel_populations.txt; head -25 $out/r04_bench_categories.txt
timeout 400 python bench.py > $out/r04_bench_line_default.json 2> $out/bench_default.err; cut -c1-200 $out/r04_bench_line_default.json
