#!/bin/bash
# round 4, GPU call h: the evidence of the round -- rocprofv3 kernel stats of the default bench command (+ steady-state categories, tile-kernel
# populations), FETCH/WRITE PMC passes (photo + regularisers), SQ counters of the tile kernel, one-rank RCCL bench row, train.py's own log
cd /root/repo
root=$PWD
export TMPDIR=/tmp
out=$root/gpurun_out/prof_r04
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -40 > $out/pytest.log
tail -5 $out/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke > $out/smoke.log; cat $out/smoke.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/losstrace -- python $root/scripts/loss_path_workload.py fine_tune 12 30 > $out/loss_workload.log 2>&1 )
python scripts/loss_kernels.py "$(find $out/losstrace -name '*kernel_trace.csv' | head -1)" 20 > $out/r04_loss_path_kernels.txt 2>&1; rm -rf $out/losstrace; cat $out/r04_loss_path_kernels.txt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -- python $root/bench.py --no_cpu_baseline > $out/bench.log 2>&1 )
grep "^{" $out/bench.log | tail -1 > $out/r04_bench_line_profiled.json
st=$(find $out/bench -name '*kernel_stats.csv' | head -1)
tr=$(find $out/bench -name '*kernel_trace.csv' | head -1)
cp "$st" $out/r04_bench_default_rocprofv3_kernel_stats.csv
python scripts/steady_state_stats.py "$tr" 10 $out/r04_bench_fine_tune_steady_kernel_stats.csv 22 > /dev/null 2>&1
python scripts/categorise_stats.py $out/r04_bench_fine_tune_steady_kernel_stats.csv > $out/r04_bench_categories.txt 2>&1
python scripts/tile_populations.py "$tr" 20 > $out/r04_tile_kernel_populations.txt 2>&1
rm -rf $out/bench
cat $out/r04_tile_kernel_populations.txt; head -12 $out/r04_bench_categories.txt; grep photo_tile $out/r04_bench_default_rocprofv3_kernel_stats.csv | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python $root/scripts/pmc_workload.py > $out/pmc_$c.log 2>&1 )
  f=$(find $out/pmc_$c -name '*counter_collection.csv' | head -1)
  python scripts/pmc_summary.py "$f" $out/r04_pmc_$c.csv > /dev/null
  rm -rf $out/pmc_$c
done
python scripts/make_traffic_json.py $out/r04_pmc_FETCH_SIZE.csv $out/r04_pmc_WRITE_SIZE.csv r04 $out/photo_traffic.json
bash scripts/pmc_reg.sh r04 > $out/pmc_reg.log 2>&1; cp gpurun_out/pmc_r04/r04_* $out/ 2>/dev/null; tail -12 $out/pmc_reg.log
bash scripts/pmc_photo.sh r04 > $out/pmc_photo.log 2>&1; cp gpurun_out/pmc_r04/SQ_photo_tile_kernel.csv $out/r04_pmc_SQ_photo_tile_kernel.csv 2>/dev/null; grep "2, false" $out/r04_pmc_SQ_photo_tile_kernel.csv | cut -c60-140 | head -30
rm -rf gpurun_out/pmc_r04/a gpurun_out/pmc_r04/b gpurun_out/pmc_r04/c
timeout 400 python bench.py > $out/r04_bench_line_default.json 2> $out/bench_default.err
DD_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 timeout 400 python bench.py --no_cpu_baseline > $out/r04_bench_line_rccl_one_rank.json 2> $out/bench_rccl.err
( cd dynamo-depth_amd && timeout 400 python train.py -d kitti --synthetic -b 12 --weights_init scratch --epoch_schedules 0 0 0 1 --epoch-size 400 --ramp_red 100000 --log_frequency 20 --num_workers 8 --log_dir /tmp/dd_trainlog --no_train_vis > $out/r04_train_py_log.txt 2>&1 )
grep "examples/s" $out/r04_train_py_log.txt | tail -5
for f in default rccl_one_rank; do cut -c1-330 $out/r04_bench_line_$f.json; done
# the recorded workloads without Find (the new default) -- against profiles/r04_sweep_bench.txt (Find on)
{ for row in "kitti_bf16 --amp bf16" "c5_fp16 --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp fp16" "c4_fp32 --dataset waymo --batch 8" "kitti_md2 --depth_model monodepthv2"; do
  set -- $row; label=$1; shift
  timeout 300 python bench.py --no_cpu_baseline --mode graph --steps 10 --warmup 3 "$@" > $out/nofind_$label.json 2> $out/nofind_$label.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/nofind_$label.json').read().strip().splitlines()[-1]); print('$label', 'find', d['config']['miopen_find'], d['value'], 'img/s', d['ms_per_step'], 'ms')
except Exception as e: print('$label failed', e)
PY
  grep "warm-up done" $out/nofind_$label.err
done; } > $out/r04_recorded_workloads_without_find.txt 2>&1
cat $out/r04_recorded_workloads_without_find.txt
