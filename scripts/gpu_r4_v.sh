#!/bin/bash
# round 4, GPU call v: rocprofv3 kernel stats of the default bench command on the final code (one-launch Adam), and a long
# fine_tune run of train.py on it (600 batches: finite? examples/s?)
cd /root/repo; out=/root/repo/gpurun_out/r4v; mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python /root/repo/bench.py --no_cpu_baseline > $out/bench_prof.log 2>&1 ) < /dev/null
grep "^{" $out/bench_prof.log | tail -1 > $out/r04_bench_line_under_rocprofv3.json
st=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); tr=$(find /tmp/prof -name '*kernel_trace.csv' | head -1)
if [ -n "$st" ]; then
  cp "$st" $out/r04_bench_default_rocprofv3_kernel_stats.csv
  python scripts/steady_state_stats.py "$tr" 10 $out/r04_bench_fine_tune_steady_kernel_stats.csv 22
  python scripts/categorise_stats.py $out/r04_bench_fine_tune_steady_kernel_stats.csv > $out/r04_bench_categories.txt 2>&1; head -14 $out/r04_bench_categories.txt
  python scripts/tile_populations.py "$tr" 20 > $out/r04_tile_kernel_populations.txt 2>&1; cat $out/r04_tile_kernel_populations.txt
  python scripts/step_timeline.py "$tr" $out/step_timeline.txt 20 > $out/step_streams.txt 2>&1; cat $out/step_streams.txt
  grep -i "photo_tile\|adam" $out/r04_bench_default_rocprofv3_kernel_stats.csv | cut -c1-220
fi
bash scripts/long_run.sh r4v_adam 600 < /dev/null 2>&1 | tail -22
cp gpurun_out/long/r4v_adam.log $out/ 2>/dev/null
