#!/bin/bash
# round 4, GPU call ah: final verification on the code with dd_adam_multi and dd_conv_small -- whole GPU suite, smoke, default bench line,
# rocprofv3 kernel stats of the default command, long fine_tune run of train.py
cd /root/repo; out=/root/repo/gpurun_out/r4ah; mkdir -p $out
PYTHONUNBUFFERED=1 timeout 1500 python -u -m pytest tests -q -m gpu -p no:cacheprovider > $out/pytest.log 2>&1 < /dev/null; echo "rc $?" >> $out/pytest.log; tail -6 $out/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 < /dev/null | grep "smoke\|SMOKE" > $out/smoke.log; tail -2 $out/smoke.log
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err < /dev/null; cut -c1-260 $out/bench_default.json; grep "bench " $out/bench_default.err | tail -4
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
  grep -i "photo_tile\|adam_multi\|conv_small_kernel<3, 9, 9>\|conv_head\|redu_" $out/r04_bench_default_rocprofv3_kernel_stats.csv | cut -c1-160
fi
bash scripts/long_run.sh r4ah_final 400 --log_frequency 50 < /dev/null 2>&1 | tail -8
cp gpurun_out/long/r4ah_final.log $out/ 2>/dev/null
