#!/bin/bash
# round 4, GPU call w: the one-launch Adam really in the replayed step (layout check fixed for 1x1 weights): tests, bench A/B against torch's
# kernel on aligned gradients, rocprofv3 stats of the default command
cd /root/repo; out=/root/repo/gpurun_out/r4w; mkdir -p $out
PYTHONUNBUFFERED=1 timeout 1400 python -u -m pytest tests/test_adam.py tests/test_trainer_gpu.py tests/test_train_loop_gpu.py tests/test_ddp_gpu.py tests/test_bench_gpu.py -q -x -m gpu -p no:cacheprovider > $out/pytest.log 2>&1 < /dev/null; echo "rc $?" >> $out/pytest.log; tail -5 $out/pytest.log
for v in new torch new torch; do
  if [ $v = torch ]; then export DD_STOCK_ADAM=1; else unset DD_STOCK_ADAM; fi
  DD_SEG_TIMING=1 timeout 300 python bench.py --no_cpu_baseline --mode graph > $out/$v.json 2> $out/$v.err < /dev/null
  python - <<PY
import json
d=json.loads(open('$out/$v.json').read().strip().splitlines()[-1]); print('$v', d['value'],'img/s',d['ms_per_step'],'ms/step', d['config']['optimizer_update'])
PY
  grep "segment optim" $out/$v.err | tail -1
done 2>&1 | tee $out/adam_ab.txt
export TMPDIR=/tmp; unset DD_STOCK_ADAM
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python /root/repo/bench.py --no_cpu_baseline > $out/bench_prof.log 2>&1 ) < /dev/null
grep "^{" $out/bench_prof.log | tail -1 > $out/r04_bench_line_under_rocprofv3.json
st=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); tr=$(find /tmp/prof -name '*kernel_trace.csv' | head -1)
if [ -n "$st" ]; then
  cp "$st" $out/r04_bench_default_rocprofv3_kernel_stats.csv
  python scripts/steady_state_stats.py "$tr" 10 $out/r04_bench_fine_tune_steady_kernel_stats.csv 22
  python scripts/categorise_stats.py $out/r04_bench_fine_tune_steady_kernel_stats.csv > $out/r04_bench_categories.txt 2>&1; head -14 $out/r04_bench_categories.txt
  python scripts/tile_populations.py "$tr" 20 > $out/r04_tile_kernel_populations.txt 2>&1; cat $out/r04_tile_kernel_populations.txt
  python scripts/step_timeline.py "$tr" $out/step_timeline.txt 20 > $out/step_streams.txt 2>&1; cat $out/step_streams.txt
  grep -i "photo_tile\|adam" $out/r04_bench_default_rocprofv3_kernel_stats.csv | cut -c1-200
fi
