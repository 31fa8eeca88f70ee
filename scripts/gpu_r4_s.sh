#!/bin/bash
# round 4, GPU call s: Adam with its prologue kernel; slice / pad clean-ups A/B on one box
cd /root/repo; out=/root/repo/gpurun_out/r4s; mkdir -p $out
PYTHONUNBUFFERED=1 timeout 600 python -u -m pytest tests/test_adam.py -q -x -m gpu -p no:cacheprovider > $out/adam.log 2>&1 < /dev/null; echo "rc $?" >> $out/adam.log; tail -4 $out/adam.log
for v in new stock_slices new stock_slices; do
  if [ $v = stock_slices ]; then export DD_STOCK_SLICES=1; else unset DD_STOCK_SLICES; fi
  DD_SEG_TIMING=1 timeout 300 python bench.py --no_cpu_baseline --mode graph > $out/$v.json 2> $out/$v.err < /dev/null
  echo "== $v"; python - <<PY
import json
d=json.loads(open('$out/$v.json').read().strip().splitlines()[-1]); print(d['value'],'img/s',d['ms_per_step'],'ms/step')
PY
  grep "segment" $out/$v.err | tail -13
done 2>&1 | tee $out/ab.txt
