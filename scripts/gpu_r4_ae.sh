#!/bin/bash
# round 4, GPU call ae: dd_redu -- parity tests, decoder / trainer tests, bench A/B
cd /root/repo; out=/root/repo/gpurun_out/r4ae; mkdir -p $out
PYTHONUNBUFFERED=1 timeout 900 python -u -m pytest tests/test_small_conv_gpu.py tests/test_trainer_gpu.py -q -x -m gpu -p no:cacheprovider > $out/pytest.log 2>&1 < /dev/null; echo "rc $?" >> $out/pytest.log; tail -12 $out/pytest.log
for v in new stock new stock; do
  if [ $v = stock ]; then export DD_STOCK_REDU=1; else unset DD_STOCK_REDU; fi
  timeout 300 python bench.py --no_cpu_baseline --mode graph 2>/dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'],'img/s',d['ms_per_step'],'ms/step', d['config']['final_loss'])"
done 2>&1 | tee $out/ab.txt
