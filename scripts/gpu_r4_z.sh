#!/bin/bash
# round 4, GPU call z: dd_conv_small -- parity tests, stand-alone timing against MIOpen, bench A/B
cd /root/repo; out=/root/repo/gpurun_out/r4z; mkdir -p $out
PYTHONUNBUFFERED=1 timeout 900 python -u -m pytest tests/test_small_conv_gpu.py -q -x -m gpu -p no:cacheprovider > $out/pytest.log 2>&1 < /dev/null; echo "rc $?" >> $out/pytest.log; tail -25 $out/pytest.log
timeout 300 python scripts/time_small_convs.py 2>&1 < /dev/null | grep -v amdgpu | tee $out/small_convs.txt
for v in new stock new stock; do
  if [ $v = stock ]; then export DD_STOCK_SMALL_CONV=1; else unset DD_STOCK_SMALL_CONV; fi
  DD_SEG_TIMING=1 timeout 300 python bench.py --no_cpu_baseline --mode graph > $out/$v.json 2> $out/$v.err < /dev/null
  python - <<PY
import json
d=json.loads(open('$out/$v.json').read().strip().splitlines()[-1]); print('$v', d['value'],'img/s',d['ms_per_step'],'ms/step', d['config']['final_loss'])
PY
  grep "segment motion" $out/$v.err | tail -2
done 2>&1 | tee $out/ab.txt
