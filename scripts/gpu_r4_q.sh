#!/bin/bash
# round 4, GPU call q: the one-launch Adam (dd_adam_multi) and the slice / pad clean-ups -- their tests, then bench A/B on one box
cd /root/repo; out=/root/repo/gpurun_out/r4q; mkdir -p $out
timeout 900 python -m pytest tests/test_adam.py tests/test_networks.py tests/test_trainer_gpu.py tests/test_train_loop_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -25 > $out/pytest.log < /dev/null; tail -25 $out/pytest.log
for v in new stock_adam new stock_adam; do
  if [ $v = stock_adam ]; then export DD_STOCK_ADAM=1; else unset DD_STOCK_ADAM; fi
  DD_SEG_TIMING=1 timeout 300 python bench.py --no_cpu_baseline --mode graph > $out/$v.json 2> $out/$v.err < /dev/null
  echo "== $v"; python - <<PY
import json
d=json.loads(open('$out/$v.json').read().strip().splitlines()[-1]); print(d['value'],'img/s',d['ms_per_step'],'ms/step')
PY
  grep "segment optim\|segment depth bwd" $out/$v.err | tail -2
done 2>&1 | tee $out/adam_ab.txt
