#!/bin/bash
cd /root/repo; out=/root/repo/gpurun_out/r4ai; mkdir -p $out
PYTHONUNBUFFERED=1 timeout 900 python -u -m pytest tests/test_trainer_gpu.py tests/test_small_conv_gpu.py tests/test_train_loop_gpu.py -q -x -m gpu -p no:cacheprovider -k "hooks_match or small_conv or motion_decoder or head or reduction or train_py or four_phase or litemono" > $out/pytest.log 2>&1 < /dev/null; echo "rc $?" >> $out/pytest.log; tail -5 $out/pytest.log
timeout 300 python bench.py --no_cpu_baseline --mode graph 2>/dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'],'img/s',d['ms_per_step'],'ms/step')"
