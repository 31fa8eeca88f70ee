#!/bin/bash
# round 4, GPU call ac: BatchNorm backward without the saved-output read (no residual): operator tests, network goldens, bench
cd /root/repo; out=/root/repo/gpurun_out/r4ac; mkdir -p $out
PYTHONUNBUFFERED=1 timeout 1200 python -u -m pytest tests/test_ops_gpu.py tests/test_trainer_gpu.py -q -x -m gpu -p no:cacheprovider > $out/pytest.log 2>&1 < /dev/null; echo "rc $?" >> $out/pytest.log; tail -4 $out/pytest.log
for i in 1 2; do timeout 300 python bench.py --no_cpu_baseline --mode graph 2>/dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'],'img/s',d['ms_per_step'],'ms/step')"; done | tee $out/bench.txt
