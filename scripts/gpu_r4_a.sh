#!/bin/bash
# round 4, first GPU call: the whole -m gpu suite (no -x: every failure shows), the yardstick measurement, the fp16 NaN probe
mkdir -p gpurun_out/r4a
cd /root/repo
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/r4a/pytest.log
timeout 600 python scripts/measure_amp_yardstick.py 3 4 5 6 7 8 > gpurun_out/r4a/yardstick.log 2>&1
for c in 1 0; do
  DD_AMP_CACHE=$c timeout 500 python scripts/probe_amp_nan.py --steps 60 --runs 3 > gpurun_out/r4a/probe_cache$c.log 2>&1
done
tail -5 gpurun_out/r4a/pytest.log; tail -3 gpurun_out/r4a/yardstick.log; grep "^run" gpurun_out/r4a/probe_cache*.log
