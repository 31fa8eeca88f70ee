#!/bin/bash
# round 4, third GPU call: the new / changed GPU tests, smoke, fp16 NaN statistics with post-mortem, default bench line
mkdir -p gpurun_out/r4c
cd /root/repo
timeout 1500 python -m pytest tests/test_ground_pin.py tests/test_ddp_gpu.py tests/test_bench_gpu.py tests/test_jpeg.py -q -m gpu -s -p no:cacheprovider 2>&1 | tail -120 > gpurun_out/r4c/pytest_new.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4c/smoke.log 2>&1
DD_AMP_CACHE=1 timeout 700 python scripts/probe_amp_nan.py --steps 8 --runs 14 > gpurun_out/r4c/probe_cache1.log 2>&1
timeout 400 python bench.py > gpurun_out/r4c/bench_default.json 2> gpurun_out/r4c/bench_default.err
tail -25 gpurun_out/r4c/pytest_new.log; grep smoke gpurun_out/r4c/smoke.log | tail -12; grep -E "^run|post-mortem" gpurun_out/r4c/probe_cache1.log | cut -c1-500; cat gpurun_out/r4c/bench_default.json | cut -c1-1500
