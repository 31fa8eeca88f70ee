#!/bin/bash
# round 4, GPU call k: what about a one-rank RCCL process group costs 7.5 ms per step?  (the collective itself takes 0.07-0.3 ms)
cd /root/repo; out=gpurun_out/r4k; mkdir -p $out
row() { label=$1; shift; env "$@" MASTER_ADDR=127.0.0.1 MASTER_PORT=29583 timeout 300 python bench.py --no_cpu_baseline --mode graph > $out/$label.json 2> $out/$label.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/$label.json').read().strip().splitlines()[-1]); print('%-30s'%'$label', d['value'], 'img/s', d['ms_per_step'], 'ms', d['config']['mode'], 'ranks', d['config']['rccl_ranks'], d['config']['dist_backend'], 'host', d['config']['host_enqueue_ms_per_step'])
except Exception as e: print('$label failed', e)
PY
}
row no_group DD_X=0
row rccl1_no_collective DD_BENCH_FORCE_DIST=1 DD_SEG_REDUCE=none
row rccl1_end DD_BENCH_FORCE_DIST=1 DD_SEG_REDUCE=end
row rccl1_end_no_watchdog DD_BENCH_FORCE_DIST=1 DD_SEG_REDUCE=end TORCH_NCCL_ENABLE_MONITORING=0 TORCH_NCCL_ASYNC_ERROR_HANDLING=0 TORCH_NCCL_DUMP_ON_TIMEOUT=0
row gloo1_end DD_BENCH_FORCE_DIST=1 DD_BENCH_BACKEND=gloo DD_SEG_REDUCE=end
row rccl1_end_timeline DD_BENCH_FORCE_DIST=1 DD_SEG_REDUCE=end DD_SEG_TIMING=1
grep segment $out/rccl1_end_timeline.err | tail -14
