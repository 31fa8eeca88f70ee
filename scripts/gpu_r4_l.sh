#!/bin/bash
# round 4, GPU call l: why do the caller's stream and the motion stream share a hardware queue once a process group exists?
cd /root/repo; out=gpurun_out/r4l; mkdir -p $out
row() { label=$1; shift; env "$@" DD_STREAM_PICK_DEBUG=1 DD_SEG_TIMING=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29585 timeout 300 python bench.py --no_cpu_baseline --mode graph > $out/$label.json 2> $out/$label.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/$label.json').read().strip().splitlines()[-1]); print('%-30s'%'$label', d['value'], 'img/s', d['ms_per_step'], 'ms', 'queues found', d['config']['distinct_hw_queues_found'], 'ranks', d['config']['rccl_ranks'])
except Exception as e: print('$label failed', e)
PY
  grep "queues\]" $out/$label.err; grep "depth fwd\|pose bwd" $out/$label.err | tail -2
}
row no_group DD_X=0
row rccl1 DD_BENCH_FORCE_DIST=1
row rccl1_repick DD_BENCH_FORCE_DIST=1 DD_STREAM_REPICK=1
