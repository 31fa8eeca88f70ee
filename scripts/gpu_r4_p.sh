#!/bin/bash
# round 4, GPU call p: WHEN the statistics-only side batch runs -- beside the forward passes (start), beside the backward passes (loss),
# or in the backward's short branch behind the pose backward (pose)
cd /root/repo; out=/root/repo/gpurun_out/r4p; mkdir -p $out
for at in start loss pose start loss pose; do
  DD_SEG_SIDE_AT=$at DD_SEG_TIMING=1 timeout 300 python bench.py --no_cpu_baseline --mode graph > $out/$at.json 2> $out/$at.err < /dev/null
  echo "== $at"; python - <<PY
import json
d=json.loads(open('$out/$at.json').read().strip().splitlines()[-1]); print(d['value'],'img/s',d['ms_per_step'],'ms/step')
PY
  grep "segment" $out/$at.err | tail -16
done 2>&1 | tee $out/side_at.txt
