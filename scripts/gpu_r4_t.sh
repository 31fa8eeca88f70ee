#!/bin/bash
# round 4, GPU call t: which of the slice / pad clean-ups costs time (each switched back to autograd's own on its own), Adam alone
cd /root/repo; out=/root/repo/gpurun_out/r4t; mkdir -p $out
timeout 200 python scripts/time_adam.py 2>&1 < /dev/null | grep -v "Trainer\|^$\|amdgpu" | tee $out/time_adam.txt
for v in 0 qkv redu pad eye 1 0 qkv; do
  DD_STOCK_SLICES=$v timeout 300 python bench.py --no_cpu_baseline --mode graph > $out/s_$v.json 2> $out/s_$v.err < /dev/null
  python - <<PY
import json
d=json.loads(open('$out/s_$v.json').read().strip().splitlines()[-1]); print('DD_STOCK_SLICES=$v', d['value'],'img/s',d['ms_per_step'],'ms/step')
PY
done 2>&1 | tee $out/ab.txt
