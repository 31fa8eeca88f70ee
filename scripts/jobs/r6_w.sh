#!/bin/bash
# round 6: the swizzled three-workgroups-per-CU layout of dd_conv_mfma.hip (-DDD_CM_SWZ=1) with THREE partial products (--matmul_precision high)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_swz3; mkdir -p $O
V=$PWD/dynamo-depth_amd/csrc/variants/cm_swz.so
DD_MFMA_PRODUCTS=3 timeout 300 python scripts/time_conv_mfma.py 9 2>&1 | grep -v "max error\|amdgpu" > $O/time_default.txt
DYNAMO_HIP_LIB=$V DD_MFMA_PRODUCTS=3 timeout 300 python scripts/time_conv_mfma.py 9 2>&1 | grep -v "max error\|amdgpu" > $O/time_swz.txt
for rep in 1 2; do
  timeout 300 python bench.py --no_cpu_baseline --steps 30 --warmup 10 --matmul_precision high 2>/dev/null | tail -1 > $O/bench_default_$rep.json
  DYNAMO_HIP_LIB=$V timeout 300 python bench.py --no_cpu_baseline --steps 30 --warmup 10 --matmul_precision high 2>/dev/null | tail -1 > $O/bench_swz_$rep.json
done
paste -d'\n' $O/time_default.txt $O/time_swz.txt | cut -c1-200
for f in $O/bench_*.json; do python -c "import json; d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'])"; done
