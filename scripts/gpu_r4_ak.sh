#!/bin/bash
# round 4, GPU call ak: the fp16 step's update through dd_adam_multi (GradScaler's device scalars): half-precision tests, fp16 rows A/B
cd /root/repo; out=/root/repo/gpurun_out/r4ak; mkdir -p $out
PYTHONUNBUFFERED=1 timeout 1200 python -u -m pytest tests/test_zz_half_precision_gpu.py -q -x -m gpu -p no:cacheprovider > $out/pytest.log 2>&1 < /dev/null; echo "rc $?" >> $out/pytest.log; tail -5 $out/pytest.log
row() { label=$1; shift; timeout 300 python bench.py --no_cpu_baseline --mode graph "$@" 2>/dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'],'img/s',d['ms_per_step'],'ms/step', d['config']['optimizer_update'], d['config']['final_loss'])"; }
{ row "kitti fp16 new" --amp fp16
DD_STOCK_ADAM=fp16 row "kitti fp16 torch" --amp fp16
row "c5 fp16 new" --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp fp16
DD_STOCK_ADAM=fp16 row "c5 fp16 torch" --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp fp16
row "kitti fp16 new" --amp fp16
DD_STOCK_ADAM=fp16 row "kitti fp16 torch" --amp fp16; } 2>&1 | tee $out/ab.txt
