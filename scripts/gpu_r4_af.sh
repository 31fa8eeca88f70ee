#!/bin/bash
# round 4, GPU call af: the other workloads on the final code (new conv / reduction / optimizer kernels at other shapes, phases and precisions)
cd /root/repo; out=/root/repo/gpurun_out/r4af; mkdir -p $out
row() {  label=$1; shift
  timeout 300 python bench.py --no_cpu_baseline "$@" 2>$out/err_last.txt < /dev/null | grep "^{" | tail -1 | python -c "
import json,sys
t=sys.stdin.read()
if not t.strip():
    print('%-40s (no result)' % '$label'); raise SystemExit
r=json.loads(t); ro=r.get('roofline') or {}; c=r['config']
print('%-40s %7.1f img/s %7.2f ms/step  mode=%s tile=%s us frac=%s opt=%s convs=%s final_loss=%s' % ('$label', r['value'], r['ms_per_step'], c.get('mode'), ro.get('avg_launch_us'), ro.get('frac'), c.get('optimizer_update','')[:14], c.get('motion_decoder_full_res_convs','')[:13], c.get('final_loss')))"
  grep -i "error\|Traceback" $out/err_last.txt | head -3
}
{
row "kitti litemono fine_tune fp32 (headline)" --mode graph
row "kitti litemono bf16 networks" --mode graph --amp bf16
row "kitti litemono fp16 networks" --mode graph --amp fp16
row "disp_init" --phase disp_init --mode graph
row "motion_init" --phase motion_init --mode graph
row "mask_init" --phase mask_init --mode graph
row "kitti monodepthv2 B=12 fp32" --depth_model monodepthv2 --mode graph
row "waymo 320x480 litemono B=8 fp32" --dataset waymo --batch 8 --mode graph
row "nuscenes md2 B=16 fp32" --dataset nuscenes --depth_model monodepthv2 --batch 16 --mode graph
row "nuscenes md2 B=16 fp16" --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp fp16 --mode graph
} 2>&1 | tee $out/sweep.txt
