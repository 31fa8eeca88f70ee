#!/bin/bash
# Back-to-back bench lines of one box (A/B rows of DESIGN.md section 6):  bash scripts/sweep_bench.sh > gpurun_out/sweep.txt
# Every row is bounded by `timeout` (hipGraph replay of the Waymo shape hung once with parallel graph branches; graph rows of
# other shapes than KITTI are therefore run with --mode eager here).
cd "$(dirname "$0")/.." || exit 1
row() {  # label, args...
  label=$1; shift
  timeout 420 python bench.py --no_cpu_baseline "$@" 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
t=sys.stdin.read()
if not t.strip():
    print('%-44s (no result within 420 s)' % '$label'); raise SystemExit
r=json.loads(t); ro=r.get('roofline') or {}; c=r['config']
print('%-44s %7.1f img/s %7.2f ms/step  mode=%s tile=%s us loss=%s us frac=%s host=%s ms final_loss=%s' % ('$label', r['value'], r['ms_per_step'], c.get('mode'), ro.get('avg_launch_us'), ro.get('dd_photo_loss_us'), ro.get('frac'), c.get('host_enqueue_ms_per_step'), c.get('final_loss')))"
}
if [ "${1:-}" = "r06" ]; then       # the rows of DESIGN.md section 6, round 6: fp32 rows beside the headline, then the half-precision networks with and without dd_conv3x3_half
row "default (auto)"
row "disp_init" --phase disp_init --mode graph
row "motion_init" --phase motion_init --mode graph
row "mask_init" --phase mask_init --mode graph
row "monodepthv2 kitti B=12" --depth_model monodepthv2 --mode graph
row "waymo litemono 320x480 B=8" --dataset waymo --batch 8 --mode graph
row "nuscenes md2 B=16 fp32" --dataset nuscenes --depth_model monodepthv2 --batch 16 --mode graph
row "nuscenes md2 B=16 fp16" --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp fp16 --mode graph
DD_HALF_MFMA_CONV=0 row "nuscenes md2 B=16 fp16, library convs" --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp fp16 --mode graph
row "nuscenes md2 B=16 bf16" --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp bf16 --mode graph
DD_HALF_MFMA_CONV=0 row "nuscenes md2 B=16 bf16, library convs" --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp bf16 --mode graph
row "kitti litemono fp16 networks" --amp fp16 --mode graph
row "kitti litemono bf16 networks" --amp bf16 --mode graph
exit 0
fi
if [ "${1:-}" = "r05" ]; then       # fp32 rows beside the headline on the final code of round 5 (the new convolution kernels serve fp32 networks)
row "default (auto)"
row "disp_init" --phase disp_init --mode graph
row "motion_init" --phase motion_init --mode graph
row "mask_init" --phase mask_init --mode graph
row "monodepthv2 kitti B=12" --depth_model monodepthv2 --mode graph
row "waymo litemono 320x480 B=8" --dataset waymo --batch 8 --mode graph
row "nuscenes md2 B=16 fp32" --dataset nuscenes --depth_model monodepthv2 --batch 16 --mode graph
DD_STOCK_MFMA_CONV=1 row "default, library 3x3 convolutions" --mode graph
exit 0
fi
if [ "${1:-}" = "r03s" ]; then      # the short list (GPU budget): phases, the other depth net, config 5's shape and precisions, the opt-in lean side frames
row "graph, encoder-only side frames" --mode graph --stats_only_side_frames
row "graph bf16 networks" --mode graph --amp bf16
row "eager fp16 networks" --amp fp16
row "disp_init" --phase disp_init --mode graph
row "motion_init" --phase motion_init --mode graph
row "mask_init" --phase mask_init --mode graph
row "monodepthv2 kitti B=12" --depth_model monodepthv2 --mode graph
row "nuscenes md2 B=16 fp32" --dataset nuscenes --depth_model monodepthv2 --batch 16 --mode graph
row "nuscenes md2 B=16 bf16" --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp bf16 --mode graph
row "nuscenes md2 B=16 fp16 (eager)" --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp fp16
exit 0
fi
if [ "${1:-}" = "r03" ]; then      # the rows of DESIGN.md section 6, round 3 (the replayed step is the default everywhere)
row "default (auto)"
row "graph (per-network graphs)" --mode graph
row "eager multi-stream" --mode eager
row "eager single-stream" --mode eager --single_stream
DD_GRAPH=whole row "whole-step graph (DD_GRAPH=whole)" --mode graph
row "graph, encoder-only side frames" --mode graph --stats_only_side_frames
row "graph bf16 networks" --mode graph --amp bf16
row "eager fp16 networks" --amp fp16
row "disp_init" --phase disp_init --mode graph
row "motion_init" --phase motion_init --mode graph
row "mask_init" --phase mask_init --mode graph
row "monodepthv2 kitti B=12" --depth_model monodepthv2 --mode graph
row "nuscenes md2 B=16 fp32" --dataset nuscenes --depth_model monodepthv2 --batch 16 --mode graph
row "nuscenes md2 B=16 bf16" --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp bf16 --mode graph
row "nuscenes md2 B=16 fp16 (eager)" --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp fp16
row "waymo 320x480 B=8" --dataset waymo --batch 8 --mode graph
exit 0
fi
row "default (auto)"
row "eager multi-stream" --mode eager
row "graph (single-stream capture)" --mode graph
row "eager single-stream" --mode eager --single_stream
row "disp_init" --phase disp_init --mode eager
row "motion_init" --phase motion_init --mode eager
row "mask_init" --phase mask_init --mode eager
row "monodepthv2 kitti B=12" --depth_model monodepthv2 --mode eager
row "nuscenes md2 B=16 fp32" --dataset nuscenes --depth_model monodepthv2 --batch 16 --mode eager
row "nuscenes md2 B=16 fp16" --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp fp16 --mode eager
row "waymo 320x480 B=8" --dataset waymo --batch 8 --mode eager
