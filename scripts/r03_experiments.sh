#!/bin/bash
# Round-3 A/B runs of the replayed step on one box (results under gpurun_out/exp/): late join of the statistics-only batch,
# the decoder glue kernel, and where the per-network graphs sit on the GPU (DD_SEG_TIMING=1).
mkdir -p gpurun_out/exp; export TMPDIR=/tmp
O=gpurun_out/exp
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --mode graph --no_cpu_baseline > $O/$name.json 2> $O/$name.err
  echo "== $name: $(python -c "import json,sys; d=json.load(open('$O/$name.json')); r=d.get('roofline') or {}; print(d['value'], 'img/s', d['ms_per_step'], 'ms  host', d['config']['host_enqueue_ms_per_step'], ' tile', r.get('avg_launch_us'), 'us frac', r.get('frac'), ' loss path', r.get('loss_path_us'), r.get('frac_loss_path'), r.get('last_timed_launch_us'))" 2>&1)"
}
run base
run timeline DD_SEG_TIMING=1
grep "segment" $O/timeline.err
run side_early DD_SEG_SIDE_LATE=0
run stock_glue DD_STOCK_DECODER_GLUE=1
run base2
# train.py's own launch line (the reference's: train.py -d kitti ...) on synthetic triplets, fine_tune only: its logged examples/s
# next to the bench line.  600 steps of batch 12; the first third of the epoch ramps the loss weights (eager steps).
(cd dynamo-depth_amd && timeout 900 python train.py -d kitti --synthetic -b 12 --weights_init scratch --epoch_schedules 0 0 0 1 --epoch-size 7200 \
   --log_frequency 50 --num_workers 8 --log_dir /tmp/dd_train_logs -n r03 --no_train_vis > ../$O/train_py.log 2>&1)
grep "examples/s" $O/train_py.log | tail -12
