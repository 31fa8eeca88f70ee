#!/bin/bash
# Long fine_tune runs of train.py on synthetic triplets (results under gpurun_out/long/): does the loss stay finite, and what does
# train.py log as examples/s?   bash scripts/long_run.sh <name> <batches> [extra train.py flags]
name=$1; n=$2; shift 2
mkdir -p gpurun_out/long; export TMPDIR=/tmp
cd dynamo-depth_amd && timeout 900 python train.py -d kitti --synthetic -b 12 --weights_init scratch --epoch_schedules 0 0 0 1 --epoch-size $n \
   --log_frequency 10 --num_workers 8 --log_dir /tmp/dd_long_logs -n $name --no_train_vis "$@" > ../gpurun_out/long/$name.log 2>&1
echo "== $name rc=$?"; grep -E "examples/s|Error|error" ../gpurun_out/long/$name.log | awk 'NR<=3 || NR%5==0' | tail -14 | cut -c1-200; tail -3 ../gpurun_out/long/$name.log | cut -c1-600
