#!/bin/bash
# Does the replayed step's loss stay finite over 600 steps?  A/B over how far the host may run ahead of the GPU (and other switches).
#   bash scripts/nan_hunt.sh <name> [ENV=VALUE ...]      -> gpurun_out/long/<name>.log, one summary line
name=$1; shift
mkdir -p gpurun_out/long; export TMPDIR=/tmp
( cd dynamo-depth_amd && env "$@" timeout 400 python train.py -d kitti --synthetic -b 12 --weights_init scratch --epoch_schedules 0 0 0 1 --epoch-size ${DD_HUNT_STEPS:-600} \
   --ramp_red 100000 --log_frequency 20 --num_workers 8 --log_dir /tmp/dd_long_logs -n $name --no_train_vis > ../gpurun_out/long/$name.log 2>&1 )
echo "== $name ($*): $(grep -c 'examples/s' gpurun_out/long/$name.log) log lines, last: $(grep 'examples/s' gpurun_out/long/$name.log | tail -1 | cut -c1-90) $(grep -o 'non-finite loss at step [0-9]*' gpurun_out/long/$name.log)"
grep "first replay with a non-finite\|no non-finite buffer" gpurun_out/long/$name.log | cut -c1-1500
