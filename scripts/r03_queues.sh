#!/bin/bash
# Hardware-queue count against the replayed step (results under gpurun_out/exp/): timeline + throughput with 4 (HIP's default) and 8 queues.
mkdir -p gpurun_out/exp; export TMPDIR=/tmp
O=gpurun_out/exp
for q in 8 4; do
  GPU_MAX_HW_QUEUES=$q DD_SEG_TIMING=1 timeout 600 python bench.py --mode graph --no_cpu_baseline > $O/queues$q.json 2> $O/queues$q.err
  echo "== $q hardware queues: $(grep '^{' $O/queues$q.json | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(d['value'], 'img/s', d['ms_per_step'], 'ms  host', d['config']['host_enqueue_ms_per_step'], ' tile', r.get('avg_launch_us'), 'us frac', r.get('frac'), ' loss path', r.get('loss_path_us'), r.get('frac_loss_path'))" 2>&1)"
  grep "segment" $O/queues$q.err
done
GPU_MAX_HW_QUEUES=8 timeout 600 python bench.py --mode eager --no_cpu_baseline > $O/queues8_eager.json 2> $O/queues8_eager.err
echo "== eager, 8 queues: $(grep '^{' $O/queues8_eager.json | tail -1 | cut -c1-200)"
