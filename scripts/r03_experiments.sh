#!/bin/bash
# Round-3 A/B runs of the replayed step on one box (results under gpurun_out/exp/): late join of the statistics-only batch,
# the decoder glue kernel, and where the per-network graphs sit on the GPU (DD_SEG_TIMING=1).
mkdir -p gpurun_out/exp; export TMPDIR=/tmp
O=gpurun_out/exp
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --mode graph --no_cpu_baseline > $O/$name.json 2> $O/$name.err
  echo "== $name: $(grep '^{' $O/$name.json | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(d['value'], 'img/s', d['ms_per_step'], 'ms  host', d['config']['host_enqueue_ms_per_step'], ' tile', r.get('avg_launch_us'), 'us frac', r.get('frac'), ' loss path', r.get('loss_path_us'), r.get('frac_loss_path'), r.get('last_timed_launch_us'))" 2>&1)"
}
run base
run timeline DD_SEG_TIMING=1
grep "segment" $O/timeline.err
run side_early DD_SEG_SIDE_LATE=0
run stock_glue DD_STOCK_DECODER_GLUE=1
run base2
