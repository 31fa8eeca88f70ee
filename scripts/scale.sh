#!/bin/bash
# Weak-scaling sweep of bench.py on ONE node: 1, 2, 4, 8 GPUs (as many as the box exposes), one JSON line each -> gpurun_out/scale_<tag>.jsonl
# Every line carries config.host_enqueue_ms_per_rank (host time per step on every rank): the step is within a few ms of being
# host-bound, so a rank whose Python side is slow (a loaded core) shows there before it shows in img/s.
#   bash scripts/scale.sh [tag] [steps] [warmup]
set -u
tag=${1:-r02}; steps=${2:-20}; warmup=${3:-5}
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/scale_$tag.jsonl; : > $out
ngpu=$(python -c "import torch; print(torch.cuda.device_count())")
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 1 2 4 8; do
  [ "$n" -gt "$ngpu" ] && break
  if [ "$n" -eq 1 ]; then
    python bench.py --gpus 1 --steps $steps --warmup $warmup --no_cpu_baseline | tail -1 >> $out
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) \
      bench.py --gpus $n --steps $steps --warmup $warmup | tail -1 >> $out
  fi
done
python - "$out" <<'PY'
import json, sys
base = base_ms = None
for ln in open(sys.argv[1]):
    r = json.loads(ln)
    base = base or r["value"]
    base_ms = base_ms or r["ms_per_step"]
    c = r["config"]
    bud = (c.get("allreduce_budget") or {}).get("exposed_ms_at_%d_gpus" % r["n_gpus"])
    if bud:          # the step's growth over N=1 beside what the collective would cost if fully exposed
        print("n=%d  step %+.2f ms over n=1; the all-reduce of %.0f MB fully exposed would cost %.2f .. %.2f ms (ring at 300 .. 150 GB/s) -> expected x%.2f .. x%.2f" % (
            r["n_gpus"], r["ms_per_step"] - base_ms, c["allreduce_budget"]["gradient_bytes"] / 1e6, bud[0], bud[1],
            r["n_gpus"] * base_ms / (base_ms + bud[0]), r["n_gpus"] * base_ms / (base_ms + bud[1])))
    print("n=%d  %.1f img/s  %.2f ms/step  x%.2f  mode %s  rccl_ranks %s  reduce_mode %s (probe ms %s)  distinct_hw_queues_found %s  host enqueue per rank (ms): %s" % (
        r["n_gpus"], r["value"], r["ms_per_step"], r["value"] / base, c.get("mode"), c.get("rccl_ranks"), c.get("reduce_mode"), c.get("reduce_probe_ms"),
        c.get("distinct_hw_queues_found"), c.get("host_enqueue_ms_per_rank")))
PY
