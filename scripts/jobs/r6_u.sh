#!/bin/bash
# round 6: the step at torch.set_float32_matmul_precision "highest" / "high" / "medium" (--matmul_precision), one box, two repetitions:
# the headline network and MonoDepth2 (the network with the most 3x3 stride-1 convolutions)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_precision; mkdir -p $O
for rep in 1 2; do
  for prec in highest high medium; do
    timeout 300 python bench.py --no_cpu_baseline --steps 40 --warmup 10 --matmul_precision $prec 2>/dev/null | tail -1 > $O/litemono_${prec}_$rep.json
    timeout 300 python bench.py --no_cpu_baseline --steps 40 --warmup 10 --depth_model monodepthv2 --matmul_precision $prec 2>/dev/null | tail -1 > $O/md2_${prec}_$rep.json
  done
done
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r6_precision/*.json")):
    try:
        d = json.load(open(f))
        print("%-28s %7.1f img/s  %6.2f ms/step  final loss %.6f  %s" % (os.path.basename(f), d["value"], d["ms_per_step"], d["config"]["final_loss"], d["dtype"][:90]))
    except Exception as e:
        print(f, "unreadable", e)
PY
