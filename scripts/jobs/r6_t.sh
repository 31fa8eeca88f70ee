#!/bin/bash
# round 6: fewer partial products (torch.set_float32_matmul_precision "high" / "medium"): tests, kernel times, step A/B on one box
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_products; mkdir -p $O
python -m pytest tests/test_conv_mfma_gpu.py -x -q -k "partial_products or small_integers_are_exact or packs or stale" -s 2>&1 | grep -v Warning | tail -40 > $O/tests.txt
for np in 6 3 1; do DD_MFMA_PRODUCTS=$np timeout 300 python scripts/time_conv_mfma.py 9 > $O/time_np$np.txt 2>&1; done
for rep in 1 2; do
  for prec in highest high medium; do
    timeout 300 python bench.py --no_cpu_baseline --steps 30 --warmup 10 --matmul_precision $prec 2>/dev/null | tail -1 > $O/bench_${prec}_$rep.json
  done
done
tail -5 $O/tests.txt
