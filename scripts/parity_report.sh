#!/bin/bash
# The parity numbers of a round as the tests print them (-s):  bash scripts/parity_report.sh <tag>  -> gpurun_out/<tag>_parity_report.txt
tag=${1:-r03}
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/${tag}_parity_report.txt
{
  echo "# Parity numbers printed by tests/test_fused_loss_gpu.py (fused loss vs the reference's goldens), tests/test_photo_gpu.py (dd_photo_loss through"
  echo "# the C ABI vs the fp32 oracle; full-size cases vs the fp64 oracle, decision-masked) and tests/test_photo_edge_gpu.py (constructed geometry:"
  echo "# points behind the camera, disparity 0 / 1, flat and identical frames, far translations, closed sparsity gate, non-finite input), MI355X, $tag:"
  echo "# python -m pytest tests/test_fused_loss_gpu.py tests/test_photo_gpu.py tests/test_photo_edge_gpu.py -q -s"
  timeout 1500 python -m pytest tests/test_fused_loss_gpu.py tests/test_photo_gpu.py tests/test_photo_edge_gpu.py -q -s -p no:cacheprovider 2>&1 | grep -v "^$" | cut -c1-220
} > $out
tail -3 $out
