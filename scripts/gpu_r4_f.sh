#!/bin/bash
# round 4, GPU call f: new tests (resize, fp16 replay), MIOpen find-db refresh for the half-precision / config-4/5 problems
mkdir -p gpurun_out/r4f
cd /root/repo
timeout 900 python -m pytest tests/test_resize_gpu.py tests/test_jpeg.py tests/test_zz_half_precision_gpu.py tests/test_ddp_gpu.py tests/test_photo_gpu.py tests/test_photo_edge_gpu.py tests/test_fused_loss_gpu.py -q -m gpu -s -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/r4f/pytest.log
tail -15 gpurun_out/r4f/pytest.log
bash scripts/refresh_miopen_db.sh r4f 900
