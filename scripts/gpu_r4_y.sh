#!/bin/bash
cd /root/repo; out=/root/repo/gpurun_out/r4y; mkdir -p $out
timeout 300 python scripts/time_small_convs.py 2>&1 < /dev/null | grep -v amdgpu | tee $out/small_convs.txt
