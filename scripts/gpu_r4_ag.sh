#!/bin/bash
cd /root/repo; out=/root/repo/gpurun_out/r4ag; mkdir -p $out
DD_PROBE_ROWS=90 timeout 400 python scripts/probe_step_ops.py > $out/ops.txt 2>&1 < /dev/null; grep -v "Warning\|warn" $out/ops.txt | tail -95
