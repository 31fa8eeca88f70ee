#!/bin/bash
# round 4, GPU call aa: every convolution of the step with its shapes and device time (eager, single stream), to find the ones the library does badly
cd /root/repo; out=/root/repo/gpurun_out/r4aa; mkdir -p $out
DD_PROBE_ROWS=200 DD_PROBE_FILTER="aten::miopen_convolution,aten::convolution_backward,aten::miopen_depthwise,aten::mm,aten::addmm,aten::bmm" timeout 400 python scripts/probe_step_ops.py > $out/convs.txt 2>&1 < /dev/null; grep -v Warning $out/convs.txt | tail -150
