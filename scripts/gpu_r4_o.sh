#!/bin/bash
# round 4, GPU call o: which operators issue the zero-fills / copies / adds / multiplies of an (eager, single-stream) training step
cd /root/repo; out=/root/repo/gpurun_out/r4o; mkdir -p $out
DD_PROBE_ROWS=140 DD_PROBE_FILTER="aten::fill_,aten::zero_,aten::copy_,aten::add_,aten::add,aten::mul,aten::cat,aten::sum,aten::mul_,aten::zeros,aten::zeros_like" timeout 400 python scripts/probe_copy_parents.py > $out/parents.txt 2>&1 < /dev/null; grep -v Warning $out/parents.txt | tail -145
