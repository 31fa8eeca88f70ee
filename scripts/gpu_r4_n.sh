#!/bin/bash
# round 4, GPU call n: per-stream timeline of one replayed step (which kernels make the depth backward chain), and which operators
# issue the zero-fills / copies / adds of a step
cd /root/repo; out=/root/repo/gpurun_out/r4n; mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python /root/repo/bench.py --no_cpu_baseline --steps 10 --warmup 3 --mode graph > $out/bench_prof.json 2> $out/bench_prof.err ) < /dev/null
f=$(find /tmp/tl -name "*kernel_trace.csv" 2>/dev/null | head -1); echo "trace [$f]"
if [ -n "$f" ]; then head -1 "$f"; timeout 200 python scripts/step_timeline.py "$f" $out/step_timeline.txt 20 < /dev/null | tee $out/step_streams.txt; else tail -5 $out/bench_prof.err; fi
DD_PROBE_ROWS=90 DD_PROBE_FILTER="aten::fill_,aten::zero_,aten::copy_,aten::add_,aten::add,aten::mul,aten::cat,aten::sum,aten::mul_" timeout 300 python scripts/probe_copy_parents.py > $out/parents.txt 2>&1 < /dev/null; tail -95 $out/parents.txt
