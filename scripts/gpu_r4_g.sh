#!/bin/bash
# round 4, GPU call g: whole GPU suite on the final kernels; bench line; scheduling experiments (critical-path priority stream); regulariser
# task costs; loss-path trace
mkdir -p gpurun_out/r4g
cd /root/repo
root=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r4g/pytest.log
tail -6 gpurun_out/r4g/pytest.log
timeout 400 python bench.py > gpurun_out/r4g/bench_default.json 2> gpurun_out/r4g/bench_default.err
DD_SEG_PRIORITY=1 DD_SEG_TIMING=1 timeout 400 python bench.py --mode graph --no_cpu_baseline > gpurun_out/r4g/bench_priority.json 2> gpurun_out/r4g/bench_priority.err
DD_SEG_TIMING=1 timeout 400 python bench.py --mode graph --no_cpu_baseline > gpurun_out/r4g/bench_graph.json 2> gpurun_out/r4g/bench_graph.err
timeout 400 python bench.py --mode graph --no_cpu_baseline --no_miopen_find > gpurun_out/r4g/bench_nofind.json 2> gpurun_out/r4g/bench_nofind.err
for f in default priority graph nofind; do python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r4g/bench_$f.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('$f', d['value'], 'img/s', d['ms_per_step'], 'ms; tile', r['avg_launch_us'], r['frac'], 'loss path', r['loss_path_us'], r['frac_loss_path'], 'replayed', r.get('loss_path_replayed_us'), 'host', d['config']['host_enqueue_ms_per_step'])
except Exception as e:
    print('$f', 'failed', e)
PY
grep "bench " gpurun_out/r4g/bench_$f.err | grep -v segment | head -8; done
grep segment gpurun_out/r4g/bench_priority.err | tail -14
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/r4g/losstrace -- python $root/scripts/loss_path_workload.py fine_tune 12 30 > $root/gpurun_out/r4g/loss_workload.log 2>&1 )
tr=$(find gpurun_out/r4g/losstrace -name '*kernel_trace.csv' | head -1)
python scripts/loss_kernels.py "$tr" 20 > gpurun_out/r4g/r04_loss_path_kernels.txt 2>&1
rm -rf gpurun_out/r4g/losstrace
cat gpurun_out/r4g/r04_loss_path_kernels.txt
timeout 600 bash scripts/reg_task_costs.sh > gpurun_out/r4g/reg_task_costs.txt 2>&1
cat gpurun_out/r4g/reg_task_costs.txt
