#!/bin/bash
# ONE parameterised entry for the GPU calls of a round:   gpurun --timeout T -- 'bash scripts/gpu_job.sh <tag> <job> [args...]; ...'
# Everything lands in gpurun_out/<tag>/ (merged back into the build container).  Jobs can be chained in one call:
#   bash scripts/gpu_job.sh r5a tests tests/test_networks_gpu.py -k abs_rel
#   bash scripts/gpu_job.sh r5a suite            # the whole -m gpu suite
#   bash scripts/gpu_job.sh r5a smoke
#   bash scripts/gpu_job.sh r5a bench [bench.py args]       # -> bench_<n>.json / .err
#   bash scripts/gpu_job.sh r5a loss                        # rocprofv3 kernel trace of the fused loss alone -> loss_path_kernels.txt
#   bash scripts/gpu_job.sh r5a photo                       # scripts/time_photo.py rows (tile kernel, all instantiations)
#   bash scripts/gpu_job.sh r5a py <script.py> [args]       # any script, logged
# Every job is bounded by `timeout` (DD_JOB_TIMEOUT, default per job) so that a hang cannot eat the box.
set -u
tag=$1; job=$2; shift 2
cd "$(dirname "$0")/.." || exit 1
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
n=$(ls $out 2>/dev/null | wc -l)
case $job in
  tests)
    timeout ${DD_JOB_TIMEOUT:-1500} python -u -m pytest "$@" -q -m gpu -p no:cacheprovider -s > $out/pytest_$n.log 2>&1 < /dev/null
    echo "rc $?" >> $out/pytest_$n.log; grep -E "passed|failed|error|rc " $out/pytest_$n.log | tail -6 ;;
  suite)
    timeout ${DD_JOB_TIMEOUT:-2400} python -u -m pytest tests -q -m gpu -p no:cacheprovider "$@" > $out/suite_$n.log 2>&1 < /dev/null
    echo "rc $?" >> $out/suite_$n.log; tail -5 $out/suite_$n.log ;;
  smoke)
    timeout ${DD_JOB_TIMEOUT:-300} python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke_$n.log 2>&1 < /dev/null
    echo "rc $?" >> $out/smoke_$n.log; grep -i "smoke\|rc " $out/smoke_$n.log | tail -4 ;;
  bench)
    timeout ${DD_JOB_TIMEOUT:-500} python bench.py "$@" > $out/bench_$n.json 2> $out/bench_$n.err < /dev/null
    echo "rc $? args: $*" >> $out/bench_$n.err; cut -c1-400 $out/bench_$n.json ;;
  loss)
    ( cd /tmp && timeout ${DD_JOB_TIMEOUT:-400} rocprofv3 --kernel-trace --output-format csv -d $out/loss_trace_$n -- python $root/scripts/loss_path_workload.py "$@" > $out/loss_$n.log 2>&1 )
    tr=$(find $out/loss_trace_$n -name '*kernel_trace.csv' | head -1)
    python scripts/loss_kernels.py "$tr" > $out/loss_path_kernels_$n.txt 2>&1; rm -rf $out/loss_trace_$n; cat $out/loss_path_kernels_$n.txt | tail -30 ;;
  photo)
    timeout ${DD_JOB_TIMEOUT:-400} python scripts/time_photo.py "$@" > $out/time_photo_$n.txt 2>&1 < /dev/null; tail -30 $out/time_photo_$n.txt ;;
  py)
    timeout ${DD_JOB_TIMEOUT:-600} python -u "$@" > $out/py_$n.log 2>&1 < /dev/null; echo "rc $?" >> $out/py_$n.log; tail -40 $out/py_$n.log ;;
  *) echo "unknown job $job"; exit 2 ;;
esac
