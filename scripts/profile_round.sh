#!/bin/bash
# The rocprofv3 evidence of a round, collected on the GPU box:  bash scripts/profile_round.sh <tag>   (tag e.g. r02)
#   1. `rocprofv3 --kernel-trace --stats` over the default bench command -> <tag>_bench_default_rocprofv3_kernel_stats.csv
#      and the steady-state aggregation of the same trace (last 10 steps) -> <tag>_bench_fine_tune_steady_kernel_stats.csv
#   2. FETCH_SIZE and WRITE_SIZE in separate --pmc passes over scripts/pmc_workload.py -> <tag>_pmc_*.csv, photo_traffic.json
# Everything lands in gpurun_out/prof_<tag>/; copy what is to be judged into profiles/.
set -u
tag=${1:-r02}
cd "$(dirname "$0")/.." || exit 1
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -- python $root/bench.py --no_cpu_baseline ${DD_BENCH_ARGS:-} > $out/bench.log 2>&1 )
grep "^{" $out/bench.log | tail -1 > $out/${tag}_bench_line.json
st=$(find $out/bench -name '*kernel_stats.csv' | head -1)
tr=$(find $out/bench -name '*kernel_trace.csv' | head -1)
cp "$st" $out/${tag}_bench_default_rocprofv3_kernel_stats.csv
python scripts/steady_state_stats.py "$tr" 10 $out/${tag}_bench_fine_tune_steady_kernel_stats.csv ${DD_TRACE_SKIP:-66}    # bench.py issues 3 x 22 loss evaluations behind the timed region (round 6: three legs)
python scripts/tile_populations.py "$tr" 20 > $out/${tag}_tile_kernel_populations.txt 2>&1
python scripts/categorise_stats.py $out/${tag}_bench_fine_tune_steady_kernel_stats.csv > $out/${tag}_categories.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python $root/scripts/pmc_workload.py > $out/pmc_$c.log 2>&1 )
  f=$(find $out/pmc_$c -name '*counter_collection.csv' | head -1)
  python scripts/pmc_summary.py "$f" $out/${tag}_pmc_$c.csv > /dev/null
done
python scripts/make_traffic_json.py $out/${tag}_pmc_FETCH_SIZE.csv $out/${tag}_pmc_WRITE_SIZE.csv $tag $out/photo_traffic.json
rm -rf $out/bench $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
grep -i "photo" $out/${tag}_bench_default_rocprofv3_kernel_stats.csv | cut -c1-200
cat $out/${tag}_categories.txt | tail -25
cat $out/${tag}_tile_kernel_populations.txt
cat $out/${tag}_bench_line.json
