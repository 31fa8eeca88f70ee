set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
out=gpurun_out/r6i; mkdir -p $out
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/c5 -- python $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --mode graph --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp fp16 > $GRAFT_REPO_ROOT/$out/c5_bench.log 2>&1 )
tr=$(find $out/c5 -name '*kernel_trace.csv' | head -1)
python scripts/steady_state_stats.py "$tr" 10 $out/c5_fp16_steady.csv 66
python scripts/categorise_stats.py $out/c5_fp16_steady.csv > $out/c5_fp16_categories.txt 2>&1
rm -rf $out/c5
cat $out/c5_fp16_categories.txt | head -24
