#!/bin/bash
# round 4, GPU call j: RCCL's one-rank all-reduce timed alone; the smoothness kernel with its XCD band remap
cd /root/repo; out=gpurun_out/r4j; mkdir -p $out; export TMPDIR=/tmp; root=$PWD
timeout 200 python scripts/time_allreduce_one_rank.py 2>&1 | grep -v "^$" | tail -12 > $out/allreduce_one_rank.txt; cat $out/allreduce_one_rank.txt
timeout 600 python -m pytest tests/test_fused_loss_gpu.py tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $root/$out/losstrace -- python $root/scripts/loss_path_workload.py fine_tune 12 30 > $root/$out/loss_workload.log 2>&1 )
python scripts/loss_kernels.py "$(find $out/losstrace -name '*kernel_trace.csv' | head -1)" 20 > $out/r04_loss_path_kernels.txt 2>&1; rm -rf $out/losstrace; cat $out/r04_loss_path_kernels.txt
