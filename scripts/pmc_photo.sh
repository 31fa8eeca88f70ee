#!/bin/bash
# SQ counter passes over the photometric kernel (run on the GPU box):  bash scripts/pmc_photo.sh <tag>
# rocprofv3 --pmc only with --kernel-trace (gpurun refuses other trace domains next to counters).
set -u
tag=${1:-r02}
cd "$(dirname "$0")/.." || exit 1
root=$PWD
mkdir -p gpurun_out/pmc_$tag
export TMPDIR=/tmp DD_PMC_PHOTO_ONLY=1
pass() {  # name, counters...
  name=$1; shift
  ( cd /tmp && rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $root/gpurun_out/pmc_$tag/$name -- python $root/scripts/pmc_workload.py > $root/gpurun_out/pmc_$tag/$name.log 2>&1 )
  f=$(find gpurun_out/pmc_$tag/$name -name '*counter_collection.csv' | head -1)
  python scripts/pmc_summary.py "$f" gpurun_out/pmc_$tag/$name.csv photo_tile > /dev/null
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass b SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
pass c SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
cat gpurun_out/pmc_$tag/a.csv gpurun_out/pmc_$tag/b.csv gpurun_out/pmc_$tag/c.csv | grep -v "^Kernel" | sed 's/void dd::photo_tile_kernel//' > gpurun_out/pmc_$tag/SQ_photo_tile_kernel.csv
cat gpurun_out/pmc_$tag/SQ_photo_tile_kernel.csv
