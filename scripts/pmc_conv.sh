#!/bin/bash
# SQ counter passes over dd_conv3x3_mfma's kernels (run on the GPU box):  bash scripts/pmc_conv.sh <tag>
# DD_PMC_WORKLOAD=scripts/pmc_conv_half_workload.py DD_PMC_NAME=conv_half: the same passes over dd_conv3x3_half's kernels (round 6)
set -u
tag=${1:-r05}
cd "$(dirname "$0")/.." || exit 1
root=$PWD
name_=${DD_PMC_NAME:-conv_mfma}
out=$root/gpurun_out/pmc_${name_}_$tag
mkdir -p $out
export TMPDIR=/tmp
pass() {  # name, counters...
  name=$1; shift
  ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python $root/${DD_PMC_WORKLOAD:-scripts/pmc_conv_workload.py} > $out/$name.log 2>&1 )
  f=$(find $out/$name -name '*counter_collection.csv' | head -1)
  python scripts/pmc_summary.py "$f" $out/$name.csv conv_ > /dev/null
  rm -rf $out/$name
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass b SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
pass c SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES
pass d GRBM_GUI_ACTIVE SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_WAVE32_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_EXP_GDS
cat $out/a.csv $out/b.csv $out/c.csv $out/d.csv | grep -v "^Kernel" | sed 's/void dd::cm:://; s/void dd::ch:://; s/(float const.*)",/",/; s/(unsigned short const.*)",/",/' > $out/SQ_${name_}.csv
cat $out/SQ_${name_}.csv
