#!/bin/bash
# What does each task kind of dd::reg_stage_kernel cost?  Builds a variant of the library whose dd_reg_losses can skip task kinds
# (-DDD_REG_DEBUG_SKIP, mask in DD_REG_SKIP: bit k = task kind k of the enum in csrc/dd_reg.hip) and times the whole fused loss with
# one kind removed at a time (the results are wrong then -- timing only).  GPU box:  bash scripts/reg_task_costs.sh
cd "$(dirname "$0")/.."
cd dynamo-depth_amd/csrc && mkdir -p variants && make -s >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DDD_REG_DEBUG_SKIP -c dd_reg.hip -o variants/regskip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls dd_*.o | grep -v dd_reg.o) variants/regskip.o -o variants/regskip.so && rm variants/regskip.o
cd ../..
names=(MEAN SPCOUNT GCAND SMOOTHALL SPGRAD GSCORE SMFOLD GCOUNT DISPFIN GFOLD)
echo "none skipped: $(DYNAMO_HIP_LIB=$PWD/dynamo-depth_amd/csrc/variants/regskip.so python scripts/loss_path_workload.py fine_tune 12 50 | tail -1)"
for k in 0 1 2 3 4 5 6 7 8; do
  echo "without ${names[$k]}: $(DD_REG_SKIP=$((1<<k)) DYNAMO_HIP_LIB=$PWD/dynamo-depth_amd/csrc/variants/regskip.so python scripts/loss_path_workload.py fine_tune 12 50 | tail -1)"
done
echo "all reg tasks skipped: $(DD_REG_SKIP=1023 DYNAMO_HIP_LIB=$PWD/dynamo-depth_amd/csrc/variants/regskip.so python scripts/loss_path_workload.py fine_tune 12 50 | tail -1)"
