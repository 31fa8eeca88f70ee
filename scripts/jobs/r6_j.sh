set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
DD_PMC_WORKLOAD=scripts/pmc_conv_half_workload.py DD_PMC_NAME=conv_half bash scripts/pmc_conv.sh r06 > gpurun_out/r6j_pmc.log 2>&1
tail -70 gpurun_out/r6j_pmc.log | cut -c1-160
