set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
bash scripts/pmc_photo.sh r06 > gpurun_out/r6q_pmc.log 2>&1
tail -30 gpurun_out/r6q_pmc.log | cut -c1-170
DD_B=96 DD_PHASES=fine_tune timeout 300 python scripts/time_photo.py 2>&1 | grep "grad=" > gpurun_out/r6q_B96.txt
cat gpurun_out/r6q_B96.txt
