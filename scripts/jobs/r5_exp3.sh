set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5x3; mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python scripts/time_mlp.py > $out/time_mlp.txt 2>&1; cat $out/time_mlp.txt
