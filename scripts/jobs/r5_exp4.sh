set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5x4; mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -u -m pytest tests/test_pw_gemm_gpu.py -q -m gpu -p no:cacheprovider -s > $out/pytest_pw.log 2>&1; grep -E "fused forward|passed|failed|Error" $out/pytest_pw.log | tail -12
timeout 900 python scripts/time_mlp.py > $out/time_mlp.txt 2>&1; head -9 $out/time_mlp.txt
