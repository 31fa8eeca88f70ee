set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5x2; mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -u -m pytest tests/test_pw_gemm_gpu.py -q -m gpu -p no:cacheprovider -s > $out/pytest_pw.log 2>&1; tail -25 $out/pytest_pw.log
timeout 600 python scripts/time_mlp.py > $out/time_mlp.txt 2>&1; cat $out/time_mlp.txt
timeout 900 python -u -m pytest tests/test_trainer_gpu.py -q -m gpu -p no:cacheprovider -s -k "hooks_match_stock" > $out/pytest_hooks.log 2>&1; tail -12 $out/pytest_hooks.log | cut -c1-600
