set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5x9; mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -u -m pytest tests/test_photo_gpu.py tests/test_photo_edge_gpu.py tests/test_fused_loss_gpu.py tests/test_abi.py -q -m gpu -p no:cacheprovider -x > $out/pytest.log 2>&1; tail -4 $out/pytest.log
for rep in 1 2; do
for pk in 0 1; do
  echo "== DD_PACKED=$pk (rep $rep)" >> $out/variants.txt
  DD_PACKED=$pk timeout 300 python scripts/time_photo.py 2>&1 | grep "grad=" >> $out/variants.txt
done
done
cat $out/variants.txt
bash scripts/gpu_job.sh r5x9 bench
DD_PACK_SOURCES=0 bash scripts/gpu_job.sh r5x9 bench
bash scripts/gpu_job.sh r5x9 bench
