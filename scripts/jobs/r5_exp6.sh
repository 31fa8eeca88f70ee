set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5x6; mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for rep in 1 2; do
for v in base ringold; do
  if [ $v = base ]; then unset DYNAMO_HIP_LIB; else export DYNAMO_HIP_LIB=$PWD/dynamo-depth_amd/csrc/variants/$v.so; fi
  echo "== $v (rep $rep)" >> $out/variants.txt
  timeout 300 python scripts/time_photo.py 2>&1 | grep "grad=" >> $out/variants.txt
done
done
cat $out/variants.txt
unset DYNAMO_HIP_LIB
timeout 900 python -u -m pytest tests/test_photo_gpu.py tests/test_photo_edge_gpu.py tests/test_fused_loss_gpu.py -q -m gpu -p no:cacheprovider -x > $out/pytest.log 2>&1; tail -5 $out/pytest.log
