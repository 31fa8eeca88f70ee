set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for rep in 1 2; do
for v in base ph_scale_inner; do
  if [ $v = base ]; then unset DYNAMO_HIP_LIB; else export DYNAMO_HIP_LIB=$PWD/dynamo-depth_amd/csrc/variants/$v.so; fi
  echo "== $v (rep $rep)" >> gpurun_out/r6l_variants.txt
  timeout 300 python scripts/time_photo.py 2>&1 | grep "grad=" >> gpurun_out/r6l_variants.txt
  timeout 300 python scripts/loss_path_workload.py fine_tune 12 30 2>&1 | tail -2 >> gpurun_out/r6l_variants.txt
done
done
export DYNAMO_HIP_LIB=$PWD/dynamo-depth_amd/csrc/variants/ph_scale_inner.so
timeout 900 python -m pytest tests/test_photo_gpu.py tests/test_fused_loss_gpu.py tests/test_fused5_gpu.py tests/test_photo_edge_gpu.py -q -m gpu 2>&1 | tail -3 >> gpurun_out/r6l_variants.txt
cat gpurun_out/r6l_variants.txt
