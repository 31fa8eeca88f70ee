set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in base ch_deep_f1 ch_deep_f3 ch_wg4_f1; do
  if [ $v = base ]; then unset DYNAMO_HIP_LIB; else export DYNAMO_HIP_LIB=$PWD/dynamo-depth_amd/csrc/variants/$v.so; fi
  echo "== $v" >> gpurun_out/r6k_variants.txt
  timeout 300 python scripts/time_half_convs.py 2>&1 | grep "(" | cut -c1-95 >> gpurun_out/r6k_variants.txt
  timeout 300 python -m pytest tests/test_conv_half_gpu.py -q -x 2>&1 | tail -1 >> gpurun_out/r6k_variants.txt
done
cat gpurun_out/r6k_variants.txt
