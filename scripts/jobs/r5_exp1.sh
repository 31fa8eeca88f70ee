set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5x1; mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=dynamo-depth_amd/csrc/variants
for rep in 1 2; do
for v in base prio_s1 prio_s2 prio_s3 prio_g1 prio_s1g1 prio_s2g1; do
  if [ $v = base ]; then lib=dynamo-depth_amd/hipops/libdynamo_hip.so; else lib=$V/$v.so; fi
  echo "== $v (rep $rep)" >> $out/photo_variants.txt
  DYNAMO_HIP_LIB=$PWD/$lib DD_PHASES=fine_tune,disp_init timeout 200 python scripts/time_photo.py 2>&1 | grep "us per call" >> $out/photo_variants.txt
done
done
echo "== timing build" >> $out/photo_variants.txt
DYNAMO_HIP_LIB=$PWD/$V/timing.so timeout 200 python scripts/time_photo.py >> $out/photo_variants.txt 2>&1
timeout 600 python -u -m pytest tests/test_conv_mfma_gpu.py -q -m gpu -p no:cacheprovider -x > $out/pytest_conv.log 2>&1; tail -3 $out/pytest_conv.log
timeout 500 python scripts/time_conv_mfma.py > $out/time_conv_mfma.txt 2>&1; tail -5 $out/time_conv_mfma.txt
cat $out/photo_variants.txt
