set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5x5; mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for rep in 1 2; do
for v in base dephase8 dephase3 dephase0 priowarp1 priowarp3; do
  if [ $v = base ]; then unset DYNAMO_HIP_LIB; else export DYNAMO_HIP_LIB=$PWD/dynamo-depth_amd/csrc/variants/$v.so; fi
  echo "== $v (rep $rep)" >> $out/variants.txt
  DD_PHASES=fine_tune,disp_init timeout 300 python scripts/time_photo.py 2>&1 | grep "grad=1" | grep -v "fine_tune    grad=1 shared=0" >> $out/variants.txt
done
done
cat $out/variants.txt
