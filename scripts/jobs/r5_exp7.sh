set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5x7; mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in base taps1 taps3 taps2 nossim nossim_taps2; do
  if [ $v = base ]; then unset DYNAMO_HIP_LIB; else export DYNAMO_HIP_LIB=$PWD/dynamo-depth_amd/csrc/variants/$v.so; fi
  echo "== $v" >> $out/variants.txt
  DD_PHASES=fine_tune,disp_init timeout 300 python scripts/time_photo.py 2>&1 | grep "grad=" | grep -v "fine_tune    grad=1 shared=0" >> $out/variants.txt
done
cat $out/variants.txt
