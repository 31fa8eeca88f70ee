set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_conv_mfma_gpu.py -q -m gpu 2>&1 | tail -2 > gpurun_out/r6n.txt
for v in base cm_swz0 cm_swz1_twoset; do
  if [ $v = base ]; then unset DYNAMO_HIP_LIB; else export DYNAMO_HIP_LIB=$PWD/dynamo-depth_amd/csrc/variants/$v.so; fi
  echo "== $v" >> gpurun_out/r6n.txt
  timeout 400 python scripts/time_conv_mfma.py 6 2>&1 | grep -E "^\(|B,cin" | cut -c1-200 >> gpurun_out/r6n.txt
  timeout 300 python scripts/time_conv_flat.py 2>&1 | tail -8 | cut -c1-200 >> gpurun_out/r6n.txt
done
unset DYNAMO_HIP_LIB
cat gpurun_out/r6n.txt
for rep in 1 2; do
for v in base cm_swz0; do
  if [ $v = base ]; then unset DYNAMO_HIP_LIB; else export DYNAMO_HIP_LIB=$PWD/dynamo-depth_amd/csrc/variants/$v.so; fi
  bash scripts/gpu_job.sh r6n bench --no_cpu_baseline --mode graph
done
done
