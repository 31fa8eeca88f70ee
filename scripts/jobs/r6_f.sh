set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in ch_wg4_f1 ch_wg3_f1; do
  export DYNAMO_HIP_LIB=$PWD/dynamo-depth_amd/csrc/variants/$v.so
  echo "== $v" >> gpurun_out/r6f_variants.txt
  timeout 300 python scripts/time_half_convs.py 2>&1 | grep "(" >> gpurun_out/r6f_variants.txt
done
unset DYNAMO_HIP_LIB
cat gpurun_out/r6f_variants.txt | cut -c1-140
for v in 1 0 1 0; do
  DD_HALF_MFMA_CONV=$v bash scripts/gpu_job.sh r6f bench --no_cpu_baseline --mode graph --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp fp16
done
for v in 1 0; do
  DD_HALF_MFMA_CONV=$v bash scripts/gpu_job.sh r6f bench --no_cpu_baseline --mode graph --amp fp16
done
