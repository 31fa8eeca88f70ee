#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the regulariser kernels (reg_stage_kernel x4, smooth_quad_kernel, finish_kernel) of one loss evaluation at
# the bench shape, separate --pmc passes as MI355X_MICROARCH.md prescribes:   bash scripts/pmc_reg.sh <tag>
# -> gpurun_out/pmc_<tag>/<tag>_pmc_reg_{FETCH,WRITE}_SIZE.csv + <tag>_reg_traffic.txt (corrected with the calibration of photo_traffic.json's run)
set -u
tag=${1:-r04}
cd "$(dirname "$0")/.." || exit 1
root=$PWD
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp DD_HOST_ISSUED=1
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/reg_$c -- python $root/scripts/loss_path_workload.py fine_tune 12 6 > $out/reg_$c.log 2>&1 )
  f=$(find $out/reg_$c -name '*counter_collection.csv' | head -1)
  python scripts/pmc_summary.py "$f" $out/${tag}_pmc_reg_$c.csv > /dev/null
  rm -rf $out/reg_$c
done
python - <<PY > $out/${tag}_reg_traffic.txt
import csv
def load(p):
    return [(r["Kernel"], float(r["MeanValue"]), int(r["Dispatches"])) for r in csv.DictReader(open(p))]
F, W = load("$out/${tag}_pmc_reg_FETCH_SIZE.csv"), load("$out/${tag}_pmc_reg_WRITE_SIZE.csv")
# FETCH_SIZE counts in KiB and shows 1/2 of the bytes on gfx950 for this access pattern (calibrated in profiles/photo_traffic.json's run); WRITE_SIZE exact
print("kernel, mean over its dispatches of one run (reg_stage_kernel: the four stage launches averaged -- per-launch figures need the trace order)")
tot = 0.0
for (k, f, n), (_, w, _) in zip(F, W):
    if not any(x in k for x in ("reg_stage", "smooth_quad", "finish_kernel", "photo_")):
        continue
    mb = (2.0 * f + w) * 1024 / 1e6
    print("%-70s dispatches %4d  fetched %8.1f KiB x2  written %8.1f KiB  -> %7.2f MB per launch" % (k[:70], n, f, w, mb))
PY
cat $out/${tag}_reg_traffic.txt
