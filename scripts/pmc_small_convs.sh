#!/bin/bash
# HBM traffic of the dd_conv_small / dd_conv_head kernels from the PMC counters (separate --pmc passes, calibrated in the same run on an
# element-wise atan of known traffic):   bash scripts/pmc_small_convs.sh <tag>
set -u
tag=${1:-r04}
cd "$(dirname "$0")/.." || exit 1
root=$PWD
out=$root/gpurun_out/pmc_convs_$tag
mkdir -p $out
export TMPDIR=/tmp DD_PMC=1
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -- python $root/scripts/time_small_convs.py > $out/$c.log 2>&1 ) < /dev/null
  f=$(find $out/$c -name '*counter_collection.csv' | head -1)
  python scripts/pmc_summary.py "$f" $out/${tag}_pmc_convs_$c.csv > /dev/null
  rm -rf $out/$c
done
python - <<PY | tee $out/${tag}_small_conv_traffic.txt
import csv
def load(p):
    return {r["Kernel"]: (float(r["MeanValue"]), int(r["Dispatches"])) for r in csv.DictReader(open(p))}
F, W = load("$out/${tag}_pmc_convs_FETCH_SIZE.csv"), load("$out/${tag}_pmc_convs_WRITE_SIZE.csv")
cal = [k for k in F if "atan" in k][0]
cf, cw = 256 * 1024.0 / F[cal][0], 256 * 1024.0 / W[cal][0]
print("calibration: FETCH_SIZE x%.3f  WRITE_SIZE x%.3f   (mean over a kernel's dispatches in the run: instantiations used at several shapes average over them)" % (cf, cw))
for k in sorted(F):
    if "dd::conv_small" in k or "dd::conv_head" in k:
        f, w = F[k][0] * cf * 1024 / 1e6, W.get(k, (0, 0))[0] * cw * 1024 / 1e6
        print("%-92s dispatches %3d  fetched %7.1f MB  written %7.1f MB" % (k[:92], F[k][1], f, w))
PY
