#!/bin/bash
# HBM traffic of dd::adam_multi_kernel from the PMC counters, separate --pmc passes as MI355X_MICROARCH.md prescribes, calibrated in the
# same run on an element-wise atan of known traffic (256 MiB read + 256 MiB written):   bash scripts/pmc_adam.sh <tag>
set -u
tag=${1:-r04}
cd "$(dirname "$0")/.." || exit 1
root=$PWD
out=$root/gpurun_out/pmc_adam_$tag
mkdir -p $out
export TMPDIR=/tmp DD_PMC=1
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -- python $root/scripts/time_adam.py > $out/$c.log 2>&1 ) < /dev/null
  f=$(find $out/$c -name '*counter_collection.csv' | head -1)
  python scripts/pmc_summary.py "$f" $out/${tag}_pmc_adam_$c.csv > /dev/null
  rm -rf $out/$c
done
python - <<PY | tee $out/${tag}_adam_traffic.txt
import csv
def load(p):
    return {r["Kernel"]: (float(r["MeanValue"]), int(r["Dispatches"])) for r in csv.DictReader(open(p))}
F, W = load("$out/${tag}_pmc_adam_FETCH_SIZE.csv"), load("$out/${tag}_pmc_adam_WRITE_SIZE.csv")
cal = [k for k in F if "atan" in k]
ad = [k for k in F if "adam_multi_kernel" in k]
known = 256 * 1024.0          # KiB read and KiB written by the calibration launch
for k in cal[:1]:
    cf, cw = known / F[k][0], known / W[k][0]
    print("calibration %-60s FETCH_SIZE %.1f KiB -> x%.3f   WRITE_SIZE %.1f KiB -> x%.3f" % (k[:60], F[k][0], cf, W[k][0], cw))
for k in ad:
    f, w = F[k][0] * cf * 1024 / 1e6, W[k][0] * cw * 1024 / 1e6
    print("%-60s dispatches %d  fetched %.1f MB  written %.1f MB  -> %.1f MB per launch" % (k[:60], F[k][1], f, w, f + w))
PY
