"""Per-kernel call count and average duration from a rocprofv3 kernel-trace CSV, skipping the first `skip` calls of each kernel.
usage: kernel_avg.py <kernel_trace.csv> [skip]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = defaultdict(list)
for r in rows:
    agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1][skip:] or [0])):
    w = v[skip:]
    if w:
        print("%6d calls  avg %8.2f us  %s" % (len(w), sum(w) / len(w) / 1e3, k[:110]))
