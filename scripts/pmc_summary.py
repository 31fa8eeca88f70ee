"""Mean PMC counter value per kernel from a rocprofv3 --pmc ... --output-format csv run.
usage: pmc_summary.py <counter_collection.csv> <out.csv> [name-filter]"""
import csv
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows = list(csv.DictReader(open(src)))
cols = rows[0].keys()
kn = "Kernel_Name" if "Kernel_Name" in cols else [c for c in cols if "ernel" in c and "ame" in c][0]
cn = "Counter_Name" if "Counter_Name" in cols else [c for c in cols if "ounter" in c and "ame" in c][0]
cv = "Counter_Value" if "Counter_Value" in cols else [c for c in cols if "alue" in c][0]
agg = defaultdict(lambda: [0, 0.0])
for r in rows:
    if flt and flt not in r[kn]:
        continue
    a = agg[(r[kn][:120], r[cn])]
    a[0] += 1
    a[1] += float(r[cv])
with open(dst, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Kernel", "Counter", "Dispatches", "MeanValue"])
    for (k, c), (n, tot) in sorted(agg.items()):
        w.writerow([k, c, n, "%.3f" % (tot / n)])
        print(k[:70], c, n, "%.3f" % (tot / n))
