"""Aggregates a rocprofv3 --kernel-trace CSV over the LAST `steps` training steps only (one dd::photo_tile_kernel launch
marks one step), so that MIOpen's solver search during warm-up does not pollute the per-kernel summary.
usage: steady_state_stats.py <kernel_trace.csv> <steps> <out.csv> [skip]     (skip: trailing step markers to leave out -- bench.py's host-issued
loss evaluations behind the timed region each launch the tile kernel once more)"""
import csv
import sys
from collections import defaultdict

src, steps, dst = sys.argv[1], int(sys.argv[2]), sys.argv[3]
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rows = list(csv.DictReader(open(src)))
name_key = "Kernel_Name" if "Kernel_Name" in rows[0] else [k for k in rows[0] if "ame" in k][0]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "photo_tile_kernel" in r[name_key]]
if skip:
    marks = marks[:-skip]
if len(marks) <= steps:
    raise SystemExit("not enough steps in the trace: %d marks" % len(marks))
lo, hi = marks[-steps - 1], marks[-1]          # from one step marker to the last: exactly `steps` steps
sel = rows[lo + 1:hi + 1]
agg = defaultdict(lambda: [0, 0])
for r in sel:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg[r[name_key]]
    a[0] += 1
    a[1] += d
total = sum(a[1] for a in agg.values())
wall = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
with open(dst, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["# steady state: last %d steps, %d dispatches, wall %.3f ms/step, sum of kernel time %.3f ms/step" % (steps, len(sel), wall / steps / 1e6, total / steps / 1e6)])
    w.writerow(["Name", "CallsPerStep", "TotalNsPerStep", "AverageNs", "Percentage"])
    for name, (calls, dur) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([name[:160], "%.1f" % (calls / steps), "%.0f" % (dur / steps), "%.0f" % (dur / calls), "%.3f" % (100.0 * dur / total)])
print("steady-state rows:", len(agg), "dispatches/step:", len(sel) / steps, "kernel ms/step: %.3f" % (total / steps / 1e6))
