"""How busy is the GPU during the steady-state steps of a kernel trace?  Union of all kernel intervals (any stream) over the last
`steps` steps (one dd::photo_tile_kernel launch marks a step), the time with >= 2 kernels in flight, and the idle remainder.
usage: gpu_busy.py <kernel_trace.csv> <steps>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
name = "Kernel_Name" if "Kernel_Name" in rows[0] else [k for k in rows[0] if "ame" in k][0]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "photo_tile_kernel" in r[name]]
lo, hi = marks[-steps - 1], marks[-1]
sel = rows[lo + 1:hi + 1]
ev = []
for r in sel:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
depth, last, busy, multi, hist = 0, t0, 0, 0, {}
for t, d in ev:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        multi += t - last
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth += d
    last = t
wall = t1 - t0
ksum = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
print("steps %d  wall %.2f ms/step  busy (>=1 kernel) %.2f ms/step (%.1f %%)  >=2 kernels %.2f ms/step  idle %.2f ms/step  sum of kernel time %.2f ms/step"
      % (steps, wall / steps / 1e6, busy / steps / 1e6, 100.0 * busy / wall, multi / steps / 1e6, (wall - busy) / steps / 1e6, ksum / steps / 1e6))
print("time by number of kernels in flight:", {k: round(v / steps / 1e6, 2) for k, v in sorted(hist.items())})
# the longest idle gaps
gaps = []
depth, last = 0, t0
for t, d in ev:
    if depth == 0 and t > last:
        gaps.append(t - last)
    depth += d
    last = t
gaps.sort(reverse=True)
print("idle gaps: %d, top %s us; gaps < 10 us: %.2f ms/step" % (len(gaps), [round(g / 1e3) for g in gaps[:8]], sum(g for g in gaps if g < 10000) / steps / 1e6))
