"""Per-kernel durations of the fused loss from a rocprofv3 kernel trace (any workload): the photo_* / reg_stage / finish / assemble
kernels of the last `steps` steps, in launch order for one step plus the per-step totals.
usage: loss_kernels.py <kernel_trace.csv> [steps]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
name = "Kernel_Name" if "Kernel_Name" in rows[0] else [k for k in rows[0] if "ame" in k][0]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keys = ("photo_tile_kernel", "photo_combine_kernel", "photo_finalize_kernel", "photo_identity_kernel", "reg_stage_kernel", "smooth_quad_kernel", "ground_score_all_kernel", "finish_kernel", "assemble_kernel",
        "fused_post_kernel")        # (fused_finish_kernel matches "finish_kernel")
sel = [r for r in rows if any(k in r[name] for k in keys)]
marks = [i for i, r in enumerate(sel) if "photo_tile_kernel" in r[name]]
marks = marks[-steps - 1:]
per_step = []
for a, b in zip(marks[:-1], marks[1:]):
    per_step.append([(r[name].split("(")[0][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))) for r in sel[a:b]])
n = len(per_step)
last = per_step[-1]
print("one evaluation (last of %d), launch order:" % n)
for j, (nm, us, grid) in enumerate(last):
    avg = sum(st[j][1] for st in per_step if len(st) == len(last)) / max(sum(1 for st in per_step if len(st) == len(last)), 1)
    print("  %-62s grid %-9s %8.1f us (avg %8.1f)" % (nm, grid, us, avg))
tot = [sum(u for _, u, _ in st) for st in per_step]
print("sum of loss-path kernel time per evaluation: avg %.1f us, min %.1f us, max %.1f us over %d evaluations" % (sum(tot) / n, min(tot), max(tot), n))
