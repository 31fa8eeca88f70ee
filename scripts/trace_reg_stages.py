"""Durations of the four dd::reg_stage_kernel launches of a step, in launch order, from a rocprofv3 kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python bench.py --no_cpu_baseline --mode eager --steps 4 --warmup 3
   python scripts/trace_reg_stages.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
name = "Kernel_Name" if "Kernel_Name" in rows[0] else [k for k in rows[0] if "ame" in k][0]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "reg_stage_kernel" in r[name] or "assemble_kernel" in r[name] or "photo_" in r[name]]
for r in sel[-28:]:
    gx = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
    print("%-40s grid=%-8s %8.1f us" % (r[name][:40], gx, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
