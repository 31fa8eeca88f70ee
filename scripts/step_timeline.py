"""One steady-state training step out of a rocprofv3 --kernel-trace CSV, as a compact per-stream timeline.
usage: step_timeline.py <kernel_trace.csv> <out.txt> [skip]   (skip: trailing tile-kernel launches that are bench.py's loss evaluations)
Writes one line per dispatch of one steady-state step (tile-kernel launch to tile-kernel launch): stream/queue id, start (us from the step's first
kernel), duration (us), short kernel name; then per stream a summary by kind (wrw / fwd / bwd-data / gemm / zero / dd / aten) with
busy time, span and gaps."""
import csv
import re
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = list(csv.DictReader(open(src)))
nk = "Kernel_Name" if "Kernel_Name" in rows[0] else [k for k in rows[0] if "ame" in k][0]
sk = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tiles = [i for i, r in enumerate(rows) if "photo_tile_kernel" in r[nk]]
if skip:
    tiles = tiles[:-skip]
# one step's worth of dispatches: from behind one step's tile kernel to the next step's tile kernel (a step in the trace's own order;
# rocprofv3 runs one kernel at a time, so durations are alone-times and the order is the dispatch order)
sel = rows[tiles[-3] + 1:tiles[-2] + 1]
t0 = int(sel[0]["Start_Timestamp"])


def kind(n):
    if "igemm_wrw" in n or "wrw" in n.lower():
        return "conv wrw"
    if "igemm_bwd" in n:
        return "conv bwd-data"
    if "igemm_fwd" in n or "ck16tensor" in n or "naive_conv" in n:
        return "conv fwd"
    if n.startswith("Cijk"):
        return "gemm"
    if "SubTensorOp" in n or "FillFunctor" in n or "fillBuffer" in n:
        return "zero/fill"
    if "dd::" in n:
        return "dd"
    return "aten/other"


def short(n):
    n = re.sub(r"at::native::|\(anonymous namespace\)::|void ", "", n)
    return n[:90]


per = defaultdict(list)
with open(dst, "w") as fh:
    fh.write("# %d dispatches, wall %.3f ms\n" % (len(sel), (int(sel[-1]["End_Timestamp"]) - t0) / 1e6))
    for r in sel:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        per[r[sk]].append((s, e, r[nk]))
        fh.write("%s %10.1f %8.1f %s\n" % (r[sk], s / 1e3, (e - s) / 1e3, short(r[nk])))
    fh.write("\n# per stream\n")
    for st, evs in sorted(per.items(), key=lambda kv: kv[1][0][0]):
        busy = sum(e - s for s, e, _ in evs)
        by = defaultdict(lambda: [0, 0])
        for s, e, n in evs:
            k = kind(n)
            by[k][0] += 1
            by[k][1] += e - s
        fh.write("# stream %s: %d kernels, first %.2f ms, last end %.2f ms, busy %.2f ms : %s\n" % (
            st, len(evs), evs[0][0] / 1e6, max(e for _, e, _ in evs) / 1e6, busy / 1e6,
            ", ".join("%s %d/%.2f" % (k, c, d / 1e6) for k, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1]))))
print(open(dst).read().split("# per stream")[1])
