"""The two populations of dd::photo_tile_kernel launches in a rocprofv3 kernel trace of `python bench.py` (round 4): the launches inside the
replayed timed steps run BESIDE the other streams' kernels (the statistics-only batch is still going when the loss starts: the kernel
shares the chip and takes longer), the host-issued evaluations directly behind the timed region run ALONE -- those are the ones bench.py
times for `roofline.avg_launch_us` (its `--stats` row averages both populations and the warm-up).
usage: tile_populations.py <kernel_trace.csv> <timed steps K> [<probe evaluations per leg, default max(K,10)+2> [<legs, default 3>]]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2])
P = int(sys.argv[3]) if len(sys.argv) > 3 else max(K, 10) + 2
name = "Kernel_Name" if "Kernel_Name" in rows[0] else [k for k in rows[0] if "ame" in k][0]
tile = sorted((r for r in rows if "photo_tile_kernel" in r[name] and "true, true, false" in r[name].replace("true, true, true, false", "true, true, false")),
              key=lambda r: int(r["Start_Timestamp"]))
tile = [r for r in tile if "<2, false, true" in r[name]] or tile
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tile]
# since round 6 bench.py issues THREE legs of P evaluations behind the timed region, in this order: the step's own network outputs
# (identity warps), SURVEY 8(d) synthetic outputs (low-frequency fields: what roofline.frac is quoted on), the same distributions per pixel
LEGS = int(sys.argv[4]) if len(sys.argv) > 4 else 3
legs = [dur[len(dur) - (LEGS - i) * P:len(dur) - (LEGS - i - 1) * P] for i in range(LEGS)]
step, before = dur[-LEGS * P - K:-LEGS * P], dur[:-LEGS * P - K]


def line(tag, d):
    if d:
        print("%-58s n %4d  avg %7.1f us  min %7.1f  max %7.1f" % (tag, len(d), sum(d) / len(d), min(d), max(d)))


print("dd::photo_tile_kernel<2,false,true,true,false> launches in launch order (%d in all):" % len(dur))
line("warm-up / auto-mode probe (eager and replayed, mixed)", before)
line("inside the %d timed replayed steps (other streams busy)" % K, step)
names = ["identity warps (the step's own network outputs)", "SURVEY 8(d) synthetic outputs, low-frequency  <- roofline.frac", "SURVEY 8(d) distributions per pixel (white noise)"]
for i, leg in enumerate(legs):
    line("alone, host-issued: " + (names[i] if LEGS == 3 else "leg %d" % i), leg[2:])
