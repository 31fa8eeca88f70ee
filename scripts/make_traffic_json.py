"""profiles/photo_traffic.json from the two PMC summaries (pmc_summary.py output): applies the calibration measured in the same
run (dd_disp_to_depth: 256 MiB read / 512 MiB written in the photometric kernel's one-dword-per-lane pattern).
usage: make_traffic_json.py <pmc_FETCH_SIZE.csv> <pmc_WRITE_SIZE.csv> <tag> <out.json>"""
import csv, json, sys


def load(path):
    return {r["Kernel"]: float(r["MeanValue"]) for r in csv.DictReader(open(path))}


F, W = load(sys.argv[1]), load(sys.argv[2])
tag, dst = sys.argv[3], sys.argv[4]
KB = 1024
find = lambda d, key: [v for k, v in d.items() if key in k][0]
cal_f, cal_w = find(F, "disp_to_depth"), find(W, "disp_to_depth")
fcorr, wcorr = 256 * 1024 / cal_f, 512 * 1024 / cal_w
out = {"collected": "round %s, MI355X," % tag[1:3].lstrip("0") + " rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over scripts/pmc_workload.py (profiles/%s_pmc_*.csv)" % tag,
       "unit": "bytes per dd_photo_loss launch (tile + combine + finalize kernels)",
       "corrections": {"FETCH_SIZE": "x%.3f (calibrated in the same run: dd_disp_to_depth reads 256 MiB with one dword per lane, counter shows %.0f KiB; 256 MiB wide copies show the same 1/2)" % (fcorr, cal_f),
                       "WRITE_SIZE": "x%.3f (512 MiB written, counter shows %.0f KiB)" % (wcorr, cal_w)},
       "workloads": {}}
for phase, tile, comb in (("fine_tune", "photo_tile_kernel<2, false, true", "photo_combine_kernel<5>"), ("disp_init", "photo_tile_kernel<0, true, true", "photo_combine_kernel<1>")):
    parts, tot = {}, 0.0
    for name, key in (("tile", tile), ("combine", comb), ("finalize", "photo_finalize")):
        f, w = find(F, key) * KB * fcorr, find(W, key) * KB * wcorr
        parts[name] = {"fetch": round(f), "write": round(w)}
        tot += f + w
    out["workloads"]["12x192x640 scales=3 phase=%s" % phase] = {"traffic_bytes_per_launch": round(tot), "kernels": parts}
    print(phase, round(tot / 1e6, 1), "MB", parts)
json.dump(out, open(dst, "w"), indent=1)
