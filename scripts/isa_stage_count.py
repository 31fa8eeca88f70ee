"""Static instruction counts of dd::photo_tile_kernel per stage, from the gfx950 ISA (no GPU needed).

Compiles csrc/dd_photo.hip with -DDD_ISA_MARKS (comment markers between the stages), extracts one instantiation and
counts VALU / LDS / global instructions between the markers.  Each marker carries the fraction of the workgroup's waves
that execute the stage (the halo ring runs on 4 of 8 waves, the halo centres on 2), so the weighted VALU total is a
per-pixel-scale estimate of the dynamic count.  `cost` weighs the instructions by their VALU pipe time in units of one
plain fp32 instruction (4 cycles per wave64): packed fp32 (v_pk_*) 2, quarter-rate (v_rcp/v_mul_lo_u32/v_mul_hi/...) 4 --
the kernel's run time tracks this sum (measured: DESIGN.md section 6).
Loops whose trip count is a run-time value (the two up-sampling-adjoint passes) are counted once.

usage: python scripts/isa_stage_count.py [ILi2ELb0ELb1ELb1E] [extra hipcc flags...]
"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "dynamo-depth_amd", "csrc", "dd_photo.hip")


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else "ILi2ELb0ELb1ELb1ELb0E"
    extra = sys.argv[2:]
    out = "/tmp/dd_photo_marks.s"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-DDD_ISA_MARKS", "-S",
           "--cuda-device-only", SRC, "-o", out] + extra
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN2dd17photo_tile_kernel" + pat) and l.rstrip().endswith(("E:", "E: ")) or
                 (l.startswith("_ZN2dd17photo_tile_kernel" + pat) and ":" in l))
    stages, cur = [], ["prologue", 1.0, 0, 0, 0, 0, 0, 0]
    for l in lines[start + 1:]:
        t = l.strip()
        m = re.match(r"; DDMARK (\S+) (\S+)", t)
        if m:
            stages.append(cur)
            cur = [m.group(1), float(m.group(2)), 0, 0, 0, 0, 0, 0]
            continue
        if t.startswith("s_endpgm"):
            break
        op = t.split(" ")[0].split("\t")[0]
        if op.startswith("v_"):
            cur[2] += 1
            if op.startswith("v_pk_"):
                cur[5] += 1
            if op.startswith("v_mov") or op.startswith("v_accvgpr"):
                cur[6] += 1
            if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_mul_lo_u32", "v_mul_hi", "v_mul_lo_i32", "v_mad_u64")):
                cur[7] += 1
        elif op.startswith("ds_"):
            cur[3] += 1
        elif op.startswith(("global_", "buffer_", "scratch_")):
            cur[4] += 1
    stages.append(cur)
    print("%-12s %5s %6s %5s %5s %5s %5s %5s %6s" % ("stage", "frac", "VALU", "pk", "slow", "mov", "LDS", "mem", "cost"))
    tot = cost_tot = 0.0
    for name, w, valu, lds, mem, pk, mov, slow in stages:
        cost = valu + pk + 3 * slow
        print("%-12s %5.2f %6d %5d %5d %5d %5d %5d %6d" % (name, w, valu, pk, slow, mov, lds, mem, cost))
        tot += w * valu
        cost_tot += w * cost
    print("weighted VALU per pixel-scale ~ %.0f   weighted cost ~ %.0f" % (tot, cost_tot))
    meta = [l for l in lines if "photo_tile_kernel" + pat in l and ".name:" in l]
    idx = lines.index(meta[0]) if meta else None
    if idx:
        for l in lines[idx:idx + 12]:
            if "vgpr_count" in l or "spill" in l or "private_segment" in l:
                print(l.strip())


if __name__ == "__main__":
    main()
