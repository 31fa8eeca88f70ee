"""Static instruction counts of dd::photo_tile_kernel per stage, from the gfx950 ISA (no GPU needed).

Compiles csrc/dd_photo.hip with -DDD_ISA_MARKS (comment markers between the stages), extracts one instantiation and
counts VALU / LDS / global instructions between the markers.  Each marker carries the fraction of the workgroup's waves
that execute the stage (the halo ring runs on 4 of 8 waves, the halo centres on 2), so the weighted VALU total is a
per-pixel-scale estimate of the dynamic count.  `cost` weighs the instructions by their VALU pipe time in units of one
plain fp32 instruction (4 cycles per wave64): packed fp32 (v_pk_*) 2, quarter-rate (v_rcp/v_mul_lo_u32/v_mul_hi/...) 4 --
the kernel's run time tracks this sum (measured: DESIGN.md section 6).
Loops whose trip count is a run-time value (the two up-sampling-adjoint passes) are counted once.

With DD_ISA_CLASSES=1 a second table splits every stage's VALU instructions into classes: floating-point arithmetic (packed and
scalar), compare / select, integer and address arithmetic, moves and cross-lane traffic, conversions -- what is and is not the
algorithm's own arithmetic ("which of the instructions are irreducible, with the ISA listing").

usage: python scripts/isa_stage_count.py [ILi2ELb0ELb1ELb1E] [extra hipcc flags...]
"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "dynamo-depth_amd", "csrc", "dd_photo.hip")


def classify(op):
    op = re.sub(r"_e32$|_e64$|_dpp$|_sdwa$", "", op)
    if op.startswith("v_pk_"):
        return "fp packed"
    if op.startswith(("v_cmp", "v_cndmask")):
        return "compare/select"
    if op.startswith(("v_mov", "v_readlane", "v_writelane", "v_readfirstlane", "v_accvgpr", "v_swap", "v_perm", "v_alignbit")):
        return "move/lane"
    if op.startswith("v_cvt") or op.startswith(("v_floor", "v_trunc", "v_rndne", "v_fract", "v_ceil")):
        return "convert/round"
    if re.match(r"v_(add|sub|subrev|mul|fma|fmac|mac|mad|max|min|med3|rcp|rsq|sqrt|exp|log|div_scale|div_fmas|div_fixup|ldexp)_(f32|legacy_f32)$", op) or op in ("v_rcp_iflag_f32",):
        return "fp scalar"
    return "integer/address"


CLASSES = ["fp packed", "fp scalar", "compare/select", "integer/address", "move/lane", "convert/round"]


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
    classes, ccur = [], dict.fromkeys(CLASSES, 0)
    for l in lines[start + 1:]:
        t = l.strip()
        m = re.match(r"; DDMARK (\S+) (\S+)", t)
        if m:
            stages.append(cur)
            classes.append(ccur)
            cur = [m.group(1), float(m.group(2)), 0, 0, 0, 0, 0, 0]
            ccur = dict.fromkeys(CLASSES, 0)
            continue
        if t.startswith("s_endpgm"):
            break
        op = t.split(" ")[0].split("\t")[0]
        if op.startswith("v_"):
            cur[2] += 1
            ccur[classify(op)] += 1
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
    classes.append(ccur)
    print("%-12s %5s %6s %5s %5s %5s %5s %5s %6s" % ("stage", "frac", "VALU", "pk", "slow", "mov", "LDS", "mem", "cost"))
    tot = cost_tot = 0.0
    for name, w, valu, lds, mem, pk, mov, slow in stages:
        cost = valu + pk + 3 * slow
        print("%-12s %5.2f %6d %5d %5d %5d %5d %5d %6d" % (name, w, valu, pk, slow, mov, lds, mem, cost))
        tot += w * valu
        cost_tot += w * cost
    print("weighted VALU per pixel-scale ~ %.0f   weighted cost ~ %.0f" % (tot, cost_tot))
    if os.environ.get("DD_ISA_CLASSES") == "1":
        print()
        print("%-12s %5s " % ("stage", "frac") + " ".join("%15s" % c for c in CLASSES))
        wsum = dict.fromkeys(CLASSES, 0.0)
        for (name, w, *_), cc in zip(stages, classes):
            print("%-12s %5.2f " % (name, w) + " ".join("%15d" % cc[c] for c in CLASSES))
            for c in CLASSES:
                wsum[c] += w * cc[c]
        print("%-12s %5s " % ("weighted", "") + " ".join("%15.0f" % wsum[c] for c in CLASSES))
    meta = [l for l in lines if "photo_tile_kernel" + pat in l and ".name:" in l]
    idx = lines.index(meta[0]) if meta else None
    if idx:
        for l in lines[idx:idx + 12]:
            if "vgpr_count" in l or "spill" in l or "private_segment" in l:
                print(l.strip())


if __name__ == "__main__":
    main()
