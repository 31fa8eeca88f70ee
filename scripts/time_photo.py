"""Quick device timing of dd_photo_loss, stand-alone (DD_B / DD_H / DD_W / DD_SCALES / DD_PHASES / DD_SMOOTH; default: the KITTI bench shape B=12, 192x640, 3 scales)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests"), ROOT, os.path.join(ROOT, "dynamo-depth_amd")):
    sys.path.insert(0, p)
import photo_case as pc  # noqa: E402
from hipops import lib as L  # noqa: E402

B = int(os.environ.get("DD_B", 12))
H, W = int(os.environ.get("DD_H", 192)), int(os.environ.get("DD_W", 640))
SCALES = [int(x) for x in os.environ.get("DD_SCALES", "0,1,2").split(",")]
for phase in os.environ.get("DD_PHASES", "disp_init,motion_init,fine_tune").split(","):
    case = pc.Case(phase, B, H, W, SCALES, seed=1)
    if os.environ.get("DD_SMOOTH", "1") == "1":      # (default since round 5: white-noise flow fields scatter the source taps and double the kernel time)
        # network-like outputs: low-frequency disparity / flow / mask instead of per-pixel white noise
        import torch.nn.functional as F
        for (kind, s), v in list(case.leaves.items()):
            if kind in ("disp", "flow", "prob"):
                coarse = F.avg_pool2d(v.detach(), 8, 8, ceil_mode=True) if v.shape[-1] >= 16 else v.detach()
                sm = F.interpolate(coarse, v.shape[-2:], mode="bilinear", align_corners=False)
                case.leaves[(kind, s)] = (sm * (0.2 if kind == "flow" else 1.0)).requires_grad_()
    case.outputs = pc.synth.leaves_to_outputs(case.leaves, case.scales, pc.orc.pose_matrix, case.cmpflow, case.motmask)
    for want_grad, shared in ((True, True), (True, False), (False, True)):
        if shared and case.mode == 0:
            continue
        args, t = case.photo_buffers("cuda", materialise=False, want_grad=want_grad, shared=shared, packed=os.environ.get("DD_PACKED", "1") == "1")
        lib = L.load()
        st = L.current_stream()
        for _ in range(5):
            lib.dd_photo_loss(C.byref(args), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            lib.dd_photo_loss(C.byref(args), st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / n
        if hasattr(lib, "dd_debug_stage_cycles"):
            cyc = (C.c_ulonglong * 8)()
            lib.dd_debug_stage_cycles(cyc, 1)            # reset
            lib.dd_photo_loss(C.byref(args), st)
            torch.cuda.synchronize()
            lib.dd_debug_stage_cycles(cyc, 1)
            tot = float(sum(cyc)) or 1.0
            names = ["0:stage+target", "1:identity", "A:warp", "B+L:ssim/select", "C:gather+chain", "C2:upsample adjoint", "R:reduce", "-"]
            print("   stages: " + "  ".join("%s %.1f%%" % (nm, 100.0 * c / tot) for nm, c in zip(names, cyc) if c))
        print("%-12s grad=%d shared=%d B=%d %dx%d S=%d  %.1f us per call (photo tile kernel + combine + finalize)" % (phase, want_grad, shared, B, H, W, len(SCALES), us))
