"""Workload for the PMC passes: a known-size device copy (calibrates FETCH_SIZE / WRITE_SIZE on this stack) followed by
dd_photo_loss launches at the bench shape (B=12, 192x640, 3 scales, fine_tune and disp_init, network-like inputs)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests"), ROOT, os.path.join(ROOT, "dynamo-depth_amd")):
    sys.path.insert(0, p)
os.environ["DD_SMOOTH"] = "1"
import photo_case as pc  # noqa: E402
from hipops import lib as L  # noqa: E402
import torch.nn.functional as F  # noqa: E402

PHOTO_ONLY = os.environ.get("DD_PMC_PHOTO_ONLY") == "1"    # skip the traffic calibrations (SQ counter passes)
src = torch.rand((1 if PHOTO_ONLY else 64) * 1024 * 1024, device="cuda")        # 256 MiB
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)                                        # calibration: 256 MiB read + 256 MiB written per launch
torch.cuda.synchronize()
# second calibration in the photometric kernel's own access pattern (one dword per lane, coalesced): dd_disp_to_depth reads
# 64 Mi floats (256 MiB) and writes two planes (512 MiB) per launch
from hipops.functions import _p  # noqa: E402
dsp = torch.rand((1 if PHOTO_ONLY else 64) * 1024 * 1024, device="cuda")
o1, o2 = torch.empty_like(dsp), torch.empty_like(dsp)
for _ in range(3):
    L.check(L.load().dd_disp_to_depth(_p(dsp), dsp.numel(), 0.1, 100.0, _p(o1), _p(o2), L.current_stream()), "dd_disp_to_depth")
torch.cuda.synchronize()
del dsp, o1, o2
for phase in ("fine_tune", "disp_init"):
    case = pc.Case(phase, 12, 192, 640, [0, 1, 2], seed=1)
    for (kind, s), v in list(case.leaves.items()):
        if kind in ("disp", "flow", "prob"):
            coarse = F.avg_pool2d(v.detach(), 8, 8, ceil_mode=True) if v.shape[-1] >= 16 else v.detach()
            case.leaves[(kind, s)] = (F.interpolate(coarse, v.shape[-2:], mode="bilinear", align_corners=False) * (0.2 if kind == "flow" else 1.0)).requires_grad_()
    case.outputs = pc.synth.leaves_to_outputs(case.leaves, case.scales, pc.orc.pose_matrix, case.cmpflow, case.motmask)
    # what the trainer hands over: one flow field / one mask tensor for both frames (the kernel's shared-tensor path)
    args, t = case.photo_buffers("cuda", materialise=False, want_grad=True, shared=True)
    for _ in range(5):
        L.load().dd_photo_loss(C.byref(args), L.current_stream())
    torch.cuda.synchronize()
print("done")
