"""How far is the fp32 oracle from itself in fp64?  (the floor under every gradient tolerance of tests/photo_case.py)

usage: python scripts/probe_oracle_fp64.py [phase B H W]      -- CPU only
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests"), ROOT, os.path.join(ROOT, "dynamo-depth_amd")):
    sys.path.insert(0, p)
import photo_case as pc  # noqa: E402
import oracle.ref_loss as orc  # noqa: E402

phase = sys.argv[1] if len(sys.argv) > 1 else "disp_init"
B, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (1, 288, 512)
scales = [0, 1, 2, 3]
case = pc.Case(phase, B, H, W, scales, seed=13).run_oracle()
g32 = {f: case.outputs[("cam_T_cam", 0, f)].grad.clone() for f in (-1, 1) if case.outputs[("cam_T_cam", 0, f)].grad is not None}
gd32 = {k: v.grad.clone() for k, v in case.leaves.items() if v.grad is not None}
pg = orc.pixel_grid
orc.pixel_grid = lambda *a, **k: pg(*a, **k).double()
case.inputs = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in case.inputs.items()}
case.leaves = {k: v.detach().double().requires_grad_() for k, v in case.leaves.items()}
if case.noise is not None:
    case.noise = {s: v.double() for s, v in case.noise.items()}
case.run_oracle()
for f in g32:
    g64 = case.outputs[("cam_T_cam", 0, f)].grad
    print("T[%d]   rel-L2 fp32 oracle vs fp64 oracle: %.3e" % (f, float((g32[f].double() - g64).norm() / g64.norm())))
for k, v in case.leaves.items():
    if v.grad is not None and k in gd32 and float(v.grad.norm()) > 0:
        print("%-18s rel-L2 %.3e" % (k, float((gd32[k].double() - v.grad).norm() / v.grad.norm())))
