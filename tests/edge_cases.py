"""Adversarial loss-path cases (VERDICT r2 item 1 / SURVEY.md section 4 item 5, section 7 "exact PyTorch semantics"):
geometry and image content that the benign seeded cases of synth.py never produce.  Each builder returns a
photo_case.Case whose inputs / stand-in network outputs were overwritten before the oracle runs.

Reference lines these exercise:
  tools.py:216         z + 1e-7 is NOT clamped: points behind the camera project through the principal point
  Trainer.py:281       grid_sample(border, align_corners=True): samples outside the image, zero gradient on the clip
  tools.py:255-257     clamp((1 - n/d)/2, 0, 1) on flat patches: n == d, sub-gradient of the clamp on the closed interval
  Trainer.py:341-352   torch.min over [identity, identity, warped, warped]: exact ties go to the first entry
  tools.py:291-298     disp exactly 0 / 1 -> depth exactly max_depth / min_depth
"""
import torch

import photo_case as pc


def _set(case, key, value):
    with torch.no_grad():
        case.leaves[key].copy_(value)


def behind_camera(phase, B=2, H=64, W=96, scales=(0, 1, 2, 3), seed=21):
    """Large rotations and translations: depth = 1 / (0.01 + 9.99 disp) lies in [0.105, 1.96] (median ~0.2), so a translation of
    ~0.2 along the optical axis puts a good share of the points behind the source camera (z + eps < 0, tools.py:216 does not
    clamp) while the rest project far outside the image."""
    case = pc.Case(phase, B, H, W, list(scales), seed=seed)
    g = torch.Generator().manual_seed(seed + 5)
    for f, sign in ((-1, 1.0), (1, -1.0)):
        aa = 0.25 * torch.randn(B, 1, 3, generator=g)
        tr = torch.randn(B, 1, 3, generator=g) * torch.tensor([0.15, 0.1, 0.05]) + torch.tensor([0.0, 0.0, sign * 0.18])
        _set(case, ("axisangle", f), aa)
        _set(case, ("translation", f), tr)
    return case


def geometry_stats(case):
    """(fraction of transformed points with z + eps <= 0, fraction of samples outside [-1, 1]) of the oracle's outputs, frame -1, scale 0."""
    o = case.outputs
    grid = o[("sample", -1, 0)].detach()
    outside = ((grid.abs() > 1).any(-1)).float().mean().item()
    pts = o[("cam_points", 0, case.scales[-1])].detach()
    T = o[("cam_T_cam", 0, -1)].detach()
    if case.mode == 1:
        z = pts[:, 2]               # motion_init applies no rigid transform (Trainer.py:270-271)
    else:
        z = torch.matmul(T, pts)[:, 2]
    behind = (z + 1e-7 <= 0).float().mean().item()
    return behind, outside


def disp_extremes(phase, B=2, H=64, W=96, scales=(0, 1, 2, 3), seed=22):
    """Disparity exactly 0 and exactly 1 in blocks (sigmoid saturates to both in fp32), random elsewhere."""
    case = pc.Case(phase, B, H, W, list(scales), seed=seed)
    for s in case.scales:
        d = case.leaves[("disp", s)].detach().clone()
        h, w = d.shape[-2:]
        d[:, :, : h // 3, : w // 2] = 0.0
        d[:, :, h // 3: 2 * h // 3, w // 2:] = 1.0
        d[:, :, -1, :] = 0.0
        d[:, :, :, 0] = 1.0
        _set(case, ("disp", s), d)
    return case


def flat_frames(phase, B=2, H=64, W=96, scales=(0, 1, 2), seed=23, value=0.5):
    """Constant-colour frames: every SSIM window has zero variance (n == d up to rounding), every L1 term is ~0."""
    case = pc.Case(phase, B, H, W, list(scales), seed=seed)
    for f in (0, -1, 1):
        case.inputs[("color", f, 0)] = torch.full((B, 3, H, W), value)
        case.inputs[("color_aug", f, 0)] = case.inputs[("color", f, 0)]
    for s in case.scales:
        if s:
            case.inputs[("color", 0, s)] = torch.full((B, 3, H >> s, W >> s), value)
    if case.automask:
        case.noise = {s: torch.zeros(B, 2, H, W) for s in case.scales}
    return case


def identical_sources(phase, B=2, H=64, W=96, scales=(0, 1, 2), seed=24, zero_noise=True):
    """Both source frames are the same image and both poses the same transform: the two warped losses are bit-identical in any
    implementation that treats the frames alike, so torch.min's first-index rule decides every pixel."""
    case = pc.Case(phase, B, H, W, list(scales), seed=seed)
    case.inputs[("color", 1, 0)] = case.inputs[("color", -1, 0)].clone()
    case.inputs[("color_aug", 1, 0)] = case.inputs[("color", 1, 0)]
    _set(case, ("axisangle", 1), case.leaves[("axisangle", -1)].detach())
    _set(case, ("translation", 1), case.leaves[("translation", -1)].detach())
    if case.automask and zero_noise:
        case.noise = {s: torch.zeros(B, 2, H, W) for s in case.scales}
    return case


def far_translation(phase, B=1, H=64, W=96, scales=(0, 2), seed=25):
    """A translation of 1e20: projected coordinates overflow to +-inf and are clipped to the border; every value stays finite."""
    case = pc.Case(phase, B, H, W, list(scales), seed=seed)
    _set(case, ("translation", 1), torch.tensor([[[1e20, -1e20, 3.0]]]).repeat(B, 1, 1))
    return case
