"""Seeded synthetic loss-path inputs: frames, intrinsics and stand-in
network outputs.  Used by the golden generator and by the parity tests (same seed -> same tensors,
torch CPU generator)."""
import numpy as np
import torch
import torch.nn.functional as F

KITTI_K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)


def make_frames(gen, B, H, W, frame_ids=(0, -1, 1), shift=5):
    base = torch.rand(B, 3, H // 8 + 2, W // 8 + 2, generator=gen)
    big = F.interpolate(base, (H + 16, W + 16), mode="bilinear", align_corners=False)
    out = {}
    for f in frame_ids:
        x0 = 8 + shift * f
        img = big[:, :, 8:8 + H, x0:x0 + W] + 0.05 * torch.rand(B, 3, H, W, generator=gen)
        out[f] = img.clamp(0, 1).contiguous()
    return out


def make_intrinsics(B, H, W, num_scales):
    out = {}
    for s in range(num_scales):
        K = KITTI_K.copy()
        K[0, :] *= W // (2 ** s)
        K[1, :] *= H // (2 ** s)
        inv_K = np.linalg.pinv(K)
        out[("K", s)] = torch.from_numpy(K).unsqueeze(0).repeat(B, 1, 1)
        out[("inv_K", s)] = torch.from_numpy(inv_K).unsqueeze(0).repeat(B, 1, 1)
    return out


def make_inputs(seed, B, H, W, scales, ts=None):
    gen = torch.Generator().manual_seed(seed)
    frames = make_frames(gen, B, H, W)
    inputs = {}
    for f, img in frames.items():
        inputs[("color", f, 0)] = img
        inputs[("color_aug", f, 0)] = img
    inputs.update(make_intrinsics(B, H, W, max(scales) + 1))
    for f in (0, -1, 1):
        inputs[("ts", f)] = torch.ones(B, dtype=torch.int64) if ts is None else torch.as_tensor(ts[f])
    # target pyramid exactly as Trainer.apply_img_resize (Trainer.py:729-734): chained bicubic+antialias, clamp
    for s in range(1, max(scales) + 1):
        if s != 0:
            h, w = H // 2 ** s, W // 2 ** s
            inputs[("color", 0, s)] = torch.clamp(
                F.interpolate(inputs[("color", 0, s - 1)], (h, w), mode="bicubic", align_corners=False, antialias=True), 0, 1)
    return inputs


def make_leaves(seed, B, H, W, scales):
    """Stand-in network outputs as autograd leaves."""
    gen = torch.Generator().manual_seed(seed + 1000)
    leaves = {}
    for s in scales:
        h, w = H // 2 ** s, W // 2 ** s
        leaves[("disp", s)] = (0.05 + 0.9 * torch.rand(B, 1, h, w, generator=gen)).requires_grad_()
        leaves[("flow", s)] = (0.05 * torch.randn(B, 3, h, w, generator=gen)).requires_grad_()
        leaves[("prob", s)] = torch.randn(B, 1, h, w, generator=gen).requires_grad_()
    for f in (-1, 1):
        leaves[("axisangle", f)] = (0.01 * torch.randn(B, 1, 3, generator=gen)).requires_grad_()
        leaves[("translation", f)] = (0.03 * torch.randn(B, 1, 3, generator=gen)).requires_grad_()
    return leaves


def leaves_to_outputs(leaves, scales, pose_fn, cmpflow, motmask):
    """Assemble the `outputs` dict the way networks.Model.forward does (networks/model.py:58-149):
    flow for frame -1 is the negated field, mask/prob tensors are shared by both frames."""
    outputs = {}
    for s in scales:
        outputs[("disp", 0, s)] = leaves[("disp", s)]
        if cmpflow:
            outputs[("complete_flow", -1, s)] = -1 * leaves[("flow", s)]
            outputs[("complete_flow", 1, s)] = 1 * leaves[("flow", s)]
        if motmask:
            m_raw = leaves[("prob", s)]
            for f in (-1, 1):
                outputs[("motion_prob", f, s)] = m_raw
                outputs[("motion_mask", f, s)] = torch.sigmoid(m_raw)
    for f in (-1, 1):
        outputs[("axisangle", 0, f)] = leaves[("axisangle", f)]
        outputs[("translation", 0, f)] = leaves[("translation", f)]
        T = pose_fn(leaves[("axisangle", f)], leaves[("translation", f)], invert=True)
        T.retain_grad()
        outputs[("cam_T_cam", 0, f)] = T
    return outputs
