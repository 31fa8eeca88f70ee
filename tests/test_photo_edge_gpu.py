"""Adversarial-geometry parity of the fused photometric kernel (through the C ABI) against the oracle: points behind the
camera, samples outside the border, disparity exactly 0 / 1, flat and identical frames (SSIM n == d, exact min ties, auto-mask
ties with zero noise), overflowing coordinates, and the closed sparsity gate / non-finite inputs through the whole fused loss.
Cases: tests/edge_cases.py.  GPU only (tests/test_hostmath.py runs the same cases through the host-compiled arithmetic)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

import edge_cases as ec
import photo_case as pc
import synth

pytestmark = pytest.mark.gpu


def run_case(case, materialise=True, shared=False):
    from hipops import lib as L
    args, t = case.photo_buffers("cuda", materialise=materialise, want_grad=True, shared=shared)
    L.check(L.load().dd_photo_loss(C.byref(args), L.current_stream()), "dd_photo_loss")
    torch.cuda.synchronize()
    return t


def judge(case, t, masked=True, resid_atol=1e-6):
    """Values and maps against the fp32 oracle; per-pixel gradients decision-masked against the fp64 oracle; pose gradients
    (sums over all pixels) in units of the fp32 oracle's own distance from fp64."""
    report = []
    fails = case.check(t, report=report, resid_atol=resid_atol)
    if masked and case.grad64 is not None:
        fails += case.check_grads_masked(t, report=report)
        # the pose gradients are sums over ALL pixels: a decision that the kernel's rounding moves and torch's does not (near z = 0
        # one pixel can carry 1e-3 of the sum) shows in them undiluted -- then the yardstick is the size of such a pixel, not
        # the fp32 oracle's rounding
        fails += case.check_grads(t, report=report, only_T=True, t_slack=4.0 if case.kernel_flips == 0 else 16.0)
    else:
        fails += case.check_grads(t, report=report)
    print("\n".join(report))
    return fails


@pytest.mark.parametrize("phase", ["disp_init", "motion_init", "mask_init"])
def test_points_behind_the_camera_and_samples_outside_the_border(phase):
    case = ec.behind_camera(phase).run_oracle(fp64=True)
    behind, outside = ec.geometry_stats(case)
    print("behind the camera: %.3f, outside the border: %.3f" % (behind, outside))
    if phase != "motion_init":          # motion_init applies no rigid transform (Trainer.py:270-271): only the flow moves points
        assert behind >= 0.10 and outside >= 0.30, (behind, outside)
    t = run_case(case, shared=phase != "disp_init")
    fails = judge(case, t)
    assert not fails, fails


def test_points_behind_the_camera_full_size():
    case = ec.behind_camera("fine_tune", B=1, H=192, W=640, scales=(0, 1, 2), seed=31).run_oracle(fp64=True)
    behind, outside = ec.geometry_stats(case)
    print("behind the camera: %.3f, outside the border: %.3f" % (behind, outside))
    assert behind >= 0.10 and outside >= 0.30, (behind, outside)
    t = run_case(case, materialise=False, shared=True)
    fails = judge(case, t)
    assert not fails, fails


@pytest.mark.parametrize("phase", ["disp_init", "mask_init"])
def test_disparity_exactly_zero_and_one(phase):
    case = ec.disp_extremes(phase).run_oracle(fp64=True)
    t = run_case(case, shared=phase != "disp_init")
    fails = judge(case, t, resid_atol=5e-5)        # |P| reaches 100: T P - P cancels to ~100 x 2^-23 in any fp32 implementation
    assert not fails, fails


@pytest.mark.parametrize("phase", ["disp_init", "mask_init"])
def test_constant_colour_frames(phase):
    """Zero-variance SSIM windows: n == d in exact arithmetic, the clamp sits on its lower edge, every gradient through the
    colours is zero (the bilinear taps of a constant image have no slope).  Under the auto-mask the identity loss is exactly 0
    and wins every pixel (ties go to the identity entries, Trainer.py:341-343)."""
    case = ec.flat_frames(phase).run_oracle()
    t = run_case(case, shared=phase != "disp_init")
    for si, s in enumerate(case.scales):
        photo = float(t["sums"][si, 0]) / (case.B * case.H * case.W)
        want = case.oracle_photo(s)
        print("scale %d: p_photo %.3e (oracle %.3e)" % (s, photo, want))
        assert abs(photo - want) < 1e-6 and photo >= 0.0
        if case.automask:
            assert photo == 0.0 and float(t["scales"][si]["out_idsel"].sum()) == 0.0
            assert float(case.outputs["identity_selection/%d" % s].sum()) == 0.0
        d = t["scales"][si]
        if case.mode != 2:          # mask phases: c_consistency still has a gradient; the photometric part is checked below
            assert float(d["g_disp"].abs().max()) < 1e-9
    fails = case.check_grads(t)
    assert not fails, fails


@pytest.mark.parametrize("noise", ["zero", "drawn"])
def test_identical_source_frames_exact_ties(noise):
    """Frames -1 and +1 are the same image under the same pose: the two warped losses are bit-identical, torch.min returns the
    FIRST index and routes the whole gradient to frame -1."""
    case = ec.identical_sources("disp_init", zero_noise=noise == "zero").run_oracle(fp64=True)
    t = run_case(case)
    g0, g1 = t["g_T"][0].cpu(), t["g_T"][1].cpu()
    print("|g_T[-1]| %.3e  |g_T[+1]| %.3e" % (g0.norm(), g1.norm()))
    want1 = case.outputs[("cam_T_cam", 0, 1)].grad
    assert want1 is None or float(want1.abs().max()) == 0.0       # the oracle sends nothing to the second frame ...
    assert float(g1.abs().max()) == 0.0                           # ... and neither does the kernel
    assert float(g0.abs().max()) > 0.0
    fails = judge(case, t)
    assert not fails, fails


def test_identical_source_frames_without_automask():
    case = ec.identical_sources("mask_init")
    for s in case.scales:
        ec._set(case, ("flow", s), torch.zeros_like(case.leaves[("flow", s)]))     # +-0 flow: both frames see the same geometry
    case.coefs["c_consistency"] = 0.0
    case.cfg.coefs["c_consistency"] = 0.0
    case.run_oracle(fp64=True)
    t = run_case(case, shared=True)
    g1 = t["g_T"][1].cpu()
    want1 = case.outputs[("cam_T_cam", 0, 1)].grad
    assert float(g1.abs().max()) == 0.0 and (want1 is None or float(want1.abs().max()) == 0.0)
    fails = judge(case, t)
    assert not fails, fails


@pytest.mark.parametrize("phase", ["disp_init", "mask_init"])
def test_overflowing_coordinates_stay_finite(phase):
    case = ec.far_translation(phase).run_oracle()
    t = run_case(case, shared=phase != "disp_init")
    for d in t["scales"]:
        assert torch.isfinite(d["g_disp"]).all()
    assert torch.isfinite(t["sums"]).all() and all(torch.isfinite(g).all() for g in t["g_T"])
    fails = judge(case, t, masked=False)
    assert not fails, fails


# ---- whole fused loss: closed sparsity gate, non-finite inputs -----------------------------------------------------------------

def fused(phase, B, H, W, scales, leaves, inputs, coefs=None, rand_idx=None):
    from hipops.fused_loss import LossPlan, fused_loss
    from hipops.functions import PoseMatrixFn
    cmp, mot, optimised, automask = pc.orc.PHASES[phase]
    dev_in = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inputs.items()}
    dev_leaves = {k: v.detach().cuda().requires_grad_() for k, v in leaves.items()}
    outputs = synth.leaves_to_outputs(dev_leaves, scales, lambda a, t, invert: PoseMatrixFn.apply(a, t, invert), cmp, mot)
    plan = LossPlan(height=H, width=W, scales=scales, min_depth=0.1, max_depth=100.0, ssim_weight=0.85, mask_disp_thrd=0.03,
                    gp_prior=0.4, gp_tol=0.005, gp_max_it=100, gp_np_per_it=5, cmpflow=cmp, motmask=mot, automask=automask,
                    optimised=optimised, coefs=coefs or pc.BASE_COEFS)
    losses = fused_loss(plan, dev_in, outputs, rand_idx=rand_idx)
    losses["loss"].backward()
    torch.cuda.synchronize()
    return dev_leaves, losses


def closed_gate_case(B=2, H=64, W=96, scales=(0, 1, 2), seed=41):
    """Image 1 moves everywhere (a large uniform flow at near-constant depth), image 0 hardly at all: the batch-global mean of disp_mag lies below
    every pixel of image 1, which therefore has ZERO static pixels -- `torch.all(sum(static) > 0)` is false and the sparsity
    term of that (scale, frame) is skipped (reference Trainer.py:397-399)."""
    scales = list(scales)
    inputs = synth.make_inputs(seed, B, H, W, scales)
    leaves = synth.make_leaves(seed, B, H, W, scales)
    with torch.no_grad():
        for s in scales:
            fl = leaves[("flow", s)]
            fl[0] *= 0.02
            fl[1] = 0.02 * fl[1] + torch.tensor([0.6, 0.3, 0.0]).view(3, 1, 1)
            d = leaves[("disp", s)]
            d[1] = 0.5 + 0.02 * (d[1] - 0.5)          # near-constant depth: the flow shifts every pixel of image 1 by about the same amount
    return inputs, leaves


@pytest.mark.parametrize("phase", ["mask_init", "fine_tune"])
def test_closed_sparsity_gate(phase):
    B, H, W, scales = 2, 64, 96, [0, 1, 2]
    inputs, leaves = closed_gate_case(B, H, W, scales)
    ridx = {s: pc.orc.ransac_indices(B, int(0.4 * (H >> s)) * (W >> s), 500) for s in scales} if phase == "fine_tune" else None
    cfg = pc.orc.LossConfig(H, W, scales, coefs=pc.BASE_COEFS)
    cmp, mot, _, _ = pc.orc.PHASES[phase]
    outputs = synth.leaves_to_outputs(leaves, scales, pc.orc.pose_matrix, cmp, mot)
    want = pc.orc.loss_path(cfg, dict(inputs), outputs, phase, None, ridx)
    want["loss"].backward()
    # the gate really is closed in the oracle: image 1 has no static pixel at any scale / frame
    for s in scales:
        for f in (-1, 1):
            h, w = H >> s, W >> s
            e = pc.orc.resize_bilinear(outputs[("sample_ego", f, s)].permute(0, 3, 1, 2), (h, w))
            k = pc.orc.resize_bilinear(outputs[("sample_complete", f, s)].permute(0, 3, 1, 2), (h, w))
            mag = ((e - k) ** 2).sum(1)
            assert int((mag[1] < mag.mean()).sum()) == 0 and int((mag[0] < mag.mean()).sum()) > 0
    assert float(want["loss_term/m_sparsity"]) == 0.0
    dev_leaves, got = fused(phase, B, H, W, scales, leaves, inputs, rand_idx=ridx)
    assert float(got["loss_term/m_sparsity"]) == 0.0
    for name in ("p_photo", "c_smooth", "c_consistency", "m_smooth", "d_smooth"):
        g, w_ = float(got["loss_term/" + name]), float(want["loss_term/" + name])
        print("%-16s got %.7f want %.7f" % (name, g, w_))
        assert abs(g - w_) <= 3e-5 * max(1.0, abs(w_)), name
    for s in scales:
        g = dev_leaves[("prob", s)].grad.cpu().double()
        w_ = leaves[("prob", s)].grad.double()
        rel = ((g - w_).norm() / w_.norm()).item()
        print("grad prob[%d] rel_l2 %.3e" % (s, rel))
        assert rel < 2e-2


def test_open_gate_control():
    """The same construction with image 1's flow scaled down keeps the gate open: the sparsity term is there and matches."""
    B, H, W, scales = 2, 64, 96, [0, 1, 2]
    inputs = synth.make_inputs(41, B, H, W, scales)
    leaves = synth.make_leaves(41, B, H, W, scales)
    cfg = pc.orc.LossConfig(H, W, scales, coefs=pc.BASE_COEFS)
    outputs = synth.leaves_to_outputs(leaves, scales, pc.orc.pose_matrix, True, True)
    want = pc.orc.loss_path(cfg, dict(inputs), outputs, "mask_init")
    _, got = fused("mask_init", B, H, W, scales, leaves, inputs)
    g, w_ = float(got["loss_term/m_sparsity"]), float(want["loss_term/m_sparsity"])
    assert w_ > 0 and abs(g - w_) <= 3e-5 * max(1.0, w_), (g, w_)


def test_non_finite_disparity_reaches_the_loss():
    """A NaN in the network output is not swallowed: the reference's loss becomes NaN (smoothness and the depth chain read it),
    and so does the fused loss."""
    B, H, W, scales = 1, 64, 96, [0, 1]
    inputs = synth.make_inputs(43, B, H, W, scales)
    leaves = synth.make_leaves(43, B, H, W, scales)
    with torch.no_grad():
        leaves[("disp", 0)][0, 0, 20, 30] = float("nan")
    cfg = pc.orc.LossConfig(H, W, scales, coefs=pc.BASE_COEFS)
    outputs = synth.leaves_to_outputs(leaves, scales, pc.orc.pose_matrix, False, False)
    noise = {s: torch.zeros(B, 2, H, W) for s in scales}
    want = pc.orc.loss_path(cfg, dict(inputs), outputs, "disp_init", noise)
    assert math.isnan(float(want["loss"]))
    _, got = fused("disp_init", B, H, W, scales, leaves, inputs)
    assert math.isnan(float(got["loss"]))
    assert np.isfinite(float(got["loss_term/p_photo"])) or math.isnan(float(got["loss_term/p_photo"]))
