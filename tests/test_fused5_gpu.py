"""dd_fused_loss (five launches: the smoothness inside the photometric tile kernel at scale 0 and inside the footprint pass at the
scales >= 1, candidates + scoring in one task, per-image folds) against dd_photo_loss + dd_reg_losses_finish (round 4's ten launches)
on the same arguments: every entry of the losses dict and every gradient, all four phases, shapes with partial tiles, four scales,
the Waymo and nuScenes shapes; and the fall-back where a scale's rows are not whole quads.  The two pipelines differ in the order of a
few additions and in the disparity smoothness being divided by (mean + eps) after the differences instead of before
(include/dynamo_hip.h) -- both stay within the tolerances the reference goldens are held to (tests/test_fused_loss_gpu.py runs those
through both).  Reference: Trainer.py:215-411, tools.py:311-326."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu

PHASES = {"disp_init": (False, False, ("Depth", "Pose"), True),
          "motion_init": (True, False, ("CmpFlow",), False),
          "mask_init": (True, True, ("Pose", "CmpFlow", "MotMask"), False),
          "fine_tune": (True, True, ("Depth", "Pose", "CmpFlow", "MotMask"), False)}
BASE = dict(p_photo=1.0, d_smooth=1e-3, d_ground=0.1, c_smooth=1e-3, c_consistency=5.0, m_sparsity=0.04, m_smooth=0.1)


def evaluate(phase, B, H, W, scales, pipeline, seed=3, materialise=False):
    from hipops import fused_loss as FL
    from hipops.functions import PoseMatrixFn
    cmp, mot, optimised, automask = PHASES[phase]
    inputs = {k: v.cuda() for k, v in synth.make_inputs(seed, B, H, W, scales).items()}
    leaves = {k: v.detach().cuda().requires_grad_() for k, v in synth.make_leaves(seed, B, H, W, scales).items()}
    outputs = synth.leaves_to_outputs(leaves, scales, lambda a, t, invert: PoseMatrixFn.apply(a, t, invert), cmp, mot)
    if cmp:
        for s in scales:
            outputs[("complete_flow_field", 1, s)] = leaves[("flow", s)]          # what networks.Model publishes: one tensor for both frames
            if mot:
                outputs[("motion_mask", -1, s)] = outputs[("motion_mask", 1, s)]     # (the reference's frames share ONE mask object, networks/model.py:148-149)
    plan = FL.LossPlan(height=H, width=W, scales=scales, min_depth=0.1, max_depth=100.0, ssim_weight=0.85, mask_disp_thrd=0.03,
                       gp_prior=0.4, gp_tol=0.005, gp_max_it=100, gp_np_per_it=5, cmpflow=cmp, motmask=mot, automask=automask,
                       optimised=optimised, coefs=dict(BASE))
    g = torch.Generator().manual_seed(11)
    noise = {s: torch.randn(B, 2, H, W, generator=g).cuda() for s in scales} if automask else None
    rs = np.random.RandomState(5)
    ridx = {s: rs.randint(0, int(0.4 * (H >> s)) * (W >> s), (B, 500)).astype(np.int64) for s in scales}
    old = FL.PIPELINE
    FL.PIPELINE = pipeline
    # the five-launch pipeline does not zero its gradient buffers (every element is written by one plain store): leave NaNs in the
    # caching allocator's free blocks, so that an element nobody writes shows up in the comparison
    poison = torch.full((64 << 20,), float("nan"), device="cuda")
    del poison
    try:
        losses = FL.fused_loss(plan, inputs, outputs, noise=noise, rand_idx=ridx, materialise=materialise)
        ran = FL.LAST_PIPELINE[0]
    finally:
        FL.PIPELINE = old
    losses["loss"].backward()
    torch.cuda.synchronize()
    vals = {k: float(v) for k, v in losses.items()}
    grads = {k: (torch.zeros_like(v) if v.grad is None else v.grad).double().cpu().numpy() for k, v in leaves.items()}
    return vals, grads, ran


SHAPES = [(2, 192, 640, [0, 1, 2]),          # KITTI
          (3, 96, 160, [0, 1, 2, 3]),        # four scales, w = 20 at the coarsest
          (1, 64, 96, [0, 1, 2]),            # three tile columns
          (2, 80, 144, [0, 1, 2]),           # partial tiles in both directions (80 = 5 x 16, 144 = 4.5 x 32)
          (2, 320, 480, [0, 1, 2]),          # Waymo (BASELINE config 4): 300 tiles
          (2, 288, 512, [0, 1, 2, 3])]       # nuScenes (config 5)


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dx%dx%d_S%d" % (s[0], s[1], s[2], len(s[3])))
@pytest.mark.parametrize("phase", list(PHASES))
def test_five_launch_pipeline_matches_the_ten_launch_one(phase, shape):
    B, H, W, scales = shape
    v5, g5, ran5 = evaluate(phase, B, H, W, scales, "fused")
    v10, g10, ran10 = evaluate(phase, B, H, W, scales, "split")
    assert ran5 == "fused5" and ran10 == "split", (ran5, ran10)
    lines, bad = [], []
    for k in sorted(v10):
        a, b = v5[k], v10[k]
        ok = abs(a - b) <= 3e-6 * max(abs(b), 1e-2)
        lines.append("%-28s fused5 %.8f split %.8f %s" % (k, a, b, "" if ok else "<-- FAIL"))
        if not ok:
            bad.append(k)
    for k in sorted(g10, key=str):
        a, b = g5[k], g10[k]
        scale = np.abs(b).max() + 1e-30
        rel = np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)
        # an element whose disparity difference to a neighbour vanishes in one rounding and not in the other takes a different
        # sub-gradient of |.| (both valid): such flips are counted, the rest must agree to rounding
        flips = np.abs(a - b) > 1e-4 * scale
        rest = np.linalg.norm((a - b) * ~flips) / (np.linalg.norm(b) + 1e-30)
        lines.append("%-22s rel_l2 %.2e without flips %.2e flips %.2e max|g| %.2e" % (k, rel, rest, flips.mean(), scale))
        if rest > 2e-5 or flips.mean() > 2e-4 or rel > 2e-3:
            bad.append(k)
    print("\n".join(lines))
    assert not bad, bad


def test_rows_that_are_not_whole_quads_fall_back():
    """W = 104: the scale-2 rows are 26 wide -- no 16-byte quads; dd_fused_loss_supported says no and the ten launches run."""
    v, g, ran = evaluate("fine_tune", 2, 64, 104, [0, 1, 2], "fused")
    assert ran == "split" and np.isfinite(v["loss"])


def test_log_step_outputs_come_from_the_five_launch_pipeline_too():
    """materialise=True (the OUT instantiation of the tile kernel, with the smoothness): same losses as the training step."""
    v, _, ran = evaluate("fine_tune", 2, 96, 160, [0, 1, 2], "fused", materialise=True)
    w, _, _ = evaluate("fine_tune", 2, 96, 160, [0, 1, 2], "fused", materialise=False)
    assert ran == "fused5"
    for k in w:
        assert abs(v[k] - w[k]) <= 1e-7 * max(abs(w[k]), 1e-3), (k, v[k], w[k])
