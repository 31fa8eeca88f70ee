"""End-to-end parity of the fused HIP loss (hipops.fused_loss -> C ABI) against the golden vectors the
UNMODIFIED reference produced (tests/golden/loss_<phase>.npz): every losses-dict entry, selected output
maps and the gradients w.r.t. the stand-in network outputs.  GPU only."""
import os

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
RAMPED = ("c_smooth", "c_consistency", "m_sparsity", "m_smooth")


def str2key(s):
    return tuple(int(p) if p.lstrip("-").isdigit() else p for p in s.split("|"))


def run_phase(golden_dir, phase, materialise=True, shared_field=False):
    """shared_field: publish the un-negated flow field the way networks.Model does (('complete_flow_field', 1, s)): both frames then
    read ONE flow tensor (the sign rides on ts), the kernels take their SHARED instantiations and the loss runs as dd_fused_loss."""
    from hipops.fused_loss import LossPlan, fused_loss
    from hipops.functions import PoseMatrixFn
    z = np.load(os.path.join(golden_dir, "loss_%s.npz" % phase))
    B, H, W = int(z["meta/B"]), int(z["meta/H"]), int(z["meta/W"])
    scales = [int(s) for s in z["meta/scales"]]
    ts = {0: [1] * B, -1: z["meta/ts_m1"].tolist(), 1: z["meta/ts_p1"].tolist()}
    inputs = {k: v.cuda() for k, v in synth.make_inputs(int(z["meta/seed"]), B, H, W, scales, ts=ts).items()}
    leaves = {str2key(k[5:]): torch.from_numpy(z[k]).cuda().requires_grad_() for k in z.files if k.startswith("leaf/")}
    cmp, mot, optimised, automask = PHASES[phase]
    outputs = synth.leaves_to_outputs(leaves, scales, lambda a, t, invert: PoseMatrixFn.apply(a, t, invert), cmp, mot)
    if shared_field and cmp:
        for s in scales:
            outputs[("complete_flow_field", 1, s)] = leaves[("flow", s)]
            if mot:
                outputs[("motion_mask", -1, s)] = outputs[("motion_mask", 1, s)]     # one mask object for both frames (networks/model.py:148-149)
    ramp = float(np.clip(3 * int(z["meta/step"]) / int(z["meta/steps_per_epoch"]), 0, 1))
    coefs = {k: v * (ramp if k in RAMPED else 1.0) for k, v in BASE.items()}
    plan = LossPlan(height=H, width=W, scales=scales, min_depth=0.1, max_depth=100.0, ssim_weight=0.85, mask_disp_thrd=0.03,
                    gp_prior=0.4, gp_tol=0.005, gp_max_it=100, gp_np_per_it=5, cmpflow=cmp, motmask=mot, automask=automask,
                    optimised=optimised, coefs=coefs)
    noise = {s: torch.from_numpy(z["noise/%d" % s]).cuda() for s in scales} if automask else None
    ridx = {s: z["rand_idx/%d" % s] for s in scales} if "rand_idx/0" in z.files else None
    losses = fused_loss(plan, inputs, outputs, noise=noise, rand_idx=ridx, materialise=materialise)
    losses["loss"].backward()
    torch.cuda.synchronize()
    return z, leaves, outputs, losses


@pytest.mark.parametrize("shared_field", [False, True])
@pytest.mark.parametrize("phase", list(PHASES))
def test_fused_loss_matches_reference_golden(golden_dir, phase, shared_field):
    from hipops import fused_loss as FL
    if shared_field and phase == "disp_init":
        pytest.skip("the rigid phase has no flow field")
    z, leaves, outputs, losses = run_phase(golden_dir, phase, shared_field=shared_field)
    # which launches ran: the five of dd_fused_loss wherever the frames share their tensors (what networks.Model publishes) and in
    # the rigid phase; round 4's ten for per-frame flow tensors (the reference's dict layout)
    assert FL.LAST_PIPELINE[0] == ("fused5" if (shared_field or phase == "disp_init") else "split"), FL.LAST_PIPELINE
    lines, fails = [], []
    for name in z.files:
        if name.startswith("losses/"):
            got, want = float(losses[name[7:]]), float(z[name])
            # d_ground goes through a RANSAC 3x3 solve (fp32 LAPACK in the reference, fp64 cofactors here)
            tol = 2e-3 if "d_ground" in name else 3e-5
            ok = abs(got - want) <= tol * max(1.0, abs(want)) if "d_ground" not in name else abs(got - want) <= tol * max(abs(want), 1e-3) + 1e-6
            lines.append("%-28s got %.7f want %.7f %s" % (name, got, want, "" if ok else "<-- FAIL"))
            if not ok and not ("d_ground" in name or (phase == "fine_tune" and name in ("losses/loss", "losses/loss_term/0", "losses/loss_term/1", "losses/loss_term/2", "losses/loss_term/3"))):
                fails.append(name)
            elif not ok:
                # totals inherit the d_ground deviation: allow its weight
                if abs(got - want) > 0.1 * 2e-3 + 3e-5:
                    fails.append(name)
        elif name.startswith("out/identity_selection"):
            mism = float((outputs[name[4:]].cpu().numpy() != z[name]).mean())
            lines.append("%-28s mismatch %.2e" % (name, mism))
            if mism > 1e-3:
                fails.append(name)
        elif name.startswith("out/") and str2key(name[4:])[0] in ("color", "sample", "depth", "residual_flow"):
            if str2key(name[4:]) not in outputs:
                continue        # residual_flow is only produced where a loss term reads it (mask phases)
            got = outputs[str2key(name[4:])].detach().cpu().numpy()
            err = np.abs(got - z[name])
            bad = float((err > 2e-5 + 1e-4 * np.abs(z[name])).mean())
            lines.append("%-28s max|err| %.2e frac_bad %.2e" % (name, err.max(), bad))
            if bad > 2e-3:
                fails.append(name)
        elif name.startswith("grad/") and not name.startswith("grad/cam_T_cam"):
            leaf = leaves[str2key(name[5:])]
            got = torch.zeros_like(leaf) if leaf.grad is None else leaf.grad
            got = got.cpu().double().numpy()
            want = z[name].astype(np.float64)
            scale = np.abs(want).max() + 1e-30
            outl = np.abs(got - want) > 1e-3 * scale + 1e-3 * np.abs(want)
            keep = ~outl
            trimmed = np.linalg.norm((got - want) * keep) / (np.linalg.norm(want * keep) + 1e-30)
            rel = np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-30)
            lines.append("%-28s rel_l2 %.2e trimmed %.2e outliers %.2e max|ref| %.2e" % (name, rel, trimmed, outl.mean(), scale))
            small = want.size <= 16          # pose vectors: sums over all pixels, judge the vector
            # ~10x what the kernels measure against these goldens (profiles/r02_parity_report.txt): bulk error 1e-5..5e-5
            # (worst 1.3e-4), pose vectors <= 2.8e-4, outliers <= 7e-4 -- the outliers and the total rel-L2 of disp_init are the
            # auto-mask's identity/warp ties flipping on last-bit differences of the photometric loss
            rel_cap = 5e-2 if phase == "disp_init" else 1e-2
            if (small and rel > 3e-3) or (not small and (trimmed > 5e-4 or outl.mean() > 5e-3 or rel > rel_cap)):
                fails.append(name)
    print("\n".join(lines))
    assert not fails, fails


def test_fused_loss_no_grad_and_unmaterialised(golden_dir):
    with torch.no_grad():
        pass
    z, leaves, outputs, losses = run_phase(golden_dir, "mask_init", materialise=False)
    assert ("color", -1, 0) not in outputs
    assert abs(float(losses["loss"]) - float(z["losses/loss"])) < 1e-4
