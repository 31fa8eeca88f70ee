"""The oracle (oracle/ref_loss.py) pinned against vectors produced by the unmodified reference
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import synth
import oracle.ref_loss as orc

PHASE_NAMES = ("disp_init", "motion_init", "mask_init", "fine_tune")
BASE_COEFS = dict(p_photo=1.0, d_smooth=1e-3, d_ground=0.1, c_smooth=1e-3, c_consistency=5.0, m_sparsity=0.04, m_smooth=0.1)
RAMPED = ['g_c_smooth', 'g_c_consistency', 'g_m_sparsity', 'g_m_smooth']


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def str2key(s):
    parts = s.split("|")
    return tuple(int(p) if p.lstrip("-").isdigit() else p for p in parts)


def run_oracle(z, phase):
    B, H, W = int(z["meta/B"]), int(z["meta/H"]), int(z["meta/W"])
    scales = [int(s) for s in z["meta/scales"]]
    ts = {0: [1] * B, -1: z["meta/ts_m1"].tolist(), 1: z["meta/ts_p1"].tolist()}
    inputs = synth.make_inputs(int(z["meta/seed"]), B, H, W, scales, ts=ts)
    leaves = synth.make_leaves(int(z["meta/seed"]), B, H, W, scales)
    for k, v in leaves.items():     # the stored leaves are authoritative
        np.testing.assert_array_equal(v.detach().numpy(), z["leaf/" + "|".join(str(x) for x in k)])
    coefs = orc.ramped_coefs(BASE_COEFS, RAMPED, 3, int(z["meta/step"]), int(z["meta/steps_per_epoch"]))
    cfg = orc.LossConfig(H, W, scales, coefs=coefs)
    cmp, mot, _, _ = orc.PHASES[phase]
    outputs = synth.leaves_to_outputs(leaves, scales, orc.pose_matrix, cmp, mot)
    noise = {s: torch.from_numpy(z["noise/%d" % s]) for s in scales if "noise/%d" % s in z.files} or None
    ridx = {s: z["rand_idx/%d" % s] for s in scales if "rand_idx/%d" % s in z.files} or None
    losses = orc.loss_path(cfg, inputs, outputs, phase, noise, ridx)
    losses["loss"].backward()
    return inputs, leaves, outputs, losses


@pytest.mark.parametrize("phase", PHASE_NAMES)
def test_loss_path_matches_reference(golden_dir, phase):
    z = load(golden_dir, "loss_%s.npz" % phase)
    inputs, leaves, outputs, losses = run_oracle(z, phase)
    for name in z.files:
        if name.startswith("losses/"):
            assert abs(float(losses[name[7:]]) - float(z[name])) <= 2e-6 * max(1.0, abs(float(z[name]))), name
        elif name.startswith("out/identity_selection"):
            got = outputs[name[4:]].numpy()
            assert (got != z[name]).mean() < 1e-4, name
        elif name.startswith("out/"):
            got = outputs[str2key(name[4:])].detach().numpy()
            np.testing.assert_allclose(got, z[name], rtol=1e-5, atol=2e-6, err_msg=name)
        elif name.startswith("grad/cam_T_cam"):
            g = outputs[str2key(name[5:])].grad
            got = np.zeros_like(z[name]) if g is None else g.numpy()
            np.testing.assert_allclose(got, z[name], rtol=1e-4, atol=1e-7 + 1e-5 * np.abs(z[name]).max(), err_msg=name)
        elif name.startswith("grad/"):
            g = leaves[str2key(name[5:])].grad
            got = np.zeros_like(z[name]) if g is None else g.numpy()
            np.testing.assert_allclose(got, z[name], rtol=1e-4, atol=1e-7 + 1e-5 * np.abs(z[name]).max(), err_msg=name)


def test_operators_match_reference(golden_dir):
    z = load(golden_dir, "ops.npz")
    t = {k: torch.from_numpy(z[k]) for k in z.files}
    B, _, h, w = t["depth"].shape
    pts = orc.backproject(t["depth"], t["inv_K"])
    np.testing.assert_allclose(pts.numpy(), z["points"], rtol=1e-6, atol=1e-6)
    T = orc.pose_matrix(t["axisangle"], t["translation"], invert=True)
    np.testing.assert_allclose(T.numpy(), z["T_inv"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(orc.pose_matrix(t["axisangle"], t["translation"], invert=False).numpy(), z["T_fwd"], rtol=1e-6, atol=1e-7)
    pix, ego = orc.project(t["points"], t["K"], t["T_inv"], h, w)
    np.testing.assert_allclose(pix.numpy(), z["pix_T"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ego.numpy(), z["ego_T"], rtol=1e-5, atol=1e-6)
    pix, ego = orc.project(t["points"], t["K"], None, h, w)
    np.testing.assert_allclose(pix.numpy(), z["pix_N"], rtol=1e-5, atol=1e-5)
    assert np.abs(ego.numpy()).max() == 0 and np.abs(z["ego_N"]).max() == 0
    np.testing.assert_allclose(orc.ssim_map(t["x"], t["y"]).numpy(), z["ssim"], rtol=1e-5, atol=1e-6)
    sd, dp = orc.disp_to_depth(t["disp"], 0.1, 100.0)
    np.testing.assert_allclose(sd.numpy(), z["scaled_disp"], rtol=1e-6)
    np.testing.assert_allclose(dp.numpy(), z["depth_from_disp"], rtol=1e-6)
    np.testing.assert_allclose(orc.depth_to_disp(dp, 0.1, 100.0).numpy(), z["disp_roundtrip"], rtol=1e-5, atol=1e-7)
    assert abs(float(orc.smooth_loss(t["smooth_inp"], t["x"])) - float(z["smooth_img"])) < 1e-6
    assert abs(float(orc.smooth_loss(t["smooth_inp"], None)) - float(z["smooth_none"])) < 1e-6
    np.testing.assert_allclose(orc.resize_bilinear(t["disp"], (h * 4, w * 4)).numpy(), z["interp_up"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(orc.resize_bilinear(t["x"], (h // 4, w // 4)).numpy(), z["interp_down"], rtol=1e-6, atol=1e-7)
    cfg = orc.LossConfig(h, w, [0])
    dist, param = orc.ground_plane(t["ground_points"], cfg, z["ground_rand_idx"])
    np.testing.assert_allclose(param.numpy(), z["ground_param"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dist.numpy(), z["ground_dist"], rtol=1e-4, atol=1e-5)
