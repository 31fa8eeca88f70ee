"""Generate golden vectors by executing the UNMODIFIED reference (/root/reference) on CPU.

Runs only in the build container (the reference never travels to the GPU box); the .npz files it
writes are committed.  Usage:  python tests/golden/make_golden.py [ops] [loss] [net] [metrics]

Each golden stores inputs (or the seed that regenerates them via tests/golden/synth.py) and the
reference's outputs / autograd gradients.  The script also checks oracle/ref_loss.py against the
reference on every key (not just the stored subset) and prints the worst deviations.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _refshim  # noqa: E402
import synth  # noqa: E402

LOSS_B, LOSS_H, LOSS_W, LOSS_SCALES = 2, 64, 96, [0, 1, 2, 3]
LOSS_SEED = 7
PHASE_STEP, STEPS_PER_EPOCH = 20, 100      # ramp factor clip(3*20/100) = 0.6


def key2str(k):
    return k if isinstance(k, str) else "|".join(str(x) for x in k)


def npy(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------------------------------------
def gen_ops(ref):
    """Operator-level vectors: tools.py operators, utils.interp, layers.transformation_from_parameters."""
    tools = ref.tools
    g = torch.Generator().manual_seed(11)
    B, h, w = 3, 24, 40
    out = {}
    depth = 0.1 + 5 * torch.rand(B, 1, h, w, generator=g)
    intr = synth.make_intrinsics(B, h, w, 1)
    K, inv_K = intr[("K", 0)], intr[("inv_K", 0)]
    bp = tools.BackprojectDepth(B + 1, h, w)        # constructed batch larger than used: tools.py:192-195 slices [:B]
    pts = bp(depth, inv_K)
    aa = 0.05 * torch.randn(B, 1, 3, generator=g)
    tr = 0.3 * torch.randn(B, 1, 3, generator=g)
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_layers", os.path.join(_refshim.REFERENCE_ROOT, "networks", "layers.py"))
    layers = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(layers)
    T_inv = layers.transformation_from_parameters(aa, tr, invert=True)
    T_fwd = layers.transformation_from_parameters(aa, tr, invert=False)
    pj = tools.Project3D(B + 1, h, w)
    pix_T, ego_T = pj(pts, K, T_inv)
    pix_N, ego_N = pj(pts, K, None)
    x = torch.rand(B, 3, h, w, generator=g)
    y = (x + 0.1 * torch.randn(B, 3, h, w, generator=g)).clamp(0, 1)
    ssim = tools.SSIM()(x, y)
    disp = torch.rand(B, 1, h, w, generator=g)
    sd, dp = tools.disp_to_depth(disp, 0.1, 100.0)
    d2 = tools.depth_to_disp(dp, 0.1, 100.0)
    inp3 = torch.randn(B, 3, h, w, generator=g)
    sm_img = tools.compute_smooth_loss(inp3, x)
    sm_none = tools.compute_smooth_loss(inp3, None)
    up = ref.utils.interp(disp, (h * 4, w * 4))
    down = ref.utils.interp(x, (h // 4, w // 4))
    # ground plane: a tilted noisy plane below the camera + clutter
    gp = tools.GroundPlane(num_points_per_it=5, max_it=100, tol=0.005, g_prior=0.4)
    ray = torch.matmul(inv_K[:, :3, :3], bp.pix_coords[:B])
    plane_depth = (1.6 / (ray[:, 1:2] - 0.02 * ray[:, 0:1] + 1e-3)).clamp(0.5, 60).reshape(B, 1, h, w)
    plane_depth = plane_depth * (1 + 0.002 * torch.randn(B, 1, h, w, generator=g))
    gpts = bp(plane_depth, inv_K)[:, :3].reshape(B, 3, h, w)
    np.random.seed(5)
    N_g = int(0.4 * h) * w
    rand_idx = np.stack([np.random.choice(np.arange(N_g), 500, replace=True) for _ in range(B)])
    np.random.seed(5)
    gdist, gparam = gp(gpts)
    out.update(depth=depth, K=K, inv_K=inv_K, points=pts, axisangle=aa, translation=tr, T_inv=T_inv, T_fwd=T_fwd,
               pix_T=pix_T, ego_T=ego_T, pix_N=pix_N, ego_N=ego_N, x=x, y=y, ssim=ssim, disp=disp,
               scaled_disp=sd, depth_from_disp=dp, disp_roundtrip=d2, smooth_inp=inp3,
               smooth_img=sm_img, smooth_none=sm_none, interp_up=up, interp_down=down,
               ground_points=gpts, ground_rand_idx=torch.from_numpy(rand_idx), ground_dist=gdist, ground_param=gparam)
    path = os.path.join(HERE, "ops.npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in out.items()})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


# ------------------------------------------------------------------------------------------------
def build_ref_trainer(ref, B, H, W, scales, depth_model="monodepthv2"):
    opt = _refshim.make_opt(ref, argv=["-d", "kitti", "--depth_model", depth_model, "-b", str(B),
                                       "--height", str(H), "--width", str(W), "--scales"] + [str(s) for s in scales])
    tr = ref.Trainer.Trainer(opt)
    tr.num_steps_per_epoch = STEPS_PER_EPOCH
    return tr, opt


def run_ref_loss(ref, tr, phase, inputs, leaves, layers_fn, noise_seed, ransac_seed):
    tr.setup_phase(phase)
    tr.bool_automask = phase == "disp_init"
    tr.step = PHASE_STEP
    cmp, mot = tr.base_model.bool_CmpFlow, tr.base_model.bool_MotMask
    outputs = synth.leaves_to_outputs(leaves, tr.opt.scales, layers_fn, cmp, mot)
    inputs = dict(inputs)
    tr.generate_images_pred(inputs, outputs)
    torch.manual_seed(noise_seed)
    np.random.seed(ransac_seed)
    losses = tr.compute_losses(inputs, outputs)
    losses["loss"].backward()
    return outputs, losses


def gen_loss(ref):
    import oracle.ref_loss as orc
    B, H, W, scales = LOSS_B, LOSS_H, LOSS_W, LOSS_SCALES
    tr, opt = build_ref_trainer(ref, B, H, W, scales)
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_layers", os.path.join(_refshim.REFERENCE_ROOT, "networks", "layers.py"))
    layers = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(layers)
    ts = {0: [1, 1], -1: [1, 2], 1: [1, 2]}
    base_coefs = {k[2:]: v for k, v in opt.__dict__.items() if k[:2] == "g_"}
    for phase in ("disp_init", "motion_init", "mask_init", "fine_tune"):
        inputs = synth.make_inputs(LOSS_SEED, B, H, W, scales, ts=ts)
        leaves = synth.make_leaves(LOSS_SEED, B, H, W, scales)
        noise_seed, ransac_seed = 123, 321
        outputs, losses = run_ref_loss(ref, tr, phase, inputs, leaves, layers.transformation_from_parameters,
                                       noise_seed, ransac_seed)
        # reproduce the RNG draws the reference made (Trainer.py:339, tools.py:125-127)
        noise, rand_idx = {}, {}
        if phase == "disp_init":
            torch.manual_seed(noise_seed)
            for s in scales:
                noise[s] = torch.randn(B, 2, H, W)
        if phase == "fine_tune":
            np.random.seed(ransac_seed)
            for s in scales:
                h, w = H // 2 ** s, W // 2 ** s
                rand_idx[s] = orc.ransac_indices(B, int(opt.gp_prior * h) * w, opt.gp_np_per_it * opt.gp_max_it)

        # ---- oracle restatement vs reference, every comparable key ----
        cfg = orc.LossConfig(H, W, scales, coefs=orc.ramped_coefs(base_coefs, opt.weight_ramp, opt.ramp_red,
                                                                  PHASE_STEP, STEPS_PER_EPOCH))
        o_inputs = synth.make_inputs(LOSS_SEED, B, H, W, scales, ts=ts)
        o_leaves = synth.make_leaves(LOSS_SEED, B, H, W, scales)
        cmp, mot, _, _ = orc.PHASES[phase]
        o_outputs = synth.leaves_to_outputs(o_leaves, scales, orc.pose_matrix, cmp, mot)
        o_losses = orc.loss_path(cfg, o_inputs, o_outputs, phase, noise or None, rand_idx or None)
        o_losses["loss"].backward()
        worst = ("", 0.0)
        for k, v in losses.items():
            a = float(v)
            b = float(o_losses[k])
            d = abs(a - b)
            if d > worst[1]:
                worst = ("losses/" + k, d)
        for k, v in outputs.items():
            if not torch.is_tensor(v) or k[0] == "cam_points" or k not in o_outputs:
                continue
            d = (v.detach() - o_outputs[k].detach()).abs().max().item()
            if d > worst[1]:
                worst = (key2str(k), d)
        for k, v in leaves.items():
            ga, gb = v.grad, o_leaves[k].grad
            if ga is None and gb is None:
                continue
            ga = torch.zeros_like(v) if ga is None else ga
            gb = torch.zeros_like(v) if gb is None else gb
            d = (ga - gb).abs().max().item() / (ga.abs().max().item() + 1e-20)
            if d > worst[1]:
                worst = ("grad/" + key2str(k), d)
        print("[{}] loss={:.7f} oracle={:.7f} worst dev: {} {:.3e}".format(
            phase, float(losses["loss"]), float(o_losses["loss"]), worst[0], worst[1]))

        # ---- store ----
        store = {"meta/B": B, "meta/H": H, "meta/W": W, "meta/scales": np.array(scales), "meta/seed": LOSS_SEED,
                 "meta/step": PHASE_STEP, "meta/steps_per_epoch": STEPS_PER_EPOCH,
                 "meta/ts_m1": np.array(ts[-1]), "meta/ts_p1": np.array(ts[1])}
        for k, v in losses.items():
            store["losses/" + k] = np.float32(float(v))
        for k, v in leaves.items():
            store["leaf/" + key2str(k)] = npy(v)
            store["grad/" + key2str(k)] = npy(v.grad if v.grad is not None else torch.zeros_like(v))
        for f in (-1, 1):
            T = outputs[("cam_T_cam", 0, f)]
            store["out/cam_T_cam|0|{}".format(f)] = npy(T)
            store["grad/cam_T_cam|0|{}".format(f)] = npy(T.grad if T.grad is not None else torch.zeros_like(T))
        keep_scales = (0, scales[-1])
        for k, v in outputs.items():
            if not torch.is_tensor(v) or isinstance(k, str):
                continue
            name, f, s = k
            if name in ("color", "sample", "residual_flow") and s in keep_scales and f != 0:
                store["out/" + key2str(k)] = npy(v)
            if name in ("sample_ego", "sample_complete") and s == scales[-1] and f != 0:
                store["out/" + key2str(k)] = npy(v)
            if name in ("depth",) and s in keep_scales:
                store["out/" + key2str(k)] = npy(v)
        for s in scales:
            k = "identity_selection/{}".format(s)
            if k in outputs:
                store["out/" + k] = npy(outputs[k]).astype(np.uint8)
            if s in noise:
                store["noise/{}".format(s)] = npy(noise[s])
            if s in rand_idx:
                store["rand_idx/{}".format(s)] = rand_idx[s].astype(np.int32)
        path = os.path.join(HERE, "loss_{}.npz".format(phase))
        np.savez_compressed(path, **store)
        print("   wrote", path, os.path.getsize(path) // 1024, "KiB")


# ------------------------------------------------------------------------------------------------
def metrics_inputs(seed=5, B=3, H=96, W=320, M=4000):
    """Seeded DepthMetrics case: smooth disparity, LiDAR-like (row, col, depth) lists with padding, two ground-truth sizes."""
    g = torch.Generator().manual_seed(seed)
    coarse = torch.rand(B, 1, H // 8, W // 8, generator=g)
    disp = torch.nn.functional.interpolate(coarse, (H, W), mode="bilinear", align_corners=False) * 0.9 + 0.01   # disp_scaled in (0.01, 0.91)
    dims = torch.tensor([[375, 1242], [370, 1226], [375, 1242]], dtype=torch.int32)[:B]
    lidar = torch.zeros(B, M, 3)
    valid = torch.zeros(B, M)
    for b in range(B):
        n = M - 500 * (b + 1)                                    # the rest is padding
        lidar[b, :n, 0] = torch.randint(0, int(dims[b, 0]), (n,), generator=g).float()
        lidar[b, :n, 1] = torch.randint(0, int(dims[b, 1]), (n,), generator=g).float()
        lidar[b, :n, 2] = torch.rand(n, generator=g) * 90.0      # some below min_depth / above max_depth
        valid[b, :n] = 1.0
    return {"disp": disp, "depth_gt": lidar, "depth_valid": valid, "gt_dim": dims}


def gen_metrics(ref):
    """tools.DepthMetrics (tools.py:6-73) of the unmodified reference on the seeded case above."""
    bound, lo, hi = [0.40810811, 0.99189189, 0.03594771, 0.96405229], 1e-3, 80.0      # options.py eval defaults (KITTI crop)
    case = metrics_inputs()
    dm = ref.tools.DepthMetrics(bound, lo, hi)
    inputs = {k: case[k] for k in ("depth_gt", "depth_valid", "gt_dim")}
    out = dm(inputs, {("disp_scaled", 0, 0): case["disp"]})
    store = {k: npy(v) for k, v in case.items()}
    store["bound"] = np.asarray(bound, np.float64)
    store["depth_range"] = np.asarray([lo, hi], np.float64)
    store["metrics"] = np.asarray([float(out[m]) for m in dm.depth_metric_names], np.float64)
    per = []
    for b in range(case["disp"].shape[0]):
        o = dm({k: v[b:b + 1] for k, v in inputs.items()}, {("disp_scaled", 0, 0): case["disp"][b:b + 1]})
        per.append([float(o[m]) for m in dm.depth_metric_names])
    store["per_sample"] = np.asarray(per, np.float64)
    path = os.path.join(HERE, "depth_metrics.npz")
    np.savez_compressed(path, **store)
    print("   wrote", path, os.path.getsize(path) // 1024, "KiB;", dict(zip(dm.depth_metric_names, store["metrics"].round(5))))


def gen_metrics_masked(ref):
    """The mask branch of tools.DepthMetrics (tools.py:23-25,58-72) of the unmodified reference: same seeded case, a blocky uint8
    label image in ground-truth pixels (labels 0, 1, 2, 5; label 7 occurs in the image but under no kept LiDAR point)."""
    z = np.load(os.path.join(HERE, "depth_metrics.npz"))
    bound, (lo, hi) = [float(v) for v in z["bound"]], [float(v) for v in z["depth_range"]]
    inputs = {k: torch.from_numpy(z[k]) for k in ("depth_gt", "depth_valid", "gt_dim")}
    disp = torch.from_numpy(z["disp"])
    B = disp.shape[0]
    gh, gw = int(z["gt_dim"][0, 0]), int(z["gt_dim"][0, 1])
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    base = ((xs // 97) + 2 * (ys // 61)) % 4
    base = torch.where(base == 3, torch.full_like(base, 5), base)
    mask = torch.stack([torch.roll(base, 13 * b, 1) for b in range(B)]).to(torch.uint8)
    mask[:, :2, :] = 7                                   # rows above the evaluation crop: a label without any kept point
    dm = ref.tools.DepthMetrics(bound, lo, hi)
    out = dm(inputs, {("disp_scaled", 0, 0): disp}, mask=mask)
    store = {"mask": mask.numpy(), "labels": np.asarray(sorted(out["de:abs_rel_mask"].keys()), np.int64),
             "metrics": np.asarray([float(out[m]) for m in dm.depth_metric_names], np.float64)}
    for m in dm.depth_metric_names:
        d = out[m + "_mask"]
        store["mask/" + m] = np.asarray([[d[l][0], d[l][1]] for l in sorted(d.keys())], np.float64)
    path = os.path.join(HERE, "depth_metrics_masked.npz")
    np.savez_compressed(path, **store)
    print("   wrote", path, os.path.getsize(path) // 1024, "KiB; labels", store["labels"], store["mask/de:abs_rel"])


if __name__ == "__main__":
    what = sys.argv[1:] or ["ops", "loss"]
    torch.set_num_threads(8)
    ref = _refshim.import_reference()
    if "ops" in what:
        gen_ops(ref)
    if "loss" in what:
        gen_loss(ref)
    if "metrics" in what:
        gen_metrics(ref)
    if "metrics_masked" in what:
        gen_metrics_masked(ref)
    if "net" in what:
        import make_golden_net
        make_golden_net.gen_net(ref)
    if "litemono_train" in what:
        import make_golden_net
        make_golden_net.gen_litemono_train(ref)
