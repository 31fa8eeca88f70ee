"""A MEASURED yardstick for the step-level parity tests (tests/test_trainer_gpu.py): this tree's networks on the CPU + the oracle
loss on the tiny_kitti golden batch, once in float64 and once in float32, same weights, same RANSAC draws / tie-break noise as the
reference goldens (net_tiny_kitti.npz, net_litemono_train.npz).  The float64 run is the yardstick's zero; |float32 - float64| is what
fp32 arithmetic costs on this step -- per loss term and per network's gradient norm -- and the GPU step is held to a small multiple of
THAT instead of a fixed per-cent tolerance.  Needs neither the reference nor a GPU:   python tests/golden/make_golden_yardstick.py"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (HERE, os.path.join(ROOT, "tests"), ROOT, os.path.join(ROOT, "dynamo-depth_amd")):
    sys.path.insert(0, p)
import oracle.ref_loss as orc  # noqa: E402
from fill import fill_state  # noqa: E402
from test_networks import batch_from_golden, make_opt  # noqa: E402

STEP, STEPS_PER_EPOCH = 50, 100


def one(depth_model, phase, dtype, z, zl):
    import networks
    opt = make_opt(depth_model, ["--synthetic"])
    model = networks.Model(opt)
    for name in sorted(model.module_names):
        fill_state(getattr(model, name), seed=3)
    for m in model.modules():
        if type(m).__name__ == "DropPath":
            m.drop_prob = 0.0
    if getattr(getattr(model, "depth_enc", None), "_drop_layers", None) is not None:
        model.depth_enc._drop_layers = None
    model.to(dtype)
    for m in model.modules():                 # LiteMono's fixed sin/cos grid is built in float32: hand it over in the run's dtype
        if type(m).__name__ == "PositionalEncodingFourier":
            m.features = (lambda f: (lambda *a, **k: f(*a, **k).to(dtype)))(m.features)
    cmpflow, motmask, nets, automask = orc.PHASES[phase]
    model.bool_CmpFlow, model.bool_MotMask = cmpflow, motmask
    model.set_train()
    params = set(id(p) for p in model.parameters_by_names(list(nets)))
    for p in model.parameters():
        p.requires_grad_(id(p) in params)
    inputs = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch_from_golden(z, opt.scales).items()}
    for s in opt.scales:
        if s:
            inputs[("color", 0, s)] = F.interpolate(inputs[("color", 0, s - 1)], (opt.height >> s, opt.width >> s), mode="bicubic",
                                                   align_corners=False, antialias=True).clamp(0, 1)
    base = {k[2:]: v for k, v in vars(opt).items() if k[:2] == "g_"}
    cfg = orc.LossConfig(opt.height, opt.width, opt.scales, coefs=orc.ramped_coefs(base, opt.weight_ramp, opt.ramp_red, STEP, STEPS_PER_EPOCH))
    noise = rand_idx = None
    if phase == "disp_init":
        torch.manual_seed(77)
        noise = {s: torch.randn(2, 2, opt.height, opt.width).to(dtype) for s in opt.scales}
    else:
        rand_idx = {s: zl["{}/fine_tune/rand_idx|{}".format(depth_model, s)] for s in opt.scales}
    outputs = model(inputs)
    grid32 = orc.pixel_grid
    orc.pixel_grid = lambda *a, **k: grid32(*a, **k).to(dtype)       # (the oracle builds its pixel grid in float32: tests/photo_case.py does the same)
    try:
        losses = orc.loss_path(cfg, inputs, outputs, phase, noise=noise, rand_idx=rand_idx)
    finally:
        orc.pixel_grid = grid32
    losses["loss"].backward()
    out = {"losses/" + k: float(v) for k, v in losses.items()}
    vecs = {}
    for name in sorted(model.module_names):
        gs = [p.grad.double().flatten() for p in getattr(model, name).parameters() if p.grad is not None]
        vecs[name] = torch.cat(gs) if gs else torch.zeros(0, dtype=torch.float64)
        out["gradnorm|" + name] = float(vecs[name].norm())
    return out, vecs


def main():
    torch.set_num_threads(8)
    z = np.load(os.path.join(HERE, "net_tiny_kitti.npz"))
    store = {}
    for depth_model, zl in (("monodepthv2", z), ("litemono", np.load(os.path.join(HERE, "net_litemono_train.npz")))):
        for phase in ("disp_init", "fine_tune"):
            r64, v64 = one(depth_model, phase, torch.float64, z, zl)
            r32, v32 = one(depth_model, phase, torch.float32, z, zl)
            pfx = "{}/{}/".format(depth_model, phase)
            # the distance of the gradient VECTORS: a norm is one number and |norm32 - norm64| can be small by cancellation (MonoDepth2
            # fine_tune, motion decoder: 1.6e-5 of the norm against 1e-3 between the vectors) -- the vectors' distance bounds the norm's
            # error (| |a| - |b| | <= |a - b|) and is the stable measure of what fp32 arithmetic costs on this gradient
            for name in v64:
                store[pfx + "gradvec_dist|" + name] = np.float64(float((v32[name] - v64[name]).norm()))
                print("%-55s |g32 - g64| %.3e  (%.2e of |g64|)" % (pfx + "gradvec_dist|" + name, store[pfx + "gradvec_dist|" + name],
                                                                  store[pfx + "gradvec_dist|" + name] / max(r64["gradnorm|" + name], 1e-30)))
            for k in r64:
                store[pfx + "f64/" + k], store[pfx + "f32/" + k] = np.float64(r64[k]), np.float64(r32[k])
                ref = zl.get(pfx + k) if hasattr(zl, "get") else (zl[pfx + k] if pfx + k in zl.files else None)
                print("%-55s f64 %.9g  f32-f64 %+.2e  reference-f64 %s" % (pfx + k, r64[k], r32[k] - r64[k],
                                                                            "%+.2e" % (float(ref) - r64[k]) if ref is not None else "-"))
    path = os.path.join(HERE, "yardstick_step.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
