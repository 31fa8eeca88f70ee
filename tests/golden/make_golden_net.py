"""Network + full process_batch goldens on assets/tiny_kitti, produced by the unmodified reference (CPU)."""
import os

import numpy as np
import torch

import _refshim
from fill import fill_state

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = ["2011_09_26/2011_09_26_drive_0001_sync 1 l", "2011_09_26/2011_09_26_drive_0001_sync 1 r"]
STEP, STEPS_PER_EPOCH = 50, 100


def collate(items):
    out = {}
    for k in items[0]:
        v0 = items[0][k]
        if torch.is_tensor(v0):
            out[k] = torch.stack([it[k] for it in items])
        else:
            out[k] = torch.tensor([it[k] for it in items])
    return out


def load_batch(ref, tr):
    ds = tr.get_dataset(FILES, is_train=False, load_depth=True, load_mask=False)
    return collate([ds[i] for i in range(len(FILES))])


def summarise(store, prefix, outputs, scales):
    for k, v in outputs.items():
        if not (isinstance(k, tuple) and torch.is_tensor(v)):
            continue
        name = "|".join(str(x) for x in k)
        if k[0] in ("axisangle", "translation", "cam_T_cam"):
            store[prefix + name] = v.detach().numpy()
        elif k[0] in ("disp", "complete_flow", "motion_prob", "motion_mask"):
            a = v.detach()
            store[prefix + "stat|" + name] = np.array([a.mean().item(), a.std().item(), a.min().item(), a.max().item()], dtype=np.float64)
            if k[2] == max(scales) and k[1] in (0, 1):
                store[prefix + name] = a.numpy()
            if k[0] == "disp" and k[1] == 0 and k[2] == 0:
                store[prefix + "sub|" + name] = a[:, :, ::8, ::8].numpy()


def gen_net(ref):
    store = {}
    saved_inputs = False
    for model_name in ("monodepthv2", "litemono"):
        opt = _refshim.make_opt(ref, argv=["-d", "kitti", "--depth_model", model_name, "-b", "2"],
                                data_path=os.path.join(_refshim.REFERENCE_ROOT, "assets", "tiny_kitti"))
        tr = ref.Trainer.Trainer(opt)
        for name in sorted(tr.base_model.module_names):
            fill_state(getattr(tr.base_model, name), seed=3)
        tr.num_steps_per_epoch = STEPS_PER_EPOCH
        batch = load_batch(ref, tr)
        if not saved_inputs:
            for f in (0, -1, 1):
                store["in/color|{}".format(f)] = (batch[("color", f, 0)] * 255).round().to(torch.uint8).numpy()
                assert torch.equal(batch[("color", f, 0)], torch.from_numpy(store["in/color|{}".format(f)]).float().div(255))
            store["in/depth_gt"] = batch["depth_gt"].numpy()
            store["in/depth_valid"] = batch["depth_valid"].numpy().astype(np.uint8)
            store["in/gt_dim"] = batch["gt_dim"].numpy()
            for s in range(4):
                store["in/K|{}".format(s)] = batch[("K", s)].numpy() if ("K", s) in batch else np.zeros(0)
            saved_inputs = True
        pfx = model_name + "/"
        # ---- eval-mode forward + depth metrics --------------------------------------------------
        tr.setup_phase("fine_tune")
        tr.set_eval()
        with torch.no_grad():
            inputs = {k: v.clone() for k, v in batch.items()}
            tr.process_inputs(inputs)
            outputs = tr.model(inputs)
            summarise(store, pfx + "eval/", outputs, opt.scales)
            sd, _ = ref.tools.disp_to_depth(outputs[("disp", 0, 0)], opt.min_depth, opt.max_depth)
            outputs[("disp_scaled", 0, 0)] = sd
            metrics = tr.depth_metrics(inputs, outputs)
            store[pfx + "eval/metrics"] = np.array([float(metrics[m]) for m in tr.depth_metrics.depth_metric_names], dtype=np.float64)
            print(model_name, "abs_rel", float(metrics["de:abs_rel"]))
        if model_name != "monodepthv2":
            continue
        # ---- train-mode full step (BatchNorm batch statistics; MD2 has no stochastic depth) ---------
        for phase in ("disp_init", "fine_tune"):
            for name in sorted(tr.base_model.module_names):
                fill_state(getattr(tr.base_model, name), seed=3)       # undo the running-stat updates of the previous pass
            tr.setup_phase(phase)
            tr.bool_automask = phase == "disp_init"
            tr.step = STEP
            tr.set_train()
            inputs = {k: v.clone() for k, v in batch.items()}
            torch.manual_seed(77)
            np.random.seed(78)
            outputs, losses = tr.process_batch(inputs)
            losses["loss"].backward()
            for k, v in losses.items():
                store[pfx + phase + "/losses/" + k] = np.float64(float(v))
            summarise(store, pfx + phase + "/", outputs, opt.scales)
            # disp_init: the tie-break noise is torch.manual_seed(77); randn(2,2,H,W) per scale (CPU generator)
            if phase != "disp_init":
                import oracle.ref_loss as orc
                np.random.seed(78)
                for s in opt.scales:
                    h, w = opt.height // 2 ** s, opt.width // 2 ** s
                    store[pfx + phase + "/rand_idx|{}".format(s)] = orc.ransac_indices(2, int(opt.gp_prior * h) * w, 500).astype(np.int32)
            # a few gradient fingerprints (norms per sub-network)
            for name in sorted(tr.base_model.module_names):
                sq = 0.0
                for p in getattr(tr.base_model, name).parameters():
                    if p.grad is not None:
                        sq += float((p.grad.double() ** 2).sum())
                        p.grad = None
                store[pfx + phase + "/gradnorm|" + name] = np.float64(sq ** 0.5)
            print(model_name, phase, "loss", float(losses["loss"]))
    path = os.path.join(HERE, "net_tiny_kitti.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_litemono_train(ref):
    """LiteMono TRAIN-mode full step (BatchNorm batch statistics, layer scale, XCA, dilated depth-wise convs) by the unmodified
    reference, for the phases the benchmark runs.  Stochastic depth is the one random element of the network
    (networks/depth_encoder.py:202,248; rates up to 0.4, networks/model.py:25): its probability is set to 0 on the reference's
    DropPath modules so that the step is a function of inputs and weights (the GPU test does the same on its side)."""
    import oracle.ref_loss as orc
    store = {}
    opt = _refshim.make_opt(ref, argv=["-d", "kitti", "--depth_model", "litemono", "-b", "2"],
                            data_path=os.path.join(_refshim.REFERENCE_ROOT, "assets", "tiny_kitti"))
    tr = ref.Trainer.Trainer(opt)
    tr.num_steps_per_epoch = STEPS_PER_EPOCH
    batch = load_batch(ref, tr)
    dropped = 0
    for m in tr.base_model.modules():
        if type(m).__name__ == "DropPath":
            m.drop_prob = 0.0
            dropped += 1
    assert dropped > 0
    for phase in ("disp_init", "fine_tune"):
        for name in sorted(tr.base_model.module_names):
            fill_state(getattr(tr.base_model, name), seed=3)
        tr.setup_phase(phase)
        tr.bool_automask = phase == "disp_init"
        tr.step = STEP
        tr.set_train()
        inputs = {k: v.clone() for k, v in batch.items()}
        torch.manual_seed(77)
        np.random.seed(78)
        outputs, losses = tr.process_batch(inputs)
        losses["loss"].backward()
        pfx = "litemono/" + phase + "/"
        for k, v in losses.items():
            store[pfx + "losses/" + k] = np.float64(float(v))
        summarise(store, pfx, outputs, opt.scales)
        if phase != "disp_init":
            np.random.seed(78)
            for s in opt.scales:
                h, w = opt.height // 2 ** s, opt.width // 2 ** s
                store[pfx + "rand_idx|{}".format(s)] = orc.ransac_indices(2, int(opt.gp_prior * h) * w, 500).astype(np.int32)
        for name in sorted(tr.base_model.module_names):
            sq = 0.0
            for p in getattr(tr.base_model, name).parameters():
                if p.grad is not None:
                    sq += float((p.grad.double() ** 2).sum())
                    p.grad = None
            store[pfx + "gradnorm|" + name] = np.float64(sq ** 0.5)
        print("litemono train-mode", phase, "loss", float(losses["loss"]))
    path = os.path.join(HERE, "net_litemono_train.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
