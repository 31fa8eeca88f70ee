"""Networks of this tree vs the reference's (goldens from tests/golden/make_golden_net.py, identical key-addressed
weights): state_dict contract, eval-mode outputs on tiny_kitti and the DepthMetrics (Abs Rel) criterion.  CPU."""
import os

import numpy as np
import pytest
import torch

from fill import fill_state


@pytest.fixture(scope="module")
def z(golden_dir):
    return np.load(os.path.join(golden_dir, "net_tiny_kitti.npz"))


def make_opt(depth_model, extra=()):
    """Options of a test run.  The GPU defaults (channels-last, multi-stream, per-network hipGraphs) are switched OFF unless the
    test asks for them by flag: every test states the configuration it checks."""
    from options import DynamoOptions
    extra = list(extra)
    for on, off in (("--hip_graph", "--no_hip_graph"), ("--channels_last", "--nchw"), ("--multi_stream", "--single_stream")):
        if on not in extra and off not in extra:
            extra.append(off)
    opt = DynamoOptions().parse(args=["-d", "kitti", "--depth_model", depth_model, "-b", "2", "--weights_init", "scratch",
                                      "--num_workers", "0", "--log_dir", "/tmp/dd_test_logs"] + extra)
    opt.print_opt = False
    return opt


def batch_from_golden(z, scales):
    inputs = {}
    for f in (0, -1, 1):
        img = torch.from_numpy(z["in/color|{}".format(f)]).float().div(255)
        inputs[("color", f, 0)] = img
        inputs[("color_aug", f, 0)] = img
        inputs[("ts", f)] = torch.ones(img.shape[0], dtype=torch.int64)
    for s in range(len(scales)):
        K = torch.from_numpy(z["in/K|{}".format(s)])
        inputs[("K", s)] = K
        inputs[("inv_K", s)] = torch.from_numpy(np.stack([np.linalg.pinv(k) for k in K.numpy()]))
    inputs["depth_gt"] = torch.from_numpy(z["in/depth_gt"])
    inputs["depth_valid"] = torch.from_numpy(z["in/depth_valid"]).float()
    inputs["gt_dim"] = torch.from_numpy(z["in/gt_dim"])
    return inputs


def compare_summary(z, prefix, outputs, rtol, atol, report):
    fails = []
    for name in z.files:
        if not name.startswith(prefix) or "/losses/" in name or "gradnorm" in name or "rand_idx" in name or name.endswith("metrics"):
            continue
        rest = name[len(prefix):]
        kind = None
        if rest.startswith("stat|"):
            kind, rest = "stat", rest[5:]
        elif rest.startswith("sub|"):
            kind, rest = "sub", rest[4:]
        key = tuple(int(p) if p.lstrip("-").isdigit() else p for p in rest.split("|"))
        v = outputs[key].detach().float().cpu()
        if kind == "stat":
            got = np.array([v.mean().item(), v.std().item(), v.min().item(), v.max().item()])
        elif kind == "sub":
            got = v[:, :, ::8, ::8].numpy()
        else:
            got = v.numpy()
        want = z[name]
        err = np.abs(got - want).max()
        ok = np.all(np.abs(got - want) <= atol + rtol * np.abs(want))
        report.append("%-52s max|err| %.2e %s" % (name, err, "" if ok else "<-- FAIL"))
        if not ok:
            fails.append(name)
    return fails


@pytest.mark.parametrize("depth_model", ["monodepthv2", "litemono"])
def test_eval_outputs_and_abs_rel_match_reference(z, depth_model):
    import networks
    from tools import DepthMetrics
    opt = make_opt(depth_model)
    model = networks.Model(opt)
    for name in sorted(model.module_names):
        fill_state(getattr(model, name), seed=3)
    model.set_eval()
    inputs = batch_from_golden(z, opt.scales)
    with torch.no_grad():
        outputs = model(inputs)
    report = []
    fails = compare_summary(z, depth_model + "/eval/", outputs, 2e-4, 2e-6, report)
    # Abs Rel criterion of the north star: within +-0.002 of the reference on identical weights
    lo, hi = 1 / opt.max_depth, 1 / opt.min_depth
    outputs[("disp_scaled", 0, 0)] = lo + (hi - lo) * outputs[("disp", 0, 0)]
    metrics = DepthMetrics(opt.eval_img_bound, opt.eval_min_depth, opt.eval_max_depth)(inputs, outputs)
    got = np.array([float(metrics[m]) for m in ["de:abs_rel", "de:sq_rel", "de:rms", "de:log_rms", "da:a1", "da:a2", "da:a3"]])
    want = z[depth_model + "/eval/metrics"]
    report.append("metrics got %s\n        want %s" % (got, want))
    print("\n".join(report))
    assert not fails, fails
    assert abs(got[0] - want[0]) < 0.002
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-4)


def test_state_dict_contract():
    """Key names / shapes of SURVEY.md Appendix E (what reference checkpoints contain)."""
    import networks
    m = networks.Model(make_opt("monodepthv2"))
    sd = m.pose_enc.state_dict()
    assert sd["encoder.conv1.weight"].shape == (64, 6, 7, 7) and "encoder.fc.weight" in sd and len(sd) == 122
    assert m.motion_enc.state_dict()["encoder.conv1.weight"].shape == (64, 9, 7, 7)
    assert len(m.depth_dec.state_dict()) == 28 and m.depth_dec.state_dict()["upconv_4_0.conv.conv.weight"].shape == (256, 512, 3, 3)
    pd = m.pose_dec.state_dict()
    assert len(pd) == 16 and torch.equal(pd["squeeze.weight"], pd["net.0.weight"])
    md = m.motion_dec.state_dict()
    assert len(md) == 38 and md["refine_motion_conv5.0.weight"].shape == (9, 12, 3, 3) and md["_residual_translation.weight"].shape == (3, 6, 1, 1)
    lite = networks.Model(make_opt("litemono"))
    ls = lite.depth_enc.state_dict()
    assert len(ls) == 263 and ls["downsample_layers.1.0.conv.weight"].shape == (128, 131, 3, 3) and ls["stages.0.3.xca.temperature"].shape == (8, 1, 1)
    assert len(lite.depth_dec.state_dict()) == 18 and lite.depth_dec.state_dict()["decoder.1.conv.conv.weight"].shape == (112, 240, 3, 3)
    n_md2 = sum(p.numel() for p in m.parameters())
    n_lite = sum(p.numel() for p in lite.parameters())
    assert abs(n_md2 / 1e6 - 52.3) < 0.1 and abs(n_lite / 1e6 - 46.2) < 0.1


def test_checkpoint_roundtrip(tmp_path):
    import networks
    opt = make_opt("litemono")
    a = networks.Model(opt)
    a.save(str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == sorted(n + ".pth" for n in a.module_names)
    assert "height" in torch.load(os.path.join(tmp_path, "depth_enc.pth"))
    opt.load_ckpt = str(tmp_path)
    b = networks.Model(opt)
    b.load(verbose=False)
    for n in a.module_names:
        for k, v in getattr(a, n).state_dict().items():
            assert torch.equal(v, getattr(b, n).state_dict()[k])


def test_options_surface():
    from options import DynamoOptions
    o = DynamoOptions().parse(args=["-d", "waymo"])
    assert (o.height, o.width, o.split, o.scales, o.epoch_size, o.batch_size) == (320, 480, "waymo", [0, 1, 2], 8000, 3)
    assert [k for k in vars(o) if k.startswith("g_")] == ["g_p_photo", "g_d_smooth", "g_d_ground", "g_c_smooth", "g_c_consistency", "g_m_sparsity", "g_m_smooth"]
    o = DynamoOptions().parse(args=["-d", "kitti", "--depth_model", "monodepthv2"])
    assert o.scales == [0, 1, 2, 3] and o.eval_img_ext == ".png" and o.eval_max_depth == 80
    # the fast configuration is the default and every part of it has an off switch (None = resolved by device in Trainer)
    assert (o.hip_graph, o.multi_stream, o.channels_last, o.miopen_find, o.device_preprocess, o.device_decode) == (None, None, None, None, True, True)
    assert (o.loader_start, o.keep_going_on_nan, o.stats_only_side_frames, o.skip_unused_depth_frames) == (None, False, False, False)
    o = DynamoOptions().parse(args=["-d", "kitti", "--no_hip_graph", "--single_stream", "--nchw", "--no_miopen_find", "--no_device_decode",
                                    "--loader_start", "fork", "--keep_going_on_nan", "--stats_only_side_frames"])
    assert (o.hip_graph, o.multi_stream, o.channels_last, o.miopen_find, o.device_decode) == (False, False, False, False, False)
    assert (o.loader_start, o.keep_going_on_nan, o.stats_only_side_frames) == ("fork", True, True)


def test_cpu_trainer_resolves_the_gpu_defaults_off():
    """On a machine without a GPU the Trainer turns the device-only defaults off instead of failing (hipGraphs, streams, channels-last)."""
    from Trainer import Trainer
    tr = Trainer(make_opt_defaults())
    if not torch.cuda.is_available():
        assert (tr.opt.hip_graph, tr.opt.multi_stream, tr.opt.channels_last) == (False, False, False)
        assert tr._worker_start() == {}
    assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"          # set at import, before any device call


def make_opt_defaults():
    from options import DynamoOptions
    opt = DynamoOptions().parse(args=["-d", "kitti", "--depth_model", "litemono", "-b", "2", "--weights_init", "scratch", "--synthetic",
                                      "--num_workers", "0", "--log_dir", "/tmp/dd_test_logs", "--height", "64", "--width", "96"])
    opt.print_opt = False
    return opt


def test_batchnorm_host_counter_matches_stock():
    """layers.BatchNorm2d defers `num_batches_tracked` to the host; outputs, running stats and checkpoints equal nn.BatchNorm2d."""
    import torch.nn as nn
    from networks.layers import BatchNorm2d
    torch.manual_seed(3)
    a, b = nn.BatchNorm2d(5), BatchNorm2d(5)
    b.load_state_dict(a.state_dict())
    for _ in range(3):
        x = torch.randn(4, 5, 6, 7)
        assert torch.equal(a(x), b(x))
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert int(sb["num_batches_tracked"]) == 3
    a.eval(); b.eval()
    x = torch.randn(2, 5, 6, 7)
    assert torch.equal(a(x), b(x))
    b.train(); b(x)
    b.load_state_dict(sa)                                  # loading drops counts not yet flushed
    assert int(b.state_dict()["num_batches_tracked"]) == 3


def test_xca_gram_formulation_equals_reference_order():
    """XCA computed from the Gram matrix on the qkv buffer (default) against the reference's permute + normalise order
    (DD_STOCK_XCA=1), outputs and input gradients, float64."""
    import os
    from networks.depth_encoder import XCA
    torch.manual_seed(0)
    for (B, N, Cc, h) in [(2, 96, 64, 8), (3, 40, 224, 8)]:
        m = XCA(Cc, num_heads=h, qkv_bias=True).double()
        with torch.no_grad():
            m.temperature.copy_(torch.rand(h, 1, 1) + 0.5)
        x = torch.randn(B, N, Cc, dtype=torch.double, requires_grad=True)
        outs = {}
        for stock in ("1", "0"):
            os.environ["DD_STOCK_XCA"] = stock
            try:
                y = m(x)
                g, = torch.autograd.grad(y.square().sum(), x)
            finally:
                os.environ.pop("DD_STOCK_XCA", None)
            outs[stock] = (y.detach(), g)
        assert torch.allclose(outs["0"][0], outs["1"][0], rtol=1e-12, atol=1e-12)
        assert torch.allclose(outs["0"][1], outs["1"][1], rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("chans,cout,stride,bias", [((3, 64), 64, 1, True), ((64, 3), 64, 2, False), ((1, 128), 128, 1, True), ((224, 32, 3), 224, 2, False)])
def test_conv_cat_aligned_equals_cat_conv(chans, cout, stride, bias):
    """layers.conv_cat_aligned (zero-padded channel count, channels-last restriding of the small inputs) is conv(cat(parts)):
    output and all gradients, float64, on the CPU with the GPU-only branch forced."""
    import torch.nn as nn
    from networks.layers import conv_cat_aligned
    torch.manual_seed(1)
    conv = nn.Conv2d(sum(chans), cout, 3, stride=stride, padding=1, bias=bias).double()
    parts_a = [torch.randn(2, c, 10, 12, dtype=torch.double).contiguous(memory_format=torch.channels_last if c > 3 else torch.contiguous_format).requires_grad_()
               for c in chans]
    parts_b = [p.detach().clone().requires_grad_() for p in parts_a]
    ya = conv(torch.cat(parts_a, 1))
    yb = conv_cat_aligned(conv, parts_b, force=True)
    assert yb.shape == ya.shape and torch.allclose(ya, yb, rtol=1e-12, atol=1e-12)
    g = torch.randn_like(ya)
    ga = torch.autograd.grad(ya, parts_a + list(conv.parameters()), g)
    gb = torch.autograd.grad(yb, parts_b + list(conv.parameters()), g)
    for a, b in zip(ga, gb):
        assert torch.allclose(a, b, rtol=1e-11, atol=1e-12)


def test_batch_groups_keeps_passes_apart():
    """layers.batch_groups: two passes sent through a BatchNorm layer as one batch give the outputs and the running statistics
    of the two separate passes (stock-operator path; the HIP path is tests/test_ops_gpu.py::test_grouped_batch_norm_...)."""
    import torch
    from networks.layers import BatchNorm2d, DeferredStats, batch_groups
    torch.manual_seed(0)
    xs = [torch.randn(3, 8, 5, 7) for _ in range(2)]
    a, b = BatchNorm2d(8), BatchNorm2d(8)
    b.load_state_dict(a.state_dict())
    a.train(), b.train()
    with torch.no_grad():
        want = torch.cat([a(x) for x in xs])
        with batch_groups(2):
            got = b(torch.cat(xs))
    assert torch.equal(got, want)
    assert torch.equal(a.running_mean, b.running_mean) and torch.equal(a.running_var, b.running_var)
    a.flush_counter(), b.flush_counter()
    assert int(a.num_batches_tracked) == int(b.num_batches_tracked) == 2
    # deferred: the groups' m*stat terms go to their collectors, apply() folds them in order
    c = BatchNorm2d(8)
    c.load_state_dict({k: v for k, v in a.state_dict().items()})
    d = BatchNorm2d(8)
    d.load_state_dict(c.state_dict())
    c.train(), d.train()
    with torch.no_grad():
        for x in xs:
            c(x)
        cols = [DeferredStats("cpu", 64), DeferredStats("cpu", 64)]
        with batch_groups(2, cols):
            d(torch.cat(xs))
    for col in cols:
        for bn, mean, var in col.pairs:
            bn.running_mean.mul_(1 - bn.momentum).add_(mean)
            bn.running_var.mul_(1 - bn.momentum).add_(var)
    assert torch.allclose(c.running_mean, d.running_mean, atol=1e-6) and torch.allclose(c.running_var, d.running_var, atol=1e-6)


def _torchvision_resnet_keys(depth, in_channels):
    """The state_dict of torchvision.models.resnet<depth> (v0.13, the reference's pin; README.md:31) written out from its
    public definition: BasicBlock for 18/34, Bottleneck (expansion 4, stride on conv2) for 50+, stage widths 64..512, a
    down-sampling 1x1 conv + BN in the first block of every stage whose input differs in stride or width, and the ImageNet
    head `fc` that the reference carries along unused.  Returns {key: shape}."""
    blocks = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}[depth]
    bottleneck = depth >= 50
    exp = 4 if bottleneck else 1
    keys = {}

    def bn(prefix, c):
        for name, shape in (("weight", (c,)), ("bias", (c,)), ("running_mean", (c,)), ("running_var", (c,)), ("num_batches_tracked", ())):
            keys["{}.{}".format(prefix, name)] = shape

    keys["conv1.weight"] = (64, in_channels, 7, 7)
    bn("bn1", 64)
    inplanes = 64
    for li, (planes, n) in enumerate(zip((64, 128, 256, 512), blocks), start=1):
        for bi in range(n):
            stride = 2 if (bi == 0 and li > 1) else 1
            pre = "layer{}.{}".format(li, bi)
            if bottleneck:
                keys[pre + ".conv1.weight"] = (planes, inplanes, 1, 1)
                bn(pre + ".bn1", planes)
                keys[pre + ".conv2.weight"] = (planes, planes, 3, 3)
                bn(pre + ".bn2", planes)
                keys[pre + ".conv3.weight"] = (planes * 4, planes, 1, 1)
                bn(pre + ".bn3", planes * 4)
            else:
                keys[pre + ".conv1.weight"] = (planes, inplanes, 3, 3)
                bn(pre + ".bn1", planes)
                keys[pre + ".conv2.weight"] = (planes, planes, 3, 3)
                bn(pre + ".bn2", planes)
            if bi == 0 and (stride != 1 or inplanes != planes * exp):
                keys[pre + ".downsample.0.weight"] = (planes * exp, inplanes, 1, 1)
                bn(pre + ".downsample.1", planes * exp)
            inplanes = planes * exp
    keys["fc.weight"] = (1000, 512 * exp)
    keys["fc.bias"] = (1000,)
    return keys


@pytest.mark.parametrize("depth,images", [(18, 1), (18, 2), (18, 3), (50, 1), (50, 2), (34, 1)])
def test_resnet_state_dict_is_torchvisions(depth, images):
    """torchvision is absent from this image, so the ResNet topology cannot be pinned by executing it:
    the next best thing is the key list and every shape of its public definition, asserted verbatim -- a reference checkpoint
    (`ckpt/*/{depth,pose,motion}_enc.pth`, keys `encoder.<torchvision key>`) then loads with strict=True."""
    from networks.resnet_encoder import ResnetEncoder
    enc = ResnetEncoder(depth, False, num_input_images=images)
    got = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    want = {"encoder." + k: v for k, v in _torchvision_resnet_keys(depth, 3 * images).items()}
    assert list(got) == list(want)            # same keys in the same order (state_dict order = registration order in torchvision)
    assert got == want
    if depth == 18:
        assert len(got) == 122


def test_datasets_survive_a_fork_server():
    """DataLoader workers start from a fork server on a GPU (Trainer._worker_start): the dataset objects are pickled to them, so
    nothing unpicklable (lambdas, open files, device handles) may hang off them -- with or without the device-side input path."""
    import pickle
    from options import DynamoOptions
    from Trainer import Trainer
    for synthetic in (False, True):
        opt = DynamoOptions().parse(args=["-d", "kitti", "--depth_model", "litemono", "-b", "2", "--weights_init", "scratch", "--num_workers", "0",
                                          "--log_dir", "/tmp/dd_test_logs", "--height", "64", "--width", "96", "--data_path", "/tmp/dd_no_such_data"] +
                                    (["--synthetic"] if synthetic else []))
        opt.print_opt = False
        tr = Trainer(opt)
        for kw in ((dict(),) if synthetic else (dict(), dict(device_preprocess=True, device_decode=True), dict(device_preprocess=True, device_decode=False))):
            ds = tr.get_dataset(["2011_09_26/2011_09_26_drive_0001_sync 5 l"], is_train=True, load_depth=False, load_mask=False, **kw)
            clone = pickle.loads(pickle.dumps(ds))
            assert type(clone) is type(ds) and len(clone) == len(ds) and (clone.height, clone.width) == (ds.height, ds.width)


def test_yardstick_float32_run_is_the_reference_step(golden_dir):
    """tests/golden/yardstick_step.npz (this tree's networks on the CPU + the oracle loss, float64 and float32: the yardstick of
    tests/test_trainer_gpu.py) against the goldens of the UNMODIFIED reference's training step: the float32 run IS the reference's step
    -- MonoDepth2 bit for bit at print precision, LiteMono within the reordering noise of its rewritten forward (batched XCA Gram
    product, cached positional features) -- so holding the GPU step to the float64 run in units of |float32 - float64| holds it to
    the reference."""
    y = np.load(os.path.join(golden_dir, "yardstick_step.npz"))
    seen = 0
    for depth_model, zf in (("monodepthv2", "net_tiny_kitti.npz"), ("litemono", "net_litemono_train.npz")):
        z = np.load(os.path.join(golden_dir, zf))
        for phase in ("disp_init", "fine_tune"):
            pfx = "{}/{}/".format(depth_model, phase)
            for name in z.files:
                if not name.startswith(pfx) or not ("/losses/" in name or "gradnorm|" in name):
                    continue
                key = name[len(pfx):]
                ref, f32, f64 = float(z[name]), float(y[pfx + "f32/" + key]), float(y[pfx + "f64/" + key])
                rel = (5e-3 if "gradnorm" in key else 2e-4) if depth_model == "litemono" else (1e-5 if "gradnorm" in key else 1e-6)
                assert abs(f32 - ref) <= rel * max(abs(ref), 1e-6), (name, f32, ref)
                # and float64 is where both float32 runs scatter around: no quantity whose fp32 realisations agree with each other
                # better than with float64 by more than the pose path's known cancellation (5 % of a gradient norm)
                assert abs(f64 - ref) <= 6e-2 * max(abs(ref), 1e-6), (name, f64, ref)
                seen += 1
    assert seen > 60, seen
