"""Per-network forward / backward time and kernel mix (GEMM+conv vs everything else) in isolation, B=12 192x640 fp32 NHWC."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd")); sys.path.insert(0, ROOT)
import bench  # noqa: F401  (sets the MIOpen environment)
import torch
from torch.profiler import profile, ProfilerActivity
from options import DynamoOptions
import networks
torch.backends.cudnn.benchmark = True
opt = DynamoOptions().parse(args=["-d", "kitti", "--depth_model", "litemono", "-b", "12", "--weights_init", "scratch", "--synthetic",
                                  "--num_workers", "0", "--log_dir", "/tmp/dd_probe", "--no_train_vis", "--channels_last"])
m = networks.Model(opt).cuda().to(memory_format=torch.channels_last)
m.set_train()
g = torch.Generator(device="cuda").manual_seed(0)
img = lambda c: torch.rand(12, c, 192, 640, device="cuda", generator=g)


def total(outs):
    if isinstance(outs, dict):
        outs = list(outs.values())
    if isinstance(outs, (list, tuple)):
        return sum(total(o) for o in outs)
    return outs.float().square().mean()


def depth():
    return total(m.depth_dec(m.depth_enc(img(3))))


def depth_enc_only():
    return total(m.depth_enc(img(3)))


def pose():
    return total(m.pose_dec([m.pose_enc(img(6))]))


def motion_enc_only():
    return total(m.motion_enc(img(9)))


def motion():
    x = img(9)
    feats = [x] + m.motion_enc(x)
    ego = torch.randn(12, 6, 1, 1, device="cuda", generator=g) * 0.01
    return total(m.motion_dec(feats, ego)) + total(m.motion_mask(feats, ego))


GEMM = ("igemm", "Cijk", "_ZN2ck", "naive_conv", "ck::")
for name, fn in (("depth enc+dec", depth), ("depth enc", depth_enc_only), ("pose enc+dec", pose), ("motion enc", motion_enc_only),
                 ("motion enc + 2 dec", motion)):
    for _ in range(4):
        fn().backward()
    m.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        loss = fn()
        torch.cuda.synchronize()
        mark = sum(e.device_time_total for e in prof.key_averages()) if False else None
        loss.backward()
        torch.cuda.synchronize()
    rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0]
    gemm = sum(t for k, c, t in rows if any(s in k for s in GEMM))
    rest = sum(t for k, c, t in rows if not any(s in k for s in GEMM))
    n = sum(c for k, c, t in rows)
    print("%-20s fwd+bwd: %5d kernels, gemm/conv %7.2f ms, other %7.2f ms" % (name, n, gemm / 1e3, rest / 1e3))
    rows.sort(key=lambda r: -r[2])
    for k, c, t in [r for r in rows if not any(s in r[0] for s in GEMM)][:int(os.environ.get("DD_PROBE_ROWS", "12"))]:
        print("      %7.1f us %4d  %s" % (t, c, k[:140]))
    m.zero_grad(set_to_none=True)

if os.environ.get("DD_PROBE_OPS"):
    fn = {"depth_enc": depth_enc_only, "depth": depth, "pose": pose, "motion": motion}[os.environ["DD_PROBE_OPS"]]
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        fn().backward()
        torch.cuda.synchronize()
    from collections import defaultdict
    agg = defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_time_total > 0 and e.name.startswith("aten::") and not any(c.device_time_total > 0 and c.name.startswith("aten::") for c in e.cpu_children):
            chain, p = [], e.cpu_parent
            while p is not None and len(chain) < 2:
                if not p.name.startswith("aten::"):
                    chain.append(p.name.replace("autograd::engine::evaluate_function: ", "bwd:"))
                p = p.cpu_parent
            a = agg[(e.name, chain[0] if chain else "fwd", str(e.input_shapes[:3])[:90])]
            a[0] += 1; a[1] += e.device_time_total
    print("---- leaf aten ops of", os.environ["DD_PROBE_OPS"])
    for (n, par, shp), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("DD_PROBE_ROWS", "12")) * 5]:
        if any(s in n for s in ("convolution", "::mm", "addmm", "bmm")) != bool(os.environ.get("DD_PROBE_CONV")):
            continue
        print("  %7.1f us %3d %-28s %-36s %s" % (t, c, n, par[:36], shp))
