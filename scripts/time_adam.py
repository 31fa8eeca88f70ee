"""dd_adam_multi alone: the parameter tensors of the four networks (shapes of the headline configuration), random gradients;
HIP events around 50 updates, against torch's fused multi-tensor Adam on the same tensors."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd")); sys.path.insert(0, ROOT)
import torch
from options import DynamoOptions
from networks.model import Model
from hipops.adam import MultiTensorAdam

opt = DynamoOptions().parse(args=["-d", "kitti", "--depth_model", "litemono", "-b", "12", "--weights_init", "scratch", "--synthetic", "--log_dir", "/tmp/dd_adam"])
model = Model(opt).cuda()
params = [p for p in model.parameters() if p.requires_grad]
n = sum(p.numel() for p in params)
flat = torch.randn(sum((p.numel() + 3) & ~3 for p in params), device="cuda") * 1e-3
off = 0
for p in params:
    p.grad = flat[off:off + p.numel()].as_strided(p.size(), p.stride()); off += (p.numel() + 3) & ~3
adam = torch.optim.Adam(params, 1e-4, capturable=True, fused=True)
adam.step()
mt = MultiTensorAdam(adam)


def time(fn, it=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


if os.environ.get("DD_PMC") == "1":              # scripts/pmc_adam.sh: a calibration kernel of known traffic, then a few updates
    a = torch.randn(64 << 20, device="cuda"); b = torch.empty_like(a)
    for _ in range(3):
        torch.atan(a, out=b)                       # 256 MiB read, 256 MiB written per launch (a kernel nothing else in this process uses)
    for _ in range(6):
        mt.step()
    torch.cuda.synchronize()
    print("calibration: atan over %d floats; %d tensors, %d parameters" % (a.numel(), len(params), n))
    sys.exit(0)
t_mine, t_torch = time(mt.step), time(adam.step)
gb = n * 28 / 1e9
print("%d tensors, %.1f M parameters, %.2f GB per update" % (len(params), n / 1e6, gb))
print("dd_adam_multi      %.3f ms  %.2f TB/s" % (t_mine, gb / t_mine))
print("torch fused Adam   %.3f ms  %.2f TB/s  (host-issued, %d tensors)" % (t_torch, gb / t_torch, len(params)))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    mt.step()
print("dd_adam_multi replayed %.3f ms" % time(g.replay))
