"""dd_depth_metrics vs the per-sample torch loop (the reference's structure) at the KITTI evaluation shape."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd"))
import torch
from tools import DepthMetrics
g = torch.Generator().manual_seed(3)
B, M = 12, 25000
disp = (torch.nn.functional.interpolate(torch.rand(B, 1, 24, 80, generator=g), (192, 640), mode="bilinear") * 0.9 + 0.01).cuda()
lidar = torch.stack([torch.randint(0, 375, (B, M), generator=g).float(), torch.randint(0, 1242, (B, M), generator=g).float(),
                     torch.rand(B, M, generator=g) * 85.0], -1).cuda()
inputs = {"depth_gt": lidar, "depth_valid": (torch.rand(B, M, generator=g) > 0.2).float().cuda(), "gt_dim": torch.tensor([[375, 1242]] * B, dtype=torch.int32).cuda()}
dm = DepthMetrics([0.40810811, 0.99189189, 0.03594771, 0.96405229], 1e-3, 80.0)
out = {("disp_scaled", 0, 0): disp}
for name, fn in (("torch loop (reference structure)", lambda: dm._forward_torch(inputs, out)), ("dd_depth_metrics", lambda: dm(inputs, out))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print("%-34s %8.3f ms per batch of %d (wall, incl. host syncs)" % (name, (time.perf_counter() - t0) / 20 * 1e3, B))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    dm.device_metrics(inputs, disp)
b.record(); torch.cuda.synchronize()
print("dd_depth_metrics device time            %8.1f us per batch" % (a.elapsed_time(b) / 20 * 1e3))
