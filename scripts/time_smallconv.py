"""dd_conv3x3_small_* vs MIOpen on the full-resolution 9-channel convs of the motion decoders (forward + backward)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd")); sys.path.insert(0, ROOT)
import bench  # noqa: F401  (MIOpen environment)
import torch
from hipops.functions import ConvBiasFn
torch.backends.cudnn.benchmark = True
for cin in (12, 10, 9):
    x = torch.randn(12, cin, 192, 640, device="cuda").to(memory_format=torch.channels_last).requires_grad_()
    w = (torch.randn(9, cin, 3, 3, device="cuda") * 0.2).to(memory_format=torch.channels_last).requires_grad_()
    b = torch.randn(9, device="cuda").requires_grad_()
    g = torch.randn(12, 9, 192, 640, device="cuda").to(memory_format=torch.channels_last)
    for stock in ("1", "0"):
        os.environ["DD_STOCK_SMALL_CONV"] = stock
        def step():
            ConvBiasFn.apply(x, w, b, (1, 1), (1, 1), (1, 1), 1).backward(g)
            x.grad = w.grad = b.grad = None
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            step()
        e1.record(); torch.cuda.synchronize()
        print("cin=%2d  %s  %8.1f us per fwd+bwd" % (cin, "MIOpen" if stock == "1" else "dd_conv3x3_small", e0.elapsed_time(e1) / 20 * 1e3))
