"""The ResNet encoders' small-spatial 3x3 convolutions (12x40 and 6x20 images, 256 / 512 channels: ~30 convolutions x three passes of the
headline step, most of what MIOpen still runs): library time per pass in NHWC (what the step uses) and in NCHW (where MIOpen's fp32
Winograd solvers apply), MIOpen Find on."""
import os
import sys
os.environ.setdefault("MIOPEN_FIND_MODE", "NORMAL")
os.environ.setdefault("MIOPEN_LOG_LEVEL", "2")
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


for (B, C, H, W) in [(12, 512, 6, 20), (12, 256, 12, 40), (24, 256, 12, 40), (12, 128, 24, 80)]:
    for fmt in ("nhwc", "nchw"):
        mf = torch.channels_last if fmt == "nhwc" else torch.contiguous_format
        x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=mf).requires_grad_()
        w = (torch.randn(C, C, 3, 3, device="cuda") / (9 * C) ** 0.5).contiguous(memory_format=mf).requires_grad_()
        y = F.conv2d(x, w, None, 1, 1)
        g = torch.randn_like(y)
        t_f = timed(lambda: F.conv2d(x, w, None, 1, 1))
        t_d = timed(lambda: torch.ops.aten.convolution_backward(g, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (True, False, False)))
        t_w = timed(lambda: torch.ops.aten.convolution_backward(g, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False)))
        fl = 2.0 * B * H * W * C * C * 9
        print("%-22s %s  fwd %6.1f us (%5.1f TF)  dgrad %6.1f us  wgrad %6.1f us" % ((B, C, H, W), fmt, t_f, fl / t_f / 1e6, t_d, t_w), flush=True)
