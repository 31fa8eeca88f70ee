"""dd_conv3x3_mfma_flat against the library on the small-image levels (forward and data gradient), per split count."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd"))
import torch
import torch.nn.functional as F
from hipops import functions as Fn
torch.backends.cudnn.benchmark = True


def timed(fn, n=40):
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


for (B, C, H, W) in [(12, 512, 6, 20), (24, 512, 6, 20), (12, 256, 12, 40), (24, 256, 12, 40)]:
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_()
    w = (torch.randn(C, C, 3, 3, device="cuda") / (9 * C) ** 0.5).contiguous(memory_format=torch.channels_last).requires_grad_()
    b = torch.randn(C, device="cuda")
    g = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    t_lf = timed(lambda: F.conv2d(x, w, b, 1, 1))
    t_ld = timed(lambda: torch.ops.aten.convolution_backward(g, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (True, False, False)))
    row = "%-20s library fwd %6.1f dgrad %6.1f us |" % ((B, C, H, W), t_lf, t_ld)
    # the C entry point alone (what a replayed step pays per pass; the pack is one launch per convolution and step)
    from hipops import lib as L
    from hipops.functions import _p
    lib = L.load()
    xd = x.detach()
    pack = torch.empty(int(lib.dd_conv3x3_mfma_pack_bytes(C, C)) // 4, device="cuda")
    sw = w.stride()
    L.check(lib.dd_conv3x3_mfma_pack(_p(w.detach()), sw[0], sw[1], sw[2], sw[3], C, C, _p(pack), None, L.current_stream()), "pack")
    y = torch.empty(B, H, W, C, device="cuda")
    for sp in os.environ.get("DD_TRY_SPLITS_C", "1,2,4,5,6,8,10,12,16").split(","):
        os.environ["DD_FLAT_SPLITS"] = sp
        nb = int(lib.dd_conv3x3_mfma_flat_workspace_bytes(B, H, W, C, C))
        ws = torch.empty(nb // 4 + 4, device="cuda")
        t = timed(lambda: lib.dd_conv3x3_mfma_flat(_p(xd), _p(pack), _p(b), B, H, W, C, C, _p(y), _p(ws), nb, L.current_stream()))
        row += " s%s %.1f" % (sp, t)
    os.environ.pop("DD_FLAT_SPLITS", None)
    print(row, flush=True)
    continue
    for sp in os.environ.get("DD_TRY_SPLITS", "auto,2,4,8").split(","):
        if sp == "auto":
            os.environ.pop("DD_FLAT_SPLITS", None)
        else:
            os.environ["DD_FLAT_SPLITS"] = sp
        Fn._WS_BYTES.clear()
        with torch.no_grad():
            t_f = timed(lambda: Fn.mfma_conv(x, w, b, 1))        # includes the weight pack (4-6 us), as the step does
        y = Fn.mfma_conv(x, w, b, 1)
        t_fb = timed(lambda: torch.autograd.grad(Fn.mfma_conv(x, w, b, 1), x, g))
        row += "  splits %-4s fwd(+pack) %6.1f fwd+dgrad(+packs) %6.1f" % (sp, t_f, t_fb)
    print(row, flush=True)
