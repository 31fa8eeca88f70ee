"""dd_conv3x3_mfma beside the library's fp32 convolution on the motion decoders' / encoders' shapes (B=12, KITTI 192x640): forward and
data gradient, us per call and TFLOP/s (2 * pixels * 9 * cin * cout), with the error of both against float64 on a small batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd")); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from hipops import lib as L
from hipops.functions import mfma_conv, _p, _ws_bytes, _dense_nhwc, _nhwc_empty
torch.backends.cudnn.benchmark = True

def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

shapes = [(12, 64, 64, 96, 320), (12, 72, 64, 96, 320), (12, 64, 64, 48, 160), (12, 72, 64, 48, 160), (12, 128, 128, 24, 80), (12, 136, 128, 24, 80),
          (12, 256, 256, 12, 40), (24, 64, 64, 48, 160), (24, 128, 128, 24, 80), (36, 32, 32, 96, 320), (36, 16, 16, 192, 640)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
lib = L.load()
NP = int(os.environ.get("DD_MFMA_PRODUCTS", "6"))          # partial products per multiply-add: 6 (fp32 accuracy), 3 (bf16x3), 1 (bf16 operands)
print("partial products per multiply-add:", NP)
print("%-28s %10s %10s %8s %8s | %10s %10s %8s %8s" % ("B,cin,cout,H,W", "own fwd", "lib fwd", "own TF", "lib TF", "own dgrad", "lib dgrad", "own TF", "lib TF"))
for (B, cin, cout, H, W) in shapes:
    x = torch.randn(B, cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device="cuda")
    g = torch.randn(B, cout, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    pf = torch.empty(_ws_bytes("dd_conv3x3_mfma_pack_bytes", cout, cin) // 4, device="cuda")
    pb = torch.empty(_ws_bytes("dd_conv3x3_mfma_pack_bytes", cin, cout) // 4, device="cuda")
    sw = w.stride()
    st = L.current_stream()
    L.check(lib.dd_conv3x3_mfma_pack(_p(w), sw[0], sw[1], sw[2], sw[3], cout, cin, _p(pf), _p(pb), st), "pack")
    xd, gd = _dense_nhwc(x), _dense_nhwc(g)
    y = _nhwc_empty(B, cout, H, W, x.device); gx = _nhwc_empty(B, cin, H, W, x.device)
    t_pack = timed(lambda: lib.dd_conv3x3_mfma_pack(_p(w), sw[0], sw[1], sw[2], sw[3], cout, cin, _p(pf), _p(pb), st))
    t_of = timed(lambda: lib.dd_conv3x3_mfma_n(_p(xd), _p(pf), _p(b), B, H, W, cin, cout, 1, NP, _p(y), st))
    t_lf = timed(lambda: F.conv2d(x, w, b, padding=1))
    t_ob = timed(lambda: lib.dd_conv3x3_mfma_n(_p(gd), _p(pb), None, B, H, W, cout, cin, 1, NP, _p(gx), st))
    t_lb = timed(lambda: torch.ops.aten.convolution_backward(g, x, w, None, (1, 1), (1, 1), (1, 1), False, [0, 0], 1, (True, False, False)))
    t_lw = timed(lambda: torch.ops.aten.convolution_backward(g, x, w, None, (1, 1), (1, 1), (1, 1), False, [0, 0], 1, (False, True, False)))
    flat = torch.empty(cout * 9 * cin, device="cuda")
    nb = int(lib.dd_conv3x3_mfma_wgrad_workspace_bytes(B, H, W, cin, cout)); wsw = torch.empty(nb // 4, device="cuda")
    t_ow = timed(lambda: lib.dd_conv3x3_mfma_bwd_weight_n(_p(xd), _p(gd), B, H, W, cin, cout, 1, NP, _p(flat), _p(wsw), nb, st)) if cout % 4 == 0 else float("nan")
    fl = 2.0 * B * H * W * 9 * cin * cout
    tf = lambda us: fl / us * 1e-6
    print("%-28s %8.1fus %8.1fus %8.1f %8.1f | %8.1fus %8.1fus %8.1f %8.1f   pack %.1fus  wgrad own %.1fus (%.1f TF) lib %.1fus (%.1f TF)" % (
        (B, cin, cout, H, W), t_of, t_lf, tf(t_of), tf(t_lf), t_ob, t_lb, tf(t_ob), tf(t_lb), t_pack, t_ow, tf(t_ow), t_lw, tf(t_lw)))
    # accuracy on one image
    x1, g1 = x[:1], g[:1]
    ref = F.conv2d(x1.double(), w.double(), b.double(), padding=1)
    own = mfma_conv(x1, w, b, 1)
    l32 = F.conv2d(x1, w, b, padding=1)
    print("    max error / max |y| against float64: own %.2e, library fp32 %.2e" % (float((own.double() - ref).abs().max() / ref.abs().max()),
                                                                                     float((l32.double() - ref).abs().max() / ref.abs().max())))
