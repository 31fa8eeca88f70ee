"""What the library's half-precision convolutions cost on config 5's shapes (nuScenes MonoDepth2 288x512, B=16: BASELINE configs[4]) --
the 3x3 stride-1 layers of the ResNet-18 encoders and the decoders: forward / data gradient / weight gradient in fp16 and bf16
(channels-last, MIOpen through torch), beside the fp32 layers through dd_conv3x3_mfma and the HBM / matrix-pipe floor of each shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd")); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True


def timed(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = int(os.environ.get("DD_B", 16))
# (cin, cout, H, W): ResNet-18 layer1..4 at 288x512, the depth decoder's Conv3x3 pairs, the motion decoders' refinement convolutions
shapes = [(64, 64, 72, 128), (128, 128, 36, 64), (256, 256, 18, 32), (512, 512, 9, 16),
          (64, 32, 144, 256), (32, 16, 288, 512), (128, 64, 72, 128), (256, 128, 36, 64),
          (64, 64, 144, 256), (72, 64, 144, 256)]
from hipops import lib as L
from hipops.functions import _p, _ws_bytes, _dense_nhwc, DTYPE_CODE
lib = L.load()
print("B=%d  %-22s | %-44s | %-44s | %s" % (B, "cin,cout,H,W", "fp16 lib fwd / dgrad / wgrad | OWN fwd / dgrad / wgrad", "bf16 lib fwd / dgrad / wgrad | OWN fwd / dgrad / wgrad", "floors: HBM at 6 TB/s (half tensors) / MFMA at 2.5 PF (us)"))
for (cin, cout, H, W) in shapes:
    row = []
    for dt in (torch.float16, torch.bfloat16):
        x = torch.randn(B, cin, H, W, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 3, 3, device="cuda", dtype=dt) / (3 * cin ** 0.5)).contiguous(memory_format=torch.channels_last)
        g = torch.randn(B, cout, H, W, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
        tf = timed(lambda: F.conv2d(x, w, None, padding=1))
        tb = timed(lambda: torch.ops.aten.convolution_backward(g, x, w, None, (1, 1), (1, 1), (1, 1), False, [0, 0], 1, (True, False, False)))
        tw = timed(lambda: torch.ops.aten.convolution_backward(g, x, w, None, (1, 1), (1, 1), (1, 1), False, [0, 0], 1, (False, True, False)))
        to = tob = tow = float("nan")
        if lib.dd_conv3x3_half_supported(cin, cout) and H >= 8 and W >= 32:
            w32 = w.float()
            pf = torch.empty(_ws_bytes("dd_conv3x3_half_pack_bytes", cout, cin) // 4, device="cuda")
            pb = torch.empty(_ws_bytes("dd_conv3x3_half_pack_bytes", cin, cout) // 4, device="cuda")
            sw, st, code = w32.stride(), L.current_stream(), DTYPE_CODE[dt]
            L.check(lib.dd_conv3x3_half_pack(_p(w32), sw[0], sw[1], sw[2], sw[3], cout, cin, code, _p(pf), _p(pb), st), "pack")
            xd, gd = _dense_nhwc(x), _dense_nhwc(g)
            y = torch.empty((B, H, W, cout), dtype=dt, device="cuda"); gx = torch.empty((B, H, W, cin), dtype=dt, device="cuda")
            to = timed(lambda: lib.dd_conv3x3_half(_p(xd), _p(pf), None, B, H, W, cin, cout, 1, code, _p(y), st))
            tob = timed(lambda: lib.dd_conv3x3_half(_p(gd), _p(pb), None, B, H, W, cout, cin, 1, code, _p(gx), st))
            tp = timed(lambda: lib.dd_conv3x3_half_pack(_p(w32), sw[0], sw[1], sw[2], sw[3], cout, cin, code, _p(pf), _p(pb), st))
            if min(cin, cout) >= 32:
                flat = torch.empty(cout * 9 * cin, device="cuda")
                nb = _ws_bytes("dd_conv3x3_half_wgrad_workspace_bytes", B, H, W, cin, cout); wsw = torch.empty(nb // 4, device="cuda")
                tow = timed(lambda: lib.dd_conv3x3_half_bwd_weight(_p(xd), _p(gd), B, H, W, cin, cout, 1, code, _p(flat), _p(wsw), nb, st))
        row.append("%6.1f %6.1f %6.1f | %6.1f %6.1f %6.1f " % (tf, tb, tw, to, tob, tow))
    nbytes = B * H * W * (cin + cout) * 2
    flops = 2.0 * B * H * W * 9 * cin * cout
    print("     %-22s | %s| %s| %6.1f / %6.1f   (%.1f GFLOP)" % ((cin, cout, H, W), row[0], row[1], nbytes / 6e12 * 1e6, flops / 2.5e15 * 1e6, flops / 1e9))
