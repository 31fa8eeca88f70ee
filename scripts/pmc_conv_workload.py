"""A few launches of dd_conv3x3_mfma (forward) and dd_conv3x3_mfma_bwd_weight at the motion decoders' half-resolution shape, for the
rocprofv3 --pmc passes of scripts/pmc_conv.sh."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd")); sys.path.insert(0, ROOT)
import torch
from hipops import lib as L
from hipops.functions import _p, _ws_bytes, _dense_nhwc, _nhwc_empty
lib = L.load()
B, cin, cout, H, W = 12, 64, 64, 96, 320
x = _dense_nhwc(torch.randn(B, cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last))
g = _dense_nhwc(torch.randn(B, cout, H, W, device="cuda").contiguous(memory_format=torch.channels_last))
w = torch.randn(cout, cin, 3, 3, device="cuda") / 24
b = torch.randn(cout, device="cuda")
pf = torch.empty(_ws_bytes("dd_conv3x3_mfma_pack_bytes", cout, cin) // 4, device="cuda")
sw = w.stride(); st = L.current_stream()
L.check(lib.dd_conv3x3_mfma_pack(_p(w), sw[0], sw[1], sw[2], sw[3], cout, cin, _p(pf), None, st), "pack")
y = _nhwc_empty(B, cout, H, W, x.device)
flat = torch.empty(cout * 9 * cin, device="cuda")
nb = int(lib.dd_conv3x3_mfma_wgrad_workspace_bytes(B, H, W, cin, cout)); ws = torch.empty(nb // 4, device="cuda")
for _ in range(int(os.environ.get("DD_PMC_REPS", "6"))):
    L.check(lib.dd_conv3x3_mfma(_p(x), _p(pf), _p(b), B, H, W, cin, cout, 1, _p(y), st), "fwd")
    L.check(lib.dd_conv3x3_mfma_bwd_weight(_p(x), _p(g), B, H, W, cin, cout, 1, _p(flat), _p(ws), nb, st), "wgrad")
torch.cuda.synchronize()
print("done")
