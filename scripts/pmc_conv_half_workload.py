"""A few launches of dd_conv3x3_half (forward) and dd_conv3x3_half_bwd_weight at config 5's 144x256 level (16 x 64 -> 64 channels, fp16), for
the rocprofv3 --pmc passes of scripts/pmc_conv.sh (DD_PMC_WORKLOAD=scripts/pmc_conv_half_workload.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd")); sys.path.insert(0, ROOT)
import torch
from hipops import lib as L
from hipops.functions import _p, _ws_bytes, _dense_nhwc
lib = L.load()
B, cin, cout, H, W = 16, 64, 64, 144, 256
dt, code = torch.float16, 1
x = _dense_nhwc(torch.randn(B, cin, H, W, device="cuda").to(dt).contiguous(memory_format=torch.channels_last))
g = _dense_nhwc(torch.randn(B, cout, H, W, device="cuda").to(dt).contiguous(memory_format=torch.channels_last))
w = torch.randn(cout, cin, 3, 3, device="cuda") / 24
pf = torch.empty(_ws_bytes("dd_conv3x3_half_pack_bytes", cout, cin) // 4, device="cuda")
sw = w.stride(); st = L.current_stream()
L.check(lib.dd_conv3x3_half_pack(_p(w), sw[0], sw[1], sw[2], sw[3], cout, cin, code, _p(pf), None, st), "pack")
y = torch.empty((B, H, W, cout), dtype=dt, device="cuda")
flat = torch.empty(cout * 9 * cin, device="cuda")
nb = int(lib.dd_conv3x3_half_wgrad_workspace_bytes(B, H, W, cin, cout)); ws = torch.empty(nb // 4, device="cuda")
for _ in range(int(os.environ.get("DD_PMC_REPS", "6"))):
    L.check(lib.dd_conv3x3_half(_p(x), _p(pf), None, B, H, W, cin, cout, 1, code, _p(y), st), "fwd")
    L.check(lib.dd_conv3x3_half_bwd_weight(_p(x), _p(g), B, H, W, cin, cout, 1, code, _p(flat), _p(ws), nb, st), "wgrad")
torch.cuda.synchronize()
print("done")
