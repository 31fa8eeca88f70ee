"""What MIOpen makes of the motion decoders' full-resolution convolutions (reference networks/motion_decoder.py:24-33: 3x3 convs on 9-12
channels at 192x640, 1x1 reductions to 3 / 1 channels): forward, data gradient and weight gradient timed separately, channels-last fp32,
against the time their bytes take at 5 TB/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd")); sys.path.insert(0, ROOT)
import torch
import miopen_env  # noqa: F401  (the shipped find-db, as the trainer sets it up)
try:
    miopen_env.setup()
except Exception as e:
    print("miopen_env.setup:", e)
torch.backends.cudnn.benchmark = os.environ.get("DD_FIND", "0") == "1"


PMC = os.environ.get("DD_PMC") == "1"          # scripts/pmc_small_convs.sh: few launches, a calibration kernel of known traffic first
if PMC:
    _a = torch.randn(64 << 20, device="cuda"); _b = torch.empty_like(_a)
    for _ in range(3):
        torch.atan(_a, out=_b)                     # 256 MiB read, 256 MiB written per launch


def time(fn, it=20):
    if PMC:
        it = 2
    for _ in range(3 if not PMC else 1):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


B = 12
for (H, W, cin, cout, k) in [(192, 640, 12, 9, 3), (192, 640, 10, 9, 3), (192, 640, 9, 9, 3), (192, 640, 9, 3, 1), (192, 640, 9, 1, 1),
                             (96, 320, 72, 64, 3), (96, 320, 64, 64, 3)]:
    x = torch.randn(B, cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, k, k, device="cuda").contiguous(memory_format=torch.channels_last)
    pad = k // 2
    y = torch.nn.functional.conv2d(x, w, None, 1, pad)
    g = torch.randn_like(y)
    t_f = time(lambda: torch.nn.functional.conv2d(x, w, None, 1, pad))
    t_d = time(lambda: torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False]))
    t_w = time(lambda: torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False]))
    mine = ""
    from hipops import lib as L
    lib = L.load()
    if lib.dd_conv_small_supported(k, cin, cout):
        nb = lib.dd_conv_small_workspace_bytes(k, cin, cout)
        ws = torch.empty(nb // 4 + 1, device="cuda")
        yy, gx = torch.empty_like(y), torch.empty_like(x)
        gw, gb = torch.empty(cout * k * k * cin, device="cuda"), torch.empty(cout, device="cuda")
        sw = w.stride()
        st = L.current_stream()
        m_f = time(lambda: lib.dd_conv_small_fwd(x.data_ptr(), w.data_ptr(), sw[0], sw[1], sw[2], sw[3], None, B, H, W, cin, cout, k, yy.data_ptr(), ws.data_ptr(), nb, st))
        m_d = time(lambda: lib.dd_conv_small_bwd_data(g.data_ptr(), w.data_ptr(), sw[0], sw[1], sw[2], sw[3], B, H, W, cin, cout, k, gx.data_ptr(), ws.data_ptr(), nb, st))
        m_w = time(lambda: lib.dd_conv_small_bwd_weight(x.data_ptr(), g.data_ptr(), B, H, W, cin, cout, k, gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), nb, st))
        mine = " || dd_conv_small: fwd %6.1f  data grad %6.1f  weight grad (+ bias grad) %6.1f us" % (m_f, m_d, m_w)
    mb = lambda *ts: sum(t.numel() * 4 for t in ts) / 1e6
    print("%dx%d %2d->%2d %dx%d | fwd %6.1f us (bytes at 5 TB/s: %5.1f) | data grad %6.1f us (%5.1f) | weight grad %6.1f us (%5.1f) | %.2f GFLOP each" % (
        H, W, cin, cout, k, k, t_f, mb(x, y) / 5, t_d, mb(g, x) / 5, t_w, mb(g, x) / 5, 2 * B * H * W * cin * cout * k * k / 1e9) + mine)

# the disparity heads (3x3 -> 1 channel on a pre-padded input)
from hipops import lib as L
lib = L.load()
for (Bh, Cc, Hp, Wp) in [(24, 32, 98, 322), (12, 32, 98, 322), (24, 64, 50, 162), (12, 64, 50, 162)]:
    x = torch.randn(Bh, Cc, Hp, Wp, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(1, Cc, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last)
    bias = torch.randn(1, device="cuda")
    y = torch.nn.functional.conv2d(x, w, bias)
    g = torch.randn_like(y)
    t_f = time(lambda: torch.nn.functional.conv2d(x, w, bias))
    t_w = time(lambda: torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))
    out, gw, gb = torch.empty_like(y), torch.empty(9 * Cc, device="cuda"), torch.empty(1, device="cuda")
    nb = lib.dd_conv_head_workspace_bytes(Bh, Hp, Wp, Cc)
    ws = torch.empty(nb // 4 + 1, device="cuda")
    sw = w.stride()
    st = L.current_stream()
    m_f = time(lambda: lib.dd_conv_head_fwd(x.data_ptr(), w.data_ptr(), sw[1], sw[2], sw[3], bias.data_ptr(), Bh, Hp, Wp, Cc, out.data_ptr(), st))
    m_w = time(lambda: lib.dd_conv_head_bwd_weight(x.data_ptr(), g.data_ptr(), Bh, Hp, Wp, Cc, gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), nb, st))
    print("head %2d x %2d ch %3dx%3d | MIOpen fwd %6.1f  weight grad %6.1f us || dd_conv_head: fwd %6.1f  weight + bias grad %6.1f us (bytes at 5 TB/s: %4.1f)" % (
        Bh, Cc, Hp, Wp, t_f, t_w, m_f, m_w, x.numel() * 4 / 5e6))
