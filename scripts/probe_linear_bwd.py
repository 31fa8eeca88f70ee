"""Micro-benchmark: weight-gradient GEMM of LiteMono's point-wise Linears (huge K, small MxN) -- plain mm vs split-K bmm
vs a 1x1 convolution weight gradient through MIOpen; and forward / data-gradient variants."""
import os
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (B, H, W, cin, cout) in [(12, 48, 160, 64, 384), (12, 48, 160, 384, 64), (12, 24, 80, 128, 768), (12, 24, 80, 768, 128),
                             (12, 12, 40, 224, 1344), (12, 12, 40, 1344, 224)]:
    K = B * H * W
    x = torch.randn(K, cin, device="cuda"); g = torch.randn(K, cout, device="cuda"); w = torch.randn(cout, cin, device="cuda")
    bias = torch.randn(cout, device="cuda")
    ref = g.t() @ x
    res = {"fwd addmm": timeit(lambda: torch.addmm(bias, x, w.t())), "dX mm": timeit(lambda: g @ w), "dW mm": timeit(lambda: g.t() @ x),
           "db sum": timeit(lambda: g.sum(0))}
    for P in (8, 16, 32, 60, 120):
        if K % P:
            continue
        f = lambda P=P: torch.bmm(g.view(P, K // P, cout).transpose(1, 2), x.view(P, K // P, cin)).sum(0)
        err = (f() - ref).abs().max().item() / ref.abs().max().item()
        res["dW bmm P=%d" % P] = timeit(f)
    x4 = x.view(B, H, W, cin).permute(0, 3, 1, 2); g4 = g.view(B, H, W, cout).permute(0, 3, 1, 2); w4 = w.view(cout, cin, 1, 1).contiguous(memory_format=torch.channels_last)
    res["fwd conv1x1"] = timeit(lambda: F.conv2d(x4, w4, bias))
    cb = lambda mask: torch.ops.aten.convolution_backward(g4, x4, w4, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, mask)
    res["dX conv1x1"] = timeit(lambda: cb([True, False, False]))
    res["dW conv1x1"] = timeit(lambda: cb([False, True, False]))
    flop = 2.0 * K * cin * cout
    print("K=%d %d->%d  (%.2f GFLOP per GEMM; rel err of bmm %.1e)" % (K, cin, cout, flop / 1e9, err))
    for k, v in res.items():
        print("   %-16s %8.1f us  %6.1f TFLOP/s" % (k, v, flop / v / 1e6 if "sum" not in k else 0))
