"""BatchNorm(+ReLU/GELU, +residual) forward+backward: stock PyTorch-ROCm (MIOpen BN + ATen element-wise) vs dd_bn_act_*."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd"))
import torch
from networks.layers import BatchNorm2d


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


CASES = [((12, 64, 96, 320), "relu", False), ((12, 64, 96, 320), "gelu", False), ((12, 64, 48, 160), "relu", True),
                        ((12, 128, 24, 80), "relu", True), ((12, 256, 12, 40), "relu", True), ((12, 512, 6, 20), "relu", True),
                        ((12, 224, 12, 40), None, False), ((12, 64, 48, 160), None, False)]
if os.environ.get('DD_BN_CASE'):
    CASES = [CASES[int(os.environ['DD_BN_CASE'])]]
for shape, act, res in CASES:
    x = torch.randn(*shape, device="cuda").to(memory_format=torch.channels_last).requires_grad_()
    r = torch.randn(*shape, device="cuda").to(memory_format=torch.channels_last).requires_grad_() if res else None
    g = torch.randn(*shape, device="cuda").to(memory_format=torch.channels_last)
    bn = BatchNorm2d(shape[1]).cuda().train()

    def step():
        y = bn(x, act=act, residual=r)
        y.backward(g)
        x.grad = None
        if r is not None:
            r.grad = None

    res_t = {}
    for stock in ("1", "0"):
        os.environ["DD_STOCK_BATCHNORM"] = stock
        res_t[stock] = timeit(step)
    mb = x.numel() * 4 / 1e6
    print("%-20s act=%-5s res=%d  %6.1f MB  stock %7.1f us   fused %7.1f us" % (shape, act, res, mb, res_t["1"], res_t["0"]))
