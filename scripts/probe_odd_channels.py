"""Does padding an odd input-channel count (cat of 3 flow channels + 64 features = 67) to a multiple of 8 help MIOpen's NHWC fp32 convs?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: F401
import torch, torch.nn.functional as F
torch.backends.cudnn.benchmark = True


def timeit(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (cin, cout, H, W, stride) in [(67, 64, 96, 320, 1), (72, 64, 96, 320, 1), (65, 64, 96, 320, 1), (131, 128, 24, 80, 1), (136, 128, 24, 80, 1),
                                  (67, 64, 48, 160, 1), (72, 64, 48, 160, 1), (67, 64, 96, 320, 2), (72, 64, 96, 320, 2), (259, 224, 24, 80, 2), (264, 224, 24, 80, 2),
                                  (12, 9, 192, 640, 1), (16, 9, 192, 640, 1), (16, 16, 192, 640, 1), (9, 64, 192, 640, 2), (16, 64, 192, 640, 2)]:
    k = 7 if (cout == 64 and H == 192) else 3
    x = torch.randn(12, cin, H, W, device="cuda").to(memory_format=torch.channels_last).requires_grad_()
    w = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).to(memory_format=torch.channels_last).requires_grad_()
    y = F.conv2d(x, w, None, stride, k // 2)
    g = torch.randn_like(y)
    tf = timeit(lambda: F.conv2d(x, w, None, stride, k // 2))
    def fb():
        F.conv2d(x, w, None, stride, k // 2).backward(g)
        x.grad = w.grad = None
    tb = timeit(fb)
    print("cin=%3d cout=%3d %dx%d k%d s%d  fwd %7.1f us   fwd+bwd %7.1f us" % (cin, cout, H, W, k, stride, tf, tb))
