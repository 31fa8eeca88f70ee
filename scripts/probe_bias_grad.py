import torch, torch.nn as nn
from torch.profiler import profile, ProfilerActivity
torch.backends.cudnn.benchmark = True
for (cin, cout, h, w) in [(12, 9, 192, 640), (64, 32, 96, 320), (32, 1, 192, 640)]:
    conv = nn.Conv2d(cin, cout, 3, padding=1).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(12, cin, h, w, device="cuda").to(memory_format=torch.channels_last).requires_grad_()
    for _ in range(3):
        conv(x).sum().backward()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            y = conv(x)
            y.backward(torch.ones_like(y))
        torch.cuda.synchronize()
    print("== conv", cin, cout, h, w)
    for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:6]:
        print("   %-90s calls %3d avg %8.1f us" % (e.key[:90], e.count, e.device_time_total / e.count))
