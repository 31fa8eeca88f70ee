"""Device timing of LiteMono's MLP blocks: dd_pw_gemm (MlpFn) against the BLAS path + ATen GELU (PointwiseLinearFn), forward and
forward + backward, at the three stages' shapes of the 192x640 workload (B = 12 target frames, 24 statistics-only side frames).
Every variant is captured into a hipGraph (ten repetitions) and the replays are timed: device time, no host launch cost -- the step
replays graphs too.  Second table: the single launches."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dynamo-depth_amd")):
    sys.path.insert(0, p)
from hipops import lib as L  # noqa: E402
from hipops.functions import mlp, mlp_fused, pointwise_linear  # noqa: E402


class Block(torch.nn.Module):
    def __init__(self, C):
        super().__init__()
        self.pwconv1 = torch.nn.Linear(C, 6 * C)
        self.act = torch.nn.GELU()
        self.pwconv2 = torch.nn.Linear(6 * C, C)


def timed(fn, reps=10, n=20):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / (n * reps)


SHAPES = [(12, 48, 160, 64), (24, 48, 160, 64), (12, 24, 80, 128), (24, 24, 80, 128), (12, 12, 40, 224), (24, 12, 40, 224)]
print("%-22s %12s %12s %14s %14s %14s" % ("B,H,W,C", "own fwd us", "lib fwd us", "own fwd+bwd", "lib fwd+bwd", "fused fwd us"))
for (B, H, W, C) in SHAPES:
    blk = Block(C).cuda()
    y = torch.randn(B, H, W, C, device="cuda", requires_grad=True)
    go = torch.randn(B, H, W, C, device="cuda")
    own = lambda: mlp(y, blk)
    lib = lambda: pointwise_linear(blk.act(pointwise_linear(y, blk.pwconv1)), blk.pwconv2)
    params = [y] + list(blk.parameters())

    def nograd(f):
        def run():
            with torch.no_grad():
                return f()
        return run
    f_own, f_lib = timed(nograd(own)), timed(nograd(lib))
    b_own = timed(lambda: torch.autograd.grad(own(), params, go))
    b_lib = timed(lambda: torch.autograd.grad(lib(), params, go))
    f_fused = timed(nograd(lambda: mlp_fused(y.detach(), blk))) if C in (64, 128) else float("nan")
    print("%-22s %12.1f %12.1f %14.1f %14.1f %14.1f" % ((B, H, W, C), f_own, f_lib, b_own, b_lib, f_fused))

print()
print("%-22s %10s %10s %10s %10s %10s %10s | %10s %10s %10s %10s" % ("B,H,W,C", "pack", "wide", "narrow+act", "narrow", "gelu_pair", "", "mm wide", "mm narrow", "gelu", "gelu_bwd"))
lib = L.load()
for (B, H, W, C) in SHAPES:
    M, hid = B * H * W, 6 * C
    w1, w2 = torch.randn(hid, C, device="cuda") / 8, torch.randn(C, hid, device="cuda") / 20
    b1, b2 = torch.randn(hid, device="cuda"), torch.randn(C, device="cuda")
    x = torch.randn(M, C, device="cuda")
    pre, post, gp = torch.randn(M, hid, device="cuda"), torch.empty(M, hid, device="cuda"), torch.randn(M, hid, device="cuda")
    out = torch.empty(M, C, device="cuda")
    nb1, nb2 = int(lib.dd_pw_gemm_pack_bytes(hid, C)), int(lib.dd_pw_gemm_pack_bytes(C, hid))
    packs = torch.empty((2 * (nb1 + nb2)) // 4, device="cuda")
    p0 = packs.data_ptr()
    st = lambda: L.current_stream()
    f_pack = lambda: lib.dd_mlp_pack(w1.data_ptr(), hid and C, 1, w2.data_ptr(), hid, 1, C, hid, p0, p0 + nb1, p0 + nb1 + nb2, p0 + 2 * nb1 + nb2, None, st())
    f_pack()
    t = [timed(f_pack),
         timed(lambda: lib.dd_pw_gemm(x.data_ptr(), p0, b1.data_ptr(), M, C, hid, 0, pre.data_ptr(), st())),
         timed(lambda: lib.dd_pw_gemm(pre.data_ptr(), p0 + nb1, b2.data_ptr(), M, hid, C, 1, out.data_ptr(), st())),
         timed(lambda: lib.dd_pw_gemm(pre.data_ptr(), p0 + nb1, b2.data_ptr(), M, hid, C, 0, out.data_ptr(), st())),
         timed(lambda: lib.dd_gelu_pair(pre.data_ptr(), gp.data_ptr(), post.data_ptr(), M * hid, st())),
         0.0,
         timed(lambda: torch.addmm(b1, x, w1.t())), timed(lambda: torch.addmm(b2, pre, w2.t())), timed(lambda: torch.nn.functional.gelu(pre)),
         timed(lambda: torch.ops.aten.gelu_backward(gp, pre))]
    print("%-22s " % ((B, H, W, C),) + " ".join("%10.1f" % v for v in t[:6]) + " | " + " ".join("%10.1f" % v for v in t[6:]))
