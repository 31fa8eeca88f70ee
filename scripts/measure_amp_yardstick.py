#!/usr/bin/env python
"""Distance of the autocast step's gradients from the fp32 step's, in units of the storage-rounding yardstick
(tests/test_zz_half_precision_gpu.py), over several weight seeds: the numbers behind that test's SLACK.
    python scripts/measure_amp_yardstick.py [seeds...]  > profiles/r04_amp_yardstick.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests"), ROOT, os.path.join(ROOT, "dynamo-depth_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("DD_MIOPEN_FIND", "0")
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import test_zz_half_precision_gpu as T  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "net_tiny_kitti.npz"))
seeds = [int(x) for x in sys.argv[1:]] or list(range(3, 11))
worst = {}
for seed in seeds:
    l32, g32, _ = T.one_step(z, "none", seed=seed)
    for amp, dtype in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        ly, gy, _ = T.one_step(z, "none", seed=seed, yardstick=dtype)
        lh, gh, _ = T.one_step(z, amp, seed=seed)
        dy, dh = T.distances(gy, g32), T.distances(gh, g32)
        print("seed %d %s: loss fp32 %.6f yardstick %+.2e autocast %+.2e" % (seed, amp, l32, ly - l32, lh - l32))
        for n in T.NETS:
            ratio = dh[n] / max(dy[n], 1e-3)
            worst[(amp, n)] = max(worst.get((amp, n), 0.0), ratio)
            print("   %-12s |g32| %.4e  yardstick %.3e  autocast %.3e  ratio %.2f  norm ratio %.2f" % (
                n, float(g32[n].norm()), dy[n], dh[n], ratio, float(gh[n].norm() / g32[n].norm())), flush=True)
print("worst ratio per (type, network):", {k: round(v, 2) for k, v in worst.items()})
