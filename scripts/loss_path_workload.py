"""The whole fused loss (dd_photo_loss + dd_reg_losses_finish, forward AND gradients) at the bench shape, alone, N times:
   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python scripts/loss_path_workload.py [phase] [B] [iters]
   python scripts/steady_state_stats.py <kernel_trace.csv> 20 <out.csv>      # per-kernel durations of one loss evaluation
Also prints the wall time per evaluation from HIP events (kernels back to back on one stream)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests"), ROOT, os.path.join(ROOT, "dynamo-depth_amd")):
    sys.path.insert(0, p)
import synth  # noqa: E402
from hipops.fused_loss import LossPlan, fused_loss  # noqa: E402
from hipops.functions import PoseMatrixFn  # noqa: E402

phase = sys.argv[1] if len(sys.argv) > 1 else "fine_tune"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 12
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
H, W, scales = int(os.environ.get("DD_H", 192)), int(os.environ.get("DD_W", 640)), [int(x) for x in os.environ.get("DD_SCALES", "0,1,2").split(",")]
PH = {"disp_init": (False, False, ("Depth", "Pose"), True), "motion_init": (True, False, ("CmpFlow",), False),
      "mask_init": (True, True, ("Pose", "CmpFlow", "MotMask"), False), "fine_tune": (True, True, ("Depth", "Pose", "CmpFlow", "MotMask"), False)}
cmp, mot, optimised, automask = PH[phase]
coefs = dict(p_photo=1.0, d_smooth=1e-3, d_ground=0.1, c_smooth=1e-3, c_consistency=5.0, m_sparsity=0.04, m_smooth=0.1)
inputs = {k: v.cuda() for k, v in synth.make_inputs(3, B, H, W, scales).items()}
if os.environ.get("DD_PACKED", "1") == "1":           # as Trainer.pack_sources: the source frames' pixel-interleaved copies (once per step, not part of the loss path)
    from hipops.inputs import pack_rgb
    for f in (-1, 1):
        inputs[("color_packed", f)] = pack_rgb(inputs[("color", f, 0)])
raw = synth.make_leaves(3, B, H, W, scales)
if os.environ.get("DD_SMOOTH", "1") == "1":          # network-like outputs: low-frequency disparity / flow / mask instead of per-pixel white noise
    import torch.nn.functional as F
    for (kind, s), v in list(raw.items()):
        if kind in ("disp", "flow", "prob"):
            coarse = F.avg_pool2d(v.detach(), 8, 8, ceil_mode=True) if v.shape[-1] >= 16 else v.detach()
            raw[(kind, s)] = F.interpolate(coarse, v.shape[-2:], mode="bilinear", align_corners=False) * (0.2 if kind == "flow" else 1.0)
leaves = {k: v.detach().cuda().requires_grad_() for k, v in raw.items()}
plan = LossPlan(height=H, width=W, scales=scales, min_depth=0.1, max_depth=100.0, ssim_weight=0.85, mask_disp_thrd=0.03, gp_prior=0.4, gp_tol=0.005,
                gp_max_it=100, gp_np_per_it=5, cmpflow=cmp, motmask=mot, automask=automask, optimised=optimised, coefs=coefs)


def once():
    outputs = synth.leaves_to_outputs(leaves, scales, lambda a, t, invert: PoseMatrixFn.apply(a, t, invert), cmp, mot)
    if cmp and os.environ.get("DD_SEPARATE", "0") != "1":          # what networks.Model publishes: ONE field / mask tensor for both frames
        for s in scales:
            outputs[("complete_flow_field", 1, s)] = leaves[("flow", s)]
            if mot:
                outputs[("motion_mask", -1, s)] = outputs[("motion_mask", 1, s)]
    return fused_loss(plan, inputs, outputs)


for _ in range(5):
    losses = once()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
if os.environ.get("DD_HOST_ISSUED", "0") == "1":
    e0.record()
    for _ in range(iters):
        losses = once()
    e1.record()
    how = "host-issued"
else:
    # one evaluation captured in a hipGraph and replayed: the kernels back to back, no Python between them
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        once()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        losses = once()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        graph.replay()
    e1.record()
    how = "graph replay"
torch.cuda.synchronize()
print("%s B=%d %dx%d: %.1f us per loss evaluation (%s, one stream), loss %.6f" % (phase, B, H, W, e0.elapsed_time(e1) * 1e3 / iters, how, float(losses["loss"].detach())))
