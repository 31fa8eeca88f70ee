"""Times the UNMODIFIED reference (through tests/golden/_refshim.py) on this container's CPU: the loss path
(generate_images_pred + compute_losses + backward, synthetic network outputs) at the benchmark shape, and a full training
step (process_batch + backward, LiteMono) at batch 2.  Runs only where /root/reference exists (the build container)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import _refshim  # noqa: E402
import make_golden as mg  # noqa: E402
import synth  # noqa: E402

threads = int(os.environ.get("DD_THREADS", "8"))
ITERS = int(os.environ.get("DD_ITERS", "13"))          # the first three are warm-up (excluded from the median)
torch.set_num_threads(threads)
ref = _refshim.import_reference()
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("_ref_layers", os.path.join(_refshim.REFERENCE_ROOT, "networks", "layers.py"))
layers = importlib.util.module_from_spec(spec)
spec.loader.exec_module(layers)

B, H, W, scales = int(os.environ.get("DD_B", "12")), 192, 640, [0, 1, 2]
tr, opt = mg.build_ref_trainer(ref, B, H, W, scales, depth_model="litemono")
ts = {0: [1] * B, -1: [1] * B, 1: [1] * B}
for phase in ("fine_tune",):
    times = []
    for it in range(ITERS):
        inputs = synth.make_inputs(7, B, H, W, scales, ts=ts)
        leaves = synth.make_leaves(7, B, H, W, scales)
        t0 = time.time()
        for attempt in range(50):
            try:
                mg.run_ref_loss(ref, tr, phase, inputs, leaves, layers.transformation_from_parameters, 123, 321 + attempt)
                break
            except torch.linalg.LinAlgError:
                t0 = time.time()                      # a singular RANSAC draw aborts the reference; redraw, do not charge it
        times.append(time.time() - t0)
    print("reference loss path fwd+bwd, %s, B=%d 192x640 S=3, %d threads: median %.2f s  (%.2f img/s)  runs %s" % (
        phase, B, threads, float(np.median(times[3:])), B / float(np.median(times[3:])), ["%.2f" % t for t in times]))

# ---- full training step of the reference (networks + loss + backward), LiteMono, batch 2 ------------------------------------
B2 = 2
tr2, opt2 = mg.build_ref_trainer(ref, B2, H, W, scales, depth_model="litemono")
tr2.setup_phase("fine_tune")
tr2.bool_automask = False
tr2.step = mg.PHASE_STEP
tr2.set_train()
times = []
for it in range(ITERS):
    inputs = synth.make_inputs(11 + it, B2, H, W, scales, ts={0: [1] * B2, -1: [1] * B2, 1: [1] * B2})
    t0 = time.time()
    for attempt in range(50):
        try:
            np.random.seed(5 + attempt)
            tr2.model.zero_grad()
            outputs, losses = tr2.process_batch(dict(inputs))
            losses["loss"].backward()
            break
        except torch.linalg.LinAlgError:
            t0 = time.time()
    times.append(time.time() - t0)
print("reference full step (process_batch + backward), fine_tune, litemono B=%d 192x640, %d threads: median %.2f s  (%.2f img/s)  runs %s" % (
    B2, threads, float(np.median(times[3:])), B2 / float(np.median(times[3:])), ["%.2f" % t for t in times]))
