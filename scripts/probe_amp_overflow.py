"""Where does a half-precision step overflow?  Runs BASELINE.json config 5's shape (nuScenes 288x512, MonoDepth2, batch 16, fine_tune) under
--amp fp16 / bf16 with forward hooks on every leaf module: prints the largest |activation| per module (top of the list) and the
first module whose output is non-finite, step by step.  GPU only; `python scripts/probe_amp_overflow.py [fp16|bf16] [steps] [batch]`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd"))
import miopen_env  # noqa: E402

miopen_env.setup()
import torch  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402
from options import DynamoOptions  # noqa: E402
from Trainer import Trainer  # noqa: E402

amp = sys.argv[1] if len(sys.argv) > 1 else "fp16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
opt = DynamoOptions().parse(args=["-d", "nuscenes", "--depth_model", "monodepthv2", "-b", str(B), "--weights_init", "scratch", "--synthetic", "--num_workers", "0",
                                  "--log_dir", "/tmp/dd_probe_logs", "--no_train_vis", "--amp", amp, "--no_hip_graph"])
opt.print_opt = False
torch.manual_seed(1234)
tr = Trainer(opt)
tr.num_steps_per_epoch = 1000
tr.setup_phase("fine_tune")
tr.bool_automask = False
tr.step = 1000
tr.set_train()
batch = next(iter(DataLoader(tr.get_dataset(["synthetic {}".format(i) for i in range(B)], is_train=False, seed=0), batch_size=B)))
tr.upload_inputs(batch)

peaks, first_bad = {}, []


def hook(name):
    def fn(mod, inp, out):
        outs = out if isinstance(out, (tuple, list)) else [out]
        for o in outs:
            if torch.is_tensor(o) and o.is_floating_point():
                m = float(o.detach().abs().max())
                peaks[name] = max(peaks.get(name, 0.0), m) if m == m else float("nan")
                if not (m == m and m < float("inf")) and not first_bad:
                    first_bad.append((name, str(o.dtype), tuple(o.shape)))
    return fn


for name, mod in tr.base_model.named_modules():
    if len(list(mod.children())) == 0:
        mod.register_forward_hook(hook(name))

for it in range(steps):
    peaks.clear()
    _, losses = tr.train_step(dict(batch))
    loss = float(losses["loss"])
    top = sorted(peaks.items(), key=lambda kv: -(kv[1] if kv[1] == kv[1] else 1e30))[:6]
    scale = float(tr._scaler.get_scale()) if getattr(tr, "_scaler", None) is not None else 1.0
    print("step %2d loss %.5f scale %.0f first non-finite %s | largest: %s" % (it, loss, scale, first_bad[:1], ", ".join("%s %.3g" % kv for kv in top)), flush=True)
    first_bad.clear()
