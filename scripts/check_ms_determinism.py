"""With library kernels that use no atomics (cudnn.deterministic), a training run is a pure function of seed and data: the
multi-stream forward/backward must then reproduce the single-stream run's weights to the last bit, every time."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd"))
from options import DynamoOptions  # noqa: E402
from Trainer import Trainer  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402


def run(multi_stream, steps=4, H=64, W=96, B=2, extra=()):
    opt = DynamoOptions().parse(args=["-d", "kitti", "--depth_model", "litemono", "-b", str(B), "--height", str(H), "--width", str(W),
                                      "--weights_init", "scratch", "--synthetic", "--num_workers", "0", "--log_dir", "/tmp/dd_msdet", "--channels_last"] + list(extra))
    opt.print_opt = False
    opt.multi_stream = multi_stream
    torch.manual_seed(5)
    tr = Trainer(opt)
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = True
    tr.num_steps_per_epoch = 10
    tr.setup_phase("fine_tune")
    tr.bool_automask = False
    tr.step = 10
    tr.set_train()
    ds = tr.get_dataset(["s {}".format(i) for i in range(B)], seed=3)
    batch = next(iter(DataLoader(ds, batch_size=B)))
    torch.manual_seed(11)
    np.random.seed(11)
    losses = []
    for _ in range(steps):
        _, l = tr.train_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
        losses.append(float(l["loss"]))
    torch.cuda.synchronize()
    params = torch.cat([p.detach().double().reshape(-1) for p in tr.base_model.parameters()])
    bufs = torch.cat([b.detach().double().reshape(-1) for b in tr.base_model.buffers()])
    return params, bufs, losses


if __name__ == "__main__":
    H, W, B = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 96, 2)))
    ref = run(False, H=H, W=W, B=B)
    again = run(False, H=H, W=W, B=B)
    print("single vs single : params max|diff| %.3e  buffers %.3e" % (float((ref[0] - again[0]).abs().max()), float((ref[1] - again[1]).abs().max())))
    for i in range(3):
        ms = run(True, H=H, W=W, B=B)
        print("multi  vs single : params max|diff| %.3e  buffers %.3e  losses %s" % (float((ref[0] - ms[0]).abs().max()), float((ref[1] - ms[1]).abs().max()), [round(x, 6) for x in ms[2]]))
    print("single losses", [round(x, 6) for x in ref[2]])
