"""Reproducer for the per-network capture: python scripts/debug_segments.py <train|eval> [phase]"""
import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd"))
import miopen_env
miopen_env.setup()
import torch
from torch.utils.data import DataLoader
from options import DynamoOptions
from Trainer import Trainer
mode = sys.argv[1] if len(sys.argv) > 1 else "train"
phase = sys.argv[2] if len(sys.argv) > 2 else "disp_init"
opt = DynamoOptions().parse(args=["-d", "kitti", "--depth_model", "litemono", "-b", "2", "--weights_init", "scratch", "--synthetic", "--height", "96", "--width", "160",
                                  "--num_workers", "0", "--log_dir", "/tmp/dd_dbg_logs", "--no_train_vis"] + sys.argv[3:])
opt.print_opt = False
torch.manual_seed(0)
tr = Trainer(opt)
tr.num_steps_per_epoch = 10
tr.setup_phase(phase)
tr.bool_automask = phase == "disp_init"
tr.step = 10
(tr.set_train if mode == "train" else tr.set_eval)()
batch = next(iter(DataLoader(tr.get_dataset(["s 0", "s 1"]), batch_size=2)))
for i in range(4):
    _, l = tr.train_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    torch.cuda.synchronize()
    print("step", i, float(l["loss"]), flush=True)
print("OK", mode, phase)
