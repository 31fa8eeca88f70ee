"""cProfile of the host side of one training step (which Python frames cost the most while enqueueing)."""
import os, sys, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd")); sys.path.insert(0, ROOT)
import bench  # noqa: F401
import torch
from options import DynamoOptions
from Trainer import Trainer
torch.backends.cudnn.benchmark = True
opt = DynamoOptions().parse(args=["-d", "kitti", "--depth_model", "litemono", "-b", "12", "--weights_init", "scratch", "--synthetic",
                                  "--num_workers", "0", "--log_dir", "/tmp/dd_probe", "--no_train_vis", "--channels_last", "--multi_stream"])
opt.print_opt = False
tr = Trainer(opt); tr.num_steps_per_epoch = 1000; tr.setup_phase("fine_tune"); tr.bool_automask = False; tr.step = 1000; tr.set_train()
batch = bench.make_batch(tr, 0)
for _ in range(6):
    tr.train_step(dict(batch))
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.train_step(dict(batch))
    torch.cuda.synchronize()          # per-step sync: the host never waits on a full queue inside the step
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats(60)
print(s.getvalue()[:12000])
