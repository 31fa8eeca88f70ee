"""debug: per-step losses of the hipGraph step with the multi-stream forward (DD_MS_DEBUG keeps branches on the current stream)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd"))
os.environ.setdefault("MIOPEN_LOG_LEVEL", "2")
import shutil
db = "/tmp/dd_miopen_db_dbg"
if not os.path.isdir(db):
    shutil.copytree(os.path.join(ROOT, "dynamo-depth_amd", "miopen_db"), db)
os.environ["MIOPEN_USER_DB_PATH"] = db
import torch
from options import DynamoOptions
from Trainer import Trainer
from torch.utils.data import DataLoader
args = ["-d", "kitti", "--depth_model", "litemono", "-b", "12", "--weights_init", "scratch", "--synthetic", "--num_workers", "0", "--log_dir", "/tmp/dd_dbg",
        "--no_train_vis", "--channels_last", "--multi_stream"] + sys.argv[1:]
torch.backends.cudnn.benchmark = True
opt = DynamoOptions().parse(args=args)
opt.print_opt = False
torch.manual_seed(0)
tr = Trainer(opt)
tr.num_steps_per_epoch = 1000
tr.setup_phase("fine_tune")
tr.bool_automask = False
tr.step = 1000
tr.set_train()
ds = tr.get_dataset(["s {}".format(i) for i in range(12)], is_train=False, seed=0)
batch = next(iter(DataLoader(ds, batch_size=12)))
tr.process_inputs(batch)
vals = []
for i in range(8):
    _, l = tr.train_step(dict(batch))
    torch.cuda.synchronize()
    vals.append(float(l["loss"]))
bad = [n for n, p in tr.base_model.named_parameters() if not torch.isfinite(p).all()]
print("DD_MS_DEBUG=%r %s losses %s  non-finite params: %d %s" % (os.environ.get("DD_MS_DEBUG", ""), sys.argv[1:], ["%.5f" % v for v in vals], len(bad), bad[:3]))

# ---- forward-only capture: is the captured multi-stream FORWARD already wrong? ----
if os.environ.get("DD_FWD_ONLY") == "1":
    torch.manual_seed(0)
    tr2 = Trainer(opt)
    tr2.num_steps_per_epoch = 1000
    tr2.setup_phase("fine_tune")
    tr2.bool_automask = False
    tr2.step = 1000
    tr2.set_train()
    from networks.depth_encoder import DropPath
    for m in tr2.base_model.modules():
        if isinstance(m, DropPath):
            m.drop_prob = 0.0
    static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                _, l0 = tr2.forward_and_losses(dict(static))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        eager = float(l0["loss"])
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            _, lg = tr2.forward_and_losses(dict(static))
        vals = []
        for _ in range(4):
            g.replay()
            torch.cuda.synchronize()
            vals.append(float(lg["loss"]))
    print("forward-only: eager %.6f  captured replays %s" % (eager, ["%.6f" % v for v in vals]))
