"""Which convolution problems still go to MIOpen in the headline step, and how often: one eager fine_tune step of the bench workload under
MIOPEN_ENABLE_LOGGING_CMD=1 (MIOpen prints a MIOpenDriver command line per call to stderr); this script runs the step in a child process
and tallies the lines by direction (-F 1 forward, 2 data gradient, 4 weight gradient) and shape."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, "dynamo-depth_amd"))
import miopen_env; miopen_env.setup()
import torch
from options import DynamoOptions
from Trainer import Trainer
from torch.utils.data import DataLoader
opt = DynamoOptions().parse(args=["-d", "kitti", "--depth_model", "litemono", "-b", "12", "--weights_init", "scratch", "--synthetic", "--num_workers", "0",
                                  "--log_dir", "/tmp/dd_probe_logs", "--no_train_vis", "--no_hip_graph", "--no_miopen_find"])
opt.print_opt = False
tr = Trainer(opt); tr.num_steps_per_epoch = 1000; tr.setup_phase("fine_tune"); tr.bool_automask = False; tr.step = 1000; tr.set_train()
ds = tr.get_dataset(["synthetic %%d" %% i for i in range(12)], is_train=False, seed=0)
batch = next(iter(DataLoader(ds, batch_size=12))); tr.upload_inputs(batch)
for _ in range(2):
    tr.train_step(dict(batch))
torch.cuda.synchronize()
print("MARK", file=sys.stderr, flush=True)
tr.train_step(dict(batch)); torch.cuda.synchronize()
''' % ROOT
env = dict(os.environ, MIOPEN_ENABLE_LOGGING_CMD="1", MIOPEN_LOG_LEVEL="3")
res = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
err = res.stderr
tail = err[err.rfind("MARK"):] if "MARK" in err else err
tally = collections.Counter()
for ln in tail.splitlines():
    m = re.search(r"MIOpenDriver (conv\w*) (.*)", ln)
    if not m:
        continue
    a = dict(re.findall(r"-(\w) (\S+)", m.group(2)))
    # -n batch -c in -H -W -k out -y -x kernel -p -q pad -u -v stride -l -j dilation -g groups -F direction
    key = "F%s  %4s x %4s -> %4s  %4sx%-4s  k%sx%s s%s p%s d%s g%s %s" % (a.get("F"), a.get("n"), a.get("c"), a.get("k"), a.get("H"), a.get("W"), a.get("y"), a.get("x"),
                                                                   a.get("u"), a.get("p"), a.get("l"), a.get("g", "1"), m.group(1))
    tally[key] += 1
print("rc", res.returncode, "lines", sum(tally.values()))
for k, v in sorted(tally.items(), key=lambda kv: (kv[0][:2], -kv[1])):
    print("%3d x  %s" % (v, k))
if res.returncode:
    print(err[-2000:])
