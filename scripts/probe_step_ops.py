"""torch.profiler view of one training step (operator level, with shapes) for the slowest reduction / copy kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_LOG_LEVEL", "2"); os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from options import DynamoOptions
from Trainer import Trainer
torch.backends.cudnn.benchmark = True
opt = DynamoOptions().parse(args=["-d", "kitti", "--depth_model", "litemono", "-b", "12", "--weights_init", "scratch", "--synthetic",
                                  "--num_workers", "0", "--log_dir", "/tmp/dd_probe", "--no_train_vis", "--channels_last", "--no_hip_graph", "--single_stream"])
opt.print_opt = False
tr = Trainer(opt); tr.num_steps_per_epoch = 1000; tr.setup_phase("fine_tune"); tr.bool_automask = False; tr.step = 1000; tr.set_train()
batch = bench.make_batch(tr, 0)
for _ in range(4):
    tr.train_step(dict(batch))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=bool(os.environ.get('DD_PROBE_STACK'))) as prof:
    tr.train_step(dict(batch))
    torch.cuda.synchronize()
rows = []
if os.environ.get("DD_PROBE_STACK"):
    # group by the innermost frames of OUR tree, to see which module issues the op
    for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
        if e.device_time_total > 0:
            own = [f for f in e.stack if "dynamo-depth_amd" in f or "bench.py" in f]
            where = " <- ".join(f.split("dynamo-depth_amd/")[-1].replace(".py(", ":").split(")")[0] + ")" for f in own[:3])
            rows.append((e.device_time_total, e.count, e.key, str(e.input_shapes)[:70] + " @ " + where))
else:
    for e in prof.key_averages(group_by_input_shape=True):
        if e.device_time_total > 0:
            rows.append((e.device_time_total, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total device us (op-level, nested ops double counted):", tot)
import os
flt = os.environ.get("DD_PROBE_FILTER", "")
shown = 0
for t, c, k, sh in rows:
    if flt and not any(f in k for f in flt.split(",")):
        continue
    print("%9.0f us %4d  %-38s %s" % (t, c, k[:38], sh))
    shown += 1
    if shown >= int(os.environ.get('DD_PROBE_ROWS', '45')):
        break
