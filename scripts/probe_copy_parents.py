"""Which operator issues each layout copy / elementwise add of one training step: walks the torch.profiler event tree
upwards from every aten::copy_ / aten::add_ / aten::cat that launched a kernel and aggregates by (ancestor chain, shape)."""
import os, sys
from collections import defaultdict
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_step(dict(batch))
    torch.cuda.synchronize()
targets = tuple(os.environ.get("DD_PROBE_FILTER", "aten::copy_,aten::cat,aten::add_,aten::add").split(","))
agg = defaultdict(lambda: [0, 0.0])

# kineto device events are linked through FunctionEvents; the simple route: use FunctionEvent.cpu_parent
evs = prof.events()
for e in evs:
    if e.name in targets and e.device_time_total > 0 and not any(c.name in targets and c.device_time_total > 0 for c in e.cpu_children):
        chain = []
        p = e.cpu_parent
        while p is not None and len(chain) < 4:
            if not p.name.startswith("aten::to") and p.name not in ("aten::contiguous", "aten::clone", "aten::_to_copy"):
                chain.append(p.name.replace("autograd::engine::evaluate_function: ", "bwd:"))
            p = p.cpu_parent
        key = (e.name, " <- ".join(chain[:3]), str(e.input_shapes[:1]))
        a = agg[key]; a[0] += 1; a[1] += e.device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("total us in targeted ops: %.0f" % sum(v[1] for v in agg.values()))
for (name, chain, shp), (n, us) in rows[:int(os.environ.get("DD_PROBE_ROWS", "70"))]:
    print("%8.0f us %4d %-12s %-34s %s" % (us, n, name, shp[:34], chain[:150]))
