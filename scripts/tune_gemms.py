"""Records PyTorch TunableOp results (best rocBLAS / hipBLASLt solution per GEMM shape) for the benchmark workload.
usage: tune_gemms.py <out.csv>   (runs two training steps with tuning on, then writes the table)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd")); sys.path.insert(0, ROOT)
import bench  # noqa: F401
import torch
import torch.cuda.tunable as tn
from options import DynamoOptions
from Trainer import Trainer
tn.enable(True); tn.tuning_enable(True)
tn.set_max_tuning_duration(8); tn.set_max_tuning_iterations(4)
tn.set_filename(sys.argv[1])
torch.backends.cudnn.benchmark = True
opt = DynamoOptions().parse(args=["-d", "kitti", "--depth_model", "litemono", "-b", "12", "--weights_init", "scratch", "--synthetic",
                                  "--num_workers", "0", "--log_dir", "/tmp/dd_tune", "--no_train_vis", "--channels_last"])
opt.print_opt = False
tr = Trainer(opt); tr.num_steps_per_epoch = 1000; tr.setup_phase("fine_tune"); tr.bool_automask = False; tr.step = 1000; tr.set_train()
batch = bench.make_batch(tr, 0)
import time
t0 = time.time()
for i in range(2):
    tr.train_step(dict(batch)); torch.cuda.synchronize()
    print("step", i, "%.1f s" % (time.time() - t0), flush=True)
print(len(tn.get_results()), "entries; the table is written to", sys.argv[1], "at exit")
