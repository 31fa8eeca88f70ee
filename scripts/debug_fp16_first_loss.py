"""Why does the first fp16 loss of the replayed step differ from the eager step's (tests/test_zz_half_precision_gpu.py)?  Prints the loss sequence
of four steps for {fp32, fp16} x {eager single-stream, eager multi-stream, replayed} on the same weights and batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests"), ROOT, os.path.join(ROOT, "dynamo-depth_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("DD_MIOPEN_FIND", "0")
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402
from fill import fill_state  # noqa: E402
from test_networks import make_opt  # noqa: E402
from Trainer import Trainer  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

for amp in ("none", "fp16"):
    for mode in ("eager-single", "eager-multi", "graph"):
        torch.manual_seed(0)
        extra = {"eager-single": [], "eager-multi": ["--multi_stream"], "graph": ["--hip_graph", "--multi_stream"]}[mode]
        opt = make_opt("monodepthv2", ["--synthetic", "--height", "96", "--width", "160", "--channels_last"] + (["--amp", amp] if amp != "none" else []) + extra)
        tr = Trainer(opt)
        for name in sorted(tr.base_model.module_names):
            fill_state(getattr(tr.base_model, name), seed=3)
        tr.base_model.to(tr.device)
        tr.base_model.to(memory_format=torch.channels_last)
        tr.num_steps_per_epoch = 10
        tr.setup_phase("fine_tune")
        tr.bool_automask = False
        tr.step = 10
        tr.set_train()
        batch = next(iter(DataLoader(tr.get_dataset(["s {}".format(i) for i in range(2)]), batch_size=2)))
        rs = np.random.RandomState(1)
        tr.rand_idx_override = {s: torch.from_numpy(rs.randint(0, int(0.4 * (opt.height >> s)) * (opt.width >> s), (2, 500)).astype(np.int32)).cuda() for s in opt.scales}
        rows = []
        for _ in range(4):
            _, l = tr.train_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
            rows.append({k: round(float(v), 6) for k, v in l.items() if k.startswith("loss_term/") and not k[-1].isdigit()} | {"loss": round(float(l["loss"]), 6)})
        print(amp, mode, [r["loss"] for r in rows])
        print("    first step terms:", rows[0], flush=True)
