"""How long does RCCL's all-reduce of the step's gradient buffer take in a ONE-rank process group (nothing to exchange)?  The one-rank
bench rows lose ~7.5 ms per step wherever the collective is issued (profiles/r04_rccl_one_rank.txt); is that the collective itself?"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29581")
dist.init_process_group("nccl", rank=0, world_size=1)
for n in (46_250_000, 12_000_000, 1_000_000):
    buf = torch.randn(n, device="cuda")
    for op, name in ((dist.ReduceOp.AVG, "AVG"), (dist.ReduceOp.SUM, "SUM")):
        for _ in range(3):
            dist.all_reduce(buf, op=op)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dist.all_reduce(buf, op=op)
        e1.record()
        torch.cuda.synchronize()
        print("all_reduce(%s) of %.1f MB, one rank: %.3f ms per call" % (name, n * 4 / 1e6, e0.elapsed_time(e1) / 10), flush=True)
t = time.perf_counter()
dist.barrier()
torch.cuda.synchronize()
print("barrier %.3f ms" % ((time.perf_counter() - t) * 1e3))
dist.destroy_process_group()
