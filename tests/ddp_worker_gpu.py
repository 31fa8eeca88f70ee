"""World-size-2 worker for tests/test_ddp_gpu.py: the Trainer's DDP path on the GPU with everything on -- channels-last LiteMono,
train-mode fused BatchNorm, every network-side HIP hook, the fused HIP loss, gradient-as-bucket-view, fused Adam.  Both ranks
share cuda:0 and talk over gloo (RCCL refuses two ranks on one device; the DDP machinery above the backend is the same); with
backend "nccl" every rank takes its own device and the gradients travel over RCCL / xGMI."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main(out_path, backend="gloo"):
    from options import DynamoOptions
    from Trainer import Trainer
    from torch.utils.data import DataLoader
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))      # RCCL: one device per rank
    dist.init_process_group(backend=backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    opt = DynamoOptions().parse(args=["-d", "kitti", "--depth_model", "litemono", "-b", "2", "--height", "64", "--width", "96",
                                      "--weights_init", "scratch", "--synthetic", "--num_workers", "0", "--log_dir", "/tmp/dd_ddp_gpu_logs_%d" % rank,
                                      "--dist_backend", backend, "--channels_last"])
    opt.print_opt = False
    opt.local_world_size, opt.ddp, opt.local_rank = world, True, rank
    opt.cuda_ids = list(range(world)) if backend == "nccl" else [0] * world
    opt.multi_stream = True                  # what bench.py runs: network branches on separate HIP streams, DDP hooks on top
    torch.manual_seed(100 + rank)
    tr = Trainer(opt)
    assert tr.device.type == "cuda"
    tr.num_steps_per_epoch = 10
    tr.setup_phase("fine_tune")
    assert isinstance(tr.model, torch.nn.parallel.DistributedDataParallel) and tr.model.static_graph
    tr.bool_automask = False
    tr.step = 10
    tr.set_train()
    ds = tr.get_dataset(["s {}".format(i) for i in range(4)], seed=3)
    batch = next(iter(DataLoader(torch.utils.data.Subset(ds, [2 * rank, 2 * rank + 1]), batch_size=2)))
    start = torch.stack([p.detach().double().sum() for p in tr.base_model.parameters()]).cpu()
    losses = []
    for _ in range(3):
        _, l = tr.train_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
        losses.append(float(l["loss"]))
    torch.cuda.synchronize()
    assert all(x == x and abs(x) < 1e6 for x in losses), losses
    # identical initial weights (DDP broadcast) + averaged gradients + the same Adam -> identical weights on both ranks
    digest = torch.stack([p.detach().double().sum() for p in tr.base_model.parameters()]).cpu()
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    same = all(torch.equal(gathered[0], g) for g in gathered[1:])
    moved = not torch.equal(start, digest)                 # the optimiser really stepped
    dist.barrier()
    if rank == 0:
        with open(out_path, "w") as fh:
            fh.write("%s same_weights=%s losses=%s\n" % ("OK" if (same and moved) else "FAIL", same, losses))
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "gloo")
