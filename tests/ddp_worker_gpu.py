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
                                      "--dist_backend", backend, "--channels_last"] +
                                     ([] if os.environ.get("DD_TEST_HIP_GRAPH", "0") == "1" else ["--no_hip_graph"]))
    opt.print_opt = False
    opt.local_world_size, opt.ddp, opt.local_rank = world, True, rank
    opt.cuda_ids = list(range(world)) if backend == "nccl" else [0] * world
    opt.multi_stream = os.environ.get("DD_TEST_MULTI_STREAM", "1") == "1"    # what bench.py runs: network branches on separate HIP streams, DDP hooks on top
    torch.manual_seed(100 + rank)
    tr = Trainer(opt)
    # for the gradient comparison: library kernels without atomics -- two runs of one backward then agree to ~1e-6, and the
    # comparison is not blurred by min-reprojection decisions flipping on a 1e-7 difference of a split-K weight gradient
    # (with the default solvers two runs of the same backward are 1e-5 ... 1e-3 apart at this size)
    benchmark = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = True
    assert tr.device.type == "cuda"
    tr.num_steps_per_epoch = 10
    ds = tr.get_dataset(["s {}".format(i) for i in range(4)], seed=3)
    batch = next(iter(DataLoader(torch.utils.data.Subset(ds, [2 * rank, 2 * rank + 1]), batch_size=2)))
    import numpy as np

    def one_backward():
        tr.bool_automask = False
        tr.step = 10
        tr.set_train()
        torch.manual_seed(7)                 # same stochastic-depth masks in both runs
        np.random.seed(7)                    # same RANSAC draws
        tr.zero_grads(set_to_none=True)                   # (flat mode: the views stay attached, the buffer is zero-filled)
        _, l = tr.process_batch({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
        l["loss"].backward()
        tr.reduce_eager_grads()              # flat mode: ONE all-reduce of the phase's gradient buffer; a no-op under torch's reducer
        torch.cuda.synchronize()
        grads = [p.grad.detach().clone() if (p.grad is not None and p.requires_grad) else None for p in tr.base_model.parameters()]
        tr.zero_grads(set_to_none=True)
        return grads

    # what every rank computes alone (no wrapper yet), on rank 0's weights ...
    tr.opt.ddp = False
    tr.setup_phase("fine_tune")
    for t in list(tr.base_model.parameters()) + list(tr.base_model.buffers()):
        dist.broadcast(t.data, 0)
    alone = one_backward()
    again = one_backward()                   # yardstick: how far two runs of the same backward are apart (split-K atomics in the
    noise = 0.0                              # library weight gradients, min-reprojection decisions flipping on the last bit)
    for a, b in zip(alone, again):
        if a is not None:
            noise = max(noise, float((a - b).abs().max()) / (float(a.abs().max()) + 1e-12))
    # ... and what the phase's DDP wrapper leaves in .grad
    tr.opt.ddp = True
    tr.setup_phase("fine_tune")
    if tr._eager_reduce_mode() == "flat":     # the GPU default: no wrapper, one flat gradient buffer (Trainer.FlatGradients)
        assert tr.model is tr.base_model and tr._flat_grads is not None
    else:
        assert isinstance(tr.model, torch.nn.parallel.DistributedDataParallel) and tr.model.static_graph
    reduced = one_backward()
    if os.environ.get("DD_TEST_SECOND_ITERATION") == "1":
        reduced = one_backward()
    # the gradients DDP leaves in .grad must be the mean over ranks of what each rank computes alone on its own data -- with
    # the branches of forward and backward on separate HIP streams (the all-reduce buckets collect from several of them)
    worst, worst_name = 0.0, ""
    names = [n for n, _ in tr.base_model.named_parameters()]
    for name, a, r in zip(names, alone, reduced):
        if a is None:
            # no gradient alone: none under the wrapper, an all-zero view in flat mode (the buffer covers every trainable parameter)
            assert r is None or float(r.abs().max()) == 0.0, name
            continue
        assert r is not None, name
        mean = a.clone()
        dist.all_reduce(mean)
        mean /= world
        scale = float(mean.abs().max()) + 1e-12
        dev = float((r - mean).abs().max()) / scale
        if dev > worst:
            worst, worst_name = dev, "{} (max |grad| {:.3e})".format(name, scale)
    grads_ok = worst < max(1e-5, 10.0 * noise)
    torch.backends.cudnn.deterministic = False           # the training steps below run on the solvers a real run uses
    torch.backends.cudnn.benchmark = benchmark
    tr.bool_automask = False
    tr.step = 10
    tr.set_train()

    start = torch.stack([p.detach().double().sum() for p in tr.base_model.parameters()]).cpu()
    losses = []
    def across_ranks(tensors):
        d = torch.stack([t.detach().double().sum() if t is not None else torch.zeros((), dtype=torch.float64, device="cuda") for t in tensors])
        g = [torch.zeros_like(d) for _ in range(world)]          # (device tensors: RCCL has no CPU collectives)
        dist.all_gather(g, d)
        d, g = d.cpu(), [x.cpu() for x in g]
        return [names[i] for i in range(len(names)) if any(g[0][i] != x[i] for x in g[1:])]

    for it in range(3):
        if os.environ.get("DD_TEST_DEBUG_STEPS") == "1":
            tr.zero_grads(set_to_none=True)
            _, l = tr.process_batch({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
            l["loss"].backward()
            tr.reduce_eager_grads()
            torch.cuda.synchronize()
            bad_g = across_ranks([p.grad for p in tr.base_model.parameters()])
            tr.optim["optimizer"].step()
            torch.cuda.synchronize()
            bad_w = across_ranks(list(tr.base_model.parameters()))
            if rank == 0:
                print("STEP %d grads differ across ranks: %s ; weights differ: %s" % (it, bad_g[:5], bad_w[:5]), flush=True)
            tr.zero_grads()
        else:
            _, l = tr.train_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
        losses.append(float(l["loss"]))
    torch.cuda.synchronize()
    assert all(x == x and abs(x) < 1e6 for x in losses), losses
    # identical initial weights (DDP broadcast) + averaged gradients + the same Adam -> identical weights on both ranks
    digest = torch.stack([p.detach().double().sum() for p in tr.base_model.parameters()])
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    digest, gathered = digest.cpu(), [g.cpu() for g in gathered]
    same = all(torch.equal(gathered[0], g) for g in gathered[1:])
    if not same and rank == 0:
        bad = [(n, float((gathered[0][i] - gathered[1][i]).abs())) for i, n in enumerate(names) if gathered[0][i] != gathered[1][i]]
        print("DIVERGED %d of %d parameters, e.g. %s" % (len(bad), len(names), bad[:6] + bad[-3:]), flush=True)
    moved = not torch.equal(start, digest)                 # the optimiser really stepped
    dist.barrier()
    if rank == 0:
        with open(out_path, "w") as fh:
            graph_steps = getattr(tr._graph, "replays", 0) if tr._graph is not None else 0
            fh.write("%s same_weights=%s grads_vs_mean_of_local=%.2e at %s (two runs of one backward differ by %.2e) losses=%s graph_steps=%d reduce_mode=%s\n"
                     % ("OK" if (same and moved and grads_ok) else "FAIL", same, worst, worst_name, noise, losses, graph_steps,
                        getattr(tr._graph, "reduce_mode_chosen", None) if tr._graph is not None else None))
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "gloo")
