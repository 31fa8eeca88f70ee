"""World-size-2 gloo worker for tests/test_ddp_gloo.py: checks that the Trainer's DDP wrapping averages the
gradients of the phase's networks across ranks, on CPU.  (The HIP loss has no CPU path, so a differentiable
surrogate of the network outputs stands in for it; the collective path under test is the same.)"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def surrogate(outputs, scales):
    """Touches every network output the phase produces (what the real loss reads)."""
    total = 0
    for s in scales:
        total = total + outputs[("disp", 0, s)].mean()
        if ("complete_flow", 1, s) in outputs:
            total = total + outputs[("complete_flow", 1, s)].abs().mean()
        if ("motion_mask", 1, s) in outputs:
            total = total + outputs[("motion_mask", 1, s)].mean()
    for f in (-1, 1):
        total = total + outputs[("cam_T_cam", 0, f)][:, :3].abs().mean()
    return total


def check_phases(tr, opt, rank, world):
    """Per phase: the wrapper is built for exactly the phase's trainable parameters (static graph, no unused-parameter
    search), two consecutive steps go through, frozen networks keep grad None, and the all-reduced gradients agree bit for
    bit on both ranks.  (The probe of SURVEY.md 2.3: which networks a phase reaches.)"""
    from torch.utils.data import DataLoader
    from Trainer import PHASE_TABLE
    ds = tr.get_dataset(["s 0", "s 1"], seed=3)
    report = []
    for phase, (_, _, nets, _) in PHASE_TABLE.items():
        tr.setup_phase(phase)
        tr.set_eval()
        ddp = tr.model
        assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel) and ddp.static_graph and not ddp.find_unused_parameters
        assert not ddp.broadcast_buffers
        trainable = set(id(p) for p in tr.base_model.parameters_by_names(nets))
        for it in range(2):
            tr.base_model.zero_grad(set_to_none=True)
            batch = next(iter(DataLoader(torch.utils.data.Subset(ds, [rank]), batch_size=1)))
            tr.process_inputs(batch)
            surrogate(tr.model(batch), opt.scales).backward()
        with_grad = 0
        digest = torch.zeros(1, dtype=torch.float64)
        for name, p in tr.base_model.named_parameters():
            if id(p) not in trainable or ".fc." in name:
                assert not p.requires_grad and p.grad is None, (phase, name)
            else:
                assert p.requires_grad, (phase, name)
                if p.grad is not None:
                    with_grad += 1
                    digest += p.grad.double().abs().sum()
        assert with_grad > 0, phase
        both = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(both, digest)
        assert all(torch.equal(both[0], b) for b in both), (phase, both)
        report.append("%s:%d" % (phase, with_grad))
    return " ".join(report)


def main(out_path):
    from fill import fill_state
    from options import DynamoOptions
    from Trainer import Trainer
    from torch.utils.data import DataLoader
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    opt = DynamoOptions().parse(args=["-d", "kitti", "--depth_model", "monodepthv2", "-b", "1", "--height", "64", "--width", "96",
                                      "--weights_init", "scratch", "--synthetic", "--num_workers", "0", "--log_dir", "/tmp/dd_ddp_logs_%d" % rank,
                                      "--dist_backend", "gloo", "--epoch-size", "4"])
    opt.print_opt = False
    opt.local_world_size, opt.ddp, opt.local_rank = world, True, rank
    opt.cuda_ids = list(range(world))
    tr = Trainer(opt)
    assert tr.device.type == "cpu"
    for name in sorted(tr.base_model.module_names):
        fill_state(getattr(tr.base_model, name), seed=5)
    flat = tr._eager_reduce_mode() == "flat"          # DD_EAGER_REDUCE=flat: Trainer.FlatGradients instead of torch's reducer (the GPU default)
    phases = "flat" if flat else check_phases(tr, opt, rank, world)
    # cross-rank loss averaging of the logging path (one all-reduce of the stacked scalars)
    red = tr.reduce_losses({"loss": torch.tensor(float(rank + 1)), "loss_term/0": torch.tensor([2.0 * rank]), "loss_coef/x": 0.5})
    assert abs(red["loss"] - (1 + world) / 2) < 1e-6 and abs(red["loss_term/0"] - (world - 1)) < 1e-6 and red["loss_coef/x"] == 0.5, red
    tr.base_model.zero_grad(set_to_none=True)
    tr.setup_phase("fine_tune")
    if flat:
        assert tr.model is tr.base_model and tr._flat_grads is not None
        assert all(p.grad is not None for p in tr.base_model.parameters() if p.requires_grad)      # the views are attached
    else:
        assert isinstance(tr.model, torch.nn.parallel.DistributedDataParallel)
    tr.set_eval()                      # deterministic BN so that the single-process reference below is exact
    # sampler sharding: the two ranks must see disjoint items
    tr.setup_train_loader()
    seen = [int(b["index"][0]) for b in tr.train_loader]
    gathered = [None] * world
    dist.all_gather_object(gathered, seen)
    assert not (set(gathered[0]) & set(gathered[1])), gathered
    ds = tr.get_dataset(["s 0", "s 1"], seed=3)
    batch = next(iter(DataLoader(torch.utils.data.Subset(ds, [rank]), batch_size=1)))
    tr.process_inputs(batch)
    loss = surrogate(tr.model(batch), opt.scales)
    loss.backward()
    tr.reduce_eager_grads()            # flat mode: one all-reduce of the whole gradient buffer; no-op under the wrapper
    mine = {n: p.grad.clone() for n, p in tr.base_model.named_parameters() if p.grad is not None and p.requires_grad}
    # reference: both items through the un-wrapped model on this rank, mean of the two losses
    tr.base_model.zero_grad(set_to_none=True)
    total = 0
    for r in range(world):
        b = next(iter(DataLoader(torch.utils.data.Subset(ds, [r]), batch_size=1)))
        tr.process_inputs(b)
        total = total + surrogate(tr.base_model(b), opt.scales) / world
    total.backward()
    worst = 0.0
    for n, p in tr.base_model.named_parameters():
        if p.grad is None:
            assert n not in mine or float(mine[n].abs().max()) == 0.0
            continue
        err = float((p.grad - mine[n]).abs().max()) / (float(p.grad.abs().max()) + 1e-12)
        worst = max(worst, err)
    dist.barrier()
    if rank == 0:
        with open(out_path, "w") as fh:
            fh.write("OK worst_rel_err=%.3e params=%d phases[%s]\n" % (worst, len(mine), phases))
    assert worst < 1e-4, worst
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
