#!/usr/bin/env python
"""Hunt for the non-finite loss of the fp16 bench rows at config 5's batch: N eager multi-stream training
steps of `bench.py --dataset nuscenes --depth_model monodepthv2 --batch 16 --amp fp16` WITHOUT a host sync, one finiteness flag
per (loss, network gradients, network weights) per step into a ring on the device (level 0), plus one flag per leaf-module output
(level 1, DD_PROBE_LEVEL=1).  The ring is read at the end: the first step and buffer that went non-finite.

    python scripts/probe_amp_nan.py [--steps 60] [--batch 16] [--amp fp16] [--single_stream] [--runs 3]
    DD_AMP_CACHE=1|0 forces autocast's weight cache on / off (default: off under multi-stream, Trainer.run_networks)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd"))
sys.path.insert(0, ROOT)
import miopen_env  # noqa: E402

miopen_env.setup()
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--amp", default="fp16")
    ap.add_argument("--dataset", default="nuscenes")
    ap.add_argument("--depth_model", default="monodepthv2")
    ap.add_argument("--single_stream", action="store_true")
    ap.add_argument("--runs", type=int, default=1)
    ap.add_argument("--seed", type=int, default=1234)
    a = ap.parse_args()
    import bench
    from options import DynamoOptions
    from Trainer import Trainer
    level = int(os.environ.get("DD_PROBE_LEVEL", "0"))
    for run in range(a.runs):
        args = ["-d", a.dataset, "--depth_model", a.depth_model, "-b", str(a.batch), "--weights_init", "scratch", "--synthetic", "--num_workers", "0",
                "--log_dir", "/tmp/dd_probe_logs", "--no_train_vis", "--no_hip_graph", "--amp", a.amp] + (["--single_stream"] if a.single_stream else [])
        opt = DynamoOptions().parse(args=args)
        opt.print_opt = False
        torch.manual_seed(a.seed + run)
        tr = Trainer(opt)
        tr.num_steps_per_epoch = 1000
        tr.setup_phase("fine_tune")
        tr.bool_automask = False
        tr.step = 1000
        tr.set_train()
        batch = bench.make_batch(tr, seed=0)
        nets = sorted(tr.base_model.module_names)
        names = ["loss"] + ["grad " + n for n in nets] + ["weights " + n for n in nets]
        mods = []
        if level >= 1:
            mods = [(n, m) for n, m in tr.base_model.named_modules() if not list(m.children())]
            names += ["out " + n for n, _ in mods]
        ring = torch.ones(a.steps, len(names), dtype=torch.bool, device=tr.device)
        peak = torch.zeros(a.steps, len(names), dtype=torch.float32, device=tr.device)
        cur = {"step": 0}
        handles = []
        for j, (n, m) in enumerate(mods):
            col = 1 + 2 * len(nets) + j

            def hook(_m, _i, out, col=col):
                if torch.is_tensor(out) and out.is_floating_point():
                    o = out.detach()
                    ring[cur["step"], col] &= torch.isfinite(o).all()
                    peak[cur["step"], col] = torch.maximum(peak[cur["step"], col], o.abs().max().float())
            handles.append(m.register_forward_hook(hook))
        optimizer = tr.optim["optimizer"]
        orig_zero = optimizer.zero_grad
        grads_seen = {}

        def zero_grad(*x, **k):            # Trainer.train_step ends with zero_grad(): look at the gradients just before they go
            i = cur["step"]
            for c, n in enumerate(nets):
                gs = [p.grad for p in getattr(tr.base_model, n).parameters() if p.grad is not None]
                if gs:
                    norms = torch._foreach_norm(gs)
                    tot = torch.stack(norms).float().norm()
                    ring[i, 1 + c] = torch.isfinite(tot)
                    peak[i, 1 + c] = tot
            return orig_zero(*x, **k)
        optimizer.zero_grad = zero_grad
        scales = []
        for i in range(a.steps):
            cur["step"] = i
            _, losses = tr.train_step(dict(batch))
            ring[i, 0] = torch.isfinite(losses["loss"].detach())
            peak[i, 0] = losses["loss"].detach().float()
            for c, n in enumerate(nets):
                ps = [p.detach() for p in getattr(tr.base_model, n).parameters()]
                tot = torch.stack(torch._foreach_norm(ps)).norm()
                ring[i, 1 + len(nets) + c] = torch.isfinite(tot)
                peak[i, 1 + len(nets) + c] = tot
            sc = tr._grad_scaler()
            scales.append(None if sc is None else sc._scale.clone())
        torch.cuda.synchronize()
        r, pk = ring.cpu(), peak.cpu()
        first = None
        for i in range(a.steps):
            bad = [names[j] for j in range(len(names)) if not bool(r[i, j])]
            if bad:
                first = (i, bad)
                break
        sc = [None if s is None else float(s) for s in scales]
        print("run {} cache={} multi_stream={} B={} {}: loss {:.5f} -> {:.5f}; loss scale {} -> {}; first non-finite: {}".format(
            run, os.environ.get("DD_AMP_CACHE", "default"), not a.single_stream, a.batch, a.amp, float(pk[0, 0]), float(pk[-1, 0]), sc[0], sc[-1], first), flush=True)
        if first is not None:
            i = first[0]
            for k in range(max(0, i - 2), min(a.steps, i + 2)):
                print("   step {}: ".format(k) + ", ".join("{} {:.3g}".format(names[j], float(pk[k, j])) for j in range(1 + 2 * len(nets))), "scale", sc[k])
            if level >= 1:
                cols = sorted(range(1 + 2 * len(nets), len(names)), key=lambda j: -float(pk[i, j]) if bool(r[i, j]) else -float("inf"))
                print("   first non-finite module outputs at step {}: {}".format(i, [names[j] for j in range(1 + 2 * len(nets), len(names)) if not bool(r[i, j])][:8]))
                print("   largest finite module outputs: {}".format([(names[j], float(pk[i, j])) for j in cols[:6]]))
        else:
            top = sorted(range(1 + 2 * len(nets), len(names)), key=lambda j: -float(pk[:, j].max()))[:5]
            if top:
                print("   largest module outputs over the run: {}".format([(names[j], float(pk[:, j].max())) for j in top]))
        for h in handles:
            h.remove()
        optimizer.zero_grad = orig_zero
        if first is not None and os.environ.get("DD_PROBE_POSTMORTEM", "1") == "1":
            post_mortem(tr, batch)
            break
        del tr


def post_mortem(tr, batch):
    """The run went non-finite and stayed so with finite weights: what state carries it?  Inputs, buffers, optimizer state, and the
    first module whose output is non-finite in a SYNCHRONISED single-stream forward on the same weights and batch (eval of the
    same function without the race: if this forward is finite, the poison is not in any persistent tensor)."""
    def bad(t):
        return torch.is_tensor(t) and t.is_floating_point() and not bool(torch.isfinite(t).all())
    torch.cuda.synchronize()
    print("   post-mortem: non-finite inputs:", [str(k) for k, v in batch.items() if bad(v)][:5])
    print("   post-mortem: non-finite parameters:", [n for n, p in tr.base_model.named_parameters() if bad(p)][:5])
    print("   post-mortem: non-finite buffers:", [n for n, b in tr.base_model.named_buffers() if bad(b)][:8])
    st = tr.optim["optimizer"].state
    print("   post-mortem: non-finite optimizer state tensors:", sum(1 for v in st.values() for t in v.values() if bad(t)))
    seen = []
    handles = []
    for n, m in tr.base_model.named_modules():
        if list(m.children()):
            continue

        def hook(_m, inp, out, n=n):
            if not seen and torch.is_tensor(out) and bad(out):
                ins = [float(t.detach().abs().max()) if torch.is_tensor(t) and t.is_floating_point() else None for t in inp]
                fin = [bool(torch.isfinite(t).all()) if torch.is_tensor(t) and t.is_floating_point() else None for t in inp]
                seen.append((n, type(_m).__name__, ins, fin, str(out.dtype)))
        handles.append(m.register_forward_hook(hook))
    for multi in (False, True):
        seen.clear()
        tr.opt.multi_stream = multi
        with torch.no_grad():
            _, losses = tr.process_batch(dict(batch))
        torch.cuda.synchronize()
        print("   post-mortem: {} forward on the same weights and batch: loss {} ; first non-finite module output: {}".format(
            "multi-stream" if multi else "single-stream", float(losses["loss"]), seen[:1]))
    for h in handles:
        h.remove()


if __name__ == "__main__":
    main()
