"""torch.optim.Adam's update for every parameter of the step in ONE launch (dd_adam_multi, csrc/dd_adam.hip).

The optimizer object stays torch's (reference Trainer.py:492-497: optim.Adam over the phase's parameter list; its state_dict is what
`adam.pth` holds, Trainer.py:707): this class only replaces what `optimizer.step()` LAUNCHES.  It reads the parameters, their .grad
tensors and the per-parameter state (`exp_avg`, `exp_avg_sq`, `step`) out of the optimizer once, builds the record table on the
device, and step() launches the kernel (and its prologue, which advances the step counters as optimizer.step() does) once per
parameter group.  Addresses are baked in: build it after the state exists and the gradients sit where they will stay (the flat gradient buffers
of segments.SegmentedStep), and build it again when either changes.  The learning rate is a launch argument, read at step() time --
under graph capture it is a constant of the captured launch, exactly like torch's fused Adam with a float lr.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L

_REC = np.dtype([("param", "<u8"), ("grad", "<u8"), ("exp_avg", "<u8"), ("exp_avg_sq", "<u8"), ("step", "<u8"), ("n", "<i8")])


def supported(optimizer):
    """Plain Adam as the reference configures it: no amsgrad, not maximising, fp32 dense parameters on one GPU, step counters on
    the device (capturable / fused)."""
    if type(optimizer) is not torch.optim.Adam:
        return False
    for g in optimizer.param_groups:
        if g.get("amsgrad") or g.get("maximize") or g.get("differentiable") or torch.is_tensor(g["lr"]):
            return False
        for p in g["params"]:
            if p.grad is None:
                continue
            st = optimizer.state.get(p)
            if not st or not torch.is_tensor(st.get("step")) or not st["step"].is_cuda or st["step"].dtype != torch.float32:
                return False
            if p.dtype != torch.float32 or not p.is_cuda or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                return False
            if not _same_dense_layout(p, p.grad) or not _same_dense_layout(p, st["exp_avg"]) or not _same_dense_layout(p, st["exp_avg_sq"]):
                return False
    return True


def _same_dense_layout(a, b):
    """element i of a's memory and element i of b's memory are the same logical element, and both cover their memory densely"""
    return a.shape == b.shape and a.stride() == b.stride() and b.dtype == torch.float32 and _dense(a)


def _dense(t):
    # a permutation of a contiguous tensor: sorting the strides gives the running products of the sizes
    dims = sorted(zip(t.stride(), t.shape), reverse=True)
    want = 1
    for st, sz in reversed(dims):
        if sz == 1:
            continue
        if st != want:
            return False
        want *= sz
    return True


class MultiTensorAdam:
    def __init__(self, optimizer):
        if not supported(optimizer):
            raise ValueError("optimizer / parameters outside what dd_adam_multi covers")
        self.optimizer = optimizer
        self.lib = L.load()
        chunk = self.lib.dd_adam_chunk()
        self.groups = []
        self.keep = []                      # every tensor whose address is in a table
        for g in optimizer.param_groups:
            recs, blocks, steps = [], [], []
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = optimizer.state[p]
                n = p.numel()
                if n == 0:
                    continue
                recs.append((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(), n))
                blocks.extend((len(recs) - 1, c) for c in range((n + chunk - 1) // chunk))
                steps.append(st["step"])
                self.keep.extend((p, p.grad, st["exp_avg"], st["exp_avg_sq"], st["step"]))
            if not recs:
                continue
            dev = steps[0].device
            table = torch.from_numpy(np.array(recs, dtype=_REC).view(np.uint8).copy()).to(dev)
            bmap = torch.tensor(blocks, dtype=torch.int32).reshape(-1, 2).contiguous().to(dev)
            aux = torch.zeros(2 * len(recs), dtype=torch.float32, device=dev)
            self.groups.append({"group": g, "table": table, "map": bmap, "n_blocks": len(blocks), "n_records": len(recs), "aux": aux})

    def step(self, grad_scale=None, found_inf=None):
        for e in self.groups:
            g = e["group"]
            b1, b2 = g["betas"]
            L.check(self.lib.dd_adam_multi(C.c_void_p(e["table"].data_ptr()), e["n_records"], C.c_void_p(e["map"].data_ptr()), e["n_blocks"],
                                           C.c_void_p(e["aux"].data_ptr()), float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]),
                                           None if grad_scale is None else C.c_void_p(grad_scale.data_ptr()),
                                           None if found_inf is None else C.c_void_p(found_inf.data_ptr()), L.current_stream()), "dd_adam_multi")
