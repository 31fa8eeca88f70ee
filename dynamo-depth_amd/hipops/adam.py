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


def ensure_state(optimizer):
    """State for the parameters that carry a gradient tensor and have never been stepped -- what torch.optim.Adam._init_group
    creates lazily inside step(): a zero step counter (a float on the parameter's device for capturable / fused optimizers) and
    zero moments laid out like the parameter.  A replayed step hands EVERY parameter of a trained network a gradient view (zeros
    where autograd reaches nothing: LiteMono's unused final norm); made here, outside the capture, the zero-fills are not part of
    the optimizer graph.  Returns how many parameters were initialised."""
    made = 0
    for g in optimizer.param_groups:
        on_device = bool(g.get("capturable") or g.get("fused"))
        for p in g["params"]:
            if p.grad is None or len(optimizer.state.get(p, {})) != 0:
                continue
            st = optimizer.state[p]
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device) if on_device else torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if g.get("amsgrad"):
                st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            made += 1
    return made


def unsupported_reason(optimizer):
    """None when dd_adam_multi covers the optimizer -- plain Adam as the reference configures it: no amsgrad, not maximising, fp32
    dense parameters on one GPU, step counters on the device (capturable / fused) -- else what is in the way."""
    if type(optimizer) is not torch.optim.Adam:
        return "not torch.optim.Adam"
    for g in optimizer.param_groups:
        if g.get("amsgrad") or g.get("maximize") or g.get("differentiable") or torch.is_tensor(g["lr"]):
            return "amsgrad / maximize / differentiable / tensor learning rate"
        for p in g["params"]:
            if p.grad is None:
                continue
            st = optimizer.state.get(p)
            if not st or not torch.is_tensor(st.get("step")) or not st["step"].is_cuda or st["step"].dtype != torch.float32:
                return "a {} parameter without optimizer state, or step counters on the host (not capturable / fused)".format(tuple(p.shape))
            if p.dtype != torch.float32 or not p.is_cuda or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                return "a parameter or gradient that is not dense fp32 on the GPU"
            for name, t in (("grad", p.grad), ("exp_avg", st["exp_avg"]), ("exp_avg_sq", st["exp_avg_sq"])):
                if not _same_dense_layout(p, t):
                    return "{} of a {} parameter (strides {}) laid out differently (strides {})".format(name, tuple(p.shape), p.stride(), t.stride())
    return None


def supported(optimizer):
    return unsupported_reason(optimizer) is None


def _same_dense_layout(a, b):
    """element i of a's memory and element i of b's memory are the same logical element, and both cover their memory densely
    (dimensions of size 1 carry arbitrary strides -- a channels-last 1x1 convolution weight and its zeros_like differ there)"""
    if a.shape != b.shape or b.dtype != torch.float32:
        return False
    if any(sa != sb for sa, sb, sz in zip(a.stride(), b.stride(), a.shape) if sz != 1):
        return False
    want = 1
    for st, sz in sorted((st, sz) for st, sz in zip(a.stride(), a.shape) if sz != 1):      # dense: running products of the sizes
        if st != want:
            return False
        want *= sz
    return True


class MultiTensorAdam:
    def __init__(self, optimizer):
        why = unsupported_reason(optimizer)
        if why is not None:
            raise ValueError("optimizer / parameters outside what dd_adam_multi covers: " + why)
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
