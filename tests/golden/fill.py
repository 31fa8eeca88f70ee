"""Deterministic, key-addressed weight fill: the same state_dict key gets the same tensor in the reference's
modules (at golden-generation time) and in this tree's modules (at test time) -- no 200 MB checkpoint needed."""
import math
import zlib

import torch


def fill_state(module, seed=0):
    sd = module.state_dict()
    out = {}
    for key, t in sd.items():
        if not torch.is_floating_point(t):
            out[key] = t.clone()
            continue
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + seed) & 0x7FFFFFFF)
        leaf = key.split(".")[-1]
        if leaf == "running_var":
            v = 1.0 + 0.2 * torch.rand(t.shape, generator=g)
        elif leaf == "running_mean":
            v = 0.05 * torch.randn(t.shape, generator=g)
        elif leaf in ("gamma", "gamma_xca"):
            v = 0.05 + 0.1 * torch.rand(t.shape, generator=g)
        elif leaf == "temperature":
            v = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
        elif t.dim() <= 1:
            v = (1.0 + 0.1 * torch.randn(t.shape, generator=g)) if leaf == "weight" else 0.02 * torch.randn(t.shape, generator=g)
        else:
            fan_in = t[0].numel()
            v = torch.randn(t.shape, generator=g) * math.sqrt(2.0 / fan_in) * 0.8
        out[key] = v.to(t.dtype)
    module.load_state_dict(out)
    return module
