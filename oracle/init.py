"""Deterministic, name-keyed weight init so fixtures need not ship weights.  TEST INFRASTRUCTURE ONLY.

Every tensor of ``module.state_dict()`` is filled from its own CPU generator seeded by
(seed, crc32 of its key); the same call on the reference model, the oracle and the
HIP model (identical keys) therefore yields identical weights.
"""
import math
import zlib

import torch


@torch.no_grad()
def deterministic_init_(module, seed=0):
    for name, p in module.state_dict().items():
        if name.endswith("inv_freq"):
            continue
        g = torch.Generator().manual_seed(seed * 1000003 + zlib.crc32(name.encode()))
        r = torch.rand(p.shape, generator=g, dtype=torch.float32) * 2 - 1
        leaf = name.rsplit(".", 1)[-1]
        is_norm = any(s in name for s in (".norm", "conv_norm_out"))
        if is_norm and leaf == "weight":
            v = 1.0 + 0.2 * r
        elif leaf == "bias":
            v = 0.1 * r
        elif "word_embedding" in name:
            v = r
        else:
            fan_in = p[0].numel() if p.dim() > 1 else p.numel()
            v = r * math.sqrt(3.0 / fan_in)   # unit-gain uniform: keeps activations O(1) through depth
        p.copy_(v.to(p.dtype))
    return module
