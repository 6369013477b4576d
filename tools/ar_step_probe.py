#!/usr/bin/env python
"""Timing probe (GPU box): HIP-event time of every C-ABI call of ONE folded AR decode step (64 prompts, d 512, 4 layers)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prompt_tts_amd.ar as par  # noqa: E402
from prompt_tts_amd import _lib as L  # noqa: E402

fused = len(sys.argv) < 2 or sys.argv[1] != "0"
par.AR_FUSED = fused
torch.manual_seed(3)
ar = par.ARCodecDecoder(512, 4, 8, 1024, 8, max_frames=64, dtype=torch.bfloat16).to("cuda:0")
ctx = (torch.randn(64, 64, 512, generator=torch.Generator().manual_seed(9)) * 0.5).to("cuda:0")
ar.generate(ctx, 8, graph=False); torch.cuda.synchronize()
recs, orig = [], {}


def wrap(name, fn):
    def inner(*a):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a); e1.record()
        info = ""
        if name in ("pt_gemm", "pt_decode_linear"):
            d = a[0]._obj; info = f"M={d.M} N={d.N} K={d.K}"
        recs.append((name, info, e0, e1))
        return r
    return inner


for name in L.SIGNATURES:
    orig[name] = getattr(L.lib, name); setattr(L.lib, name, wrap(name, orig[name]))
try:
    ar.generate(ctx, 40, graph=False); torch.cuda.synchronize()
finally:
    for n, f in orig.items():
        setattr(L.lib, n, f)
import collections
agg = collections.OrderedDict()
per_frame = len(recs) // 40
for name, info, e0, e1 in recs[-per_frame:]:
    k = (name, info)
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
tot = 0.0
for (name, info), (c, us) in agg.items():
    print(f"{name:20s} {info:28s} x{c:3d}  {us / c:7.1f} us each  {us:8.1f} us"); tot += us
print(f"fused={fused}: {per_frame} C-ABI calls in the last frame, {tot:.0f} us inside them")
