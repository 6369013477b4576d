#!/usr/bin/env python
"""Timing probe (GPU box): where the Encodec ENCODE leg of bench.py (32 waveforms x 12 s, f32) spends its time, per C-ABI entry
point (HIP events around every call, in issue order, same-named calls summed) and host-side.  Diagnostic only."""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import encode_codec   # noqa: E402
from prompt_tts_amd import _lib as L   # noqa: E402

dev = torch.device("cuda:0")
enc = encode_codec.load_encoder(None, torch.float32, dev, random_weights=True)
wav = (torch.randn(32, 1, 24000 * 12, generator=torch.Generator().manual_seed(7)) * 0.3).to(dev)
for _ in range(2):
    enc.encode(wav)
torch.cuda.synchronize()
recs, orig = [], {}


def wrap(name, fn):
    def inner(*a):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a); e1.record()
        info = ""
        if name == "pt_gemm":
            d = a[0]._obj; info = f"M={d.M} N={d.N} K={d.K}"
        elif name == "pt_rowconv":
            d = a[0]._obj; info = f"rows={d.B * d.n_rows} cin={d.cin} taps={d.taps} N={d.N}"
        recs.append((name, info, e0, e1))
        return r
    return inner


for name in L.SIGNATURES:
    orig[name] = getattr(L.lib, name); setattr(L.lib, name, wrap(name, orig[name]))
try:
    enc.encode(wav); torch.cuda.synchronize()
finally:
    for n, f in orig.items():
        setattr(L.lib, n, f)
agg = collections.OrderedDict()
for name, info, e0, e1 in recs:
    k = (name, info); ms = e0.elapsed_time(e1)
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += ms
tot = 0.0
for (name, info), (n, ms) in agg.items():
    tot += ms
    print(f"{name:22s} {info:44s} x{n:<4d} {ms:8.3f} ms")
print(f"sum of launches {tot:.3f} ms")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enc.encode(wav); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"encode 32 x 12 s: host enqueue {1e3 * (t1 - t0):.2f} ms, then +{1e3 * (t2 - t1):.2f} ms to completion = {1e3 * (t2 - t0):.2f} ms")
