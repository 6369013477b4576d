#!/usr/bin/env python
"""Timing probe (GPU box): where the Encodec decode of BASELINE configs[3] (64 x 1024 frames, bf16) spends its time, per C-ABI
entry point and, for the conv stack, per launch in issue order.  Diagnostic only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from decode_codec import random_decoder_weights   # noqa: E402
from prompt_tts_amd import _lib as L, ops   # noqa: E402
from prompt_tts_amd.encodec import EncodecDecoder   # noqa: E402


def main():
    dev = torch.device("cuda:0")
    dtype = torch.float32 if len(sys.argv) > 1 and sys.argv[1] == "f32" else torch.bfloat16
    dec = EncodecDecoder(random_decoder_weights(0), device=dev, dtype=dtype)
    codes = torch.randint(0, 1024, (64, 8, 1024), generator=torch.Generator().manual_seed(7)).to(dev)
    for _ in range(2):
        dec.decode(codes)
    torch.cuda.synchronize()
    recs = []
    orig = {}

    def wrap(name, fn):
        def inner(*a):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); r = fn(*a); e1.record()
            info = ""
            if name == "pt_gemm":
                d = a[0]._obj; info = f"M={d.M} N={d.N} K={d.K}"
            elif name == "pt_rowconv":
                d = a[0]._obj; info = f"rows={d.B * d.n_rows} cin={d.cin} taps={d.taps} cin2={d.cin2} N={d.N}"
            recs.append((name, info, e0, e1))
            return r
        return inner
    for name in L.SIGNATURES:
        orig[name] = getattr(L.lib, name); setattr(L.lib, name, wrap(name, orig[name]))
    try:
        dec.decode(codes); torch.cuda.synchronize()
    finally:
        for n, f in orig.items():
            setattr(L.lib, n, f)
    tot = 0.0
    for name, info, e0, e1 in recs:
        ms = e0.elapsed_time(e1); tot += ms
        print(f"{name:18s} {info:48s} {ms:8.3f} ms")
    print(f"sum of launches {tot:.3f} ms")
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dec.decode(codes)
    e1.record(); torch.cuda.synchronize()
    print(f"decode 64 x 1024: {e0.elapsed_time(e1) / 5:.3f} ms per batch")
    import time
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        wav, st = dec._decode(codes); t1 = time.perf_counter()
        st.event.synchronize(); t2 = time.perf_counter()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        print(f"  host: enqueue {1e3 * (t1 - t0):.2f} ms, LSTM status event +{1e3 * (t2 - t1):.2f} ms, rest of the stack +{1e3 * (t3 - t2):.2f} ms")


if __name__ == "__main__":
    main()
