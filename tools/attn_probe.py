#!/usr/bin/env python
"""Times the attention kernels on the shapes of the config-B training step (diagnostic; run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_tts_amd import ops  # noqa: E402

SHAPES = [("self N1024", 32, 8, 1024, 1024, 64), ("self N512", 32, 8, 512, 512, 64), ("cross 1024x256", 32, 8, 1024, 256, 64),
          ("cross 512x256", 32, 8, 512, 256, 64), ("text 256x256", 32, 8, 256, 256, 64)]
if os.environ.get("PT_PROBE_OTHER_D") == "1":      # configs[4] (D = 128, N = 2048) and configs[0] (D = 32) shapes
    SHAPES = [("E self N2048 D128", 8, 8, 2048, 2048, 128), ("E cross 2048x256 D128", 8, 8, 2048, 256, 128),
              ("E text 256 D64 H16", 8, 16, 256, 256, 64), ("A self N1024 D32", 4, 8, 1024, 1024, 32), ("A cross D32", 4, 8, 1024, 256, 32)]


def t(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, B, H, Nq, Nk, D in SHAPES:
    C = H * D
    mk = lambda n: torch.randn(B * n, C, device="cuda", dtype=torch.bfloat16)
    q, k, v, do = mk(Nq), mk(Nk), mk(Nk), mk(Nq)
    o = torch.empty_like(q); lse = torch.empty(B, H, Nq, device="cuda"); delta = torch.empty_like(lse)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    f = t(lambda: ops.attn_fwd(q, k, v, o, lse, B, H, Nq, Nk, D, D ** -0.5))
    b = t(lambda: ops.attn_bwd(q, k, v, o, lse, do, delta, dq, dk, dv, B, H, Nq, Nk, D, D ** -0.5))
    fl = 4.0 * B * H * Nq * Nk * D
    print(f"{name:16s} fwd {f:8.1f} us {fl / f / 1e6:7.0f} TF   bwd {b:8.1f} us {3.5 * fl / b / 1e6:7.0f} TF(7 products) {2.5 * fl / b / 1e6:7.0f} TF(5)", flush=True)
