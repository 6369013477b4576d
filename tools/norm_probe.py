#!/usr/bin/env python
"""Times the GroupNorm / LayerNorm kernels ALONE on the shapes of the config-B training step (diagnostic; run on the GPU box):
per call, achieved HBM GB/s against the algorithmic bytes (every operand read / written once)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_tts_amd import ops  # noqa: E402


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


dev = "cuda"
bf = torch.bfloat16
for B, N, C1, C2, G, silu, dres in [(32, 1024, 512, 0, 32, 1, 1), (32, 1024, 512, 0, 32, 1, 0), (32, 1024, 512, 512, 32, 1, 0),
                                    (32, 512, 512, 0, 32, 1, 1), (32, 256, 512, 0, 32, 1, 1), (32, 1024, 256, 0, 32, 1, 1)]:
    C = C1 + C2
    x1 = torch.randn(B * N, C1, device=dev, dtype=bf); x2 = torch.randn(B * N, C2, device=dev, dtype=bf) if C2 else None
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    y = torch.empty(B * N, C, device=dev, dtype=bf); mean = torch.empty(B * G, device=dev); rstd = torch.empty(B * G, device=dev)
    f = t(lambda: ops.groupnorm_fwd(x1, x2, gamma, beta, y, mean, rstd, B, N, G, 1e-5, silu))
    dy = torch.randn_like(y); dr = torch.randn(B * N, C1, device=dev, dtype=bf) if dres and not C2 else None
    dx1 = torch.empty_like(x1); dx2 = torch.empty_like(x2) if C2 else None
    dg = torch.zeros(16 * C, device=dev); db = torch.zeros(16 * C, device=dev); ws = torch.zeros(B * G * 2 + 64, device=dev)
    b = t(lambda: ops.groupnorm_bwd(dy, x1, x2, mean, rstd, gamma, beta, dr, dx1, dx2, dg, db, ws, B, N, G, silu, n_rep=16, rep_stride=C))
    fb = 2 * B * N * C * 2; bb = (3 + (1 if dr is not None else 0)) * B * N * C * 2
    print(f"GN B{B} N{N} C{C1}+{C2} dres={int(dr is not None)}: fwd {f:7.1f} us {fb / f / 1e3:6.0f} GB/s   bwd {b:7.1f} us {bb / b / 1e3:6.0f} GB/s", flush=True)

for M, C, dres in [(32768, 512, 1), (32768, 512, 0), (16384, 512, 1), (8192, 1024, 1)]:
    x = torch.randn(M, C, device=dev, dtype=bf); dy = torch.randn_like(x); dr = torch.randn_like(x) if dres else None
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    y = torch.empty_like(x); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev); dx = torch.empty_like(x)
    dg = torch.zeros(16 * C, device=dev); db = torch.zeros(16 * C, device=dev)
    f = t(lambda: ops.layernorm_fwd(x, gamma, beta, y, mean, rstd, 1e-5))
    b = t(lambda: ops.layernorm_bwd(dy, x, mean, rstd, gamma, dr, dx, dg, db, n_rep=16, rep_stride=C))
    fb = 2 * M * C * 2; bb = (3 + dres) * M * C * 2
    print(f"LN M{M} C{C} dres={dres}: fwd {f:7.1f} us {fb / f / 1e3:6.0f} GB/s   bwd {b:7.1f} us {bb / b / 1e3:6.0f} GB/s", flush=True)
