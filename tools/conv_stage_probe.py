#!/usr/bin/env python
"""What bounds the conv k3 forward GEMM of the training step (M = 32 x 1024 tokens, cin = cout = 512: K = 1536)?  Times the launch
under PT_GEMM_TILE = 0 (default: eight-phase) / 512 (two-stage 256 x 256), and a plain GEMM of the same M, N, K; run once per
ablation build (PT_TTS_LIB=.../build/exp/ablN/lib.so: 1 no MFMAs / LDS reads, 2 no staging loads -- two-stage kernels only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_tts_amd import _lib as L, ops   # noqa: E402

dev = torch.device("cuda:0")
B, n, cin, cout = 32, 1024, 512, 512
M, N, K = B * n, cout, 3 * cin
x = torch.randn(M, cin, device=dev, dtype=torch.bfloat16)
xp = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * K ** -0.5
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)


def timeit(f, iters=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


conv = lambda: ops.gemm(M, N, K, ops.conv(x, cin, n, n, L.PT_MAP_S1, taps=3), ops.plain(w), out, L.PT_BF16)
plain = lambda: ops.gemm(M, N, K, ops.plain(xp), ops.plain(w), out, L.PT_BF16)
print(f"tile {os.environ.get('PT_GEMM_TILE', '0')} lib {os.environ.get('PT_TTS_LIB', 'product')[-20:]}: conv k3 {timeit(conv):.1f} us, plain K=1536 {timeit(plain):.1f} us "
      f"({2 * M * N * K / 1e6:.0f} MFLOP)")
