#!/usr/bin/env python
"""Diagnostic: forward GEMM time per tile configuration (PT_GEMM_TILE is read once per process: run once per setting).
Times: with a residual operand / without.  Usage: PT_GEMM_TILE=128|256|512|8 python tools/tile_probe.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_tts_amd import ops, _lib as L

dev = torch.device("cuda:0")


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


row = f"tile={os.environ.get('PT_GEMM_TILE', 'auto'):>4}:"
for M, N, K in [(32768, 512, 512), (16384, 512, 512), (16384, 512, 1536), (16384, 1536, 512), (16384, 512, 2048), (32768, 1536, 512), (32768, 4096, 512), (32768, 512, 2048)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t = timed(lambda: ops.gemm(M, N, K, ops.plain(a), ops.plain(w), out, L.PT_BF16, bias=bias, residual=res, ldr=N))
    t0 = timed(lambda: ops.gemm(M, N, K, ops.plain(a), ops.plain(w), out, L.PT_BF16, bias=bias))
    row += f"  {M}x{N}x{K} {t*1e3:6.1f} / {t0*1e3:6.1f} us"
print(row, flush=True)
