#!/usr/bin/env python
"""Diagnostic: fp8 vs bf16 GEMM time (and the quantisation cost) at the linear-layer shapes of workload E (d = 1024).
Usage: python tools/fp8_probe.py [tokens]"""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_tts_amd import ops, _lib as L

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for N, K in [(1024, 1024), (3072, 1024), (8192, 1024), (1024, 4096), (4096, 4096)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    a8 = torch.empty(M, K, dtype=torch.uint8, device=dev); w8 = torch.empty(N, K, dtype=torch.uint8, device=dev)
    sa = torch.empty(258, device=dev); sw = torch.empty(258, device=dev)
    ops.fp8_quantize(w, w8, sw, 0)
    t_q = timed(lambda: ops.fp8_quantize(a, a8, sa, 0))
    t_8 = timed(lambda: ops.gemm_fp8(M, N, K, a8, w8, out, sa, sw))
    t_b = timed(lambda: ops.gemm(M, N, K, ops.plain(a), ops.plain(w), out, L.PT_BF16))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: bf16 {t_b*1e3:7.1f} us ({fl/t_b/1e9:6.0f} TF)  fp8 {t_8*1e3:7.1f} us ({fl/t_8/1e9:6.0f} TF)  quantize A {t_q*1e3:6.1f} us", flush=True)

for N, K in [(8192, 1024), (1024, 4096)]:                  # weight quantisation (with the transposed copy), once per step each
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    w8 = torch.empty(N, K, dtype=torch.uint8, device=dev); w8t = torch.empty(K, N, dtype=torch.uint8, device=dev)
    sw = torch.empty(258, device=dev)
    print(f"weight [{N}][{K}] quantize + transpose: {timed(lambda: ops.fp8_quantize(w, w8, sw, 0, out_t=w8t))*1e3:6.1f} us", flush=True)
