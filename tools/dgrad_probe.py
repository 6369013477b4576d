#!/usr/bin/env python
"""Diagnostic: data-gradient GEMM dx = dy W with W read transposed in place (T-mode B operand) vs a pre-transposed copy W^T
(plain K-contiguous B operand), at the linear-layer shapes of workload B.  Usage: python tools/dgrad_probe.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_tts_amd import ops, _lib as L

dev = torch.device("cuda:0")


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


for M, N, K in [(32768, 512, 512), (32768, 1536, 512), (32768, 512, 2048), (32768, 4096, 512), (16384, 512, 512), (16384, 4096, 512), (8192, 512, 512)]:
    # dy [M][N] (N = out features = reduction), W [N][K] -> dx [M][K]
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    wt = w.t().contiguous()
    dx = torch.empty(M, K, dtype=torch.bfloat16, device=dev); dx2 = torch.empty_like(dx)
    t_t = timed(lambda: ops.gemm(M, K, N, ops.plain(dy), ops.plain(w, trans=True), dx, L.PT_BF16))
    t_n = timed(lambda: ops.gemm(M, K, N, ops.plain(dy), ops.plain(wt), dx2, L.PT_BF16))
    assert torch.equal(dx, dx2) or float((dx.float() - dx2.float()).abs().max()) < 1e-2 * float(dx.float().abs().max())
    fl = 2.0 * M * N * K
    print(f"dy [{M}][{N}] W [{N}][{K}]: T-mode {t_t*1e3:7.1f} us ({fl/t_t/1e9:5.0f} TF)   pre-transposed {t_n*1e3:7.1f} us ({fl/t_n/1e9:5.0f} TF)", flush=True)

# conv k3 dgrad: dx[m][ci] = sum_tap sum_co dy[m + 1 - tap... (gather)][co] W[co][2 - tap][ci]
for B, n, C in [(32, 1024, 512), (32, 512, 512), (32, 1024, 1024)]:
    cin = cout = C if C == 512 else 512
    cin_d = C                                                     # channels dy carries (the conv's cout)
    M = B * n
    dy = torch.randn(M, cin_d, device=dev).to(torch.bfloat16)
    w3 = (torch.randn(cin_d, 3, cin, device=dev) * 0.03).to(torch.bfloat16)          # shadow layout [Cout][3][Cin]
    w3m = w3.view(cin_d, 3 * cin)
    wd = w3.flip(1).permute(2, 1, 0).reshape(cin, 3 * cin_d).contiguous()               # [ci][tap*cout + co] = W[co][2-tap][ci]
    dx = torch.empty(M, cin, dtype=torch.bfloat16, device=dev); dx2 = torch.empty_like(dx)
    a = ops.conv(dy, cin_d, n, n, L.PT_MAP_S1)
    t_t = timed(lambda: ops.gemm(M, cin, 3 * cin_d, a, ops.wflip(w3m, cin_d, cin), dx, L.PT_BF16))
    t_n = timed(lambda: ops.gemm(M, cin, 3 * cin_d, a, ops.plain(wd), dx2, L.PT_BF16))
    assert float((dx.float() - dx2.float()).abs().max()) < 1e-2 * float(dx.float().abs().max())
    fl = 2.0 * M * cin * 3 * cin_d
    print(f"conv dgrad B={B} n={n} cout={cin_d} cin={cin}: wflip T-mode {t_t*1e3:7.1f} us ({fl/t_t/1e9:5.0f} TF)   pre-transposed {t_n*1e3:7.1f} us ({fl/t_n/1e9:5.0f} TF)", flush=True)
