#!/usr/bin/env python
"""Diagnostic: write-only / read-only / copy bandwidth of this MI355X with torch's streaming kernels (what a GEMM epilogue's write
phase can hope for).  Usage: python tools/write_bw_probe.py"""
import torch
dev = torch.device("cuda:0")


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for mb in (32, 128, 1024):
    n = mb * (1 << 20) // 2
    x = torch.empty(n, dtype=torch.bfloat16, device=dev); y = torch.empty_like(x)
    t_fill = timed(lambda: x.fill_(1.0))
    t_sum = timed(lambda: x.sum())
    t_copy = timed(lambda: y.copy_(x))
    print(f"{mb:5d} MB: fill {mb / t_fill / 1e3:6.2f} TB/s ({t_fill * 1e3:6.1f} us)   sum(read) {mb / t_sum / 1e3:6.2f} TB/s   copy {2 * mb / t_copy / 1e3:6.2f} TB/s (r+w)", flush=True)
