#!/usr/bin/env python
"""Diagnostic: grouped conv k3 weight gradient with the per-item row map (first two / last k-tile of every batch item take the
address-recompute path) vs one flat item (B*n rows: the regular pointer walk everywhere; the cross-item products it adds are
removed by two rank-(B-1) correction GEMMs in the product).  Usage: python tools/convwgrad_probe.py"""
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


B, n, C = 32, 1024, 512
ws = torch.empty(ops.wgrad_group_ws_floats(256), dtype=torch.float32, device=dev)
for nconv, cin in [(4, 512), (4, 1024), (2, 1024)]:
    keep = []
    for flat in (False, True):
        descs = []
        for _ in range(nconv):
            dy = torch.randn(B * n, C, device=dev).to(torch.bfloat16); x = torch.randn(B * n, cin, device=dev).to(torch.bfloat16)
            gw = torch.zeros(C, 3 * cin, dtype=torch.float32, device=dev)
            keep.append((dy, x, gw))
            nn_ = B * n if flat else n
            descs.append(ops.gemm_desc(C, 3 * cin, B * n, ops.plain(dy, trans=True), ops.conv(x, cin, nn_, nn_, L.PT_MAP_S1, trans=True), gw,
                                       ldc=3 * cin, out_kind=L.PT_OUT_F32_ATOMIC))
        t = timed(lambda: ops.wgrad_group(descs, ws, 256))
        fl = nconv * 2.0 * B * n * C * 3 * cin
        print(f"{nconv} convs cin={cin} {'flat' if flat else 'per-item'}: {t*1e3:7.1f} us  {fl/t/1e9:5.0f} TF", flush=True)
