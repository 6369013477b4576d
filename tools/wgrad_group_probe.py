#!/usr/bin/env python
"""Timing probe (GPU box): the weight gradients of one transformer block / one resnet of BASELINE configs[1] as (a) one
pt_gemm launch each (split-K f32 atomics, the round-1 path) and (b) one pt_wgrad_group launch + fold.  Diagnostic only."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_tts_amd import _lib as L, engine as E, ops   # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    T, S = 32768, 8192
    bf = torch.bfloat16
    groups = {
        "transformer block (qkv,out,q,kv,ff1,ff2)": [(T, 1536, 512), (T, 512, 512), (T, 512, 512), (S, 1024, 512), (T, 4096, 512), (T, 512, 2048)],
        "lin 512x512 alone": [(T, 512, 512)],
        "ff1 alone": [(T, 4096, 512)],
        "two transformer blocks (8 biggest)": [(T, 1536, 512), (T, 512, 512), (T, 4096, 512), (T, 512, 2048)] * 2,
    }
    ws = torch.empty(ops.wgrad_group_ws_floats(512), dtype=torch.float32, device=dev)
    for name, probs in groups.items():
        descs, keep, old = [], [], []
        flops = 0.0
        for red, nout, kin in probs:
            dy = torch.randn(red, nout, device=dev).to(bf); x = torch.randn(red, kin, device=dev).to(bf)
            dw = torch.zeros(nout, kin, dtype=torch.float32, device=dev); gb = torch.zeros(nout, dtype=torch.float32, device=dev)
            keep.append((dy, x, dw, gb)); flops += 2.0 * red * nout * kin
            descs.append(ops.gemm_desc(nout, kin, red, ops.plain(dy, trans=True), ops.plain(x, trans=True), dw,
                                       out_kind=L.PT_OUT_F32_ATOMIC, arow_sum=gb, arow_n=nout))
            old.append((nout, kin, red, dy, x, dw, gb))

        def run_old(tgt):
            for nout, kin, red, dy, x, dw, gb in old:
                import math
                tiles = math.ceil(nout / 128) * math.ceil(kin / 128)
                sk = max(1, min(tgt // tiles, red // 64 // 4, 64))
                ops.gemm(nout, kin, red, ops.plain(dy, trans=True), ops.plain(x, trans=True), dw, L.PT_BF16,
                         out_kind=L.PT_OUT_F32_ATOMIC, split_k=sk, arow_sum=gb, arow_n=nout)
        row = f"{name:45s} {flops / 1e9:7.1f} GF |"
        for tgt in (256, 512):
            us = timeit(lambda: run_old(tgt)); row += f" atomics@{tgt}: {us:7.1f} us {flops / us / 1e6:5.0f} TF |"
        for tgt in (256, 512):
            us = timeit(lambda: ops.wgrad_group(descs, ws, tgt)); row += f" group@{tgt}: {us:7.1f} us {flops / us / 1e6:5.0f} TF |"
        print(row, flush=True)
    # resnet convs
    B = 32
    for name, cases in {"resnet (conv1 512, conv2)": [(1024, 512, 512), (1024, 512, 512)],
                        "up resnet (conv1 1024->512, conv2)": [(1024, 1024, 512), (1024, 512, 512)],
                        "three resnets": [(1024, 512, 512)] * 6}.items():
        descs, keep, old = [], [], []
        flops = 0.0
        for n, cin, cout in cases:
            x = torch.randn(B * n, cin, device=dev).to(bf); dy = torch.randn(B * n, cout, device=dev).to(bf)
            dw = torch.zeros(cout, 3 * cin, dtype=torch.float32, device=dev); gb = torch.zeros(cout, dtype=torch.float32, device=dev)
            keep.append((x, dy, dw, gb)); flops += 2.0 * B * n * cout * 3 * cin
            descs.append(ops.gemm_desc(cout, 3 * cin, B * n, ops.plain(dy, trans=True), ops.conv(x, cin, n, n, L.PT_MAP_S1, trans=True),
                                       dw, ldc=3 * cin, out_kind=L.PT_OUT_F32_ATOMIC, arow_sum=gb, arow_n=cout))
            old.append((cout, cin, n, dy, x, dw, gb))

        def run_old(tgt):
            import math
            for cout, cin, n, dy, x, dw, gb in old:
                tiles = math.ceil(cout / 128) * math.ceil(3 * cin / 128)
                sk = max(1, min(tgt // tiles, B * n // 64 // 4, 64))
                ops.gemm(cout, 3 * cin, B * n, ops.plain(dy, trans=True), ops.conv(x, cin, n, n, L.PT_MAP_S1, trans=True), dw, L.PT_BF16,
                         ldc=3 * cin, out_kind=L.PT_OUT_F32_ATOMIC, split_k=sk, arow_sum=gb, arow_n=cout)
        row = f"{name:45s} {flops / 1e9:7.1f} GF |"
        for tgt in (256, 512):
            us = timeit(lambda: run_old(tgt)); row += f" atomics@{tgt}: {us:7.1f} us {flops / us / 1e6:5.0f} TF |"
        for tgt in (256, 512):
            us = timeit(lambda: ops.wgrad_group(descs, ws, tgt)); row += f" group@{tgt}: {us:7.1f} us {flops / us / 1e6:5.0f} TF |"
        print(row, flush=True)


if __name__ == "__main__":
    main()
