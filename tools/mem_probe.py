#!/usr/bin/env python
"""Isolated timing of the HBM-bound kernels (norms, column sums, GEGLU) at the UNet's per-level shapes.
Diagnostic only (run on the GPU box); never imported by the product."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_tts_amd import engine as E, ops   # noqa: E402

dev = "cuda"
bf = torch.bfloat16


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
    B, G = 32, 32
    print(f"{'kernel':22s} {'tokens':>7s} {'C':>5s} {'us':>8s} {'GB/s':>8s}")
    levels = (1024, 512, 256, 128, 64) if len(sys.argv) < 2 else tuple(int(a) for a in sys.argv[1].split(","))
    for n in levels:
        for Cc in (512, 1024) if len(sys.argv) < 3 else (int(sys.argv[2]),):
            M = B * n
            x = torch.randn(M, Cc, device=dev, dtype=bf); dy = torch.randn(M, Cc, device=dev, dtype=bf)
            gamma = torch.ones(Cc, device=dev); beta = torch.zeros(Cc, device=dev)
            gg = torch.zeros(Cc, device=dev); gb = torch.zeros(Cc, device=dev)
            nb = M * Cc * 2
            y, s = E.groupnorm_fwd(x, None, gamma, beta, B, n, G, 1e-5, True)
            rows = [
                ("colsum", lambda: ops.colsum(dy, gb, M, Cc), nb),
                ("gn_fwd(stats+apply)", lambda: E.groupnorm_fwd(x, None, gamma, beta, B, n, G, 1e-5, True), 3 * nb),
                ("gn_bwd(sums+apply)", lambda: E.groupnorm_bwd(dy, x, None, s, gamma, beta, gg, gb, B, n, G, True), 5 * nb),
            ]
            if Cc == 512:
                yl, st = E.layernorm_fwd(x, gamma, beta)
                rows += [("ln_fwd", lambda: E.layernorm_fwd(x, gamma, beta), 2 * nb),
                         ("ln_bwd", lambda: E.layernorm_bwd(dy, x, st, gamma, gg, gb), 3 * nb)]
                proj = torch.randn(M, 8 * Cc, device=dev, dtype=bf); out = torch.empty(M, 4 * Cc, device=dev, dtype=bf)
                dproj = torch.empty_like(proj)
                rows += [("geglu_fwd", lambda: ops.geglu_fwd(proj, out), 12 * nb),
                         ("geglu_bwd", lambda: ops.geglu_bwd(out, proj, dproj), 20 * nb)]
            for name, fn, bytes_ in rows:
                us = timeit(fn)
                print(f"{name:22s} {M:7d} {Cc:5d} {us:8.1f} {bytes_ / us / 1e3:8.0f}", flush=True)


if __name__ == "__main__":
    main()
