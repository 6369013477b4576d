#!/usr/bin/env python
"""Timing probe (GPU box): ARCodecDecoder.generate, ms per frame -- folded decode step vs the training kernels, graph replay vs
launch by launch (d 512, 4 layers, 8 codebooks, 64 prompts: bench.py's `ar_generate`)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prompt_tts_amd.ar as par  # noqa: E402


def run(fused, graph, frames=64, prompts=64, oplist=True):
    par.AR_FUSED = fused
    par.AR_OPLIST = oplist
    torch.manual_seed(3)
    ar = par.ARCodecDecoder(512, 4, 8, 1024, 8, max_frames=frames, dtype=torch.bfloat16).to("cuda:0")
    ctx = (torch.randn(prompts, 64, 512, generator=torch.Generator().manual_seed(9)) * 0.5).to("cuda:0")
    ar.generate(ctx, 8, graph=graph); torch.cuda.synchronize()
    t0 = time.perf_counter()
    codes = ar.generate(ctx, frames, graph=graph); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3, codes


if __name__ == "__main__":
    ref = None
    for fused, oplist in ((False, False), (True, False), (True, True)):
        for graph in (False, True):
            ms, codes = run(fused, graph, oplist=oplist)
            ref = codes if ref is None else ref
            print(f"folded={fused} one-call-per-frame={oplist} graph={graph}: {ms:.3f} ms per frame, same codes as the first run: {bool(torch.equal(codes, ref))}", flush=True)
