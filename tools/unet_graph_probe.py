#!/usr/bin/env python
"""Experiment (GPU box): the UNet forward of one reverse-diffusion step (config B, B = 32, T = 1024), launch by launch vs captured
once as a HIP graph and replayed -- what would graph replay save in prompt_tts_amd/sampler.sample?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, make_config, synthetic_batch   # noqa: E402
from prompt_tts_amd import engine as E, ops   # noqa: E402
from prompt_tts_amd.tts.models import TTSSingleSpeaker   # noqa: E402

dev = torch.device("cuda:0")
wl = WORKLOADS["B"]; S = 256
cfg = make_config(wl["d"], wl["L"], wl["text_layers"], wl["n_q"], wl["T"])
torch.manual_seed(0)
model = TTSSingleSpeaker(cfg, dtype=torch.bfloat16).to(dev)
x0, noise, t, ids, mask = [x.to(dev) for x in synthetic_batch(wl["B"], wl["n_q"], wl["T"], S, 1)]
st = model.store
B, T, n_q, cpad = wl["B"], wl["T"], wl["n_q"], model.unet.cpad
with torch.no_grad():
    st.ensure_shadow_fresh()
    text_emb, _ = model.text_encoder.fwd(st, ids.to(torch.int32).contiguous(), mask, B, S)
    xt = torch.randn(B * T, cpad, device=dev).to(st.dtype)
    t_dev = torch.full((B,), 500, dtype=torch.int64, device=dev)
    with E.cross_kv_cache():
        def fwd():
            return model.unet.fwd(st, xt, t_dev, text_emb, B, T, S)[0]
        for _ in range(3):
            y = fwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            y = fwd()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 10
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            yg = fwd()
        g.replay(); torch.cuda.synchronize()
        print("max |graph - eager| =", float((yg.float() - y.float()).abs().max()))
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        gr = (time.perf_counter() - t0) / 10
print(f"UNet forward B={B} T={T}: launch by launch {eager * 1e3:.2f} ms, graph replay {gr * 1e3:.2f} ms")
