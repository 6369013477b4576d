#!/usr/bin/env python
"""Soak test (GPU box): repeated decodes of BASELINE configs[3] (64 x 1024 frames) in bf16 and f32; every run must return the
SAME waveform bit for bit (the XCD-local hand-off changes which workgroup owns which units from launch to launch, never the
arithmetic) and no LSTM status word may fire (EncodecDecoder.decode raises if one does)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from decode_codec import random_decoder_weights   # noqa: E402
from prompt_tts_amd.encodec import EncodecDecoder   # noqa: E402

os.environ["PT_LSTM_RETRY"] = "0"          # a timed-out hand-off must RAISE here, not be retried with the per-step kernels
dev = torch.device("cuda:0")
codes = torch.randint(0, 1024, (64, 8, 1024), generator=torch.Generator().manual_seed(7)).to(dev)
for dtype, n in ((torch.bfloat16, 40), (torch.float32, int(os.environ.get("SOAK_F32", "150")))):
    dec = EncodecDecoder(random_decoder_weights(0), device=dev, dtype=dtype)
    ref = dec.decode(codes).clone()
    bad = 0
    for i in range(n):
        out = dec.decode(codes)
        if not torch.equal(out, ref):
            bad += 1
            print(f"{dtype}: run {i} differs: max |d| = {float((out - ref).abs().max()):.3e}", flush=True)
    print(f"{dtype}: {n} decodes, {bad} differing from the first, finite = {bool(torch.isfinite(ref).all())}", flush=True)
