#!/usr/bin/env python
"""In-kernel phase times of the fused f32-class Encodec stage kernels (diagnostic build:
`make -C prompt_tts_amd/csrc exp EXP_NAME=x2trace EXP_SRCS=encodec_x2 EXP_FLAGS=-DX2_TRACE=1`, then
PT_TTS_LIB=prompt_tts_amd/csrc/build/exp/x2trace/lib.so python tools/x2_trace.py).  Lane 0 of waves 0 and 5 (tail: 0 and 3) of
workgroup 0 stamps s_memtime around every barrier of its first 8 tiles; printed are cycles per phase (work up to the barrier, then
the wait at the barrier), tiles 2 .. 7 averaged."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from decode_codec import random_decoder_weights   # noqa: E402
from prompt_tts_amd import _lib as L   # noqa: E402
from prompt_tts_amd.encodec import EncodecDecoder   # noqa: E402

dev = torch.device("cuda:0")
dec = EncodecDecoder(random_decoder_weights(0), device=dev, dtype=torch.float32)
codes = torch.randint(0, 1024, (64, 8, 1024), generator=torch.Generator().manual_seed(7)).to(dev)
for _ in range(2):
    dec.decode(codes)
torch.cuda.synchronize()
lib = C.CDLL(L.LIB_PATH)
buf = (C.c_ulonglong * 768)()
assert lib.pt_debug_x2_trace(buf, 768) == 0
st = list(buf)
phases = {
    1: ("stage2", ["top barrier", "tconv", "bar", "k3", "bar", "1x1+stage", "bar", "fill+store", "to top"]),
    2: ("tail", ["top barrier", "B tconv", "bar", "C k3", "bar", "D 1x1", "bar", "E final", "bar", "F fill+sum+store", "to top"]),
}
for k, (name, names) in phases.items():
    for w in range(2):
        tot = [0] * len(names); n = 0; whole = 0
        for it in range(2, 7):
            s = st[((k * 2 + w) * 8 + it) * 16:((k * 2 + w) * 8 + it) * 16 + 16]
            nxt = st[((k * 2 + w) * 8 + it + 1) * 16]
            d = [s[i + 1] - s[i] for i in range(len(names) - 1)] + [nxt - s[len(names) - 1]]
            tot = [a + b for a, b in zip(tot, d)]; n += 1; whole += nxt - s[0]
        print(f"{name} wave {'0' if w == 0 else 'other'}: " + "  ".join(f"{nm} {x / n:6.0f}" for nm, x in zip(names, tot)) + f"   tile {whole / n:.0f} cycles")
        if k == 1:      # inside the transposed conv: MFMAs of row tile 0 done (9), its epilogue issued (10), MFMAs of row tile 1 done (11)
            sub = [0, 0, 0, 0]
            for it in range(2, 7):
                s = st[((k * 2 + w) * 8 + it) * 16:((k * 2 + w) * 8 + it) * 16 + 16]
                sub = [a + b for a, b in zip(sub, [s[9] - s[1], s[10] - s[9], s[11] - s[10], s[2] - s[11]])]
            print("    tconv: " + "  ".join(f"{nm} {x / 5:6.0f}" for nm, x in zip(["products 0", "epilogue 0", "products 1", "epilogue 1"], sub)))
            fill = sum(st[((k * 2 + w) * 8 + it) * 16 + 13] - st[((k * 2 + w) * 8 + it) * 16 + 12] for it in range(2, 7)) / 5
            print(f"    fill (wait for the prefetched rows + LDS writes) {fill:.0f}")
