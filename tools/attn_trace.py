#!/usr/bin/env python
"""In-kernel phase times of the attention backward dQ kernel (diagnostic build: `make -C prompt_tts_amd/csrc trace`, then
PT_TTS_LIB=prompt_tts_amd/csrc/build/trace/libprompt_tts_hip_trace.so python tools/attn_trace.py).  One wave of one workgroup in
the middle of the grid stamps s_memtime at: tile top, after the LDS-DMA issue, after the scores' arithmetic, after the dQ MFMAs
were issued, after the DMA wait, after the barrier."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_tts_amd import _lib as L, ops  # noqa: E402

B, H, N, D = 32, 8, 1024, 64
Cc = H * D
mk = lambda n: torch.randn(B * n, Cc, device="cuda", dtype=torch.bfloat16)
q, k, v, do = mk(N), mk(N), mk(N), mk(N)
o = torch.empty_like(q); lse = torch.empty(B, H, N, device="cuda"); delta = torch.empty_like(lse)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
ops.attn_fwd(q, k, v, o, lse, B, H, N, N, D, D ** -0.5)
for _ in range(3):
    ops.attn_bwd(q, k, v, o, lse, do, delta, dq, dk, dv, B, H, N, N, D, D ** -0.5)
torch.cuda.synchronize()
lib = C.CDLL(L.LIB_PATH)
buf = (C.c_ulonglong * 512)()
assert lib.pt_debug_attn2_trace(buf, 512) == 0
st = list(buf)
if os.environ.get("A2_KV") == "1":     # library built with `make trace A2_TRACE=2`: the dK/dV kernel's stamps
    names = ["blk0 S,dP,exp", "blk0 dV,dK", "blk1 S,dP,exp", "blk1 dV,dK", "(end)", "DMA wait", "barrier"]
    for t in range(16):
        s = st[8 * t + 1: 8 * t + 9]
        d = [s[i + 1] - s[i] for i in range(7)]
        print(f"step {t:2d}: " + "  ".join(f"{n} {x:5d}" for n, x in zip(names, d)) + f"   total {s[7] - s[0]}")
    print("(every stamp costs ~180 cycles itself)")
    sys.exit(0)
names = ["reads + 8 MFMA", "dma issue", "8 MFMA + exp + mul", "dQ MFMAs issued", "DMA wait", "barrier", "loop"]
print(f"prologue -> first tile: {st[1] - st[0]} cycles")
for t in range(16):
    s = st[8 * t + 1: 8 * t + 7]
    nxt = st[8 * (t + 1) + 1] if t < 15 else s[-1]
    s7 = st[8 * t + 7]
    d = [s[1] - s[0], s7 - s[1], s[2] - s7, s[3] - s[2], s[4] - s[3], s[5] - s[4], nxt - s[5]]
    print(f"tile {t:2d}: " + "  ".join(f"{n} {x:5d}" for n, x in zip(names, d)) + f"   total {nxt - s[0]}")
print("(every stamp costs ~180 cycles itself)")
