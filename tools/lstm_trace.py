#!/usr/bin/env python
"""In-kernel phase times of the persistent Encodec LSTM (diagnostic build:
`make -C prompt_tts_amd/csrc exp EXP_NAME=lstmtrace EXP_SRCS=encodec EXP_FLAGS=-DLSTM_TRACE=1`, then
PT_TTS_LIB=prompt_tts_amd/csrc/build/exp/lstmtrace/lib.so python tools/lstm_trace.py).  Thread 0 of workgroups 0 / 10 / 20 / 30 of
cluster 0 stamps s_memtime over ticks 200 .. 263: tick top, poll done (+ rounds), MFMAs issued, barrier 1, partial sums in LDS +
barrier 2, gate math done, publish issued."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from decode_codec import random_decoder_weights   # noqa: E402
from prompt_tts_amd import _lib as L   # noqa: E402
from prompt_tts_amd.encodec import EncodecDecoder   # noqa: E402

dev = torch.device("cuda:0")
dec = EncodecDecoder(random_decoder_weights(0), device=dev, dtype=torch.bfloat16)
codes = torch.randint(0, 1024, (64, 8, 1024), generator=torch.Generator().manual_seed(7)).to(dev)
for _ in range(2):
    dec.decode(codes)
torch.cuda.synchronize()
lib = C.CDLL(L.LIB_PATH)
buf = (C.c_ulonglong * 2048)()
assert lib.pt_debug_lstm_trace(buf, 2048) == 0
st = list(buf)
names = ["poll", "mfma issue", "barrier1", "red+barrier2", "gate math", "publish", "to next top"]
for slot in range(4):
    print(f"workgroup {10 * slot} of cluster 0:")
    tot = [0] * 7; rounds = 0; n = 0
    for t in range(2, 62):
        s = st[(slot * 64 + t) * 8:(slot * 64 + t) * 8 + 8]
        nxt = st[(slot * 64 + t + 1) * 8]
        d = [s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], s[6] - s[5], nxt - s[6]]
        if t < 8:
            print(f"  tick {200 + t}: " + "  ".join(f"{nm} {x:5d}" for nm, x in zip(names, d)) + f"  rounds {s[7] & 0xffff}  1st round {s[7] >> 16}  total {nxt - s[0]}")
        tot = [a + b for a, b in zip(tot, d)]; rounds += s[7] & 0xffff; n += 1
    print("  mean:     " + "  ".join(f"{nm} {x / n:7.0f}" for nm, x in zip(names, tot)) + f"  rounds {rounds / n:.2f}  total {sum(tot) / n:.0f}")
print("(stamps go to LDS and leave at the end of the kernel; `1st round` = the first poll round alone, `rounds` = failed rounds before the good one)")
