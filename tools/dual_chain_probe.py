#!/usr/bin/env python
"""Would two half-batch training chains on two streams beat one full-batch chain?  (two independent model replicas, B = 16
each, steps issued alternately on two streams, versus one replica at B = 32).  Diagnostic only."""
import os, sys, time
os.environ["PT_MAIN_PRIORITY"] = "0"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from prompt_tts_amd.tts.models import TTSSingleSpeaker

wl = dict(bench.WORKLOADS["B"])
cfg = bench.make_config(wl["d"], wl["L"], wl["text_layers"], wl["n_q"], wl["T"], 256)
dev = torch.device("cuda", 0)


def run(models, batches, streams, steps=6):
    for _ in range(2):
        for m, b, s in zip(models, batches, streams):
            with torch.cuda.stream(s):
                m.train_step(*b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for m, b, s in zip(models, batches, streams):
            with torch.cuda.stream(s):
                m.train_step(*b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


torch.manual_seed(0)
m32 = TTSSingleSpeaker(cfg, dtype=torch.bfloat16).to(dev)
b32 = [x.to(dev) for x in bench.synthetic_batch(32, wl["n_q"], wl["T"], 256, 1)]
print("one chain  B=32: %.1f ms" % run([m32], [b32], [torch.cuda.Stream()]), flush=True)
b16 = [x.to(dev) for x in bench.synthetic_batch(16, wl["n_q"], wl["T"], 256, 2)]
print("one chain  B=16: %.1f ms" % run([m32], [b16], [torch.cuda.Stream()]), flush=True)
m2 = TTSSingleSpeaker(cfg, dtype=torch.bfloat16).to(dev)
b16b = [x.to(dev) for x in bench.synthetic_batch(16, wl["n_q"], wl["T"], 256, 3)]
print("two chains B=16+16: %.1f ms per pair" % run([m32, m2], [b16, b16b], [torch.cuda.Stream(), torch.cuda.Stream()]), flush=True)
