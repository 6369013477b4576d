#!/usr/bin/env python
"""How long does the host take to ISSUE one training step (config B) versus the device to run it?  Diagnostic only."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from prompt_tts_amd.tts.models import TTSSingleSpeaker

wl = dict(bench.WORKLOADS["B"])
cfg = bench.make_config(wl["d"], wl["L"], wl["text_layers"], wl["n_q"], wl["T"], 256)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = TTSSingleSpeaker(cfg, dtype=torch.bfloat16).to(dev)
batch = [x.to(dev) for x in bench.synthetic_batch(wl["B"], wl["n_q"], wl["T"], 256, 1234)]
for _ in range(3):
    model.train_step(*batch); torch.cuda.synchronize()
for trial in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ts = []
    for _ in range(5):
        model.train_step(*batch); ts.append(time.perf_counter())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("host issue per step (ms):", [round((b - a) * 1e3, 1) for a, b in zip([t0] + ts[:-1], ts)],
          "total incl. drain %.1f ms/step" % ((t1 - t0) / 5 * 1e3), flush=True)
if len(sys.argv) > 1:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3):
        model.train_step(*batch)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
