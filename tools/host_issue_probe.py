#!/usr/bin/env python
"""Diagnostic: host time to ISSUE one training step (all launches queued, no wait) vs the device time of the step:
if the two are close the step is launch-bound.  Usage: python tools/host_issue_probe.py"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from prompt_tts_amd.tts.models import TTSSingleSpeaker

dev = torch.device("cuda:0")
wl = bench.WORKLOADS["B"]
cfg = bench.make_config(wl["d"], wl["L"], wl["text_layers"], wl["n_q"], wl["T"], 256)
torch.manual_seed(0)
model = TTSSingleSpeaker(cfg, dtype=torch.bfloat16).to(dev)
batch = [x.to(dev) for x in bench.synthetic_batch(wl["B"], wl["n_q"], wl["T"], 256, 1234)]
for _ in range(3):
    model.train_step(*batch)
torch.cuda.synchronize()
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.train_step(*batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"issue {1e3 * (t1 - t0):6.1f} ms   issue + drain {1e3 * (t2 - t0):6.1f} ms", flush=True)
# back-to-back: the host runs ahead of the device across steps
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    model.train_step(*batch)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"10 steps: issue {1e2 * (t1 - t0):6.1f} ms/step, wall {1e2 * (t2 - t0):6.1f} ms/step")
