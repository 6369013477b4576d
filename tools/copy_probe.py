#!/usr/bin/env python
"""Diagnostic (GPU box): which host call sites of one training step end up as ATen copy / fill launches
(`__amd_rocclr_copyBuffer`, elementwise copy kernels, fills)."""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from prompt_tts_amd.tts.models import TTSSingleSpeaker  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    wl = dict(bench.WORKLOADS["B"])
    cfg = bench.make_config(wl["d"], wl["L"], wl["text_layers"], wl["n_q"], wl["T"], 256)
    torch.manual_seed(0)
    model = TTSSingleSpeaker(cfg, dtype=torch.bfloat16).to(dev)
    batch = [x.to(dev) for x in bench.synthetic_batch(wl["B"], wl["n_q"], wl["T"], 256, 1234)]
    for _ in range(2):
        model.train_step(*batch)
    torch.cuda.synchronize()
    hist = collections.Counter()

    def site():
        for fr in reversed(traceback.extract_stack(limit=12)[:-2]):
            if "prompt_tts_amd" in fr.filename:
                return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:70]}"
        return "?"
    names = ["copy_", "clone", "contiguous", "zero_", "fill_", "to", "index_copy_", "add_", "mul_", "__setitem__", "__getitem__", "sum", "float", "expand"]
    orig = {n: getattr(torch.Tensor, n) for n in names}

    def wrap(n):
        def f(self, *a, **k):
            if self.is_cuda and n not in ("__getitem__", "expand"):
                if not (n == "contiguous" and self.is_contiguous()) and not (n in ("to", "float") and False):
                    hist[(n, site())] += 1
            return orig[n](self, *a, **k)
        return f
    for n in names:
        setattr(torch.Tensor, n, wrap(n))
    tfn = {n: getattr(torch, n) for n in ("zeros", "zeros_like", "cat", "full", "ones")}
    for n, fn in tfn.items():
        def g(*a, _fn=fn, _n=n, **k):
            hist[(_n, site())] += 1
            return _fn(*a, **k)
        setattr(torch, n, g)
    try:
        model.train_step(*batch)
        torch.cuda.synchronize()
    finally:
        for n in names:
            setattr(torch.Tensor, n, orig[n])
        for n, fn in tfn.items():
            setattr(torch, n, fn)
    for (n, s), c in hist.most_common(40):
        print(f"{c:5d}  {n:12s} {s}")


if __name__ == "__main__":
    main()
