"""CPU, world_size 2 over gloo: the data-parallel reducer (bucketing by module, coverage, sum semantics)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from prompt_tts_amd.parallel import GradReducer
    mods = [nn.Linear(10, 10) for _ in range(5)]
    spans = {id(m): (i * 128, i * 128 + 110) for i, m in enumerate(mods)}         # 18-element alignment gaps
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)                   # tail 640..1000 never announced
    r = GradReducer(flat, lambda m: spans[id(m)], bucket_bytes=4 * 200)
    r.begin()
    for m in reversed(mods):                                                       # backward order
        r.on_ready(m)
    mid = list(r.launched)
    r.finish()
    want = torch.arange(1000, dtype=torch.float32) * 3.0                           # (1 + 2) * base: plain SUM
    q.put((rank, bool(torch.equal(flat, want)), mid, list(r.launched), r.grad_scale))
    dist.destroy_process_group()


def test_reducer_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, mid, launched, scale in res:
        assert ok, f"rank {rank}: reduced buffer differs from the sum"
        assert scale == 0.5
        assert mid, "buckets must be launched during backward, not only at finish()"
        cover = sorted(launched)
        assert cover[0][0] == 0 and cover[-1][1] == 1000
        assert all(a[1] <= b[0] for a, b in zip(cover, cover[1:])), "no element reduced twice"
        assert sum(hi - lo for lo, hi in cover) == 1000


def test_reducer_single_process_is_noop():
    from prompt_tts_amd.parallel import GradReducer
    flat = torch.ones(16)
    r = GradReducer(flat, lambda m: (0, 16))
    r.begin(); r.on_ready(nn.Linear(2, 2)); r.finish()
    assert r.grad_scale == 1.0 and torch.equal(flat, torch.ones(16)) and r.launched == []
