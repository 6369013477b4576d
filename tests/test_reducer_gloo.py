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


# ---- reduce-scatter + sharded optimizer + all-gather (parallel.ShardedGradReducer) ---------------------------------------------
def _sharded_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from prompt_tts_amd.parallel import ShardedGradReducer
    mods = [nn.Linear(10, 10) for _ in range(5)]
    spans = {id(m): (i * 128, i * 128 + 110) for i, m in enumerate(mods)}
    n = 1003                                                                        # not a multiple of 4 W: the last bucket keeps a tail
    base = torch.arange(n, dtype=torch.float32)
    geos = []
    for step in range(2):
        flat = base * (rank + 1) + step
        r = ShardedGradReducer(flat, lambda m: spans[id(m)], bucket_bytes=4 * 200) if step == 0 else r
        r.flat = flat
        r.begin()
        for m in reversed(mods):
            r.on_ready(m)
        mid = list(r.launched)
        r.finish()
        own, tails, foreign = r.owned_ranges(), r.tail_ranges(), r.foreign_ranges()
        want = base * 3.0 + world * step
        ok_own = all(torch.equal(flat[lo:hi], want[lo:hi]) for lo, hi in own + tails)   # summed exactly once where this rank needs it
        # "optimizer": the owner publishes f(sum) into its slices (tails: every rank), then one all-gather per bucket
        for lo, hi in own + tails:
            flat[lo:hi] = flat[lo:hi] * 2.0 + 1.0
        r.allgather_published()
        ok_all = bool(torch.equal(flat, want * 2.0 + 1.0))                          # every rank holds every updated value
        geos.append((own, tails, foreign, mid, list(r.launched), ok_own, ok_all))
    q.put((rank, geos))
    dist.destroy_process_group()


def test_sharded_reducer_world2_gloo_every_element_once_and_all_gathered():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for step in range(2):
        (own0, tails0, for0, mid0, l0, ok_own0, ok_all0), (own1, tails1, for1, mid1, l1, ok_own1, ok_all1) = res[0][step], res[1][step]
        assert ok_own0 and ok_own1 and ok_all0 and ok_all1
        assert mid0 and l0 == l1 and tails0 == tails1
        # ownership partitions the buffer: rank 0's slices + rank 1's slices + the shared tails cover [0, n) exactly once
        cover = sorted(own0 + own1 + tails0)
        assert cover[0][0] == 0 and cover[-1][1] == 1003 and sum(hi - lo for lo, hi in cover) == 1003
        assert all(a[1] <= b[0] for a, b in zip(cover, cover[1:]))
        assert sorted(for0) == sorted(own1) and sorted(for1) == sorted(own0)        # what one rank imports is what the other owns
        assert all(lo % 4 == 0 for lo, _ in own0 + own1)                            # quad-aligned slices (16-byte optimizer accesses)
        assert tails0 and sum(hi - lo for lo, hi in tails0) < (8 + 4) * len(l0)     # < 4 W (+ a head of < 4) elements per bucket
    assert res[0][0][:3] == res[0][1][:3]                                           # same geometry every step
