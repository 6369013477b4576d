"""GPU: north_star ops without a reference counterpart (SURVEY 8a'), pinned bit-exactly to torch / numpy."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_codes_round_trip_bit_exact(dev):
    from oracle import collate as oc
    from prompt_tts_amd import ops
    rng = np.random.default_rng(0)
    codes = rng.integers(0, 1024, (4, 8, 900))
    x = torch.from_numpy(oc.normalise_codes(codes)).to(dev)
    assert np.array_equal(ops.codes_from_continuous(x).cpu().numpy(), codes)                       # exact inverse
    noisy = (x + 0.3 * torch.randn(x.shape, device=dev)).clamp(-1.5, 1.5)
    assert np.array_equal(ops.codes_from_continuous(noisy).cpu().numpy(), oc.denormalise_to_codes(noisy.cpu().numpy()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("V,k", [(1024, 1), (1024, 32), (149, 5), (2048, 64)])
def test_greedy_and_topk_indices_bit_exact(dev, dtype, V, k):
    from oracle import collate as oc
    from prompt_tts_amd import ops
    g = torch.Generator().manual_seed(V + k)
    R = 777
    logits = (torch.randn(R, V, generator=g) * 3).to(dtype)            # distinct values with probability ~1 in f32
    if dtype == torch.bfloat16:       # bf16 randn has ties (torch.topk's tie order is unspecified): plant 64 distinct leaders
        logits = logits.float().clamp(max=2.0)
        pos = torch.rand(R, V, generator=g).argsort(dim=1)[:, :min(64, V)]
        lead = 4.0 + 0.25 * torch.arange(pos.shape[1], dtype=torch.float32)          # exactly representable, all distinct
        logits.scatter_(1, pos, lead[None, :].expand(R, -1))
        logits = logits.to(dtype)
    u = torch.rand(R, generator=g)
    got = ops.sample_topk(logits.to(dev), k, u.to(dev), temperature=0.7)
    want = oc.sample_topk(logits.float(), k, u, temperature=0.7)
    assert torch.equal(got.cpu(), want)
    # ties: greedy must return the FIRST maximum, as torch.argmax
    flat = torch.zeros(8, V, dtype=dtype); flat[:, 5] = 1; flat[:, 9] = 1
    assert torch.equal(ops.sample_topk(flat.to(dev), 1).cpu(), torch.full((8,), 5))


def test_packed_key_topk_edge_cases(dev):
    """The bf16 packed-key kernel (V <= 1024): a row length that is not a multiple of 16 (the last lanes take the element-wise load
    path), -0.0 against +0.0 (equal: the lower column wins), -inf entries, a row stride that is not a multiple of 8, and k = V."""
    from oracle import collate as oc
    from prompt_tts_amd import ops
    g = torch.Generator().manual_seed(5)
    # argmax with signed zeros and -inf: first maximum wins
    z = torch.full((4, 1000), float("-inf"), dtype=torch.bfloat16)
    z[0, 7] = -0.0; z[0, 3] = 0.0; z[0, 900] = 0.0                      # max 0 at columns 3, 7 (-0), 900 -> 3
    z[1, 999] = -5.0                                                     # a single finite entry in the very last column
    z[2, :] = -1.0; z[2, 512] = -0.5
    z[3, 17] = 3.0; z[3, 16] = 3.0
    assert ops.sample_topk(z.to(dev), 1).cpu().tolist() == [3, 999, 512, 16]
    # top-k with distinct values on V = 1000 and on a strided view (ld = 1003)
    base = (torch.arange(1000, dtype=torch.int16) + 0x3C00).view(torch.bfloat16)    # 1000 consecutive (distinct, finite) bf16 values
    base = torch.where(torch.arange(1000) % 3 == 0, -base, base)                    # mixed signs, still distinct
    rows = torch.stack([base[torch.randperm(1000, generator=g)] for _ in range(33)])
    assert rows.dtype == torch.bfloat16 and all(len(torch.unique(r.float())) == 1000 for r in rows)
    u = torch.rand(33, generator=g)
    for k in (2, 31, 64):
        want = oc.sample_topk(rows.float(), k, u, temperature=1.3)
        assert torch.equal(ops.sample_topk(rows.to(dev), k, u.to(dev), temperature=1.3).cpu(), want)
    wide = torch.zeros(33, 1003, dtype=torch.bfloat16); wide[:, :1000] = rows
    view = wide.to(dev)[:, :1000]                                                   # ld = 1003: element-wise loads
    assert torch.equal(ops.sample_topk(view, 31, u.to(dev), temperature=1.3).cpu(), oc.sample_topk(rows.float(), 31, u, temperature=1.3))
