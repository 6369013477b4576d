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
