"""CPU: the restated third-party arithmetic (diffusers 0.15.x, absent here) cross-checked against torch primitives
and closed forms -- the reference itself has no tests for it ("parity unpinned" by the reference)."""
import math

import torch
import torch.nn.functional as F
from torch import nn

from oracle import blocks as ob


def test_attention_matches_sdpa_and_multihead():
    torch.manual_seed(0)
    a = ob.Attention(64, None, heads=4, dim_head=16)
    x = torch.randn(2, 10, 64)
    q, k, v = (l(x).view(2, 10, 4, 16).transpose(1, 2) for l in (a.to_q, a.to_k, a.to_v))
    want = a.to_out[0](F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(2, 10, 64))
    assert torch.allclose(a(x), want, atol=1e-5)
    mha = nn.MultiheadAttention(64, 4, bias=True, batch_first=True)
    with torch.no_grad():
        mha.in_proj_weight.copy_(torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight]))
        mha.in_proj_bias.zero_()
        mha.out_proj.weight.copy_(a.to_out[0].weight); mha.out_proj.bias.copy_(a.to_out[0].bias)
    assert torch.allclose(a(x), mha(x, x, x, need_weights=False)[0], atol=1e-5)
    ctx = torch.randn(2, 7, 32)
    c = ob.Attention(64, 32, heads=4, dim_head=16)
    assert c(x, ctx).shape == (2, 10, 64)
    causal = a(x, is_causal=True)
    q2, k2, v2 = q, k, v
    want_c = a.to_out[0](F.scaled_dot_product_attention(q2, k2, v2, is_causal=True).transpose(1, 2).reshape(2, 10, 64))
    assert torch.allclose(causal, want_c, atol=1e-5)


def test_block_structure_and_ignored_positional_mask():
    torch.manual_seed(1)
    blk = ob.BasicTransformerBlock(32, 2, 16)
    x = torch.randn(2, 6, 32)
    h = x + blk.attn1(blk.norm1(x))
    h = h + blk.ff(blk.norm3(h))
    assert torch.allclose(blk(x), h, atol=1e-6)
    mask = torch.full((2, 1, 6), -10000.0)
    # the reference passes its mask as positional arg #2 = encoder_hidden_states: without attn2 it changes nothing
    assert torch.equal(blk(x, mask), blk(x))
    assert set(dict(blk.named_parameters())) >= {"norm1.weight", "attn1.to_q.weight", "attn1.to_out.0.bias",
                                                  "ff.net.0.proj.weight", "ff.net.2.bias", "norm3.bias"}
    assert not any(k.startswith("norm2") or k.startswith("attn2") for k in dict(blk.named_parameters()))
    g = ob.GEGLU(8, 16)
    y = torch.randn(3, 8)
    h2, gate = g.proj(y).chunk(2, -1)
    assert torch.allclose(g(y), h2 * 0.5 * gate * (1 + torch.erf(gate / math.sqrt(2))), atol=1e-6)


def test_timestep_embedding_closed_form():
    e = ob.timestep_embedding(torch.tensor([0, 3]), 8, flip_sin_to_cos=True, downscale_freq_shift=0)
    assert torch.allclose(e[0], torch.tensor([1., 1, 1, 1, 0, 0, 0, 0]))         # [cos | sin] at t = 0
    w = torch.exp(-math.log(10000.0) * torch.arange(4) / 4)
    assert torch.allclose(e[1], torch.cat([torch.cos(3 * w), torch.sin(3 * w)]), atol=1e-6)
    e2 = ob.timestep_embedding(torch.tensor([3]), 8, flip_sin_to_cos=False)
    assert torch.allclose(e2[0], torch.cat([torch.sin(3 * w), torch.cos(3 * w)]), atol=1e-6)


def test_ddpm_schedule_and_add_noise():
    ac = ob.ddpm_alphas_cumprod()
    assert ac.shape == (1000,) and abs(float(ac[0]) - 0.9999) < 1e-7
    assert abs(float(ac[-1]) - 4.0358e-05) < 1e-8                                 # linear 1e-4..0.02, 1000 steps
    assert bool((ac[1:] < ac[:-1]).all())
    x0 = torch.randn(3, 2, 5); eps = torch.randn(3, 2, 5); t = torch.tensor([0, 500, 999])
    xt = ob.add_noise(x0, eps, t)
    for b in range(3):
        assert torch.allclose(xt[b], ac[t[b]].sqrt() * x0[b] + (1 - ac[t[b]]).sqrt() * eps[b], atol=1e-7)
    # variance preserving: a^2 + s^2 = 1 ; linear in (x0, eps)
    assert torch.allclose(ob.add_noise(2 * x0, 2 * eps, t), 2 * xt, atol=1e-6)


def test_lr_lambdas():
    f = ob.lr_lambda("constant_with_warmup", 4, 100)
    assert [f(s) for s in (0, 2, 4, 50)] == [0.0, 0.5, 1.0, 1.0]
    g = ob.lr_lambda("linear", 0, 10)
    assert g(0) == 1.0 and g(5) == 0.5 and g(10) == 0.0
    c = ob.lr_lambda("cosine", 0, 10)
    assert abs(c(5) - 0.5) < 1e-12 and c(0) == 1.0
