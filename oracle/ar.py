"""Oracle for the build-defined autoregressive codec-token decoder (prompt_tts_amd/ar.py).  TEST INFRASTRUCTURE ONLY.

parity unpinned: the reference contains no autoregressive model, logits head or sampler (SURVEY 0, 8a').  This is a plain
PyTorch f32 statement of the semantics the build defines -- the oracle's diffusers-style BasicTransformerBlock with a causal
self-attention mask, F.linear logits head, torch.argmax / torch.topk + softmax + inverse-CDF sampling with injected uniforms --
so that the HIP path is checked against torch ops, as SURVEY 8a' prescribes.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .blocks import Attention, FeedForward


def sinusoid(T, d):
    pos = torch.arange(T, dtype=torch.float32)[:, None]
    inv = torch.exp(-math.log(10000.0) * torch.arange(0, d, 2, dtype=torch.float32) / d)[None, :]
    tab = torch.zeros(T, d)
    tab[:, 0::2] = torch.sin(pos * inv); tab[:, 1::2] = torch.cos(pos * inv)
    return tab


class _Block(nn.Module):
    def __init__(self, d, heads, cross):
        super().__init__()
        self.attn1 = Attention(d, None, heads, d // heads)
        self.ff = FeedForward(d)
        self.attn2 = Attention(d, cross, heads, d // heads)
        self.norm2 = nn.LayerNorm(d)
        self.norm1 = nn.LayerNorm(d)
        self.norm3 = nn.LayerNorm(d)

    def forward(self, h, ctx):
        h = h + self.attn1(self.norm1(h), is_causal=True)
        h = h + self.attn2(self.norm2(h), context=ctx)
        return h + self.ff(self.norm3(h))


class _Head(nn.Module):
    def __init__(self, d, n_q, bins):
        super().__init__()
        self.proj = nn.Linear(d, n_q * bins)


class ARCodecDecoder(nn.Module):
    """Same parameter names / registration order as prompt_tts_amd.ar.ARCodecDecoder (state_dicts interchange)."""

    def __init__(self, d_model=512, n_layers=4, n_q=8, bins=1024, heads=8, cross_attention_dim=None, max_frames=1024):
        super().__init__()
        self.d, self.n_q, self.bins, self.max_frames = d_model, n_q, bins, max_frames
        self.code_embedding = nn.Parameter(torch.randn(n_q, bins, d_model) * 0.02)
        self.bos = nn.Parameter(torch.randn(d_model) * 0.02)
        self.blocks = nn.ModuleList([_Block(d_model, heads, cross_attention_dim or d_model) for _ in range(n_layers)])
        self.norm_out = nn.LayerNorm(d_model)
        self.head = _Head(d_model, n_q, bins)

    def forward(self, codes, ctx):
        B, n_q, T = codes.shape
        prev = torch.roll(codes, 1, dims=2)
        x = sum(self.code_embedding[q][prev[:, q]] for q in range(n_q))            # (B, T, d)
        x = torch.cat([self.bos.expand(B, 1, -1), x[:, 1:]], dim=1) + sinusoid(T, self.d).to(x.device)[None]
        for blk in self.blocks:
            x = blk(x, ctx)
        return F.linear(self.norm_out(x), self.head.proj.weight, self.head.proj.bias).view(B, T, n_q, self.bins)
