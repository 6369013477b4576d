"""Oracle restatement of the third-party (diffusers ^0.15.1) arithmetic the reference calls.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  diffusers is not vendored
under /root/reference and is not installed in this image, so these follow the
published 0.15.x algorithm; call sites in the reference:

  BasicTransformerBlock   tts/models.py:8,95-100 ; tts/ldm/transformer_1d.py:8-11,165-178
  Timesteps / TimestepEmbedding   tts/ldm/unet_1d_condition.py:17-22,209,216-222
  DDPMScheduler.add_noise / get_scheduler   train.py:8-9,32-36,60-65,96-98

Module attribute names are the diffusers ones so state_dict keys interchange.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn


class Attention(nn.Module):
    """softmax(q k^T * dim_head**-0.5) v ; to_q/to_k/to_v bias-free, to_out.0 with bias."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = heads * dim_head
        kv_dim = query_dim if cross_attention_dim is None else cross_attention_dim
        self.heads, self.dim_head = heads, dim_head
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(kv_dim, inner, bias=False)
        self.to_v = nn.Linear(kv_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

    def forward(self, x, context=None, additive_mask=None, is_causal=False):
        ctx = x if context is None else context
        b, n, _ = x.shape
        s = ctx.shape[1]
        q = self.to_q(x).view(b, n, self.heads, self.dim_head).transpose(1, 2)
        k = self.to_k(ctx).view(b, s, self.heads, self.dim_head).transpose(1, 2)
        v = self.to_v(ctx).view(b, s, self.heads, self.dim_head).transpose(1, 2)
        scores = torch.matmul(q, k.transpose(-1, -2)) * self.scale
        if additive_mask is not None:  # (b, 1, s) additive, broadcast over heads and queries
            scores = scores + additive_mask[:, None, :, :]
        if is_causal:
            tri = torch.ones(n, s, dtype=torch.bool, device=x.device).tril()
            scores = scores.masked_fill(~tri, float("-inf"))
        p = scores.softmax(dim=-1)
        o = torch.matmul(p, v).transpose(1, 2).reshape(b, n, self.heads * self.dim_head)
        return self.to_out[1](self.to_out[0](o))


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)  # erf GELU


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4, dropout=0.0):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(dropout), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    """pre-LN residual self-attn -> [pre-LN residual cross-attn] -> pre-LN residual GEGLU FF.

    0.15.x forward signature: (hidden_states, encoder_hidden_states=None, timestep=None,
    attention_mask=None, cross_attention_kwargs=None, class_labels=None).  The reference's
    TextEncoder passes its additive mask as positional arg #2 (tts/models.py:117-118), i.e. as
    ``encoder_hidden_states``; with no attn2 that argument is unused, hence text self-attention
    is UNMASKED under the pinned dependency.  ``attention_mask`` here reproduces the >=0.17
    binding when a caller opts in (mask_mode="additive" in oracle/model.py).
    """

    def __init__(self, dim, num_attention_heads, attention_head_dim, dropout=0.0, cross_attention_dim=None, **_unused):
        super().__init__()
        self.attn1 = Attention(dim, None, num_attention_heads, attention_head_dim, dropout)
        self.ff = FeedForward(dim, dropout=dropout)
        if cross_attention_dim is not None:
            self.attn2 = Attention(dim, cross_attention_dim, num_attention_heads, attention_head_dim, dropout)
            self.norm2 = nn.LayerNorm(dim)
        else:
            self.attn2 = None
            self.norm2 = None
        self.norm1 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)

    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, attention_mask=None,
                cross_attention_kwargs=None, class_labels=None, is_causal=False):
        h = hidden_states
        h = h + self.attn1(self.norm1(h), None, attention_mask, is_causal)
        if self.attn2 is not None:
            h = h + self.attn2(self.norm2(h), encoder_hidden_states, attention_mask)
        h = h + self.ff(self.norm3(h))
        return h


def timestep_embedding(timesteps, dim, flip_sin_to_cos=True, downscale_freq_shift=0.0, max_period=10000):
    half = dim // 2
    expo = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
    expo = expo / (half - downscale_freq_shift)
    ang = timesteps[:, None].float() * torch.exp(expo)[None, :]
    emb = torch.cat([ang.sin(), ang.cos()], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1))
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels, self.flip, self.shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, timesteps):
        return timestep_embedding(timesteps, self.num_channels, self.flip, self.shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", **_unused):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


# ---- DDPM (train.py:32-36, 96-98) -------------------------------------------------------------

def ddpm_alphas_cumprod(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02):
    betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    return torch.cumprod(1.0 - betas, dim=0)


def add_noise(x0, noise, timesteps, alphas_cumprod=None):
    ac = ddpm_alphas_cumprod() if alphas_cumprod is None else alphas_cumprod
    ac = ac.to(device=x0.device, dtype=x0.dtype)
    a = ac[timesteps] ** 0.5
    s = (1 - ac[timesteps]) ** 0.5
    shape = (-1,) + (1,) * (x0.dim() - 1)
    return a.view(shape) * x0 + s.view(shape) * noise


# ---- LR schedules (diffusers.optimization.get_scheduler; train.py:60-65) ------------------------

def lr_lambda(name, num_warmup_steps, num_training_steps, num_cycles=0.5, power=1.0):
    """Multiplicative LR factor as a function of the optimizer step (LambdaLR semantics)."""
    w, n = num_warmup_steps, num_training_steps
    if name == "constant":
        return lambda step: 1.0
    if name == "constant_with_warmup":
        return lambda step: float(step) / float(max(1.0, w)) if step < w else 1.0
    if name == "linear":
        return lambda step: (float(step) / float(max(1, w)) if step < w
                             else max(0.0, float(n - step) / float(max(1, n - w))))
    if name == "cosine":
        def f(step):
            if step < w:
                return float(step) / float(max(1, w))
            prog = float(step - w) / float(max(1, n - w))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * prog)))
        return f
    raise ValueError(f"unknown lr scheduler {name}")
