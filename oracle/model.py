"""Oracle restatement of the reference-owned model code (fp32, CPU, plain PyTorch).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows, as text:
  tts/models.py:11-70     positional encoding (with its permute quirk)
  tts/models.py:73-120    TextEncoder
  tts/models.py:123-172   TTSSingleSpeaker
  tts/ldm/resnet.py:11-49,52-96,99-283          Upsample1D / Downsample1D / ResnetBlock1D
  tts/ldm/transformer_1d.py:64-190,199-310      Transformer1DModel (proj_out built, never applied)
  tts/ldm/unet_blocks.py:131-620                the five block types
  tts/ldm/unet_1d_condition.py:111-412,553-739  Unet1DConditionModel
Attribute names match the reference so ``state_dict`` keys are identical.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

from .blocks import BasicTransformerBlock, Timesteps, TimestepEmbedding


def positional_table(seq_len_cfg, S, d, inv_freq=None):
    """(S, d) table the reference adds to the embeddings.

    tts/models.py:55-70 permutes (B,S,d)->(B,d,S) before PositionalEncoding1D(seq_len_cfg), so
    "position" runs over the FEATURE index k and "channel" over the TIME index s:
        pos[s, k] = sin(k * w_{s//2}) if s even else cos(k * w_{s//2}),  w_j = 10000^(-2j/ch)
    with ch = 2*ceil(seq_len_cfg/2); needs S <= ch.
    """
    ch = int(math.ceil(seq_len_cfg / 2) * 2)
    if S > ch:
        raise RuntimeError(f"text length {S} exceeds positional channels {ch}")
    if inv_freq is None:
        inv_freq = 1.0 / (10000 ** (torch.arange(0, ch, 2).float() / ch))
    ang = torch.arange(d, dtype=inv_freq.dtype, device=inv_freq.device)[:, None] * inv_freq[None, :]      # (d, ch/2)
    tab = torch.stack((ang.sin(), ang.cos()), dim=-1).flatten(-2, -1)             # (d, ch)
    return tab[:, :S].transpose(0, 1).contiguous()                                # (S, d)


class _Penc(nn.Module):
    def __init__(self, channels):
        super().__init__()
        ch = int(math.ceil(channels / 2) * 2)
        self.org_channels = channels
        self.register_buffer("inv_freq", 1.0 / (10000 ** (torch.arange(0, ch, 2).float() / ch)))


class _PencPermute(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.penc = _Penc(channels)

    def forward(self, emb):  # emb (B,S,d)
        _, S, d = emb.shape
        return positional_table(self.penc.org_channels, S, d, self.penc.inv_freq).to(emb.dtype)[None]


class TextEncoder(nn.Module):
    def __init__(self, vocab_len, seq_len, dim, attention_head_dim, dropout=0.0, num_layers=1, mask_mode="ignored"):
        super().__init__()
        if dim % attention_head_dim != 0:
            raise ValueError("dim must be a multiple of attention_head_dim")
        self.mask_mode = mask_mode
        self.word_embedding = nn.Embedding(vocab_len, dim)
        self.pos_embedding = _PencPermute(seq_len)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(dim, dim // attention_head_dim, attention_head_dim, dropout=dropout)
             for _ in range(num_layers)])

    def forward(self, input_ids, attention_mask):
        add = None
        if attention_mask is not None and self.mask_mode == "additive":
            add = ((1 - attention_mask.to(torch.float32)) * -10000.0).unsqueeze(1)
        e = self.word_embedding(input_ids)
        h = e + self.pos_embedding(e)
        for blk in self.transformer_blocks:
            # pinned diffusers binding: the mask lands in `encoder_hidden_states` and is unused.
            h = blk(h, attention_mask=add)
        return h


class Upsample1D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv1d(channels, channels, 3, padding=1)

    def forward(self, x, output_size=None):
        if output_size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=output_size, mode="nearest")
        return self.conv(x)


class Downsample1D(nn.Module):
    def __init__(self, channels, padding=1):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv1d(channels, channels, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:
            x = F.pad(x, (0, 1))
        return self.conv(x)


class ResnetBlock1D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5, output_scale_factor=1.0):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv1d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = nn.Conv1d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv1d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.output_scale_factor = output_scale_factor

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return (x + h) / self.output_scale_factor


class Transformer1DModel(nn.Module):
    def __init__(self, num_heads, head_dim, in_channels, cross_attention_dim, groups=32):
        super().__init__()
        inner = num_heads * head_dim
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Conv1d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, num_heads, head_dim, cross_attention_dim=cross_attention_dim)])
        self.proj_out = nn.Conv1d(inner, in_channels, 1)  # constructed, NEVER applied (transformer_1d.py:275-279)

    def forward(self, x, encoder_hidden_states):
        h = self.proj_in(self.norm(x)).permute(0, 2, 1)
        for blk in self.transformer_blocks:
            h = blk(h, encoder_hidden_states=encoder_hidden_states)
        return h.permute(0, 2, 1) + x


def _attn(channels, heads, cross_dim, groups):
    return Transformer1DModel(heads, channels // heads, channels, cross_dim, groups)


class DownBlock1D(nn.Module):
    has_cross_attention = False

    def __init__(self, num_layers, in_channels, out_channels, temb_channels, add_downsample, eps, groups, **_):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock1D(in_channels if i == 0 else out_channels, out_channels,
                                                    temb_channels, groups, eps) for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample1D(out_channels)]) if add_downsample else None

    def forward(self, h, temb, encoder_hidden_states=None):
        outs = ()
        for r in self.resnets:
            h = r(h, temb)
            outs += (h,)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs += (h,)
        return h, outs


class CrossAttnDownBlock1D(nn.Module):
    has_cross_attention = True

    def __init__(self, num_layers, in_channels, out_channels, temb_channels, add_downsample, eps, groups,
                 cross_attention_dim, heads):
        super().__init__()
        self.attentions = nn.ModuleList([_attn(out_channels, heads, cross_attention_dim, groups)
                                         for _ in range(num_layers)])
        self.resnets = nn.ModuleList([ResnetBlock1D(in_channels if i == 0 else out_channels, out_channels,
                                                    temb_channels, groups, eps) for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample1D(out_channels)]) if add_downsample else None

    def forward(self, h, temb, encoder_hidden_states=None):
        outs = ()
        for r, a in zip(self.resnets, self.attentions):
            h = a(r(h, temb), encoder_hidden_states)
            outs += (h,)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs += (h,)
        return h, outs


class UpBlock1D(nn.Module):
    has_cross_attention = False

    def __init__(self, num_layers, in_channels, out_channels, prev_output_channel, temb_channels, add_upsample,
                 eps, groups, **_):
        super().__init__()
        rs = []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            cin = prev_output_channel if i == 0 else out_channels
            rs.append(ResnetBlock1D(cin + skip, out_channels, temb_channels, groups, eps))
        self.resnets = nn.ModuleList(rs)
        self.upsamplers = nn.ModuleList([Upsample1D(out_channels)]) if add_upsample else None

    def forward(self, h, skips, temb, encoder_hidden_states=None, upsample_size=None):
        for r in self.resnets:
            h = r(torch.cat([h, skips[-1]], dim=1), temb)
            skips = skips[:-1]
        if self.upsamplers is not None:
            h = self.upsamplers[0](h, upsample_size)   # only UpBlock1D honours upsample_size (unet_blocks.py:198-200)
        return h


class CrossAttnUpBlock1D(nn.Module):
    has_cross_attention = True

    def __init__(self, num_layers, in_channels, out_channels, prev_output_channel, temb_channels, add_upsample,
                 eps, groups, cross_attention_dim, heads):
        super().__init__()
        rs, at = [], []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            cin = prev_output_channel if i == 0 else out_channels
            rs.append(ResnetBlock1D(cin + skip, out_channels, temb_channels, groups, eps))
            at.append(_attn(out_channels, heads, cross_attention_dim, groups))
        self.attentions, self.resnets = nn.ModuleList(at), nn.ModuleList(rs)
        self.upsamplers = nn.ModuleList([Upsample1D(out_channels)]) if add_upsample else None

    def forward(self, h, skips, temb, encoder_hidden_states=None, upsample_size=None):
        for r, a in zip(self.resnets, self.attentions):
            h = a(r(torch.cat([h, skips[-1]], dim=1), temb), encoder_hidden_states)
            skips = skips[:-1]
        if self.upsamplers is not None:
            h = self.upsamplers[0](h)                  # upsample_size ignored here (unet_blocks.py:525-527)
        return h


class UNetMidBlock1DCrossAttn(nn.Module):
    def __init__(self, in_channels, temb_channels, eps, groups, cross_attention_dim, heads, output_scale_factor=1.0):
        super().__init__()
        self.attentions = nn.ModuleList([_attn(in_channels, heads, cross_attention_dim, groups)])
        self.resnets = nn.ModuleList([ResnetBlock1D(in_channels, in_channels, temb_channels, groups, eps,
                                                    output_scale_factor) for _ in range(2)])

    def forward(self, h, temb, encoder_hidden_states=None):
        h = self.resnets[0](h, temb)
        h = self.attentions[0](h, encoder_hidden_states)
        return self.resnets[1](h, temb)


_DOWN = {"CrossAttnDownBlock1D": CrossAttnDownBlock1D, "DownBlock1D": DownBlock1D}
_UP = {"CrossAttnUpBlock1D": CrossAttnUpBlock1D, "UpBlock1D": UpBlock1D}


class Unet1DConditionModel(nn.Module):
    def __init__(self, sample_size=None, in_channels=4, out_channels=4, layers_per_block=2,
                 block_out_channels=(320, 640), down_block_types=("CrossAttnDownBlock1D", "DownBlock1D"),
                 mid_block_type="UNetMidBlock1DCrossAttn", up_block_types=("UpBlock1D", "CrossAttnUpBlock1D"),
                 cross_attention_dim=1280, attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5,
                 flip_sin_to_cos=True, freq_shift=0):
        super().__init__()
        nb = len(down_block_types)
        if len(up_block_types) != nb:
            raise ValueError("Must provide the same number of `down_block_types` as `up_block_types`.")
        if len(block_out_channels) != nb:
            raise ValueError("Must provide the same number of `block_out_channels` as `down_block_types`.")
        boc = list(block_out_channels)
        lpb = [layers_per_block] * nb if isinstance(layers_per_block, int) else list(layers_per_block)
        heads = attention_head_dim  # the reference hands `attention_head_dim` on as the NUMBER of heads
        ted = boc[0] * 4
        self.conv_in = nn.Conv1d(in_channels, boc[0], 3, padding=1)
        self.time_proj = Timesteps(boc[0], flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        self.down_blocks = nn.ModuleList()
        self.up_blocks = nn.ModuleList()   # registered before mid_block, as in the reference (key order)
        oc = boc[0]
        for i, t in enumerate(down_block_types):
            ic, oc = oc, boc[i]
            if t not in _DOWN:
                raise ValueError(f"{t} does not exist.")
            self.down_blocks.append(_DOWN[t](num_layers=lpb[i], in_channels=ic, out_channels=oc, temb_channels=ted,
                                             add_downsample=i != nb - 1, eps=norm_eps, groups=norm_num_groups,
                                             cross_attention_dim=cross_attention_dim, heads=heads))
        if mid_block_type == "UNetMidBlock1DCrossAttn":
            self.mid_block = UNetMidBlock1DCrossAttn(boc[-1], ted, norm_eps, norm_num_groups, cross_attention_dim, heads)
        elif mid_block_type is None:
            self.mid_block = None
        else:
            raise ValueError(f"unknown mid_block_type : {mid_block_type}")
        self.num_upsamplers = 0
        rboc, rlpb = boc[::-1], lpb[::-1]
        oc = rboc[0]
        for i, t in enumerate(up_block_types):
            prev, oc = oc, rboc[i]
            ic = rboc[min(i + 1, nb - 1)]
            last = i == nb - 1
            self.num_upsamplers += 0 if last else 1
            if t not in _UP:
                raise ValueError(f"{t} does not exist.")
            self.up_blocks.append(_UP[t](num_layers=rlpb[i] + 1, in_channels=ic, out_channels=oc,
                                         prev_output_channel=prev, temb_channels=ted, add_upsample=not last,
                                         eps=norm_eps, groups=norm_num_groups,
                                         cross_attention_dim=cross_attention_dim, heads=heads))
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, boc[0], eps=norm_eps)
        self.conv_out = nn.Conv1d(boc[0], out_channels, 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states, attention_mask=None, return_dict=True, **_):
        # attention_mask is turned additive by the reference (:597-599) and then never reaches attention.
        factor = 2 ** self.num_upsamplers
        forward_size = any(s % factor != 0 for s in sample.shape[-2:])
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float64 if isinstance(t, float) else torch.int64)
        elif t.dim() == 0:
            t = t[None]
        t = t.expand(sample.shape[0])
        emb = self.time_embedding(self.time_proj(t).to(sample.dtype))
        h = self.conv_in(sample)
        skips = (h,)
        for blk in self.down_blocks:
            h, outs = blk(h, emb, encoder_hidden_states)
            skips += outs
        if self.mid_block is not None:
            h = self.mid_block(h, emb, encoder_hidden_states)
        for i, blk in enumerate(self.up_blocks):
            n = len(blk.resnets)
            res, skips = skips[-n:], skips[:-n]
            size = skips[-1].shape[2:] if (i != len(self.up_blocks) - 1 and forward_size) else None
            h = blk(h, res, emb, encoder_hidden_states, upsample_size=size)
        h = self.conv_out(F.silu(self.conv_norm_out(h)))
        return SimpleNamespace(sample=h) if return_dict else (h,)


class TTSSingleSpeaker(nn.Module):
    def __init__(self, config, mask_mode="ignored"):
        super().__init__()
        self.text_encoder = TextEncoder(config["cmu_vocab_len"], config["cmu_seq_len"], config["cross_attention_dim"],
                                        config["attention_head_dim"], config["text_encoder_dropout"],
                                        config["text_encoder_layers"], mask_mode=mask_mode)
        self.unet = Unet1DConditionModel(
            sample_size=config["sample_size"], in_channels=config["in_channels"], out_channels=config["out_channels"],
            layers_per_block=config["layers_per_block"], block_out_channels=config["block_out_channels"],
            down_block_types=config["down_block_types"], mid_block_type=config["mid_block_type"],
            up_block_types=config["up_block_types"], cross_attention_dim=config["cross_attention_dim"])

    def forward(self, sample, timestep, text_seq_ids, attention_mask, cross_attention_kwargs=None, return_dict=True):
        text_emb = self.text_encoder(text_seq_ids, attention_mask)
        return self.unet(sample, timestep, text_emb, attention_mask, return_dict=return_dict)


# ---- BASELINE configs (SURVEY.md 8d mapping) ----------------------------------------------------

def make_config(d, L, text_layers, n_q, T, S=256, grad_accum=1):
    return {
        "cmu_vocab_len": 149, "cmu_seq_len": S, "cross_attention_dim": d, "attention_head_dim": 64,
        "text_encoder_dropout": 0.0, "text_encoder_layers": text_layers, "sample_size": T,
        "in_channels": n_q, "out_channels": n_q, "layers_per_block": L, "block_out_channels": [d, d],
        "down_block_types": ["CrossAttnDownBlock1D", "DownBlock1D"], "mid_block_type": "UNetMidBlock1DCrossAttn",
        "up_block_types": ["UpBlock1D", "CrossAttnUpBlock1D"],
        "gradient_accumulation_steps": grad_accum, "num_train_epochs": 1, "lr_scheduler": "constant_with_warmup",
        "lr_warmup_steps": 0, "save_per_epochs": 1,
    }


CONFIG_A = dict(d=256, L=1, text_layers=1, n_q=2, T=1024)
CONFIG_B = dict(d=512, L=5, text_layers=2, n_q=8, T=1024)
CONFIG_E = dict(d=1024, L=11, text_layers=4, n_q=8, T=2048)
