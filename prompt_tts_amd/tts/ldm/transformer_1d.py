"""Transformer1DModel on the HIP kernels (reference: tts/ldm/transformer_1d.py:64-190,199-310).

GN(eps 1e-6) -> 1x1 conv proj_in -> BasicTransformerBlock(self-attn, cross-attn over the text, GEGLU FF)
-> + residual.  `proj_out` is constructed and NEVER applied, exactly as in the reference (:275-279): its
parameters exist in the state_dict, get no gradient and are frozen in the optimizer.
The reference's two (B,C,N)<->(B,N,C) permutes vanish: everything is already token-major.
"""
from torch import nn

from ... import engine as E
from ... import ops
from .attention import BasicTransformerBlock


class Transformer1DModel(nn.Module):
    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, num_layers=1,
                 cross_attention_dim=None, norm_num_groups=32, use_linear_projection=False, **_ignored):
        super().__init__()
        if use_linear_projection:
            raise NotImplementedError("use_linear_projection=True is broken in the reference (no permute); conv path only")
        if num_layers != 1:
            raise NotImplementedError("the reference's blocks always build num_layers=1")
        inner = num_attention_heads * attention_head_dim
        self.in_channels, self.inner, self.groups = in_channels, inner, norm_num_groups
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6)
        self.proj_in = nn.Conv1d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim=cross_attention_dim)])
        self.proj_out = nn.Conv1d(inner, in_channels, 1)

    def fwd(self, st, x, ctx, B, N, S):
        hn, s = E.groupnorm_fwd(x, None, st.f(self.norm.weight), st.f(self.norm.bias), B, N, self.groups, 1e-6, False, arena=st.arena_active)
        h0 = E.linear_fwd(hn, st.w(self.proj_in.weight), st.f(self.proj_in.bias))
        out, sv = self.transformer_blocks[0].fwd(st, h0, ctx, B, N, S, final_residual=x)
        return out, (x, hn, s, sv, B, N)

    def bwd(self, st, saved, dout, dctx_accum):
        x, hn, s, sv, B, N = saved
        dh0, dctx = self.transformer_blocks[0].bwd(st, sv, dout, dctx_accum)
        gw = st.g(self.proj_in.weight).view(self.inner, self.in_channels)
        dhn = E.linear_bwd(dh0, hn, st.w(self.proj_in.weight), gw, st.g(self.proj_in.bias))
        dx, _ = E.groupnorm_bwd(dhn, x, None, s, st.f(self.norm.weight), st.f(self.norm.bias), st.g(self.norm.weight),
                                st.g(self.norm.bias), B, N, self.groups, False, dres=dout, arena=st.arena_active)
        return dx, dctx
