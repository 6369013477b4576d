"""BasicTransformerBlock (diffusers ^0.15.1 semantics) on the HIP kernels.

Replaces the third-party block the reference instantiates at tts/models.py:95-100 and
tts/ldm/transformer_1d.py:165-178.  Parameter names follow diffusers so state_dict keys match:
norm1, attn1.{to_q,to_k,to_v,to_out.0}, [norm2, attn2.*], norm3, ff.net.0.proj, ff.net.2.
"""
import torch
from torch import nn

from ... import engine as E
from ... import _lib as L
from ... import ops


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = heads * dim_head
        kv = query_dim if cross_attention_dim is None else cross_attention_dim
        self.heads, self.dim_head, self.is_cross = heads, dim_head, cross_attention_dim is not None
        self.scale = dim_head ** -0.5
        self.p_drop = float(dropout)
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(kv, inner, bias=False)
        self.to_v = nn.Linear(kv, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

    # x: (B*Nq, C) normalised input; ctx_in: (B*Nk, d) or None (self); h: residual stream (B*Nq, C)
    def fwd(self, st, x, ctx_in, h, B, Nq, Nk, causal=False, kv_len=None):
        C = self.heads * self.dim_head
        if self.dim_head not in (32, 64, 128):
            raise ValueError(f"head dim {self.dim_head} not supported by the MI355X attention kernels (32, 64, 128)")
        if not self.is_cross:
            fw = st.fused([self.to_q.weight, self.to_k.weight, self.to_v.weight])
            if fw is not None:
                qkv = E.linear_fwd(x, fw[0])
                q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
            else:
                q, k, v = (E.linear_fwd(x, st.w(p.weight)) for p in (self.to_q, self.to_k, self.to_v))
                qkv = None
            kvbuf = None
        else:
            q = E.linear_fwd(x, st.w(self.to_q.weight))

            def project_kv():
                fw = st.fused([self.to_k.weight, self.to_v.weight])
                if fw is not None:
                    buf = E.linear_fwd(ctx_in, fw[0])
                    return buf[:, :C], buf[:, C:], buf
                return E.linear_fwd(ctx_in, st.w(self.to_k.weight)), E.linear_fwd(ctx_in, st.w(self.to_v.weight)), None
            pre = E.batched_kv[0].get(id(self)) if E.batched_kv[0] is not None else None
            if pre is not None:                  # projected for every layer at once by the UNet (packed K/V weights)
                k, v = pre
                kvbuf = "batched"
            else:
                k, v, kvbuf = E.cached_cross_kv(self, ctx_in, project_kv)
            qkv = None
        o = torch.empty(B * Nq, C, dtype=x.dtype, device=x.device)
        lse = torch.empty(B, self.heads, Nq, dtype=torch.float32, device=x.device)
        ops.attn_fwd(q, k, v, o, lse, B, self.heads, Nq, Nk, self.dim_head, self.scale, causal, kv_len)
        keep = None
        if self.p_drop > 0.0 and self.training:
            # to_out = [Linear, Dropout]: out = h + dropout(o W^T + b): the residual moves from the GEMM epilogue to the dropout
            y = E.linear_fwd(o, st.w(self.to_out[0].weight), st.f(self.to_out[0].bias))
            keep = E.dropout_keep_mask(y.shape, self.p_drop, y.device)
            out = torch.empty_like(y)
            ops.dropout(y, keep, out, 1.0 / (1.0 - self.p_drop), residual=h)
        else:
            out = E.linear_fwd(o, st.w(self.to_out[0].weight), st.f(self.to_out[0].bias), residual=h)
        kv_mode = 2 if isinstance(kvbuf, str) else (1 if kvbuf is not None else 0)
        return out, (x, ctx_in, q, k, v, o, lse, B, Nq, Nk, causal, kv_len, qkv is not None, kv_mode, keep)

    # dout: grad of (h + attn(x)); returns (dx wrt normalised input, dctx or None); dctx_accum accumulates in place
    def bwd(self, st, saved, dout, dctx_accum=None):
        x, ctx_in, q, k, v, o, lse, B, Nq, Nk, causal, kv_len, fused_qkv, kv_mode, keep = saved
        fused_kv = kv_mode == 1
        C = self.heads * self.dim_head
        wo = self.to_out[0]
        if keep is not None:                          # through the dropout: d(o W^T + b) = dout * keep / (1 - p)
            dy = torch.empty_like(dout)
            ops.dropout(dout, keep, dy, 1.0 / (1.0 - self.p_drop))
            dout = dy
        do = E.linear_bwd(dout, o, st.w(wo.weight), st.g(wo.weight), st.g(wo.bias))
        delta = torch.empty_like(lse)
        if fused_qkv:
            dqkv = torch.empty(B * Nq, 3 * C, dtype=x.dtype, device=x.device)
            dq, dk, dv = dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:]
        else:
            dq = torch.empty(B * Nq, C, dtype=x.dtype, device=x.device)
            if kv_mode == 2:                       # slices of the UNet's d(kv) buffer; their dgrad / wgrad run once, batched
                dk, dv = E.batched_dkv[0][id(self)]
            elif fused_kv:
                dkv = torch.empty(B * Nk, 2 * C, dtype=x.dtype, device=x.device)
                dk, dv = dkv[:, :C], dkv[:, C:]
            else:
                dk = torch.empty(B * Nk, C, dtype=x.dtype, device=x.device); dv = torch.empty_like(dk)
        ops.attn_bwd(q, k, v, o, lse, do, delta, dq, dk, dv, B, self.heads, Nq, Nk, self.dim_head, self.scale,
                     causal, kv_len)
        if fused_qkv:
            w, gw = st.fused([self.to_q.weight, self.to_k.weight, self.to_v.weight])
            return E.linear_bwd(dqkv, x, w, gw), None
        if not self.is_cross:
            dx = E.linear_bwd(dq, x, st.w(self.to_q.weight), st.g(self.to_q.weight))
            dx = E.linear_bwd(dk, x, st.w(self.to_k.weight), st.g(self.to_k.weight), dx_accum=dx)
            dx = E.linear_bwd(dv, x, st.w(self.to_v.weight), st.g(self.to_v.weight), dx_accum=dx)
            return dx, None
        dx = E.linear_bwd(dq, x, st.w(self.to_q.weight), st.g(self.to_q.weight))
        if kv_mode == 2:
            return dx, dctx_accum
        if fused_kv:
            w, gw = st.fused([self.to_k.weight, self.to_v.weight])
            dctx = E.linear_bwd(dkv, ctx_in, w, gw, dx_accum=dctx_accum)
        else:
            dctx = E.linear_bwd(dk, ctx_in, st.w(self.to_k.weight), st.g(self.to_k.weight), dx_accum=dctx_accum)
            dctx = E.linear_bwd(dv, ctx_in, st.w(self.to_v.weight), st.g(self.to_v.weight), dx_accum=dctx)
        return dx, dctx


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4, dropout=0.0):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(dropout), nn.Linear(dim * mult, dim)])
        self.p_drop = float(dropout)

    def _fused(self, st, M):
        """GEGLU in the GEMM epilogues (bf16, whole 256-row tiles): the projection weight's shadow rows are interleaved.
        With an active dropout between GEGLU and the second Linear the stand-alone kernels run instead."""
        return id(self.net[0].proj.weight) in st.geglu_ids and M % 256 == 0 and not (self.p_drop > 0.0 and self.training)

    def fwd(self, st, x, h, residual2=None):
        p1, p2 = self.net[0].proj, self.net[2]
        M, F2 = x.shape[0], p1.weight.shape[0]
        act = torch.empty(M, F2 // 2, dtype=x.dtype, device=x.device)
        if self._fused(st, M):
            # one launch: proj (interleaved columns, kept for the backward) and act = value * gelu(gate) from the same tile
            proj = torch.empty(M, F2, dtype=x.dtype, device=x.device)
            if st.fp8 and E.fp8_pays(M, F2, x.shape[1]):
                w8, _, sw = st.w8(p1.weight)
                x8, sx = E.fp8_quantize_act(x, L.PT_FP8_E4M3)
                ops.gemm_fp8(M, F2, x.shape[1], x8, w8, proj, sx, sw, bias=st.f(p1.bias), act=2, out2=act, ldc2=F2 // 2)
            else:
                ops.gemm(M, F2, x.shape[1], ops.plain(x), ops.plain(st.w(p1.weight)), proj, ops.pt_dtype(x), bias=st.f(p1.bias),
                         act=2, out2=act, ldc2=F2 // 2)
        elif id(p1.weight) in st.geglu_ids:
            # interleaved weight, but a row count the fused epilogue does not take: stand-alone GEGLU over interleaved columns
            proj = E.linear_fwd(x, st.w(p1.weight))
            ops.geglu_fwd(proj, act, bias=st.f(p1.bias), interleaved=True)
        else:
            proj = E.linear_fwd(x, st.w(p1.weight), st.f(p1.bias))
            ops.geglu_fwd(proj, act)
        keep = None
        if self.p_drop > 0.0 and self.training:       # net = [GEGLU, Dropout, Linear]
            keep = E.dropout_keep_mask(act.shape, self.p_drop, act.device)
            ops.dropout(act, keep, act, 1.0 / (1.0 - self.p_drop))
        out = E.linear_fwd(act, st.w(p2.weight), st.f(p2.bias), residual=h, residual2=residual2)
        return out, (x, proj, act, keep)

    def bwd(self, st, saved, dout):
        x, proj, act, keep = saved
        p1, p2 = self.net[0].proj, self.net[2]
        M, F2 = proj.shape
        if keep is None and self._fused(st, M):
            # ff2: weight gradient as usual; its dgrad GEMM turns d(act) into d(proj) in the epilogue (d(act) never reaches HBM)
            E.linear_bwd(dout, act, st.w(p2.weight), st.g(p2.weight), st.g(p2.bias), need_dx=False)
            dproj = torch.empty_like(proj)
            if st.fp8 and E.fp8_pays(M, F2 // 2, dout.shape[1]):
                _, w8t, sw = st.w8(p2.weight)                      # W2^T [F][d]: the reduction (d) contiguous
                d8, sd = E.fp8_quantize_act(dout, L.PT_FP8_E5M2)
                ops.gemm_fp8(M, F2 // 2, dout.shape[1], d8, w8t, dproj, sd, sw, a_format=L.PT_FP8_E5M2, ldc=F2, act=3,
                             residual=proj, ldr=F2)
            else:
                w2t = st.wt(st.w(p2.weight))                       # W2^T [F][d] (plain operand) when the store keeps one
                ops.gemm(M, F2 // 2, dout.shape[1], ops.plain(dout), ops.plain(w2t) if w2t is not None else ops.plain(st.w(p2.weight), trans=True),
                         dproj, ops.pt_dtype(x), ldc=F2, act=3, residual=proj, ldr=F2)
            return E.linear_bwd(dproj, x, st.w(p1.weight), st.g(p1.weight), st.g(p1.bias), geglu_rows=F2 // 2)
        il = id(p1.weight) in st.geglu_ids
        dact = E.linear_bwd(dout, act, st.w(p2.weight), st.g(p2.weight), st.g(p2.bias))
        if keep is not None:
            ops.dropout(dact, keep, dact, 1.0 / (1.0 - self.p_drop))
        dproj = torch.empty_like(proj)
        ops.geglu_bwd(dact, proj, dproj, interleaved=il)
        return E.linear_bwd(dproj, x, st.w(p1.weight), st.g(p1.weight), st.g(p1.bias), geglu_rows=F2 // 2 if il else 0)


class BasicTransformerBlock(nn.Module):
    """h += attn1(LN1 h); [h += attn2(LN2 h, ctx)]; h += ff(LN3 h).  dropout > 0 (the reference forwards
    text_encoder_dropout, tts/models.py:95-100) acts in training mode after each attention's output projection and between
    GEGLU and the second feed-forward Linear, as in diffusers' Attention.to_out[1] / FeedForward.net[1]."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, dropout=0.0, cross_attention_dim=None):
        super().__init__()
        if not 0.0 <= dropout < 1.0:
            raise ValueError("dropout must be in [0, 1)")
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

    def fwd(self, st, h, ctx, B, N, S, final_residual=None, self_kv_len=None, causal=False):
        sv = {}
        n1, sv["ln1"] = E.layernorm_fwd(h, st.f(self.norm1.weight), st.f(self.norm1.bias))
        h1, sv["a1"] = self.attn1.fwd(st, n1, None, h, B, N, N, causal=causal, kv_len=self_kv_len)
        sv["h0"] = h
        if self.attn2 is not None:
            n2, sv["ln2"] = E.layernorm_fwd(h1, st.f(self.norm2.weight), st.f(self.norm2.bias))
            h2, sv["a2"] = self.attn2.fwd(st, n2, ctx, h1, B, N, S)
            sv["h1"] = h1
        else:
            h2 = h1
        n3, sv["ln3"] = E.layernorm_fwd(h2, st.f(self.norm3.weight), st.f(self.norm3.bias))
        out, sv["ff"] = self.ff.fwd(st, n3, h2, residual2=final_residual)
        sv["h2"] = h2
        return out, sv

    def bwd(self, st, sv, dout, dctx_accum=None):
        """Returns (dh, dctx).  dout is also the gradient of any `final_residual` (identity), handled by the caller."""
        dn3 = self.ff.bwd(st, sv["ff"], dout)
        dh2 = E.layernorm_bwd(dn3, sv["h2"], sv["ln3"], st.f(self.norm3.weight), st.g(self.norm3.weight),
                              st.g(self.norm3.bias), dres=dout)
        dctx = dctx_accum
        if self.attn2 is not None:
            dn2, dctx = self.attn2.bwd(st, sv["a2"], dh2, dctx_accum)
            dh1 = E.layernorm_bwd(dn2, sv["h1"], sv["ln2"], st.f(self.norm2.weight), st.g(self.norm2.weight),
                                  st.g(self.norm2.bias), dres=dh2)
        else:
            dh1 = dh2
        dn1, _ = self.attn1.bwd(st, sv["a1"], dh1)
        dh = E.layernorm_bwd(dn1, sv["h0"], sv["ln1"], st.f(self.norm1.weight), st.g(self.norm1.weight),
                             st.g(self.norm1.bias), dres=dh1)
        return dh, dctx
