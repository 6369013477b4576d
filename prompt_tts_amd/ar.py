"""Autoregressive codec-token decoder: the ops BASELINE.json's north_star names that the reference does NOT contain
(SURVEY 0, 8a': causal self-attention, cross-attention over the prompt, RVQ-codebook logits projection, greedy / top-k sampling,
AR decode loop with a KV cache).  Build-defined semantics, pinned to plain PyTorch (oracle/ar.py) -- "parity unpinned" by the
reference by construction.  Everything runs on the same hand-written kernels as the denoiser (pt_gemm, pt_attn_fwd with the
causal / kv_len masks, pt_layernorm_fwd, pt_rvq_decode as the code-embedding gather-sum, pt_sample_topk).

Model: frame t's input is the sum over the n_q codebooks of an embedding of frame t-1's codes (a learned BOS row for t = 0)
plus a sinusoidal position; L diffusers-style BasicTransformerBlocks (self-attention CAUSAL, cross-attention over the text
encoder output, GEGLU feed-forward -- the same block class the reference instantiates); LayerNorm; logits head
Linear(d, n_q * bins): the logits of all n_q codebooks of frame t in parallel.

  forward(codes, ctx)            teacher-forced logits (B, T, n_q, bins) over the whole sequence (causal attention kernel)
  generate(ctx, T, k, uniforms)  the AR loop: one frame per step against per-layer K/V caches (keys/values of earlier frames
                                 are never recomputed; cross-attention K/V are projected once), greedy (k = 1) or top-k
                                 sampling with INJECTED uniforms -> codes (B, n_q, T) int64, ready for EncodecDecoder.decode
"""
import math

import torch
from torch import nn

from . import engine as E
from . import ops
from .tts.ldm.attention import BasicTransformerBlock


def sinusoid(T, d):
    pos = torch.arange(T, dtype=torch.float32)[:, None]
    inv = torch.exp(-math.log(10000.0) * torch.arange(0, d, 2, dtype=torch.float32) / d)[None, :]
    tab = torch.zeros(T, d)
    tab[:, 0::2] = torch.sin(pos * inv); tab[:, 1::2] = torch.cos(pos * inv)
    return tab


class LogitsHead(nn.Module):
    """RVQ-codebook logits projection: Linear(d -> n_q * bins) on (rows, d) hidden states (north_star; oracle F.linear)."""

    def __init__(self, d_model, n_q=8, bins=1024):
        super().__init__()
        self.n_q, self.bins = n_q, bins
        self.proj = nn.Linear(d_model, n_q * bins)

    def fwd(self, st, h, out_f32=False):
        """h (rows, d) in the store's dtype -> logits (rows, n_q * bins)."""
        return E.linear_fwd(h, st.w(self.proj.weight), st.f(self.proj.bias), out_f32=out_f32)


AR_GRAPH = __import__("os").environ.get("PT_AR_GRAPH", "1") != "0"     # capture the decode step of generate() as a HIP graph


class ARCodecDecoder(nn.Module):
    def __init__(self, d_model=512, n_layers=4, n_q=8, bins=1024, heads=8, cross_attention_dim=None, max_frames=1024,
                 dtype=torch.bfloat16):
        super().__init__()
        if d_model % heads or (d_model // heads) not in (32, 64, 128):
            raise ValueError("d_model / heads must be 32, 64 or 128 (MI355X attention kernels)")
        if d_model % 8:
            raise ValueError("d_model must be a multiple of 8")
        self.d, self.n_q, self.bins, self.heads, self.max_frames = d_model, n_q, bins, heads, max_frames
        self.compute_dtype = dtype
        self.code_embedding = nn.Parameter(torch.randn(n_q, bins, d_model) * 0.02)     # [q][code][d]: pt_rvq_decode's codebook layout
        self.bos = nn.Parameter(torch.randn(d_model) * 0.02)
        self.blocks = nn.ModuleList([BasicTransformerBlock(d_model, heads, d_model // heads, cross_attention_dim=cross_attention_dim or d_model)
                                     for _ in range(n_layers)])
        self.norm_out = nn.LayerNorm(d_model)
        self.head = LogitsHead(d_model, n_q, bins)
        self._store = None
        self._pos = None

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._store = None
        return out

    @property
    def store(self):
        p0 = next(self.parameters())
        if not p0.is_cuda:
            raise RuntimeError("ARCodecDecoder computes on an MI355X only: move it with .to('cuda') first (there is no CPU fallback)")
        if self._store is None or self._store.device != p0.device:
            self._store = E.ParamStore(self, p0.device, self.compute_dtype)
            self._pos = sinusoid(self.max_frames, self.d).to(p0.device)
        return self._store

    # ---- inputs: frame t sees the embedding of frame t-1 (BOS at t = 0) + position t -----------------------------------------
    def _embed(self, st, codes, t0=0):
        """codes (B, n_q, n) int64 = the PREVIOUS frames of positions t0 .. t0+n-1 (ignored where the position is 0) -> (B*n, d)."""
        B, n_q, n = codes.shape
        x = torch.empty(B * n, self.d, dtype=st.dtype, device=codes.device)
        ops.rvq_decode(codes.contiguous(), st.w(self.code_embedding), x, B, n_q, n, self.bins, self.d)
        x = x.view(B, n, self.d)
        if t0 == 0:
            x[:, 0] = st.w(self.bos)
        x = x + self._pos[t0:t0 + n].to(st.dtype)[None]
        return x.reshape(B * n, self.d).contiguous()

    # ---- teacher-forced pass ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, codes, ctx):
        """codes (B, n_q, T) int64; ctx (B, S, d_ctx) text-encoder output -> logits (B, T, n_q, bins) f32; position t is
        conditioned on frames < t only."""
        st = self.store
        st.ensure_shadow_fresh()
        B, n_q, T = codes.shape
        if n_q != self.n_q or T > self.max_frames:
            raise ValueError(f"codes must be (B, {self.n_q}, T <= {self.max_frames})")
        S = ctx.shape[1]
        ctx2 = ctx.to(st.dtype).reshape(B * S, -1).contiguous()
        prev = torch.roll(codes, 1, dims=2)                                  # prev[..., t] = codes[..., t-1]; t = 0 -> BOS
        h = self._embed(st, prev)
        for blk in self.blocks:
            h, _ = blk.fwd(st, h, ctx2, B, T, S, causal=True)
        n, _ = E.layernorm_fwd(h, st.f(self.norm_out.weight), st.f(self.norm_out.bias))
        return self.head.fwd(st, n, out_f32=True).view(B, T, self.n_q, self.bins)

    # ---- one decode step of a block against its K/V cache ---------------------------------------------------------------------
    def _block_step(self, st, blk, h, cache, t, kv_len, ctx2, B, S):
        C = self.d
        a1 = blk.attn1
        n1, _ = E.layernorm_fwd(h, st.f(blk.norm1.weight), st.f(blk.norm1.bias))
        fw = st.fused([a1.to_q.weight, a1.to_k.weight, a1.to_v.weight])
        if fw is not None:
            qkv = E.linear_fwd(n1, fw[0]); q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        else:
            q, k, v = (E.linear_fwd(n1, st.w(p.weight)) for p in (a1.to_q, a1.to_k, a1.to_v))
        kc, vc = cache
        if torch.is_tensor(t):                                               # device-resident frame index (captured decode step)
            kc.index_copy_(1, t, k.unsqueeze(1)); vc.index_copy_(1, t, v.unsqueeze(1))
        else:
            kc[:, t] = k; vc[:, t] = v                                       # append this frame's key / value
        Tm = kc.shape[1]
        o = torch.empty(B, C, dtype=h.dtype, device=h.device)
        lse = torch.empty(B, self.heads, 1, dtype=torch.float32, device=h.device)
        ops.attn_fwd(q, kc.view(B * Tm, C), vc.view(B * Tm, C), o, lse, B, self.heads, 1, Tm, C // self.heads, a1.scale, False, kv_len)
        h1 = E.linear_fwd(o, st.w(a1.to_out[0].weight), st.f(a1.to_out[0].bias), residual=h)
        n2, _ = E.layernorm_fwd(h1, st.f(blk.norm2.weight), st.f(blk.norm2.bias))
        h2, _ = blk.attn2.fwd(st, n2, ctx2, h1, B, 1, S)                     # K/V of the prompt: projected once (cross_kv_cache)
        n3, _ = E.layernorm_fwd(h2, st.f(blk.norm3.weight), st.f(blk.norm3.bias))
        out, _ = blk.ff.fwd(st, n3, h2)
        return out

    @torch.no_grad()
    def generate(self, ctx, T, k=1, uniforms=None, temperature=1.0, graph=None):
        """ctx (B, S, d_ctx) -> codes (B, n_q, T) int64.  k = 1: greedy argmax; k > 1: top-k sampling, one INJECTED uniform per
        (frame, prompt, codebook): uniforms (T, B * n_q) f32 in [0, 1).
        graph (default: PT_AR_GRAPH, on): frames 0 and 1 run launch by launch, then ONE decode step -- ~25 launches per layer of a
        few microseconds each, a launch-bound loop -- is captured as a HIP graph whose frame index, K/V length, previous codes and
        uniforms row live on the device, and replayed for every remaining frame; same kernels in the same order, same codes."""
        st = self.store
        st.ensure_shadow_fresh()
        B, S = ctx.shape[0], ctx.shape[1]
        if T > self.max_frames:
            raise ValueError(f"T <= {self.max_frames}")
        if k > 1 and (uniforms is None or tuple(uniforms.shape) != (T, B * self.n_q)):
            raise ValueError("top-k sampling needs injected uniforms of shape (T, B * n_q)")
        dev = ctx.device
        ctx2 = ctx.to(st.dtype).reshape(B * S, -1).contiguous()
        caches = [(torch.zeros(B, T, self.d, dtype=st.dtype, device=dev), torch.zeros(B, T, self.d, dtype=st.dtype, device=dev))
                  for _ in self.blocks]
        codes = torch.zeros(B, self.n_q, T, dtype=torch.int64, device=dev)
        prev = torch.zeros(B, self.n_q, 1, dtype=torch.int64, device=dev)
        if graph is None:
            graph = AR_GRAPH
        n_eager = T if not graph or T < 4 else 2
        with E.cross_kv_cache():
            for t in range(n_eager):
                h = self._embed(st, prev, t0=t)
                kv_len = torch.full((B,), t + 1, dtype=torch.int32, device=dev)
                for blk, cache in zip(self.blocks, caches):
                    h = self._block_step(st, blk, h, cache, t, kv_len, ctx2, B, S)
                n, _ = E.layernorm_fwd(h, st.f(self.norm_out.weight), st.f(self.norm_out.bias))
                logits = self.head.fwd(st, n).view(B * self.n_q, self.bins)
                idx = ops.sample_topk(logits, k=k, uniforms=uniforms[t].contiguous() if k > 1 else None, temperature=temperature)
                prev = idx.view(B, self.n_q, 1)
                codes[:, :, t] = prev[:, :, 0]
            if n_eager < T:
                # ---- the decode step with every per-frame quantity on the device ----
                t_dev = torch.full((1,), n_eager, dtype=torch.int64, device=dev)
                kv_len = torch.full((B,), n_eager + 1, dtype=torch.int32, device=dev)
                prev_buf = prev.contiguous().clone()
                pos = self._pos.to(st.dtype)
                emb = st.w(self.code_embedding)

                def step():
                    x = torch.empty(B, self.d, dtype=st.dtype, device=dev)
                    ops.rvq_decode(prev_buf, emb, x, B, self.n_q, 1, self.bins, self.d)
                    h = (x + pos.index_select(0, t_dev)).contiguous()
                    for blk, cache in zip(self.blocks, caches):
                        h = self._block_step(st, blk, h, cache, t_dev, kv_len, ctx2, B, S)
                    n, _ = E.layernorm_fwd(h, st.f(self.norm_out.weight), st.f(self.norm_out.bias))
                    logits = self.head.fwd(st, n).view(B * self.n_q, self.bins)
                    u = uniforms.index_select(0, t_dev).view(-1) if k > 1 else None
                    idx = ops.sample_topk(logits, k=k, uniforms=u, temperature=temperature)
                    prev_buf.copy_(idx.view(B, self.n_q, 1))
                    codes.index_copy_(2, t_dev, prev_buf)
                    t_dev.add_(1); kv_len.add_(1)

                if k > 1:
                    uniforms = uniforms.contiguous()
                g = None
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        step()
                except RuntimeError as e:                    # e.g. another capture in progress on this device: same step, launch by launch
                    import warnings
                    warnings.warn(f"ARCodecDecoder.generate: HIP graph capture failed ({e}); running the decode step launch by launch")
                    g = None
                for _ in range(n_eager, T):
                    if g is not None:
                        g.replay()
                    else:
                        step()
        return codes
