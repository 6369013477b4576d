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
# the decode step on the folded launches of csrc/decode_step.hip (LN + Linear + epilogue in one launch, K / V appended by the
# projection's epilogue, embedding / bookkeeping kernels): ~37 launches per frame instead of ~70.  PT_AR_FUSED=0: the training kernels.
AR_FUSED = __import__("os").environ.get("PT_AR_FUSED", "1") != "0"


AR_OPLIST = __import__("os").environ.get("PT_AR_OPLIST", "1") != "0"   # the folded step as ONE pt_run_ops call per frame (else per launch)


class _FoldedStep:
    """The folded decode step as a FIXED list of launches (pt_run_ops): every buffer is allocated once, every descriptor built once
    -- the per-frame quantities (frame index, K/V length, previous codes, uniforms row) live on the device -- and a frame is one
    C-ABI call that issues the ~36 launches back to back (no Python between them, nothing to capture)."""

    def __init__(self, model, st, caches, ctx2, B, S, T, k, uniforms, temperature, t_dev, kv_len, prev_buf, codes):
        import ctypes as C
        from . import _lib as L
        self.C, self.L = C, L
        dev, bf, d = ctx2.device, st.dtype, model.d
        self.keep, self.ops = [], []
        new = lambda *shape, dtype=bf: self._keep(torch.empty(*shape, dtype=dtype, device=dev))
        pos = self._keep(model._pos.to(bf)); emb = st.w(model.code_embedding)
        h = new(B, d)
        e = L.pt_ar_embed_desc()
        e.prev, e.emb, e.pos, e.t_dev, e.out = prev_buf.data_ptr(), emb.data_ptr(), pos.data_ptr(), t_dev.data_ptr(), h.data_ptr()
        e.B, e.n_q, e.bins, e.dim = B, model.n_q, model.bins, d
        self._op(L.PT_OP_AR_EMBED, e)
        lse = new(B, model.heads, 1, dtype=torch.float32)
        for blk, (kc, vc) in zip(model.blocks, caches):
            a1, a2, ff = blk.attn1, blk.attn2, blk.ff
            Tm = kc.shape[1]
            q, o, h1, q2, o2, h2, out = (new(B, d) for _ in range(7))
            fw = st.fused([a1.to_q.weight, a1.to_k.weight, a1.to_v.weight])[0]
            self._lin(h, fw, q, B, 3 * d, d, ln=blk.norm1, st=st, seg_cols=d, y2=kc, ld2=Tm * d, y3=vc, ld3=Tm * d, t_dev=t_dev, t_stride=d)
            self._attn(q, kc.view(B * Tm, d), vc.view(B * Tm, d), o, lse, B, model.heads, Tm, d // model.heads, a1.scale, kv_len)
            self._lin(o, st.w(a1.to_out[0].weight), h1, B, d, d, bias=st.f(a1.to_out[0].bias), residual=h)
            self._lin(h1, st.w(a2.to_q.weight), q2, B, d, d, ln=blk.norm2, st=st)
            k2, v2 = self._cross_kv(st, a2, ctx2, d)
            self._attn(q2, k2, v2, o2, lse, B, a2.heads, S, a2.dim_head, a2.scale, None)
            self._lin(o2, st.w(a2.to_out[0].weight), h2, B, d, d, bias=st.f(a2.to_out[0].bias), residual=h1)
            p1, p2 = ff.net[0].proj, ff.net[2]
            F2 = p1.weight.shape[0]
            act = new(B, F2 // 2)
            self._lin(h2, st.w(p1.weight), act, B, F2, d, ln=blk.norm3, st=st, bias=st.f(p1.bias), geglu=True)
            self._lin(act, st.w(p2.weight), out, B, d, F2 // 2, bias=st.f(p2.bias), residual=h2)
            h = out
        NV = model.n_q * model.bins
        logits = new(B, NV)
        self._lin(h, st.w(model.head.proj.weight), logits, B, NV, d, ln=model.norm_out, st=st, bias=st.f(model.head.proj.bias))
        idx = new(B * model.n_q, dtype=torch.int64)
        u = None
        if k > 1:
            u = new(B * model.n_q, dtype=torch.float32)
            r = L.pt_row_select_desc()
            r.src, r.ld, r.t_dev, r.dst, r.n = uniforms.data_ptr(), uniforms.stride(0), t_dev.data_ptr(), u.data_ptr(), B * model.n_q
            self._op(L.PT_OP_ROW_SELECT, r)
        sd = L.pt_sample_desc()
        sd.logits, sd.ld, sd.uniforms, sd.out = logits.data_ptr(), model.bins, (u.data_ptr() if u is not None else None), idx.data_ptr()
        sd.R, sd.V, sd.k, sd.temperature, sd.dtype = B * model.n_q, model.bins, k, temperature, ops._DT[bf]
        self._op(L.PT_OP_SAMPLE_TOPK, sd)
        ad = L.pt_ar_advance_desc()
        ad.idx, ad.prev, ad.codes, ad.t_dev, ad.kv_len = idx.data_ptr(), prev_buf.data_ptr(), codes.data_ptr(), t_dev.data_ptr(), kv_len.data_ptr()
        ad.B, ad.n_q, ad.T = B, model.n_q, T
        self._op(L.PT_OP_AR_ADVANCE, ad)
        self.keep += [uniforms, t_dev, kv_len, prev_buf, codes, caches, ctx2]
        self.arr = (L.pt_op * len(self.ops))(*self.ops)

    def _keep(self, t):
        self.keep.append(t)
        return t

    def _op(self, kind, desc, dtype=0):
        self.keep.append(desc)
        o = self.L.pt_op()
        o.kind, o.dtype, o.desc = kind, dtype, self.C.cast(self.C.pointer(desc), self.C.c_void_p)
        self.ops.append(o)

    def _lin(self, x, w, out, M, N, K, ln=None, st=None, bias=None, residual=None, geglu=False, seg_cols=0, y2=None, ld2=0, y3=None, ld3=0,
             t_dev=None, t_stride=0):
        d = self.L.pt_decode_linear_desc()
        d.M, d.N, d.K, d.x, d.ldx = M, N, K, x.data_ptr(), x.stride(0)
        if ln is not None:
            g, b = st.f(ln.weight), st.f(ln.bias)
            d.ln_gamma, d.ln_beta, d.ln_eps = g.data_ptr(), b.data_ptr(), ln.eps
            self.keep += [g, b]
        d.geglu, d.w, d.ldw = int(geglu), w.data_ptr(), w.stride(0)
        if bias is not None:
            d.bias = bias.data_ptr()
        if residual is not None:
            d.residual, d.ldr = residual.data_ptr(), residual.stride(0)
        d.y, d.ldy, d.seg_cols = out.data_ptr(), out.stride(0), seg_cols
        if seg_cols:
            d.y2, d.ld2, d.y3, d.ld3, d.t_dev, d.t_stride = y2.data_ptr(), ld2, y3.data_ptr(), ld3, t_dev.data_ptr(), t_stride
        self.keep += [x, w, out, bias, residual]
        self._op(self.L.PT_OP_DECODE_LINEAR, d)

    def _attn(self, q, kk, vv, o, lse, B, H, Nk, D, scale, kv_len):
        d = ops.attn_desc(q, kk, vv, o, lse, B, H, 1, Nk, D, scale, False, kv_len)
        self.keep += [q, kk, vv, o, lse, kv_len]
        self._op(self.L.PT_OP_ATTN_FWD, d, ops._DT[q.dtype])

    def _cross_kv(self, st, a2, ctx2, C_):
        def project_kv():
            fkv = st.fused([a2.to_k.weight, a2.to_v.weight])
            if fkv is not None:
                buf = E.linear_fwd(ctx2, fkv[0])
                return buf[:, :C_], buf[:, C_:], buf
            return E.linear_fwd(ctx2, st.w(a2.to_k.weight)), E.linear_fwd(ctx2, st.w(a2.to_v.weight)), None
        k2, v2, buf = E.cached_cross_kv(a2, ctx2, project_kv)
        self.keep += [k2, v2, buf]
        return k2, v2

    def run(self):
        from ._lib import lib, check
        check(lib.pt_run_ops(self.arr, len(self.ops), ops._stream()), "pt_run_ops")


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

    def _fusable(self, st, B):
        """The folded decode step: bf16, <= 64 prompts, d_model <= 512 (the LayerNorm prologue holds a row per wave), fused q|k|v
        weights and interleaved GEGLU projections in the store."""
        if not AR_FUSED or st.dtype != torch.bfloat16 or B > 64 or self.d not in (256, 512):
            return False
        for blk in self.blocks:
            a1 = blk.attn1
            if st.fused([a1.to_q.weight, a1.to_k.weight, a1.to_v.weight]) is None:
                return False
            if id(blk.ff.net[0].proj.weight) not in st.geglu_ids or blk.attn2 is None:
                return False
        return True

    def _block_step_fused(self, st, blk, h, cache, t_dev, kv_len, ctx2, B, S):
        """One decode step of a block in 8 launches: [LN1 + q|k|v + cache append] attention [out + residual] [LN2 + q] cross-attention
        [out + residual] [LN3 + GEGLU projection] [ff2 + residual] -- same arithmetic as _block_step, rounding point by rounding point."""
        C, dev = self.d, h.device
        a1, a2, ff = blk.attn1, blk.attn2, blk.ff
        kc, vc = cache
        Tm = kc.shape[1]
        q = torch.empty(B, C, dtype=h.dtype, device=dev)
        fw = st.fused([a1.to_q.weight, a1.to_k.weight, a1.to_v.weight])[0]
        ops.decode_linear(h, fw, q, B, 3 * C, C, ln=(st.f(blk.norm1.weight), st.f(blk.norm1.bias)), seg_cols=C,
                          y2=kc, ld2=Tm * C, y3=vc, ld3=Tm * C, t_dev=t_dev, t_stride=C)
        o = torch.empty(B, C, dtype=h.dtype, device=dev)
        lse = torch.empty(B, self.heads, 1, dtype=torch.float32, device=dev)
        ops.attn_fwd(q, kc.view(B * Tm, C), vc.view(B * Tm, C), o, lse, B, self.heads, 1, Tm, C // self.heads, a1.scale, False, kv_len)
        h1 = torch.empty_like(h)
        ops.decode_linear(o, st.w(a1.to_out[0].weight), h1, B, C, C, bias=st.f(a1.to_out[0].bias), residual=h)
        q2 = torch.empty(B, C, dtype=h.dtype, device=dev)
        ops.decode_linear(h1, st.w(a2.to_q.weight), q2, B, C, C, ln=(st.f(blk.norm2.weight), st.f(blk.norm2.bias)))

        def project_kv():
            fkv = st.fused([a2.to_k.weight, a2.to_v.weight])
            if fkv is not None:
                buf = E.linear_fwd(ctx2, fkv[0])
                return buf[:, :C], buf[:, C:], buf
            return E.linear_fwd(ctx2, st.w(a2.to_k.weight)), E.linear_fwd(ctx2, st.w(a2.to_v.weight)), None
        k2, v2, _ = E.cached_cross_kv(a2, ctx2, project_kv)               # the prompt's K / V: projected once per generate()
        o2 = torch.empty(B, C, dtype=h.dtype, device=dev)
        ops.attn_fwd(q2, k2, v2, o2, lse, B, a2.heads, 1, S, a2.dim_head, a2.scale, False, None)
        h2 = torch.empty_like(h)
        ops.decode_linear(o2, st.w(a2.to_out[0].weight), h2, B, C, C, bias=st.f(a2.to_out[0].bias), residual=h1)
        p1, p2 = ff.net[0].proj, ff.net[2]
        F2 = p1.weight.shape[0]
        act = torch.empty(B, F2 // 2, dtype=h.dtype, device=dev)
        ops.decode_linear(h2, st.w(p1.weight), act, B, F2, C, ln=(st.f(blk.norm3.weight), st.f(blk.norm3.bias)), bias=st.f(p1.bias), geglu=True)
        out = torch.empty_like(h)
        ops.decode_linear(act, st.w(p2.weight), out, B, C, F2 // 2, bias=st.f(p2.bias), residual=h2)
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
        fused = self._fusable(st, B)
        if graph is None:
            # the folded step as ONE pt_run_ops call per frame issues its launches back to back from C: faster than replaying them as a
            # HIP graph (0.376 vs 0.415 ms per frame, tools/ar_probe.py), so the graph is the default only for the unfolded step
            graph = AR_GRAPH and not (fused and AR_OPLIST)
        n_eager = T if not graph or T < 4 else 2
        # frames on the training kernels: all the launch-by-launch ones, or -- with the folded decode step -- only frame 0 (the BOS row);
        # the folded step then runs launch by launch up to n_eager and as a replayed graph after it: the same kernels either way
        first = min(1, T) if fused else n_eager
        with E.cross_kv_cache():
            for t in range(first):
                h = self._embed(st, prev, t0=t)
                kv_len = torch.full((B,), t + 1, dtype=torch.int32, device=dev)
                for blk, cache in zip(self.blocks, caches):
                    h = self._block_step(st, blk, h, cache, t, kv_len, ctx2, B, S)
                n, _ = E.layernorm_fwd(h, st.f(self.norm_out.weight), st.f(self.norm_out.bias))
                logits = self.head.fwd(st, n).view(B * self.n_q, self.bins)
                idx = ops.sample_topk(logits, k=k, uniforms=uniforms[t].contiguous() if k > 1 else None, temperature=temperature)
                prev = idx.view(B, self.n_q, 1)
                codes[:, :, t] = prev[:, :, 0]
            if first < T:
                # ---- the decode step with every per-frame quantity on the device ----
                t_dev = torch.full((1,), first, dtype=torch.int64, device=dev)
                kv_len = torch.full((B,), first + 1, dtype=torch.int32, device=dev)
                prev_buf = prev.contiguous().clone()
                pos = self._pos.to(st.dtype)
                emb = st.w(self.code_embedding)

                folded = None
                if fused and AR_OPLIST:
                    if k > 1:
                        uniforms = uniforms.contiguous()
                    folded = _FoldedStep(self, st, caches, ctx2, B, S, T, k, uniforms, temperature, t_dev, kv_len, prev_buf, codes)

                def step():
                    if folded is not None:
                        folded.run()
                        return
                    if fused:
                        h = torch.empty(B, self.d, dtype=st.dtype, device=dev)
                        ops.ar_embed(prev_buf, emb, pos, t_dev, h, B, self.n_q, self.bins, self.d)
                        for blk, cache in zip(self.blocks, caches):
                            h = self._block_step_fused(st, blk, h, cache, t_dev, kv_len, ctx2, B, S)
                        logits = torch.empty(B, self.n_q * self.bins, dtype=st.dtype, device=dev)
                        ops.decode_linear(h, st.w(self.head.proj.weight), logits, B, self.n_q * self.bins, self.d,
                                          ln=(st.f(self.norm_out.weight), st.f(self.norm_out.bias)), bias=st.f(self.head.proj.bias))
                        u = uniforms.index_select(0, t_dev).view(-1) if k > 1 else None
                        idx = ops.sample_topk(logits.view(B * self.n_q, self.bins), k=k, uniforms=u, temperature=temperature)
                        ops.ar_advance(idx, prev_buf, codes, t_dev, kv_len, B, self.n_q, T)
                        return
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
                for _ in range(first, n_eager):
                    step()
                g = None
                if n_eager < T:
                    try:
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g):
                            step()
                    except RuntimeError as e:                # e.g. another capture in progress on this device: same step, launch by launch
                        import warnings
                        warnings.warn(f"ARCodecDecoder.generate: HIP graph capture failed ({e}); running the decode step launch by launch")
                        g = None
                for _ in range(n_eager, T):
                    if g is not None:
                        g.replay()
                    else:
                        step()
        return codes
