"""Thin host wrappers: torch device tensors in, C-ABI calls out (on torch's current HIP stream).

PyTorch is plumbing here (device memory + streams); every device op below is a hand-written HIP
kernel in prompt_tts_amd/csrc reached through include/prompt_tts_hip.h.  No CPU fallbacks.
"""
import ctypes as C

import torch

from . import _lib as L
from ._lib import lib, check

_DT = {torch.float32: L.PT_F32, torch.bfloat16: L.PT_BF16}


def pt_dtype(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"activation dtype must be float32 or bfloat16, got {t.dtype}") from None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("prompt_tts_amd ops need tensors on the GPU; there is no CPU path")


# ---- operand builders ---------------------------------------------------------------------------------

def plain(t, ld=None, trans=False):
    """2-D row-major view [rows][cols] with leading dimension ld (elements)."""
    o = L.pt_operand()
    o.p = t.data_ptr(); o.ld = ld if ld is not None else t.stride(0)
    o.kind = L.PT_V_PLAIN; o.trans = int(trans)
    return o


def concat(t1, t2, trans=False):
    o = L.pt_operand()
    o.p = t1.data_ptr(); o.ld = t1.stride(0); o.p2 = t2.data_ptr(); o.ld2 = t2.stride(0)
    o.c_split = t1.shape[1]; o.kind = L.PT_V_CONCAT; o.trans = int(trans)
    return o


def conv(t, cin, n_out, n_in, rowmap=L.PT_MAP_S1, taps=3, trans=False, stride=1):
    o = L.pt_operand()
    o.p = t.data_ptr(); o.ld = t.stride(0); o.kind = L.PT_V_CONV; o.trans = int(trans)
    o.taps = taps; o.cin = cin; o.rowmap = rowmap; o.n_out = n_out; o.n_in = n_in; o.stride = stride
    return o


def wflip(w3, cout, cin_pad, trans=True):
    """conv weight shadow [Cout][3][cin_pad] read as V[tap*Cout + co][ci] = W[co][2-tap][ci]."""
    o = L.pt_operand()
    o.p = w3.data_ptr(); o.ld = cin_pad; o.kind = L.PT_V_WFLIP; o.trans = int(trans); o.cin = cout; o.taps = 3
    return o


def gemm(M, N, K, A, B, out, dtype, ldc=None, out_kind=L.PT_OUT_T, split_k=1, bias=None, row_bias=None,
         row_bias_rows=0, row_bias_ld=0, residual=None, ldr=0, residual2=None, ldr2=0, conv_wgrad_cin=0, conv_wgrad_cin_store=0, alpha=1.0,
         act=0, out2=None, ldc2=0, act2=0, arow_sum=None, arow_n=0, arow_rep=1, arow_stride=0, x3=False, x2_block=0):
    d = L.pt_gemm_desc()
    d.f32_x3 = int(x3)             # f32 only: bf16 x 3 products instead of the exact f32 MFMA (inference at fp32 tolerance)
    d.x2_block = x2_block          # PT_BF16X2 only: plane blocking of the output columns (0 = N)
    d.M, d.N, d.K = M, N, K
    d.A, d.B = A, B
    d.C = out.data_ptr(); d.ldc = ldc if ldc is not None else N
    d.out_kind = out_kind; d.split_k = split_k
    d.bias = _p(bias); d.row_bias = _p(row_bias); d.row_bias_rows = row_bias_rows; d.row_bias_ld = row_bias_ld
    d.residual = _p(residual); d.ldr = ldr
    d.residual2 = _p(residual2); d.ldr2 = ldr2
    d.conv_wgrad_cin = conv_wgrad_cin; d.conv_wgrad_cin_store = conv_wgrad_cin_store
    d.alpha = alpha
    d.act = act; d.act2 = act2; d.C2 = _p(out2); d.ldc2 = ldc2
    d.arow_sum = _p(arow_sum); d.arow_n = arow_n; d.arow_rep = arow_rep; d.arow_stride = arow_stride
    check(lib.pt_gemm(C.byref(d), dtype, _stream()), "pt_gemm")


def fp8_quantize(x, out, state, fmt=L.PT_FP8_E4M3, out_t=None):
    """bf16 [rows][cols] -> fp8 bytes (uint8 tensor) with the per-tensor scale left in state[1] (state[0] = amax)."""
    _dev(x, out, state)
    rows, cols = x.shape
    check(lib.pt_fp8_quantize(_p(x), rows, cols, x.stride(0), _p(out), out.stride(0), _p(out_t), out_t.stride(0) if out_t is not None else 0,
                              _p(state), fmt, _stream()), "pt_fp8_quantize")


def gemm_fp8(M, N, K, a8, b8, out, state_a, state_b, a_format=L.PT_FP8_E4M3, ldc=None, bias=None, residual=None, ldr=0,
             residual2=None, ldr2=0, alpha=1.0, act=0, out2=None, ldc2=0, act2=0):
    """out (bf16) = scale_a scale_b a8 [M][K] b8[N][K]^T + epilogue; a8 / b8 uint8 tensors of fp8 bytes, state_* from fp8_quantize."""
    d = L.pt_gemm_desc()
    d.M, d.N, d.K = M, N, K
    d.A, d.B = plain(a8), plain(b8)
    d.C = out.data_ptr(); d.ldc = ldc if ldc is not None else N
    d.out_kind = L.PT_OUT_T; d.split_k = 1
    d.bias = _p(bias); d.residual = _p(residual); d.ldr = ldr; d.residual2 = _p(residual2); d.ldr2 = ldr2
    d.alpha = alpha
    d.act = act; d.act2 = act2; d.C2 = _p(out2); d.ldc2 = ldc2
    check(lib.pt_gemm_fp8(C.byref(d), a_format, C.c_void_p(state_a.data_ptr() + 4), C.c_void_p(state_b.data_ptr() + 4), _stream()), "pt_gemm_fp8")


def gemm_desc(M, N, K, A, B, out, ldc=None, out_kind=L.PT_OUT_T, split_k=1, alpha=1.0, arow_sum=None, arow_n=0, arow_rep=1,
              arow_stride=0, geglu_rows=0):
    """A bare pt_gemm_desc (weight-gradient form) for wgrad_group()."""
    d = L.pt_gemm_desc()
    d.M, d.N, d.K = M, N, K
    d.A, d.B = A, B
    d.C = out.data_ptr(); d.ldc = ldc if ldc is not None else N
    d.out_kind = out_kind; d.split_k = split_k; d.alpha = alpha
    d.arow_sum = _p(arow_sum); d.arow_n = arow_n; d.arow_rep = arow_rep; d.arow_stride = arow_stride
    d.geglu_rows = geglu_rows
    return d


WGRAD_GROUP_MAX = 8


def wgrad_group_ws_floats(target_wgs=256):
    return int(lib.pt_wgrad_group_ws_floats(target_wgs))


def wgrad_group(descs, ws, target_wgs=256):
    """descs: list of weight-gradient pt_gemm_desc (<= WGRAD_GROUP_MAX, one B operand class); ws: f32 scratch tensor."""
    arr = (L.pt_gemm_desc * len(descs))(*descs)
    check(lib.pt_wgrad_group(arr, len(descs), _p(ws), ws.numel(), target_wgs, _stream()), "pt_wgrad_group")


def attn_desc(q, k, v, o, lse, B, H, Nq, Nk, D, scale, causal=False, kv_len=None):
    d = L.pt_attn_desc()
    d.B, d.H, d.Nq, d.Nk, d.D = B, H, Nq, Nk, D
    d.q, d.ldq = q.data_ptr(), q.stride(0)
    d.k, d.ldk = k.data_ptr(), k.stride(0)
    d.v, d.ldv = v.data_ptr(), v.stride(0)
    d.o, d.ldo = o.data_ptr(), o.stride(0)
    d.lse = lse.data_ptr(); d.scale = scale; d.causal = int(causal); d.kv_len = _p(kv_len)
    return d


def attn_fwd(q, k, v, o, lse, B, H, Nq, Nk, D, scale, causal=False, kv_len=None):
    """q: [B*Nq, >=H*D] rows (column slices allowed), k/v: [B*Nk, ...]; o like q; lse f32 [B,H,Nq]."""
    _dev(q, k, v, o, lse)
    d = attn_desc(q, k, v, o, lse, B, H, Nq, Nk, D, scale, causal, kv_len)
    check(lib.pt_attn_fwd(C.byref(d), pt_dtype(q), _stream()), "pt_attn_fwd")


def attn_bwd(q, k, v, o, lse, do, delta, dq, dk, dv, B, H, Nq, Nk, D, scale, causal=False, kv_len=None):
    _dev(q, k, v, o, lse, do, delta, dq, dk, dv)
    d = attn_desc(q, k, v, o, lse, B, H, Nq, Nk, D, scale, causal, kv_len)
    d.d_o, d.lddo = do.data_ptr(), do.stride(0)
    d.delta = delta.data_ptr()
    d.dq, d.lddq = dq.data_ptr(), dq.stride(0)
    d.dk, d.lddk = dk.data_ptr(), dk.stride(0)
    d.dv, d.lddv = dv.data_ptr(), dv.stride(0)
    check(lib.pt_attn_bwd(C.byref(d), pt_dtype(q), _stream()), "pt_attn_bwd")


# ---- norms / elementwise ---------------------------------------------------------------------------------

def layernorm_fwd(x, gamma, beta, y, mean, rstd, eps=1e-5):
    M, Cc = x.shape
    check(lib.pt_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, Cc, eps, pt_dtype(x),
                               _stream()), "pt_layernorm_fwd")


def layernorm_bwd(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, n_rep=1, rep_stride=0):
    """n_rep > 1: dgamma / dbeta are replica 0 of replicated destinations `rep_stride` floats apart (fold_replicas)."""
    M, Cc = x.shape
    check(lib.pt_layernorm_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx), _p(dgamma),
                               _p(dbeta), M, Cc, n_rep, rep_stride, pt_dtype(x), _stream()), "pt_layernorm_bwd")


def groupnorm_stats(x1, x2, mean, rstd, B, N, G, eps):
    C1 = x1.shape[-1]; C2 = x2.shape[-1] if x2 is not None else 0
    check(lib.pt_groupnorm_stats(_p(x1), _p(x2), _p(mean), _p(rstd), B, N, C1, C2, G, eps, pt_dtype(x1), _stream()),
          "pt_groupnorm_stats")


def groupnorm_fwd(x1, x2, gamma, beta, y, mean, rstd, B, N, G, eps, silu):
    """Statistics + normalisation [+ SiLU] in one call; mean / rstd come back finalized."""
    C1 = x1.shape[-1]; C2 = x2.shape[-1] if x2 is not None else 0
    check(lib.pt_groupnorm_fwd(_p(x1), _p(x2), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), B, N, C1, C2, G, eps,
                               int(silu), pt_dtype(x1), _stream()), "pt_groupnorm_fwd")


def groupnorm_apply(x1, x2, mean, rstd, gamma, beta, y, xcat, B, N, G, silu, raw_eps=-1.0):
    """raw_eps >= 0: mean/rstd hold the raw (sum, sum of squares) of groupnorm_stats(eps < 0)."""
    C1 = x1.shape[-1]; C2 = x2.shape[-1] if x2 is not None else 0
    check(lib.pt_groupnorm_apply(_p(x1), _p(x2), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(y), _p(xcat),
                                 B, N, C1, C2, G, int(silu), raw_eps, pt_dtype(x1), _stream()), "pt_groupnorm_apply")


def groupnorm_bwd(dy, x1, x2, mean, rstd, gamma, beta, dres, dx1, dx2, dgamma, dbeta, ws, B, N, G, silu,
                  accumulate_dx2=False, raw_eps=-1.0, ws_zeroed=False, n_rep=1, rep_stride=0, item_sum=None):
    """item_sum (B, >= C1) f32 or None: item_sum[b][c] += sum_n dx1[(b, n)][c] (a resnet's time-embedding gradient)."""
    C1 = x1.shape[-1]; C2 = x2.shape[-1] if x2 is not None else 0
    check(lib.pt_groupnorm_bwd(_p(dy), _p(x1), _p(x2), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dres), _p(dx1),
                               _p(dx2), _p(dgamma), _p(dbeta), _p(ws), B, N, C1, C2, G, int(silu),
                               int(accumulate_dx2), raw_eps, int(ws_zeroed), n_rep, rep_stride,
                               _p(item_sum), item_sum.stride(0) if item_sum is not None else 0, pt_dtype(x1), _stream()),
          "pt_groupnorm_bwd")


def geglu_fwd(proj, out, bias=None, interleaved=False):
    """out = value * gelu(gate); `bias` (f32, original column order) is added to proj in place first."""
    M, F2 = proj.shape
    check(lib.pt_geglu_fwd(_p(proj), _p(bias), _p(out), M, F2 // 2, int(interleaved), pt_dtype(proj), _stream()), "pt_geglu_fwd")


def geglu_bwd(dout, proj, dproj, interleaved=False):
    M, F2 = proj.shape
    check(lib.pt_geglu_bwd(_p(dout), _p(proj), _p(dproj), M, F2 // 2, int(interleaved), pt_dtype(proj), _stream()), "pt_geglu_bwd")


def silu_fwd(x, y):
    check(lib.pt_silu_fwd(_p(x), _p(y), x.numel(), pt_dtype(x), _stream()), "pt_silu_fwd")


def silu_bwd(dy, x, dx):
    check(lib.pt_silu_bwd(_p(dy), _p(x), _p(dx), x.numel(), pt_dtype(x), _stream()), "pt_silu_bwd")


def dropout(x, keep, y, scale, residual=None):
    """y = x * keep * scale [+ residual]; keep: uint8 tensor of x's shape (1 = kept)."""
    _dev(x, keep, y)
    check(lib.pt_dropout(_p(x), _p(keep), _p(residual), _p(y), x.numel(), scale, pt_dtype(x), _stream()), "pt_dropout")


def add(a, b, y):
    check(lib.pt_add(_p(a), _p(b), _p(y), a.numel(), pt_dtype(a), _stream()), "pt_add")


def pairsum_rows(x, y):
    rows, Cc = y.shape
    check(lib.pt_pairsum_rows(_p(x), _p(y), rows, Cc, pt_dtype(x), _stream()), "pt_pairsum_rows")


def colsum(dy, out, M=None, N=None, seg_rows=None, ld_out=0, n_rep=1, rep_stride=0):
    M = dy.shape[0] if M is None else M
    N = dy.shape[1] if N is None else N
    check(lib.pt_colsum(_p(dy), dy.stride(0), _p(out), ld_out, M, N, M if seg_rows is None else seg_rows, n_rep, rep_stride,
                        pt_dtype(dy), _stream()), "pt_colsum")


def fold_replicas(arena, dst, segs_dev, n_segs, n_rep, max_n):
    check(lib.pt_fold_replicas(_p(arena), _p(dst), _p(segs_dev), n_segs, n_rep, max_n, _stream()), "pt_fold_replicas")


def transpose_batch(table_dev, n_seg, n_tiles):
    check(lib.pt_transpose_batch(_p(table_dev), n_seg, n_tiles, L.PT_BF16, _stream()), "pt_transpose_batch")


def embedding_fwd(ids, W, pos, out, S):
    BS, d = out.shape
    check(lib.pt_embedding_fwd(_p(ids), _p(W), _p(pos), _p(out), BS, S, d, W.shape[0], pt_dtype(out), _stream()),
          "pt_embedding_fwd")


def embedding_bwd(ids, dout, dW):
    BS, d = dout.shape
    check(lib.pt_embedding_bwd(_p(ids), _p(dout), _p(dW), BS, d, dW.shape[0], pt_dtype(dout), _stream()),
          "pt_embedding_bwd")


def timestep_embedding(t, out, flip_sin_to_cos=True, shift=0.0):
    B, Cc = out.shape
    check(lib.pt_timestep_embedding(_p(t), _p(out), B, Cc, int(flip_sin_to_cos), shift, pt_dtype(out), _stream()),
          "pt_timestep_embedding")


def add_noise(x0, noise, t, alphas_cumprod, xt, n_q, T, cpad):
    B = x0.shape[0]
    check(lib.pt_add_noise(_p(x0), _p(noise), _p(t), _p(alphas_cumprod), _p(xt), B, n_q, T, cpad, pt_dtype(xt),
                           _stream()), "pt_add_noise")


def tokens_to_bct(x, out, B, n_q, T, cpad):
    check(lib.pt_tokens_to_bct(_p(x), _p(out), B, n_q, T, cpad, pt_dtype(x), _stream()), "pt_tokens_to_bct")


def bct_to_tokens(x, out, B, n_q, T, cpad):
    check(lib.pt_bct_to_tokens(_p(x), _p(out), B, n_q, T, cpad, pt_dtype(out), _stream()), "pt_bct_to_tokens")


def mse_loss(pred, noise, loss, dpred, gscale, B, n_q, T, cpad):
    check(lib.pt_mse_loss(_p(pred), _p(noise), _p(loss), _p(dpred), gscale, B, n_q, T, cpad, pt_dtype(pred),
                          _stream()), "pt_mse_loss")


def sumsq(g, out):
    check(lib.pt_sumsq(_p(g), _p(out), g.numel(), _stream()), "pt_sumsq")


def adamw_step(p, g, m, v, shadow, seg_dev, n_seg, gnorm_sq, max_norm, lr, beta1, beta2, eps, wd, step):
    check(lib.pt_adamw_step(_p(p), _p(g), _p(m), _p(v), _p(shadow), _p(seg_dev), n_seg, p.numel(), _p(gnorm_sq),
                            max_norm, lr, beta1, beta2, eps, wd, step, pt_dtype(shadow), _stream()), "pt_adamw_step")


def adamw_step_range(p, g, m, v, shadow, seg_dev, n_seg, lo, hi, gnorm_sq, max_norm, lr, beta1, beta2, eps, wd, step, publish):
    """The fused AdamW over flat positions [lo, hi) in gradient order; publish: the new values overwrite g[lo:hi]."""
    check(lib.pt_adamw_step_range(_p(p), _p(g), _p(m), _p(v), _p(shadow), _p(seg_dev), n_seg, p.numel(), lo, hi, _p(gnorm_sq),
                                  max_norm, lr, beta1, beta2, eps, wd, step, int(publish), pt_dtype(shadow), _stream()),
          "pt_adamw_step_range")


def import_params_range(p, values, shadow, seg_dev, n_seg, lo, hi):
    """master / shadow <- values[lo:hi] (gradient order; what a peer rank published with adamw_step_range)."""
    check(lib.pt_import_params_range(_p(p), _p(values), _p(shadow), _p(seg_dev), n_seg, p.numel(), lo, hi, pt_dtype(shadow),
                                     _stream()), "pt_import_params_range")


def pack_shadow(p, shadow, seg_dev, n_seg):
    check(lib.pt_pack_shadow(_p(p), _p(shadow), _p(seg_dev), n_seg, p.numel(), pt_dtype(shadow), _stream()),
          "pt_pack_shadow")


def ddpm_step(x, eps, z, out, c_eps, c_inv, clip, c_x0, c_xt, sigma):
    check(lib.pt_ddpm_step(_p(x), _p(eps), _p(z), _p(out), x.numel(), c_eps, c_inv, clip, c_x0, c_xt, sigma, _stream()),
          "pt_ddpm_step")


def rvq_decode(codes, codebooks, out, B, n_q, T, bins, dim):
    """out[(b,t)][:] = sum_q codebooks[q][codes[b][q][t]][:]  (codes int64 (B, n_q, T); codebooks [n_q][bins][dim])."""
    _dev(codes, codebooks, out)
    check(lib.pt_rvq_decode(_p(codes), _p(codebooks), _p(out), B, n_q, T, bins, dim, pt_dtype(out), _stream()), "pt_rvq_decode")


def codes_from_continuous(x, bins=1024):
    """(B, n_q, T) f32 in [-1, 1] -> int64 code indices (inverse of the collate normalisation)."""
    x = x.contiguous().float()
    out = torch.empty(x.shape, dtype=torch.int64, device=x.device)
    check(lib.pt_codes_from_continuous(_p(x), _p(out), x.numel(), bins, _stream()), "pt_codes_from_continuous")
    return out


def sample_topk(logits, k=1, uniforms=None, temperature=1.0):
    """logits (R, V) -> int64 (R,): greedy (k=1) or top-k sampling with injected uniforms."""
    R, V = logits.shape
    out = torch.empty(R, dtype=torch.int64, device=logits.device)
    check(lib.pt_sample_topk(_p(logits), logits.stride(0), _p(uniforms), _p(out), R, V, k, temperature, pt_dtype(logits),
                             _stream()), "pt_sample_topk")
    return out


# ---- per-symbol device timing (bench.py --kernel-timing): HIP events around every C-ABI call of one step ----------

def profile_one_step(step_fn, capture=None):
    """Run step_fn() once with every pt_* call bracketed by HIP events on the launch stream.
    Returns {label: {calls, ms_total, ms_avg, tflops (GEMM/attention), gflop_avg}} sorted by time.
    capture: a list that receives (label, entry point, argument copies, flops) of every pt_gemm / pt_wgrad_group call
    (replayed by replay_captured)."""
    recs = []
    originals = {}

    def gemm_flops(d):
        return 2.0 * d.M * d.N * d.K

    def label_and_flops(name, args):
        if name == "pt_gemm":
            d = args[0]._obj; dt = {L.PT_BF16: "bf16", L.PT_BF16X2: "f32-class/x2"}.get(args[1], "f32")
            kind = ("N", "T")[d.A.trans] + ("N", "T")[d.B.trans]
            conv = "conv" if L.PT_V_CONV in (d.A.kind, d.B.kind) else "plain"
            return f"gemm<{dt},{kind},{'atomic' if d.out_kind == L.PT_OUT_F32_ATOMIC else 'store'}>/{conv}", gemm_flops(d)
        if name == "pt_gemm_fp8":
            d = args[0]._obj
            return f"gemm_fp8<{('e4m3', 'e5m2')[args[1]]} x e4m3>", gemm_flops(d)
        if name == "pt_wgrad_group":
            arr, n = args[0], args[1]
            conv = "conv" if any(arr[i].B.kind == L.PT_V_CONV for i in range(n)) else "plain"
            return f"wgrad_group<bf16>/{conv}", sum(gemm_flops(arr[i]) for i in range(n))
        if name in ("pt_attn_fwd", "pt_attn_bwd"):
            d = args[0]._obj
            return name, (4.0 if name == "pt_attn_fwd" else 14.0) * d.B * d.H * d.Nq * d.Nk * d.D
        return name, 0.0

    def wrap(name, fn):
        def inner(*args):
            lab, fl = label_and_flops(name, args)
            if capture is not None and name == "pt_gemm":
                d = args[0]._obj
                capture.append((lab, name, (C.byref(type(d).from_buffer_copy(d)),) + tuple(args[1:2]), fl))
            elif capture is not None and name == "pt_wgrad_group":
                arr = args[0]
                capture.append((lab, name, (type(arr).from_buffer_copy(arr),) + tuple(args[1:5]), fl))
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*args)
            e1.record()
            recs.append((lab, fl, e0, e1))
            return r
        return inner

    for name in L.SIGNATURES:
        originals[name] = getattr(lib, name)
        setattr(lib, name, wrap(name, originals[name]))
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        for name, fn in originals.items():
            setattr(lib, name, fn)
    agg = {}
    for lab, fl, e0, e1 in recs:
        a = agg.setdefault(lab, {"calls": 0, "ms_total": 0.0, "flops": 0.0})
        a["calls"] += 1; a["ms_total"] += e0.elapsed_time(e1); a["flops"] += fl
    out = {}
    for lab, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms_total"]):
        out[lab] = {"calls": a["calls"], "ms_total": round(a["ms_total"], 3), "ms_avg": round(a["ms_total"] / a["calls"], 4)}
        if a["flops"]:
            out[lab]["tflops"] = round(a["flops"] / (a["ms_total"] * 1e-3) / 1e12, 1)
            out[lab]["gflop_avg"] = round(a["flops"] / a["calls"] / 1e9, 2)
    return out


def replay_captured(captured, label, rounds=3):
    """Re-issue the captured launches with `label` back to back on the current stream (alone on the chip) and return
    (calls, average microseconds, TFLOP/s).  The operand buffers of the captured step may have been recycled by the caching
    allocator: they are still mapped, their contents are irrelevant to the timing, and the outputs (gradient buffers, scratch)
    are overwritten by the next real step anyway.  Timing only -- never used where values are checked."""
    sel = [(getattr(lib, fn), args, fl) for lab, fn, args, fl in captured if lab == label]
    if not sel:
        return 0, 0.0, 0.0
    st = _stream()
    for fn, args, _ in sel:
        check(fn(*args, st), "replay")
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        for fn, args, _ in sel:
            fn(*args, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (rounds * len(sel))
    return len(sel), us, sum(f for _, _, f in sel) / len(sel) / us / 1e6


# ---- autoregressive decode step (M <= 64 rows; csrc/decode_step.hip) ---------------------------------------------------------
def decode_linear(x, w, out, M, N, K, ln=None, eps=1e-5, bias=None, residual=None, geglu=False, seg_cols=0, y2=None, ld2=0, y3=None, ld3=0,
                  t_dev=None, t_stride=0):
    """out = epilogue([LayerNorm(x)] w^T): one launch for LN + Linear + bias + residual / GEGLU / the K-V cache scatter."""
    _dev(x, w, out)
    d = L.pt_decode_linear_desc()
    d.M, d.N, d.K = M, N, K
    d.x, d.ldx = x.data_ptr(), x.stride(0)
    if ln is not None:
        d.ln_gamma, d.ln_beta, d.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), eps
    d.geglu = int(geglu)
    d.w, d.ldw = w.data_ptr(), w.stride(0)
    d.bias = _p(bias)
    if residual is not None:
        d.residual, d.ldr = residual.data_ptr(), residual.stride(0)
    d.y, d.ldy = out.data_ptr(), out.stride(0)
    d.seg_cols = seg_cols
    if seg_cols:
        d.y2, d.ld2, d.t_dev, d.t_stride = y2.data_ptr(), ld2, t_dev.data_ptr(), t_stride
        if y3 is not None:
            d.y3, d.ld3 = y3.data_ptr(), ld3
    check(lib.pt_decode_linear(C.byref(d), _stream()), "pt_decode_linear")


def ar_embed(prev, emb, pos, t_dev, out, B, n_q, bins, dim):
    check(lib.pt_ar_embed(_p(prev), _p(emb), _p(pos), _p(t_dev), _p(out), B, n_q, bins, dim, _stream()), "pt_ar_embed")


def ar_advance(idx, prev, codes, t_dev, kv_len, B, n_q, T):
    check(lib.pt_ar_advance(_p(idx), _p(prev), _p(codes), _p(t_dev), _p(kv_len), B, n_q, T, _stream()), "pt_ar_advance")
