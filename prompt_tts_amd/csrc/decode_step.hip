// Autoregressive decode step (north_star ops with no reference counterpart, SURVEY 8a'): one frame for <= 64 prompts is ~70
// dependent launches of a few microseconds when it runs on the training kernels (LayerNorm, GEMM, two K/V-cache copies, attention,
// GEMM, ... per layer), launch-bound even as a replayed HIP graph (0.75 ms per frame).  The kernels here fold the launches that
// only exist because a 64-row problem was expressed with large-batch building blocks:
//   pt_decode_linear : [LayerNorm ->] Linear [-> bias] [-> + residual] [-> GEGLU] [-> scatter of column segments into K / V caches at
//                      the device-resident frame index] for M <= 64 rows, one launch
//   pt_ar_embed      : code-embedding gather-sum + position row of the device-resident frame index
//   pt_ar_advance    : what follows the sampler: previous codes, codes[:, :, t], t += 1, kv_len += 1
// Arithmetic is the separate launches' arithmetic, rounding point by rounding point (LayerNorm as ln_fwd_kernel: one wave per row,
// same lane -> column map and reduction order; the product as pt_gemm: one f32 accumulator per output element advanced by
// v_mfma_f32_16x16x32_bf16 steps of ascending k; bias, residual and GEGLU as gemm_epilogue / geglu_fwd_kernel), so that the fused
// step makes the decisions of the launch-by-launch step bit for bit (tests/test_ar_gpu.py).
#include <type_traits>
#include "mma.h"

namespace {

constexpr int DL_ROWS = 64, DL_KC = 512, DL_PITCH = (DL_KC + 8) * 2;      // LDS image of the activations: [64][512 + 8] bf16

struct DlinParams {
  int M, N, K;
  const bf16_t* x; int64_t ldx;
  const float* gamma; const float* beta; float eps;        // LayerNorm prologue (K <= 512) or NULL
  const bf16_t* w; int64_t ldw;
  const float* bias;
  const bf16_t* residual; int64_t ldr;
  bf16_t* y; int64_t ldy;
  int geglu;                                               // N = 2 F interleaved weight rows -> y[m][F]
  int seg_cols; bf16_t* y2; int64_t ld2; bf16_t* y3; int64_t ld3; const int64_t* t_dev; int64_t t_stride;
};

__device__ __forceinline__ float dl_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

// Workgroup = 16 output columns, wave w = row tile w (rows 16 w .. 16 w + 15): N / 16 workgroups stream the weights (a 64-column
// workgroup put 8 CUs on an N = 512 layer and walked K = 2048 in four dependent chunks: 17 / 46 us per launch), and every output
// element is still ONE accumulator advanced in ascending k.  With the LayerNorm prologue (LN_NKS = K / 32 = 8 or 16) a wave
// normalises its own 16 rows (all requested at once) into its private LDS rows and reads them back as A fragments; without it
// (LN_NKS = 0, K a multiple of 256) the A fragments come straight from global memory, in chunks of 256 columns requested one
// chunk ahead of the products that use them (as are the weights).  NO load is predicated: rows past M and columns past N read a
// valid row instead and are simply never stored (per-load conditions made the compiler wait for every load by itself and branch
// around every product: 12 us per launch, the same kernel as a straight run of loads then products: see DESIGN.md).
template <int LN_NKS>
__global__ __launch_bounds__(256) void dlin_kernel(const DlinParams p) {
  __shared__ __attribute__((aligned(16))) char xs[LN_NKS ? DL_ROWS * DL_PITCH : 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
  // weight row of this lane's B-operand column.  plain: 16 blockIdx + li.  GEGLU (interleaved rows: per 64, 32 values then their 32
  // gates): workgroup j = (q, w) = (j / 4, j % 4) takes value rows 64 q + 8 w + li (li < 8) and the gates of the same eight activation
  // columns (li >= 8), so that accumulator rows 4 g + r of lane groups g and g + 2 are value and gate of one column
  const int jq = blockIdx.x >> 2, jw = blockIdx.x & 3;
  int nrow = p.geglu ? 64 * jq + (li < 8 ? 8 * jw + li : 32 + 8 * jw + (li - 8)) : 16 * blockIdx.x + li;
  nrow = nrow < p.N ? nrow : p.N - 1;
  const bf16_t* wrow = p.w + (int64_t)nrow * p.ldw + 8 * g;
  const int m = 16 * wave + li;                                 // this lane's activation row (A fragment) / output row
  const bool mok = m < p.M;
  f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // what the epilogue adds is requested up front (loaded where it is used, every launch ended in two more load round trips)
  const int nb = 16 * blockIdx.x + 4 * g;                       // plain epilogue: this lane's 4 consecutive output columns
  const bool nbok = nb < p.N;
  float eb[4] = {0.f, 0.f, 0.f, 0.f};
  u32x2_t er = {0u, 0u};
  int64_t tcur = 0;
  if (!p.geglu) {
    if (p.bias) {
      const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(p.bias + (nbok ? nb : 0));
      eb[0] = bv[0]; eb[1] = bv[1]; eb[2] = bv[2]; eb[3] = bv[3];
    }
    if (p.residual) er = *reinterpret_cast<const u32x2_t*>(p.residual + (int64_t)(mok ? m : 0) * p.ldr + (nbok ? nb : 0));
    if (p.seg_cols > 0) tcur = p.t_dev[0];
  } else if (p.bias) {
    const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(p.bias + (g < 2 ? 0 : (p.N >> 1)) + 32 * jq + 8 * jw + 4 * (g & 1));
    eb[0] = bv[0]; eb[1] = bv[1]; eb[2] = bv[2]; eb[3] = bv[3];
  }
  if constexpr (LN_NKS > 0) {
    // LayerNorm of a row by one wave (ln_fwd_kernel<bf16_t, 1>: lane holds the 8 columns lane * 8 ..): rows 16 wave .. + 15
    constexpr int K = 32 * LN_NKS;
    char* xw = xs + 16 * wave * DL_PITCH;
    const int col = lane * 8;
    const bool cok = col < K;
    const int colc = cok ? col : 0;
    Vec16<bf16_t> vin[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int r = 16 * wave + i; r = r < p.M ? r : p.M - 1;
      vin[i] = load16(p.x + (int64_t)r * p.ldx + colc);
    }
    Frag<bf16_t> wf[LN_NKS];                                    // the weights travel while the rows are normalised
#pragma unroll
    for (int ks = 0; ks < LN_NKS; ++ks) frag_load_global(wf[ks], wrow + 32 * ks);
    float gm[8], bt[8];
    {
      const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(p.gamma + colc), g1 = *reinterpret_cast<const f32x4_t*>(p.gamma + colc + 4);
      const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(p.beta + colc), b1 = *reinterpret_cast<const f32x4_t*>(p.beta + colc + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { gm[e] = g0[e]; gm[4 + e] = g1[e]; bt[e] = b0[e]; bt[4 + e] = b1[e]; }
    }
    // the 16 rows' wave reductions advance TOGETHER (each is wave_sum's butterfly, v += shfl_xor(v, 32, 16, .. 1), unchanged per row:
    // one row after the other, 32 dependent cross-lane steps of ~100 cycles each were 8 of a launch's 20 us)
    float mu[16], rs[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += vin[i].get(e);
      mu[i] = cok ? s : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int i = 0; i < 16; ++i) mu[i] += __shfl_xor(mu[i], o, 64);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      mu[i] = mu[i] / (float)K;
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = vin[i].get(e) - mu[i]; ss = pt_ln_sq_acc(ss, d); }
      rs[i] = cok ? ss : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int i = 0; i < 16; ++i) rs[i] += __shfl_xor(rs[i], o, 64);
    if (cok) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        Vec16<bf16_t> o;
        const float rstd = pt_ln_rstd(rs[i], (float)K, p.eps);
#pragma unroll
        for (int e = 0; e < 8; ++e) o.set(e, pt_ln_apply(vin[i].get(e), mu[i], rstd, gm[e], bt[e]));
        *reinterpret_cast<u32x4_t*>(xw + i * DL_PITCH + col * 2) = o.raw;
      }
    }
    __syncthreads();
    Frag<bf16_t> fa[LN_NKS];
#pragma unroll
    for (int ks = 0; ks < LN_NKS; ++ks) fa[ks].v = *reinterpret_cast<const bf16x8_t*>(xw + li * DL_PITCH + (32 * ks + 8 * g) * 2);
#pragma unroll
    for (int ks = 0; ks < LN_NKS; ++ks) mma16(acc, wf[ks], fa[ks]);   // D[row = column 4 g + r of the 16][col = row 16 wave + li]
  } else {
    constexpr int CH = 8;                                       // k-steps per chunk (256 columns)
    const bf16_t* xrow = p.x + (int64_t)(mok ? m : p.M - 1) * p.ldx + 8 * g;
    const int nch = p.K / (32 * CH);
    Frag<bf16_t> wa[2][CH], xa[2][CH];
    auto request = [&](auto buf_c, int ch) {
      constexpr int buf = decltype(buf_c)::value;
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        frag_load_global(wa[buf][k], wrow + 32 * (CH * ch + k));
        frag_load_global(xa[buf][k], xrow + 32 * (CH * ch + k));
      }
    };
    request(std::integral_constant<int, 0>{}, 0);
    for (int ch = 0; ch < nch; ch += 2) {                       // two chunks per trip: buffer indices stay compile-time
      if (ch + 1 < nch) request(std::integral_constant<int, 1>{}, ch + 1);
#pragma unroll
      for (int k = 0; k < CH; ++k) mma16(acc, wa[0][k], xa[0][k]);
      if (ch + 2 < nch) request(std::integral_constant<int, 0>{}, ch + 2);
      if (ch + 1 < nch) {
#pragma unroll
        for (int k = 0; k < CH; ++k) mma16(acc, wa[1][k], xa[1][k]);
      }
    }
  }
  // ---- epilogue: lane (li, g) holds, for row m, the workgroup's columns 4 g + r ----
  if (p.geglu) {
    const int c0 = 32 * jq + 8 * jw + 4 * (g & 1);                 // activation columns c0 .. c0 + 3 (g < 2: value side)
    uint32_t mine[4], other[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // the separate launches: proj = bf16(acc) (pt_gemm), then proj = bf16(proj + bias) in place and act = bf16(h gelu(g)) (geglu_fwd_kernel)
      float v = bf16_bits_to_f32(f32_to_bf16_bits(acc[r]));
      if (p.bias) v = bf16_bits_to_f32(f32_to_bf16_bits(v + eb[r]));
      mine[r] = __float_as_uint(v);
      other[r] = (uint32_t)__shfl_xor((int)mine[r], 32, 64);                      // lane groups g and g + 2 swap
    }
    if (g < 2 && mok) {
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = __uint_as_float(mine[r]) * dl_gelu(__uint_as_float(other[r]));
      store4<bf16_t>(p.y + (int64_t)m * p.ldy + c0, o[0], o[1], o[2], o[3]);
    }
    return;
  }
  if (!nbok || !mok) return;
  bf16_t* dst = p.y; int64_t ldd = p.ldy; int ncol = nb;
  if (p.seg_cols > 0) {                                         // column segments: 0 -> y, 1 -> y2 (+ t), 2 -> y3 (+ t)
    const int seg = nb / p.seg_cols;
    ncol = nb - seg * p.seg_cols;
    if (seg == 1) { dst = p.y2 + tcur * p.t_stride; ldd = p.ld2; }
    else if (seg == 2) { dst = p.y3 + tcur * p.t_stride; ldd = p.ld3; }
  }
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = acc[r] + eb[r];
  if (p.residual) {
    v[0] += __uint_as_float(er[0] << 16); v[1] += __uint_as_float(er[0] & 0xffff0000u);
    v[2] += __uint_as_float(er[1] << 16); v[3] += __uint_as_float(er[1] & 0xffff0000u);
  }
  store4<bf16_t>(dst + (int64_t)m * ldd + ncol, v[0], v[1], v[2], v[3]);
}

// x[b][:] = bf16(bf16(sum_q emb[q][prev[b][q]][:]) + pos[t][:])  (pt_rvq_decode, then the bf16 add of the position row)
__global__ __launch_bounds__(256) void ar_embed_kernel(const int64_t* __restrict__ prev, const bf16_t* __restrict__ emb,
                                                       const bf16_t* __restrict__ pos, const int64_t* __restrict__ t_dev,
                                                       bf16_t* __restrict__ out, int B, int nq, int bins, int dim) {
  const int cpr = dim / 8;
  const int64_t t = t_dev[0];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < B * cpr; i += gridDim.x * 256) {
    const int b = i / cpr, c = (i - b * cpr) * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int q = 0; q < nq; ++q) {
      int64_t idx = prev[(int64_t)b * nq + q];
      idx = idx < 0 ? 0 : (idx >= bins ? bins - 1 : idx);
      const Vec16<bf16_t> v = load16(emb + ((int64_t)q * bins + idx) * dim + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v.get(e);
    }
    const Vec16<bf16_t> pv = load16(pos + t * dim + c);
    Vec16<bf16_t> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.set(e, bf16_bits_to_f32(f32_to_bf16_bits(acc[e])) + pv.get(e));
    store16(out + (int64_t)b * dim + c, o);
  }
}

// after the sampler: prev[i] = idx[i]; codes[b][q][t] = idx[b * nq + q]; then t += 1, kv_len[b] += 1 (one workgroup: the
// increments follow every read of t)
__global__ __launch_bounds__(256) void ar_advance_kernel(const int64_t* __restrict__ idx, int64_t* __restrict__ prev,
                                                         int64_t* __restrict__ codes, int64_t* __restrict__ t_dev,
                                                         int32_t* __restrict__ kv_len, int B, int nq, int64_t T) {
  const int64_t t = t_dev[0];
  for (int i = threadIdx.x; i < B * nq; i += 256) {
    const int64_t v = idx[i];
    prev[i] = v;
    if (t < T) codes[(int64_t)i * T + t] = v;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += 256) kv_len[b] += 1;
  if (threadIdx.x == 0) t_dev[0] = t + 1;
}

__global__ __launch_bounds__(256) void row_select_kernel(const float* __restrict__ src, int64_t ld, const int64_t* __restrict__ t_dev,
                                                         float* __restrict__ dst, int64_t n) {
  const float* row = src + t_dev[0] * ld;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = row[i];
}

}  // namespace

extern "C" int pt_row_select(const float* src, int64_t ld, const int64_t* t_dev, float* dst, int64_t n, pt_stream stream) {
  if (!src || !t_dev || !dst || n <= 0) return PT_ERR_ARG;
  int64_t blocks = (n + 255) / 256; if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(row_select_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, ld, t_dev, dst, n);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

extern "C" int pt_run_ops(const pt_op* ops, int64_t n, pt_stream stream) {
  if (!ops || n <= 0) return PT_ERR_ARG;
  for (int64_t i = 0; i < n; ++i) {
    const pt_op& o = ops[i];
    if (!o.desc) return PT_ERR_ARG;
    int st;
    switch (o.kind) {
      case PT_OP_DECODE_LINEAR: st = pt_decode_linear(static_cast<const pt_decode_linear_desc*>(o.desc), stream); break;
      case PT_OP_ATTN_FWD: st = pt_attn_fwd(static_cast<const pt_attn_desc*>(o.desc), o.dtype, stream); break;
      case PT_OP_AR_EMBED: { const auto* d = static_cast<const pt_ar_embed_desc*>(o.desc);
        st = pt_ar_embed(d->prev, d->emb, d->pos, d->t_dev, d->out, d->B, d->n_q, d->bins, d->dim, stream); } break;
      case PT_OP_SAMPLE_TOPK: { const auto* d = static_cast<const pt_sample_desc*>(o.desc);
        st = pt_sample_topk(d->logits, d->ld, d->uniforms, d->out, d->R, d->V, d->k, d->temperature, d->dtype, stream); } break;
      case PT_OP_AR_ADVANCE: { const auto* d = static_cast<const pt_ar_advance_desc*>(o.desc);
        st = pt_ar_advance(d->idx, d->prev, d->codes, d->t_dev, d->kv_len, d->B, d->n_q, d->T, stream); } break;
      case PT_OP_ROW_SELECT: { const auto* d = static_cast<const pt_row_select_desc*>(o.desc);
        st = pt_row_select(d->src, d->ld, d->t_dev, d->dst, d->n, stream); } break;
      default: return PT_ERR_ARG;
    }
    if (st != PT_OK) return st;
  }
  return PT_OK;
}

extern "C" int pt_decode_linear(const pt_decode_linear_desc* d, pt_stream stream) {
  if (!d) return PT_ERR_ARG;
  if (d->M <= 0 || d->M > DL_ROWS || d->N <= 0 || d->K <= 0 || d->K % 32 != 0 || d->N % 4 != 0) return PT_ERR_SHAPE;
  if (!d->x || !d->w || !d->y) return PT_ERR_ARG;
  if (!pt_aligned16(d->x) || (d->ldx * 2) % 16 || !pt_aligned16(d->w) || (d->ldw * 2) % 16 || (reinterpret_cast<uintptr_t>(d->y) & 7u) || d->ldy % 4) return PT_ERR_ALIGN;
  // LayerNorm prologue: K = 256 or 512 (a row per wave, a compile-time number of k-steps); without it K is walked in chunks of 256
  if (d->ln_gamma ? (!d->ln_beta || (d->K != 256 && d->K != 512)) : (d->K % 256 != 0)) return PT_ERR_SHAPE;
  if (d->ln_gamma && ((reinterpret_cast<uintptr_t>(d->ln_gamma) & 15u) || (reinterpret_cast<uintptr_t>(d->ln_beta) & 15u))) return PT_ERR_ALIGN;
  if (d->bias && (reinterpret_cast<uintptr_t>(d->bias) & 15u)) return PT_ERR_ALIGN;
  if (d->residual && ((reinterpret_cast<uintptr_t>(d->residual) & 7u) || d->ldr % 4)) return PT_ERR_ALIGN;
  if (d->geglu && (d->N % 64 != 0 || d->residual || d->seg_cols)) return PT_ERR_ARG;
  if (d->seg_cols) {
    if (d->seg_cols % 64 != 0 || d->N > 3 * d->seg_cols || !d->y2 || !d->t_dev || (d->N > 2 * d->seg_cols && !d->y3)) return PT_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(d->y2) & 7u) || d->ld2 % 4 || d->t_stride % 4 || (d->y3 && ((reinterpret_cast<uintptr_t>(d->y3) & 7u) || d->ld3 % 4))) return PT_ERR_ALIGN;
  }
  DlinParams p;
  p.M = (int)d->M; p.N = (int)d->N; p.K = (int)d->K;
  p.x = (const bf16_t*)d->x; p.ldx = d->ldx; p.gamma = d->ln_gamma; p.beta = d->ln_beta; p.eps = d->ln_eps;
  p.w = (const bf16_t*)d->w; p.ldw = d->ldw; p.bias = d->bias; p.residual = (const bf16_t*)d->residual; p.ldr = d->ldr;
  p.y = (bf16_t*)d->y; p.ldy = d->ldy; p.geglu = d->geglu ? 1 : 0;
  p.seg_cols = (int)d->seg_cols; p.y2 = (bf16_t*)d->y2; p.ld2 = d->ld2; p.y3 = (bf16_t*)d->y3; p.ld3 = d->ld3; p.t_dev = d->t_dev; p.t_stride = d->t_stride;
  const dim3 grid((unsigned)((d->N + 15) / 16));
  if (d->ln_gamma) {
    if (d->K == 256) hipLaunchKernelGGL(dlin_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(dlin_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, p);
  } else {
    hipLaunchKernelGGL(dlin_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, p);
  }
  PT_LAUNCH_CHECK();
  return PT_OK;
}

extern "C" int pt_ar_embed(const int64_t* prev, const void* emb, const void* pos, const int64_t* t_dev, void* out, int64_t B, int64_t n_q,
                           int64_t bins, int64_t dim, pt_stream stream) {
  if (B <= 0 || n_q <= 0 || bins <= 0 || dim <= 0 || dim % 8 != 0 || B * dim >= (1ll << 31)) return PT_ERR_SHAPE;
  if (!prev || !t_dev || !pt_aligned16(emb) || !pt_aligned16(pos) || !pt_aligned16(out)) return PT_ERR_ALIGN;
  int64_t blocks = (B * (dim / 8) + 255) / 256; if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(ar_embed_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, prev, (const bf16_t*)emb, (const bf16_t*)pos, t_dev,
                     (bf16_t*)out, (int)B, (int)n_q, (int)bins, (int)dim);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

extern "C" int pt_ar_advance(const int64_t* idx, int64_t* prev, int64_t* codes, int64_t* t_dev, int32_t* kv_len, int64_t B, int64_t n_q,
                             int64_t T, pt_stream stream) {
  if (B <= 0 || n_q <= 0 || T <= 0 || B * n_q >= (1ll << 31)) return PT_ERR_SHAPE;
  if (!idx || !prev || !codes || !t_dev || !kv_len) return PT_ERR_ARG;
  hipLaunchKernelGGL(ar_advance_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, idx, prev, codes, t_dev, kv_len, (int)B, (int)n_q, T);
  PT_LAUNCH_CHECK();
  return PT_OK;
}
