// Encodec decoder kernels that do not fit the 128x128 GEMM: RVQ gather-sum, row-streaming conv for few output
// channels (the 24 kHz end of the SEANet decoder: 16..64 channels, HBM-bound), and the 2-layer LSTM recurrence.
#include "mma.h"

namespace {

// ---- RVQ decode ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rvq_kernel(const int64_t* __restrict__ codes, const T* __restrict__ cb,
                                                  T* __restrict__ out, int64_t B, int nq, int64_t Tn, int bins, int dim) {
  constexpr int EPC = Vec16<T>::N;
  const int cpr = dim / EPC;
  const int64_t total = B * Tn * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t bt = i / cpr; const int c = (int)(i - bt * cpr) * EPC;
    const int64_t b = bt / Tn, t = bt - b * Tn;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    for (int q = 0; q < nq; ++q) {
      int64_t idx = codes[(b * nq + q) * Tn + t];
      idx = idx < 0 ? 0 : (idx >= bins ? bins - 1 : idx);      // validated on the host; clamp keeps the read in bounds
      Vec16<T> v = load16(cb + ((int64_t)q * bins + idx) * dim + c);
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] += v.get(e);
    }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.set(e, acc[e]);
    store16(out + bt * dim + c, o);
  }
}

// ---- RVQ encode: one stage of the nearest-codeword search (one wave per token) -------------------------------
__global__ __launch_bounds__(256) void rvq_search_kernel(const float* __restrict__ scores, const float* __restrict__ cb,
                                                         float* __restrict__ residual, int64_t* __restrict__ codes,
                                                         int64_t M, int64_t nq, int64_t Tn, int64_t q, int bins, int dim) {
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* sr = scores + m * bins;
  float best = -INFINITY; int bi = 0x7fffffff;
  for (int j = lane; j < bins; j += 64) {                   // ascending j: strict '>' keeps the first maximum of this lane
    const float v = sr[j];
    if (v > best) { best = v; bi = j; }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {                  // larger value wins; equal values: smaller index (torch.max)
    const float ov = __shfl_xor(best, off, 64); const int oi = __shfl_xor(bi, off, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (bi >= bins) bi = 0;                                    // every score NaN: defined output
  const int64_t b = m / Tn, t = m - b * Tn;
  if (lane == 0) codes[(b * nq + q) * Tn + t] = bi;
  for (int d = lane; d < dim; d += 64) residual[m * dim + d] -= cb[(int64_t)bi * dim + d];
}

// ---- row-streaming conv -----------------------------------------------------------------------------------------
struct RowConvParams {
  int64_t M; int n_rows;
  const char* x; int64_t ldx; int cin, taps, rowmap, elu_x, stride, n_in;
  const char* x2; int64_t ldx2; int cin2, elu_x2;
  const char* w; int64_t ldw; const float* bias; int N, act;
  char* y; int64_t ldy; int y_f32;
};

__device__ __forceinline__ float elu_f(float v) { return v < 0.f ? (__expf(v) - 1.f) : v; }

template <typename T> __device__ __forceinline__ void frag_elu(Frag<T>& f);
template <> __device__ __forceinline__ void frag_elu<bf16_t>(Frag<bf16_t>& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)elu_f((float)f.v[j]);
}
template <> __device__ __forceinline__ void frag_elu<float>(Frag<float>& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = elu_f(f.v[j]);
}

template <typename T, int NT, int KS>
__global__ __launch_bounds__(256) void rowconv_kernel(const RowConvParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
  // weights -> registers, once: B-operand fragment (col = output channel li of tile nt, k = 32*ks + 8g + j)
  Frag<T> wf[NT][KS];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = 16 * nt + li;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (n < p.N) frag_load_global(wf[nt][ks], reinterpret_cast<const T*>(p.w) + (int64_t)n * p.ldw + 32 * ks + 8 * g);
      else frag_zero(wf[nt][ks]);
    }
  }
  const int K1 = p.taps * p.cin;
  const int64_t nblk = (p.M + 63) / 64;
  for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < nblk; blk += (int64_t)gridDim.x * 4) {
    const int64_t r0 = blk * 64;
    f32x4_t acc[4][NT];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = (f32x4_t){0, 0, 0, 0};
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int64_t m = r0 + 16 * rt + li;
      const bool mok = m < p.M;
      const int64_t b = mok ? m / p.n_rows : 0; const int n = mok ? (int)(m - b * p.n_rows) : 0;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k0 = 32 * ks + 8 * g;
        Frag<T> fa;
        bool have = false; int elu = 0;
        if (mok && k0 < K1) {
          const int tap = k0 / p.cin, ci = k0 - tap * p.cin;
          int ns; bool ok;
          if (p.rowmap == PT_MAP_CAUSAL_REFLECT) { const int v = n + tap - (p.taps - 1); ns = v < 0 ? -v : v; ok = ns < p.n_rows; }
          else if (p.rowmap == PT_MAP_STRIDED_REFLECT) { const int v = n * p.stride + tap - (p.taps - p.stride); ns = v < 0 ? -v : v; ok = ns < p.n_in; }
          else { ns = n - tap; ok = ns >= 0; }                                   // PT_MAP_BACK
          if (ok) {
            frag_load_global(fa, reinterpret_cast<const T*>(p.x) + (b * p.n_in + ns) * p.ldx + ci);
            have = true; elu = p.elu_x;
          }
        } else if (mok && p.x2 && k0 - K1 < p.cin2) {
          frag_load_global(fa, reinterpret_cast<const T*>(p.x2) + m * p.ldx2 + (k0 - K1));
          have = true; elu = p.elu_x2;
        }
        if (!have) frag_zero(fa);
        if (elu) frag_elu<T>(fa);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) mma16(acc[rt][nt], wf[nt][ks], fa);        // D[row = n][col = m]
      }
    }
    // epilogue: lane holds, for row m = r0 + 16rt + li, channels n = 16nt + 4g + r
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int64_t m = r0 + 16 * rt + li;
      if (m >= p.M) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n0 = 16 * nt + 4 * g;
        if (n0 >= p.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + r;
          float x = acc[rt][nt][r] + ((p.bias && n < p.N) ? p.bias[n] : 0.f);
          v[r] = p.act == 1 ? elu_f(x) : x;
        }
        if (n0 + 3 < p.N) {
          if (p.y_f32) *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(p.y) + m * p.ldy + n0) = (f32x4_t){v[0], v[1], v[2], v[3]};
          else store4<T>(reinterpret_cast<T*>(p.y) + m * p.ldy + n0, v[0], v[1], v[2], v[3]);
        } else {
          for (int r = 0; r < 4 && n0 + r < p.N; ++r) {
            if (p.y_f32) reinterpret_cast<float*>(p.y)[m * p.ldy + n0 + r] = v[r];
            else reinterpret_cast<T*>(p.y)[m * p.ldy + n0 + r] = from_f32<T>(v[r]);
          }
        }
      }
    }
  }
}

template <typename T, int NT, int KS> int launch_rowconv(const RowConvParams& p, hipStream_t s) {
  int64_t blocks = (p.M + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((rowconv_kernel<T, NT, KS>), dim3((unsigned)blocks), dim3(256), 0, s, p);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

template <typename T> int dispatch_rowconv(const RowConvParams& p, int nt, int ks, hipStream_t s) {
#define RC(NT_, KS_) if (nt == NT_ && ks == KS_) return launch_rowconv<T, NT_, KS_>(p, s)
  RC(1, 3); RC(1, 7); RC(2, 2); RC(2, 6); RC(4, 3); RC(4, 4); RC(1, 1); RC(1, 2); RC(2, 3); RC(4, 2); RC(4, 6);
#undef RC
  return PT_ERR_SHAPE;
}

// ---- 2-layer LSTM step --------------------------------------------------------------------------------------------
struct LstmParams {
  int B, T, H;
  const char* x; const char* xg0; const char* whh0; const char* wcat1; const float* bias1;
  char* h0_seq; char* h1_seq; float* c0; float* c1; char* out_elu;
};

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

// launch s: blockIdx.y = layer, its time step t = s - layer; blockIdx.x = slice of 16 hidden units; blockIdx.z = 16 batch
// rows.  The four waves of a workgroup are the four gates (i, f, g, o) of those 16 units; each wave owns ONE 16 x 16
// accumulator tile.  The recurrence is latency-bound (T+1 dependent launches), so the k loop is unrolled by 8 with all
// fragment loads (L2-resident weights and h rows) issued ahead of the MFMAs, and the batch is spread over many small
// workgroups (H/16 * 2 * B/16 of them) instead of a few big ones.
template <typename T>
__global__ __launch_bounds__(256) void lstm2_step_kernel(const LstmParams p, int s) {
  const int layer = blockIdx.y, t = s - layer;
  if (t < 0 || t >= p.T) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
  const int H = p.H, j0 = blockIdx.x * 16, b0 = blockIdx.z * 16;
  __shared__ float sg[4][16][17];

  const T* h0 = reinterpret_cast<const T*>(p.h0_seq);
  const T* h1 = reinterpret_cast<const T*>(p.h1_seq);
  const int gcol = wave * H + j0 + li;      // row of the weight matrix = gate column (wave = gate i,f,g,o)
  f32x4_t acc;                              // natural layout: col = gate column li, rows = batch 4g + r
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + 4 * g + r;
    float init = 0.f;
    if (b < p.B) init = layer == 0 ? to_f32<T>(reinterpret_cast<const T*>(p.xg0)[((int64_t)b * p.T + t) * 4 * H + gcol]) : p.bias1[gcol];
    acc[r] = init;
  }
  const T* W = layer == 0 ? reinterpret_cast<const T*>(p.whh0) : reinterpret_cast<const T*>(p.wcat1);
  const int ldw = layer == 0 ? H : 2 * H;
  const int brow = b0 + li;
  const bool bok = brow < p.B;
  // source of the A operand for a k range: layer 0: h0_{t-1}; layer 1: [h0_t | h1_{t-1}]
  auto a_ptr = [&](int k0) -> const T* {
    const T* src; int tt, kk = k0;
    if (layer == 0) { src = h0; tt = t - 1; }
    else if (k0 < H) { src = h0; tt = t; }
    else { src = h1; tt = t - 1; kk = k0 - H; }
    return (tt >= 0 && bok) ? src + ((int64_t)brow * p.T + tt) * H + kk : nullptr;
  };
  constexpr int U = 8;
  for (int ks0 = 0; ks0 < ldw / 32; ks0 += U) {      // H multiple of 256 -> ldw/32 multiple of 8
    Frag<T> fw[U], fa[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k0 = 32 * (ks0 + u) + 8 * g;
      frag_load_global(fw[u], W + (int64_t)gcol * ldw + k0);
      const T* ap = a_ptr(k0);
      if (ap) frag_load_global(fa[u], ap); else frag_zero(fa[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) mma16(acc, fa[u], fw[u]);      // D[row = batch][col = gate column]
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) sg[wave][4 * g + r][li] = acc[r];
  __syncthreads();
  float* c = layer == 0 ? p.c0 : p.c1;
  T* hseq = reinterpret_cast<T*>(layer == 0 ? p.h0_seq : p.h1_seq);
  {
    const int bl = threadIdx.x >> 4, j = threadIdx.x & 15, b = b0 + bl;     // 256 threads = 16 batch rows x 16 units
    if (b < p.B) {
      const float ig = sigmoid_f(sg[0][bl][j]), fg = sigmoid_f(sg[1][bl][j]), gg = tanhf(sg[2][bl][j]), og = sigmoid_f(sg[3][bl][j]);
      const int64_t ci = (int64_t)b * H + j0 + j;
      const float cprev = t == 0 ? 0.f : c[ci];
      const float cn = fg * cprev + ig * gg;
      const float hn = og * tanhf(cn);
      c[ci] = cn;
      const int64_t oi = ((int64_t)b * p.T + t) * H + j0 + j;
      hseq[oi] = from_f32<T>(hn);
      if (layer == 1) {
        const float v = hn + to_f32<T>(reinterpret_cast<const T*>(p.x)[oi]);
        reinterpret_cast<T*>(p.out_elu)[oi] = from_f32<T>(v < 0.f ? (__expf(v) - 1.f) : v);
      }
    }
  }
}

}  // namespace

extern "C" int pt_rvq_decode(const int64_t* codes, const void* codebooks, void* out, int64_t B, int64_t n_q, int64_t T,
                             int64_t bins, int64_t dim, int dtype, pt_stream stream) {
  if (B <= 0 || n_q <= 0 || T <= 0 || bins <= 0 || dim <= 0 || dim % 8 != 0) return PT_ERR_SHAPE;
  if (!codes || !pt_aligned16(codebooks) || !pt_aligned16(out)) return PT_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  int64_t blocks = (B * T * (dim / 4) + 255) / 256; if (blocks > 4096) blocks = 4096;
  if (dtype == PT_F32) hipLaunchKernelGGL((rvq_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, codes, (const float*)codebooks, (float*)out, B, (int)n_q, T, (int)bins, (int)dim);
  else if (dtype == PT_BF16) hipLaunchKernelGGL((rvq_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, s, codes, (const bf16_t*)codebooks, (bf16_t*)out, B, (int)n_q, T, (int)bins, (int)dim);
  else return PT_ERR_DTYPE;
  PT_LAUNCH_CHECK();
  return PT_OK;
}

extern "C" int pt_rvq_search(const float* scores, const float* codebook, float* residual, int64_t* codes, int64_t B, int64_t n_q,
                             int64_t T, int64_t q, int64_t bins, int64_t dim, pt_stream stream) {
  if (B <= 0 || n_q <= 0 || T <= 0 || q < 0 || q >= n_q || bins <= 0 || bins > (1 << 30) || dim <= 0) return PT_ERR_SHAPE;
  if (!scores || !codebook || !residual || !codes) return PT_ERR_ARG;
  const int64_t M = B * T;
  hipLaunchKernelGGL(rvq_search_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, scores, codebook,
                     residual, codes, M, n_q, T, q, (int)bins, (int)dim);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

extern "C" int pt_rowconv(const pt_rowconv_desc* d, int dtype, pt_stream stream) {
  if (!d) return PT_ERR_ARG;
  if (dtype != PT_F32 && dtype != PT_BF16) return PT_ERR_DTYPE;
  const int es = dtype == PT_F32 ? 4 : 2;
  if (d->B <= 0 || d->n_rows <= 0 || d->N <= 0 || d->N > 64 || d->cin <= 0 || d->cin % 8 != 0 || d->taps < 1) return PT_ERR_SHAPE;
  if (d->rowmap != PT_MAP_CAUSAL_REFLECT && d->rowmap != PT_MAP_BACK && d->rowmap != PT_MAP_STRIDED_REFLECT) return PT_ERR_ARG;
  if (d->rowmap == PT_MAP_STRIDED_REFLECT && (d->stride < 1 || d->taps < d->stride || d->n_rows * d->stride < d->taps)) return PT_ERR_SHAPE;
  if (d->rowmap == PT_MAP_CAUSAL_REFLECT && d->n_rows < d->taps) return PT_ERR_SHAPE;
  if (d->x2 && (d->cin2 <= 0 || d->cin2 % 8 != 0)) return PT_ERR_SHAPE;
  const int64_t K = (int64_t)d->taps * d->cin + (d->x2 ? d->cin2 : 0);
  if (d->ldw % 32 != 0 || d->ldw < K) return PT_ERR_SHAPE;
  if (!pt_aligned16(d->x) || (d->ldx * es) % 16 || !pt_aligned16(d->w) || (d->x2 && (!pt_aligned16(d->x2) || (d->ldx2 * es) % 16))) return PT_ERR_ALIGN;
  if (!d->y || (reinterpret_cast<uintptr_t>(d->y) & 15u)) return PT_ERR_ALIGN;
  if (d->N >= 4 && d->ldy % 4 != 0) return PT_ERR_ALIGN;
  RowConvParams p;
  p.M = d->B * d->n_rows; p.n_rows = (int)d->n_rows;
  p.x = (const char*)d->x; p.ldx = d->ldx; p.cin = d->cin; p.taps = d->taps; p.rowmap = d->rowmap; p.elu_x = d->elu_x;
  p.stride = d->rowmap == PT_MAP_STRIDED_REFLECT ? d->stride : 1; p.n_in = p.n_rows * p.stride;
  p.x2 = (const char*)d->x2; p.ldx2 = d->ldx2; p.cin2 = d->x2 ? d->cin2 : 0; p.elu_x2 = d->elu_x2;
  p.w = (const char*)d->w; p.ldw = d->ldw; p.bias = d->bias; p.N = d->N; p.act = d->act;
  p.y = (char*)d->y; p.ldy = d->ldy; p.y_f32 = d->y_f32;
  const int nt = d->N <= 16 ? 1 : (d->N <= 32 ? 2 : 4), ks = (int)(d->ldw / 32);
  hipStream_t s = (hipStream_t)stream;
  return dtype == PT_F32 ? dispatch_rowconv<float>(p, nt, ks, s) : dispatch_rowconv<bf16_t>(p, nt, ks, s);
}

extern "C" int pt_lstm2_forward(const pt_lstm2_desc* d, int dtype, pt_stream stream) {
  if (!d) return PT_ERR_ARG;
  if (dtype != PT_F32 && dtype != PT_BF16) return PT_ERR_DTYPE;
  if (d->B <= 0 || d->T <= 0 || d->H <= 0 || d->H % 256 != 0 || d->B > 16 * 65535) return PT_ERR_SHAPE;
  if (!d->x || !d->xg0 || !d->whh0 || !d->wcat1 || !d->bias1 || !d->h0_seq || !d->h1_seq || !d->c0 || !d->c1 || !d->out_elu) return PT_ERR_ARG;
  const void* ptrs[] = {d->x, d->xg0, d->whh0, d->wcat1, d->h0_seq, d->h1_seq, d->out_elu};
  for (const void* q : ptrs) if (!pt_aligned16(q)) return PT_ERR_ALIGN;
  LstmParams p;
  p.B = (int)d->B; p.T = (int)d->T; p.H = (int)d->H;
  p.x = (const char*)d->x; p.xg0 = (const char*)d->xg0; p.whh0 = (const char*)d->whh0; p.wcat1 = (const char*)d->wcat1;
  p.bias1 = d->bias1; p.h0_seq = (char*)d->h0_seq; p.h1_seq = (char*)d->h1_seq; p.c0 = d->c0; p.c1 = d->c1; p.out_elu = (char*)d->out_elu;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(d->H / 16), 2, (unsigned)((d->B + 15) / 16));
  for (int step = 0; step <= (int)d->T; ++step) {
    if (dtype == PT_F32) hipLaunchKernelGGL((lstm2_step_kernel<float>), grid, dim3(256), 0, s, p, step);
    else hipLaunchKernelGGL((lstm2_step_kernel<bf16_t>), grid, dim3(256), 0, s, p, step);
  }
  PT_LAUNCH_CHECK();
  return PT_OK;
}
