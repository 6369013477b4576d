// Elementwise / gather / reduction kernels of the denoiser training step.  All HBM-bound; 16-byte accesses,
// grid capped at ~2048 blocks with grid-stride loops.
#include "common.h"

namespace {

constexpr int NT = 256;
inline unsigned grid_for(int64_t work_items) {
  int64_t b = (work_items + NT - 1) / NT;
  if (b < 1) b = 1;
  return (unsigned)(b > 4096 ? 4096 : b);
}

// ---- GEGLU --------------------------------------------------------------------------------------------
// IL = interleaved projection columns (the layout the fused GEMM epilogues use, pt_gemm act 2 / 3): value of act column
// 32 q + t at column 64 q + t, its gate at 64 q + 32 + t.  This kernel pair is the fallback for row counts the fused
// epilogues do not take (M not a multiple of 256) and the f32 parity mode; `bias` (original order, may be NULL) is added to
// the projection in place, because a GEMM over the interleaved weight cannot add it in its own epilogue.
template <bool IL> __device__ __forceinline__ int64_t geglu_col(int64_t c, int64_t F, bool gate) {
  if (!IL) return gate ? F + c : c;
  return 64 * (c >> 5) + (c & 31) + (gate ? 32 : 0);
}
template <typename T, bool IL>
__global__ __launch_bounds__(NT) void geglu_fwd_kernel(T* __restrict__ proj, const float* __restrict__ bias, T* __restrict__ out,
                                                       int64_t M, int64_t F) {
  constexpr int EPC = Vec16<T>::N;
  const int64_t cpr = F / EPC, total = M * cpr;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    const int64_t m = i / cpr, c = (i - m * cpr) * EPC;
    T* hp = proj + m * 2 * F + geglu_col<IL>(c, F, false); T* gp = proj + m * 2 * F + geglu_col<IL>(c, F, true);
    Vec16<T> h = load16(hp), g = load16(gp), o;
    if (bias) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) { h.set(e, h.get(e) + bias[c + e]); g.set(e, g.get(e) + bias[F + c + e]); }
      store16(hp, h); store16(gp, g);
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.set(e, h.get(e) * gelu_erf_f(g.get(e)));
    store16(out + m * F + c, o);
  }
}
template <typename T, bool IL>
__global__ __launch_bounds__(NT) void geglu_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ proj,
                                                       T* __restrict__ dproj, int64_t M, int64_t F) {
  constexpr int EPC = Vec16<T>::N;
  const int64_t cpr = F / EPC, total = M * cpr;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    const int64_t m = i / cpr, c = (i - m * cpr) * EPC;
    const int64_t hc = m * 2 * F + geglu_col<IL>(c, F, false), gc = m * 2 * F + geglu_col<IL>(c, F, true);
    Vec16<T> h = load16(proj + hc), g = load16(proj + gc), d = load16(dout + m * F + c), dh, dg;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float gv = g.get(e), dv = d.get(e);
      dh.set(e, dv * gelu_erf_f(gv));
      dg.set(e, dv * h.get(e) * gelu_erf_grad_f(gv));
    }
    store16(dproj + hc, dh);
    store16(dproj + gc, dg);
  }
}

// ---- DDPM ancestral step (diffusers DDPMScheduler.step, epsilon prediction, fixed_small variance, clip_sample) -----------
// x_prev = c_x0 * clamp((x - c_eps * eps) * c_inv, -clip, clip) + c_xt * x + sigma * z      (all f32, n elements)
__global__ __launch_bounds__(NT) void ddpm_step_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                       const float* __restrict__ z, float* __restrict__ out, int64_t n,
                                                       float c_eps, float c_inv, float clip, float c_x0, float c_xt, float sigma) {
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    const float xv = x[i];
    float x0 = (xv - c_eps * eps[i]) * c_inv;
    if (clip > 0.f) x0 = fminf(fmaxf(x0, -clip), clip);
    float r = c_x0 * x0 + c_xt * xv;
    if (z) r += sigma * z[i];
    out[i] = r;
  }
}

// ---- dropout (torch.nn.Dropout in the text encoder's BasicTransformerBlocks, tts/models.py:95-100): y = x * mask * scale
// [+ residual]; the keep mask (1 byte per element) is drawn by the host side (device RNG or injected), so forward and
// backward -- the same kernel on dy -- see the same mask.  n multiple of the 16-byte vector.
template <typename T>
__global__ __launch_bounds__(NT) void dropout_kernel(const T* __restrict__ x, const uint8_t* __restrict__ mask, const T* __restrict__ res,
                                                     T* __restrict__ y, int64_t n, float scale) {
  constexpr int EPC = Vec16<T>::N;
  const int64_t nv = n / EPC;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < nv; i += (int64_t)gridDim.x * NT) {
    Vec16<T> v = load16(x + i * EPC), r, o;
    if (res) r = load16(res + i * EPC);
    uint8_t m[EPC];
    if (EPC == 8) *reinterpret_cast<uint64_t*>(m) = *reinterpret_cast<const uint64_t*>(mask + i * EPC);
    else *reinterpret_cast<uint32_t*>(m) = *reinterpret_cast<const uint32_t*>(mask + i * EPC);
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.set(e, (m[e] ? v.get(e) * scale : 0.f) + (res ? r.get(e) : 0.f));
    store16(y + i * EPC, o);
  }
}

// ---- flat unary/binary ------------------------------------------------------------------------------------
template <typename T, int OP>   // 0 silu fwd (a=x) ; 1 silu bwd (a=dy, b=x) ; 2 add
__global__ __launch_bounds__(NT) void flat_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, int64_t n) {
  constexpr int EPC = Vec16<T>::N;
  const int64_t nv = n / EPC;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < nv; i += (int64_t)gridDim.x * NT) {
    Vec16<T> va = load16(a + i * EPC), vb, o;
    if (OP != 0) vb = load16(b + i * EPC);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float r;
      if (OP == 0) r = silu_f(va.get(e));
      else if (OP == 1) r = va.get(e) * silu_grad_f(vb.get(e));
      else r = va.get(e) + vb.get(e);
      o.set(e, r);
    }
    store16(y + i * EPC, o);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int64_t i = nv * EPC; i < n; ++i) {
      const float x = to_f32<T>(a[i]);
      float r;
      if (OP == 0) r = silu_f(x);
      else if (OP == 1) r = x * silu_grad_f(to_f32<T>(b[i]));
      else r = x + to_f32<T>(b[i]);
      y[i] = from_f32<T>(r);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void pairsum_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t rows, int64_t C) {
  constexpr int EPC = Vec16<T>::N;
  const int64_t cpr = C / EPC, total = rows * cpr;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    const int64_t r = i / cpr, c = (i - r * cpr) * EPC;
    Vec16<T> a = load16(x + (2 * r) * C + c), b = load16(x + (2 * r + 1) * C + c), o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.set(e, a.get(e) + b.get(e));
    store16(y + r * C + c, o);
  }
}

// out[seg][n] += sum over the segment's rows of dy[m][n]: block = run of rows inside ONE segment, thread = fixed chunk.
// Float atomics execute at the memory side and every block adding into ONE short row is ~14x below the chip-wide atomic
// rate (measured: 12 of this kernel's 20 us at 32768 x 512): block b therefore adds into replica (b % n_rep) of the
// destination, `rep_stride` floats apart, and pt_fold_replicas sums the replicas once per step.
template <typename T>
__global__ __launch_bounds__(NT) void colsum_kernel(const T* __restrict__ dy, int64_t ld, float* __restrict__ out, int64_t ld_out,
                                                    int64_t seg_rows, int N, int rows_per_block, int n_rep, int64_t rep_stride) {
  constexpr int EPC = Vec16<T>::N;
  const int CC = (N + EPC - 1) / EPC;
  const int CW = CC < NT ? CC : NT, RP = NT / CW;
  const int cw = threadIdx.x % CW, rr = threadIdx.x / CW;
  const int64_t seg = blockIdx.y;
  const int64_t s0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t s1 = s0 + rows_per_block < seg_rows ? s0 + rows_per_block : seg_rows;
  extern __shared__ float sh[];                 // [RP][CC * EPC] partial sums of the row-lanes
  const int NP = CC * EPC;
  if (rr < RP) {
    for (int c = cw; c < CC; c += CW) {
      float acc[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
#pragma unroll 8
      for (int64_t r = s0 + rr; r < s1; r += RP) {
        Vec16<T> v = load16(dy + (seg * seg_rows + r) * ld + (int64_t)c * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += v.get(e);
      }
#pragma unroll
      for (int e = 0; e < EPC; ++e) sh[rr * NP + c * EPC + e] = acc[e];
    }
  }
  __syncthreads();
  float* dst = out + (int64_t)(blockIdx.x % n_rep) * rep_stride + seg * ld_out;
  for (int i = threadIdx.x; i < N; i += NT) {
    float a = 0.f;
    for (int k = 0; k < RP; ++k) a += sh[k * NP + i];
#ifndef PT_DIAG_NO_COLATOMICS
    unsafeAtomicAdd(dst + i, a);
#endif
  }
}

// dst[seg.dst_off + i] += sum_r arena[seg.rep_off + r * seg.rep_stride + i]: one launch per step folds every replicated
// small gradient (biases, norm scales / shifts) into the flat gradient buffer.
__global__ __launch_bounds__(NT) void fold_replicas_kernel(const float* __restrict__ arena, float* __restrict__ dst,
                                                           const pt_fold_seg* __restrict__ segs, int n_rep) {
  const pt_fold_seg sg = segs[blockIdx.y];
  for (int i = blockIdx.x * NT + threadIdx.x; i < sg.n; i += gridDim.x * NT) {
    float a = 0.f;
    for (int r = 0; r < n_rep; ++r) a += arena[sg.rep_off + (int64_t)r * sg.rep_stride + i];
    dst[sg.dst_off + i] += a;
  }
}

// ---- embedding ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void embedding_fwd_kernel(const int32_t* __restrict__ ids, const T* __restrict__ W,
                                                           const float* __restrict__ pos, T* __restrict__ out,
                                                           int64_t BS, int64_t S, int64_t d, int64_t vocab) {
  constexpr int EPC = Vec16<T>::N;
  const int64_t cpr = d / EPC, total = BS * cpr;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    const int64_t r = i / cpr, c = (i - r * cpr) * EPC;
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);      // ids are validated on the host; clamp keeps the read in bounds
    const int64_t s = r % S;
    Vec16<T> w = load16(W + id * d + c), o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.set(e, w.get(e) + pos[s * d + c + e]);
    store16(out + r * d + c, o);
  }
}
template <typename T>
__global__ __launch_bounds__(NT) void embedding_bwd_kernel(const int32_t* __restrict__ ids, const T* __restrict__ dout,
                                                           float* __restrict__ dW, int64_t BS, int64_t d, int64_t vocab) {
  const int64_t total = BS * d;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    const int64_t r = i / d, c = i - r * d;
    const int64_t id = ids[r];
    if (id >= 0 && id < vocab) unsafeAtomicAdd(dW + id * d + c, to_f32<T>(dout[i]));
  }
}

template <typename T>
__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, T* __restrict__ out, int64_t B, int C, int flip, float shift) {
  const int half = C / 2;
  const int64_t total = B * half;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / half; const int k = (int)(i - b * half);
    const float freq = expf(-9.210340371976184f * (float)k / ((float)half - shift));   // ln(10000)
    const float ang = (float)t[b] * freq;
    const float sn = sinf(ang), cs = cosf(ang);
    T* o = out + b * C;
    if (flip) { o[k] = from_f32<T>(cs); o[half + k] = from_f32<T>(sn); }
    else      { o[k] = from_f32<T>(sn); o[half + k] = from_f32<T>(cs); }
    if ((C & 1) && k == 0) o[C - 1] = from_f32<T>(0.f);
  }
}

// ---- training-step ends -----------------------------------------------------------------------------------
template <typename T>
__global__ void add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const int64_t* __restrict__ t,
                                 const float* __restrict__ ac, T* __restrict__ xt, int64_t B, int nq, int64_t Tn, int cpad) {
  const int64_t total = B * Tn * cpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpad); const int64_t bt = i / cpad; const int64_t b = bt / Tn, n = bt - b * Tn;
    float v = 0.f;
    if (c < nq) {
      const float a = ac[t[b]];
      const int64_t src = (b * nq + c) * Tn + n;
      v = sqrtf(a) * x0[src] + sqrtf(1.f - a) * noise[src];
    }
    xt[i] = from_f32<T>(v);
  }
}
template <typename T>
__global__ void tokens_to_bct_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t B, int nq, int64_t Tn, int cpad) {
  const int64_t total = B * nq * Tn;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i % Tn; const int64_t bc = i / Tn; const int64_t b = bc / nq; const int c = (int)(bc - b * nq);
    out[i] = to_f32<T>(x[(b * Tn + n) * cpad + c]);
  }
}
template <typename T>
__global__ void bct_to_tokens_kernel(const float* __restrict__ x, T* __restrict__ out, int64_t B, int nq, int64_t Tn, int cpad) {
  const int64_t total = B * Tn * cpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpad); const int64_t bt = i / cpad; const int64_t b = bt / Tn, n = bt - b * Tn;
    out[i] = from_f32<T>(c < nq ? x[(b * nq + c) * Tn + n] : 0.f);
  }
}
template <typename T>
__global__ __launch_bounds__(NT) void mse_kernel(const T* __restrict__ pred, const float* __restrict__ noise, float* __restrict__ loss,
                                                 T* __restrict__ dpred, float gscale, int64_t B, int nq, int64_t Tn, int cpad) {
  const int64_t total = B * Tn * cpad;
  const float inv = 1.f / (float)(B * nq * Tn);
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    const int c = (int)(i % cpad); const int64_t bt = i / cpad; const int64_t b = bt / Tn, n = bt - b * Tn;
    float g = 0.f;
    if (c < nq) {
      const float d = to_f32<T>(pred[i]) - noise[(b * nq + c) * Tn + n];
      acc += d * d;
      g = 2.f * d * inv * gscale;
    }
    if (dpred) dpred[i] = from_f32<T>(g);
  }
  __shared__ float sc[NT / 64];
  acc = block_sum<NT>(acc, sc);
  if (threadIdx.x == 0) unsafeAtomicAdd(loss, acc * inv);
}

__global__ __launch_bounds__(NT) void sumsq_kernel(const float* __restrict__ g, float* __restrict__ out, int64_t n) {
  const int64_t nv = n / 4;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < nv; i += (int64_t)gridDim.x * NT) {
    f32x4_t v = *reinterpret_cast<const f32x4_t*>(g + i * 4);
    acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t i = nv * 4; i < n; ++i) acc += g[i] * g[i];
  __shared__ float sc[NT / 64];
  acc = block_sum<NT>(acc, sc);
  if (threadIdx.x == 0) unsafeAtomicAdd(out, acc);
}

}  // namespace

#define PT_DISPATCH(dtype, CALL_F32, CALL_BF16)          \
  do {                                                   \
    if ((dtype) == PT_F32) { CALL_F32; }                 \
    else if ((dtype) == PT_BF16) { CALL_BF16; }          \
    else return PT_ERR_DTYPE;                            \
    PT_LAUNCH_CHECK();                                   \
    return PT_OK;                                        \
  } while (0)

extern "C" int pt_ddpm_step(const float* x, const float* eps, const float* z, float* out, int64_t n, float c_eps, float c_inv,
                            float clip, float c_x0, float c_xt, float sigma, pt_stream stream) {
  if (!x || !eps || !out || n <= 0) return PT_ERR_ARG;
  hipLaunchKernelGGL(ddpm_step_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, x, eps, z, out, n, c_eps, c_inv, clip,
                     c_x0, c_xt, sigma);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

extern "C" int pt_geglu_fwd(void* proj, const float* bias, void* out, int64_t M, int64_t F, int interleaved, int dtype, pt_stream stream) {
  if (M <= 0 || F <= 0 || F % 8 != 0 || (interleaved && F % 32 != 0)) return PT_ERR_SHAPE;
  if (!pt_aligned16(proj) || !pt_aligned16(out)) return PT_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (interleaved) {
    PT_DISPATCH(dtype,
                hipLaunchKernelGGL((geglu_fwd_kernel<float, true>), dim3(grid_for(M * F / 4)), dim3(NT), 0, s, (float*)proj, bias, (float*)out, M, F),
                hipLaunchKernelGGL((geglu_fwd_kernel<bf16_t, true>), dim3(grid_for(M * F / 8)), dim3(NT), 0, s, (bf16_t*)proj, bias, (bf16_t*)out, M, F));
  }
  PT_DISPATCH(dtype,
              hipLaunchKernelGGL((geglu_fwd_kernel<float, false>), dim3(grid_for(M * F / 4)), dim3(NT), 0, s, (float*)proj, bias, (float*)out, M, F),
              hipLaunchKernelGGL((geglu_fwd_kernel<bf16_t, false>), dim3(grid_for(M * F / 8)), dim3(NT), 0, s, (bf16_t*)proj, bias, (bf16_t*)out, M, F));
}
extern "C" int pt_geglu_bwd(const void* dout, const void* proj, void* dproj, int64_t M, int64_t F, int interleaved, int dtype, pt_stream stream) {
  if (M <= 0 || F <= 0 || F % 8 != 0 || (interleaved && F % 32 != 0)) return PT_ERR_SHAPE;
  if (!pt_aligned16(proj) || !pt_aligned16(dout) || !pt_aligned16(dproj)) return PT_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (interleaved) {
    PT_DISPATCH(dtype,
                hipLaunchKernelGGL((geglu_bwd_kernel<float, true>), dim3(grid_for(M * F / 4)), dim3(NT), 0, s, (const float*)dout, (const float*)proj, (float*)dproj, M, F),
                hipLaunchKernelGGL((geglu_bwd_kernel<bf16_t, true>), dim3(grid_for(M * F / 8)), dim3(NT), 0, s, (const bf16_t*)dout, (const bf16_t*)proj, (bf16_t*)dproj, M, F));
  }
  PT_DISPATCH(dtype,
              hipLaunchKernelGGL((geglu_bwd_kernel<float, false>), dim3(grid_for(M * F / 4)), dim3(NT), 0, s, (const float*)dout, (const float*)proj, (float*)dproj, M, F),
              hipLaunchKernelGGL((geglu_bwd_kernel<bf16_t, false>), dim3(grid_for(M * F / 8)), dim3(NT), 0, s, (const bf16_t*)dout, (const bf16_t*)proj, (bf16_t*)dproj, M, F));
}

template <int OP> static int flat_launch(const void* a, const void* b, void* y, int64_t n, int dtype, pt_stream stream) {
  if (n <= 0) return PT_ERR_SHAPE;
  if (!pt_aligned16(a) || !pt_aligned16(y) || (b && !pt_aligned16(b))) return PT_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  PT_DISPATCH(dtype,
              hipLaunchKernelGGL((flat_kernel<float, OP>), dim3(grid_for(n / 4 + 1)), dim3(NT), 0, s, (const float*)a, (const float*)b, (float*)y, n),
              hipLaunchKernelGGL((flat_kernel<bf16_t, OP>), dim3(grid_for(n / 8 + 1)), dim3(NT), 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n));
}
extern "C" int pt_silu_fwd(const void* x, void* y, int64_t n, int dtype, pt_stream stream) { return flat_launch<0>(x, nullptr, y, n, dtype, stream); }
extern "C" int pt_silu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype, pt_stream stream) {
  if (!x) return PT_ERR_ARG;
  return flat_launch<1>(dy, x, dx, n, dtype, stream);
}
extern "C" int pt_add(const void* a, const void* b, void* y, int64_t n, int dtype, pt_stream stream) {
  if (!b) return PT_ERR_ARG;
  return flat_launch<2>(a, b, y, n, dtype, stream);
}

extern "C" int pt_dropout(const void* x, const uint8_t* mask, const void* residual, void* y, int64_t n, float scale, int dtype,
                          pt_stream stream) {
  const int epc = dtype == PT_F32 ? 4 : 8;
  if (n <= 0 || n % epc != 0) return PT_ERR_SHAPE;
  if (!x || !mask || !y) return PT_ERR_ARG;
  if (!pt_aligned16(x) || !pt_aligned16(y) || (residual && !pt_aligned16(residual)) || (reinterpret_cast<uintptr_t>(mask) & 7u)) return PT_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  PT_DISPATCH(dtype,
              hipLaunchKernelGGL((dropout_kernel<float>), dim3(grid_for(n / 4)), dim3(NT), 0, s, (const float*)x, mask, (const float*)residual, (float*)y, n, scale),
              hipLaunchKernelGGL((dropout_kernel<bf16_t>), dim3(grid_for(n / 8)), dim3(NT), 0, s, (const bf16_t*)x, mask, (const bf16_t*)residual, (bf16_t*)y, n, scale));
}

extern "C" int pt_pairsum_rows(const void* x, void* y, int64_t rows, int64_t C, int dtype, pt_stream stream) {
  if (rows <= 0 || C <= 0 || C % 8 != 0) return PT_ERR_SHAPE;
  if (!pt_aligned16(x) || !pt_aligned16(y)) return PT_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  PT_DISPATCH(dtype,
              hipLaunchKernelGGL((pairsum_kernel<float>), dim3(grid_for(rows * C / 4)), dim3(NT), 0, s, (const float*)x, (float*)y, rows, C),
              hipLaunchKernelGGL((pairsum_kernel<bf16_t>), dim3(grid_for(rows * C / 8)), dim3(NT), 0, s, (const bf16_t*)x, (bf16_t*)y, rows, C));
}

extern "C" int pt_fold_replicas(const float* arena, float* dst, const pt_fold_seg* segs_dev, int64_t n_segs, int n_rep,
                                int64_t max_n, pt_stream stream) {
  if (!arena || !dst || !segs_dev || n_segs <= 0 || n_segs > 65535 || n_rep <= 0 || max_n <= 0) return PT_ERR_ARG;
  const int64_t bx = (max_n + NT - 1) / NT;
  dim3 grid((unsigned)(bx < 16 ? bx : 16), (unsigned)n_segs);
  hipLaunchKernelGGL(fold_replicas_kernel, grid, dim3(NT), 0, (hipStream_t)stream, arena, dst, segs_dev, n_rep);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

extern "C" int pt_colsum(const void* dy, int64_t ld, float* out, int64_t ld_out, int64_t M, int64_t N, int64_t seg_rows,
                         int n_rep, int64_t rep_stride, int dtype, pt_stream stream) {
  if (ld_out <= 0) ld_out = N;
  if (n_rep < 1 || (n_rep > 1 && rep_stride <= 0)) return PT_ERR_ARG;
  if (M <= 0 || N <= 0 || N > 16384 || seg_rows <= 0 || M % seg_rows != 0 || M / seg_rows > 65535) return PT_ERR_SHAPE;
  const int es = dtype == PT_F32 ? 4 : 2, epc = 16 / es;
  if (!pt_aligned16(dy) || (ld * es) % 16 != 0) return PT_ERR_ALIGN;
  if (ld < (N + epc - 1) / epc * epc) return PT_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int rpb = 64;
  dim3 grid((unsigned)((seg_rows + rpb - 1) / rpb), (unsigned)(M / seg_rows));
  const int cc = (int)((N + epc - 1) / epc), cw = cc < NT ? cc : NT, rp = NT / cw;
  const size_t dyn = sizeof(float) * (size_t)rp * cc * epc;
  if (dyn > 64 * 1024) return PT_ERR_SHAPE;
  PT_DISPATCH(dtype,
              hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(NT), dyn, s, (const float*)dy, ld, out, ld_out, seg_rows, (int)N, rpb, n_rep, rep_stride),
              hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, dim3(NT), dyn, s, (const bf16_t*)dy, ld, out, ld_out, seg_rows, (int)N, rpb, n_rep, rep_stride));
}

extern "C" int pt_embedding_fwd(const int32_t* ids, const void* W, const float* pos, void* out, int64_t BS, int64_t S,
                                int64_t d, int64_t vocab, int dtype, pt_stream stream) {
  if (BS <= 0 || S <= 0 || d <= 0 || d % 8 != 0 || vocab <= 0 || BS % S != 0) return PT_ERR_SHAPE;
  if (!pt_aligned16(W) || !pt_aligned16(out)) return PT_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  PT_DISPATCH(dtype,
              hipLaunchKernelGGL((embedding_fwd_kernel<float>), dim3(grid_for(BS * d / 4)), dim3(NT), 0, s, ids, (const float*)W, pos, (float*)out, BS, S, d, vocab),
              hipLaunchKernelGGL((embedding_fwd_kernel<bf16_t>), dim3(grid_for(BS * d / 8)), dim3(NT), 0, s, ids, (const bf16_t*)W, pos, (bf16_t*)out, BS, S, d, vocab));
}
extern "C" int pt_embedding_bwd(const int32_t* ids, const void* dout, float* dW, int64_t BS, int64_t d, int64_t vocab,
                                int dtype, pt_stream stream) {
  if (BS <= 0 || d <= 0 || vocab <= 0) return PT_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  PT_DISPATCH(dtype,
              hipLaunchKernelGGL((embedding_bwd_kernel<float>), dim3(grid_for(BS * d)), dim3(NT), 0, s, ids, (const float*)dout, dW, BS, d, vocab),
              hipLaunchKernelGGL((embedding_bwd_kernel<bf16_t>), dim3(grid_for(BS * d)), dim3(NT), 0, s, ids, (const bf16_t*)dout, dW, BS, d, vocab));
}

extern "C" int pt_timestep_embedding(const int64_t* t, void* out, int64_t B, int64_t C, int flip, float shift, int dtype, pt_stream stream) {
  if (B <= 0 || C < 2) return PT_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  PT_DISPATCH(dtype,
              hipLaunchKernelGGL((timestep_embedding_kernel<float>), dim3(grid_for(B * C / 2)), dim3(NT), 0, s, t, (float*)out, B, (int)C, flip, shift),
              hipLaunchKernelGGL((timestep_embedding_kernel<bf16_t>), dim3(grid_for(B * C / 2)), dim3(NT), 0, s, t, (bf16_t*)out, B, (int)C, flip, shift));
}

extern "C" int pt_add_noise(const float* x0, const float* noise, const int64_t* t, const float* ac, void* xt, int64_t B,
                            int64_t n_q, int64_t T, int64_t cpad, int dtype, pt_stream stream) {
  if (B <= 0 || n_q <= 0 || T <= 0 || cpad < n_q) return PT_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  PT_DISPATCH(dtype,
              hipLaunchKernelGGL((add_noise_kernel<float>), dim3(grid_for(B * T * cpad)), dim3(NT), 0, s, x0, noise, t, ac, (float*)xt, B, (int)n_q, T, (int)cpad),
              hipLaunchKernelGGL((add_noise_kernel<bf16_t>), dim3(grid_for(B * T * cpad)), dim3(NT), 0, s, x0, noise, t, ac, (bf16_t*)xt, B, (int)n_q, T, (int)cpad));
}
extern "C" int pt_tokens_to_bct(const void* x, float* out, int64_t B, int64_t n_q, int64_t T, int64_t cpad, int dtype, pt_stream stream) {
  if (B <= 0 || n_q <= 0 || T <= 0 || cpad < n_q) return PT_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  PT_DISPATCH(dtype,
              hipLaunchKernelGGL((tokens_to_bct_kernel<float>), dim3(grid_for(B * T * n_q)), dim3(NT), 0, s, (const float*)x, out, B, (int)n_q, T, (int)cpad),
              hipLaunchKernelGGL((tokens_to_bct_kernel<bf16_t>), dim3(grid_for(B * T * n_q)), dim3(NT), 0, s, (const bf16_t*)x, out, B, (int)n_q, T, (int)cpad));
}
extern "C" int pt_bct_to_tokens(const float* x, void* out, int64_t B, int64_t n_q, int64_t T, int64_t cpad, int dtype, pt_stream stream) {
  if (B <= 0 || n_q <= 0 || T <= 0 || cpad < n_q) return PT_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  PT_DISPATCH(dtype,
              hipLaunchKernelGGL((bct_to_tokens_kernel<float>), dim3(grid_for(B * T * cpad)), dim3(NT), 0, s, x, (float*)out, B, (int)n_q, T, (int)cpad),
              hipLaunchKernelGGL((bct_to_tokens_kernel<bf16_t>), dim3(grid_for(B * T * cpad)), dim3(NT), 0, s, x, (bf16_t*)out, B, (int)n_q, T, (int)cpad));
}
extern "C" int pt_mse_loss(const void* pred, const float* noise, float* loss, void* dpred, float gscale, int64_t B,
                           int64_t n_q, int64_t T, int64_t cpad, int dtype, pt_stream stream) {
  if (B <= 0 || n_q <= 0 || T <= 0 || cpad < n_q) return PT_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  PT_DISPATCH(dtype,
              hipLaunchKernelGGL((mse_kernel<float>), dim3(grid_for(B * T * cpad)), dim3(NT), 0, s, (const float*)pred, noise, loss, (float*)dpred, gscale, B, (int)n_q, T, (int)cpad),
              hipLaunchKernelGGL((mse_kernel<bf16_t>), dim3(grid_for(B * T * cpad)), dim3(NT), 0, s, (const bf16_t*)pred, noise, loss, (bf16_t*)dpred, gscale, B, (int)n_q, T, (int)cpad));
}
extern "C" int pt_sumsq(const float* g, float* out, int64_t n, pt_stream stream) {
  if (n <= 0) return PT_ERR_SHAPE;
  if (!pt_aligned16(g)) return PT_ERR_ALIGN;
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n / 4 + 1)), dim3(NT), 0, (hipStream_t)stream, g, out, n);
  PT_LAUNCH_CHECK();
  return PT_OK;
}
