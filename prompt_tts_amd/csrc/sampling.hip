// Ops BASELINE.json's north_star names that the reference never implemented (SURVEY 8a'): greedy / top-k sampling
// over RVQ-codebook logits (wavefront shuffle top-k) and the continuous -> code-index rounding that inverts the collate
// normalisation.  Build-defined semantics, pinned to torch.argmax / torch.topk / numpy in the tests.
#include "common.h"

namespace {

// idx = clamp(rint((x + 1) / 2 * 1023), 0, 1023)   (inverse of (code/1023 - 0.5)/0.5, tts/dataloader.py:64,143)
__global__ void codes_kernel(const float* __restrict__ x, int64_t* __restrict__ codes, int64_t n, int bins) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = rintf((x[i] + 1.0f) * 0.5f * (float)(bins - 1));
    v = fminf(fmaxf(v, 0.f), (float)(bins - 1));
    codes[i] = (int64_t)v;
  }
}

// One wave per row of logits[R][V] (V <= 64*PER).  k == 1: argmax (lowest index wins ties, as torch.argmax).
// k > 1: the k largest in descending order (lowest index first among equals), softmax over them at `temperature`,
// inverse-CDF draw with the row's injected uniform u in [0,1): first j with cdf_j > u.
template <typename T, int PER>
__global__ __launch_bounds__(256) void topk_kernel(const T* __restrict__ logits, int64_t ld, const float* __restrict__ uniforms,
                                                   int64_t* __restrict__ out, int64_t R, int V, int k, float inv_temp) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  float v[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int c = lane + 64 * j;
    v[j] = c < V ? to_f32<T>(logits[row * ld + c]) : -INFINITY;
  }
  float top_v = 0.f; int top_i = 0;          // lane j (< k) keeps the j-th selected (value, index)
  float vmax = 0.f;
  for (int sel = 0; sel < k; ++sel) {
    float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int c = lane + 64 * j;
      if (v[j] > bv || (v[j] == bv && c < bi)) { bv = v[j]; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (sel == 0) vmax = bv;
    if (lane == sel) { top_v = bv; top_i = bi; }
#pragma unroll
    for (int j = 0; j < PER; ++j)
      if (lane + 64 * j == bi) v[j] = -INFINITY;          // remove the winner
  }
  if (k == 1) { if (lane == 0) out[row] = top_i; return; }
  const float e = lane < k ? __expf((top_v - vmax) * inv_temp) : 0.f;
  float cdf = e;                                          // inclusive prefix sum over lanes 0..k-1
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(cdf, o, 64); if (lane >= o) cdf += t; }
  const float total = __shfl(cdf, 63, 64);
  const float u = uniforms[row] * total;
  const unsigned long long m = __ballot(lane < k && cdf > u);
  int pick = m ? __builtin_ctzll(m) : k - 1;
  const int idx = __shfl(top_i, pick, 64);
  if (lane == 0) out[row] = idx;
}

}  // namespace

extern "C" int pt_codes_from_continuous(const float* x, int64_t* codes, int64_t n, int64_t bins, pt_stream stream) {
  if (n <= 0 || bins < 2 || !x || !codes) return PT_ERR_SHAPE;
  int64_t b = (n + 255) / 256; if (b > 4096) b = 4096;
  hipLaunchKernelGGL(codes_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, x, codes, n, (int)bins);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

extern "C" int pt_sample_topk(const void* logits, int64_t ld, const float* uniforms, int64_t* out, int64_t R, int64_t V,
                              int64_t k, float temperature, int dtype, pt_stream stream) {
  if (R <= 0 || V <= 0 || V > 64 * 32 || k < 1 || k > 64 || k > V || ld < V) return PT_ERR_SHAPE;
  if (!logits || !out || (k > 1 && !uniforms) || !(temperature > 0.f)) return PT_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((R + 3) / 4));
  const float it = 1.f / temperature;
#define TK(TT, PER) hipLaunchKernelGGL((topk_kernel<TT, PER>), grid, dim3(256), 0, s, (const TT*)logits, ld, uniforms, out, R, (int)V, (int)k, it)
  if (dtype == PT_F32) { if (V <= 1024) TK(float, 16); else TK(float, 32); }
  else if (dtype == PT_BF16) { if (V <= 1024) TK(bf16_t, 16); else TK(bf16_t, 32); }
  else return PT_ERR_DTYPE;
#undef TK
  PT_LAUNCH_CHECK();
  return PT_OK;
}
