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

// ---- bf16 logits, V <= 1024: packed-key form ---------------------------------------------------------------------------------
// The generic kernel above spends ~130 instructions per selected element (a 16-element scan with index tie-breaks, six
// ds_bpermute shuffle rounds, a 16-element removal scan): 4.5 ms for top-32 over configs[3]'s 524 288 rows of 1 024 logits.
// A bf16 logit and its column fit ONE 32-bit key -- (order-preserving image of the 16 value bits) << 16 | (0xFFFF - column) --
// whose unsigned order IS "larger value first, lower column first among equals", so:
//   * a lane loads 16 CONTIGUOUS columns (two 16-byte loads instead of sixteen 2-byte ones) and sorts its 16 keys once
//     (bitonic network, 80 compare-exchanges = v_max_u32 / v_min_u32 pairs);
//   * one selection = wave maximum of the lanes' heads (4 DPP row_shr maxima + 4 v_readlane + 3 s_max_u32: no LDS) and a pop
//     of the winning lane's list (keys are unique, so `head == maximum` names exactly one lane): ~30 instructions.
// Same results as the generic kernel bit for bit (-0 is read as +0, as the float comparison reads it).
__device__ __forceinline__ uint32_t tk_row_max(uint32_t m) {          // lane 15 of every row of 16 lanes ends with the row's maximum
  uint32_t t;
  t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x111, 0xf, 0xf, true); m = m > t ? m : t;     // row_shr:1
  t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x112, 0xf, 0xf, true); m = m > t ? m : t;     // row_shr:2
  t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x114, 0xf, 0xf, true); m = m > t ? m : t;     // row_shr:4
  t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x118, 0xf, 0xf, true); m = m > t ? m : t;     // row_shr:8
  return m;
}
__device__ __forceinline__ uint32_t tk_wave_max(uint32_t head) {
  const uint32_t m = tk_row_max(head);
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)m, 15), b = (uint32_t)__builtin_amdgcn_readlane((int)m, 31);
  const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)m, 47), d = (uint32_t)__builtin_amdgcn_readlane((int)m, 63);
  const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}
__device__ __forceinline__ uint32_t tk_key(uint32_t bits, int col) {   // bits: the 16 bits of a bf16
  if (bits == 0x8000u) bits = 0u;                                       // -0 compares equal to +0
  const uint32_t o = (bits & 0x8000u) ? (~bits & 0xffffu) : (bits | 0x8000u);
  return (o << 16) | (0xffffu - (uint32_t)col);
}
__device__ __forceinline__ float tk_value(uint32_t key) {
  const uint32_t o = key >> 16, bits = (o & 0x8000u) ? (o & 0x7fffu) : (~o & 0xffffu);
  return bf16_bits_to_f32((uint16_t)bits);
}

__global__ __launch_bounds__(256) void topk_bf16_kernel(const bf16_t* __restrict__ logits, int64_t ld, const float* __restrict__ uniforms,
                                                        int64_t* __restrict__ out, int64_t R, int V, int k, float inv_temp, int vec) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const bf16_t* src = logits + row * ld + 16 * lane;
  uint32_t key[16];
  if (vec && 16 * lane + 16 <= V) {
    const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(src), hi = *reinterpret_cast<const u32x4_t*>(src + 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t w = j < 4 ? lo[j & 3] : hi[j & 3];
      key[2 * j] = tk_key(w & 0xffffu, 16 * lane + 2 * j);
      key[2 * j + 1] = tk_key(w >> 16, 16 * lane + 2 * j + 1);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = 16 * lane + j;
      key[j] = c < V ? tk_key(src[j].bits, c) : 0u;                    // below every real key (a real -inf has a larger image)
    }
  }
  if (k == 1) {
    uint32_t head = key[0];
#pragma unroll
    for (int j = 1; j < 16; ++j) head = head > key[j] ? head : key[j];
    const uint32_t best = tk_wave_max(head);
    if (lane == 0) out[row] = (int64_t)(0xffffu - (best & 0xffffu));
    return;
  }
  // descending bitonic sort of the lane's 16 keys
#pragma unroll
  for (int kk = 2; kk <= 16; kk <<= 1)
#pragma unroll
    for (int jj = kk >> 1; jj > 0; jj >>= 1)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int l = i ^ jj;
        if (l > i) {
          const uint32_t hi = key[i] > key[l] ? key[i] : key[l], lo = key[i] > key[l] ? key[l] : key[i];
          if ((i & kk) == 0) { key[i] = hi; key[l] = lo; } else { key[i] = lo; key[l] = hi; }
        }
      }
  uint32_t top = 0u;                         // lane j (< k) keeps the j-th selected key
  for (int sel = 0; sel < k; ++sel) {
    const uint32_t best = tk_wave_max(key[0]);
    if (lane == sel) top = best;
    if (key[0] == best) {                    // exactly one lane: its list moves up
#pragma unroll
      for (int j = 0; j < 15; ++j) key[j] = key[j + 1];
      key[15] = 0u;
    }
  }
  const float vmax = tk_value((uint32_t)__builtin_amdgcn_readlane((int)top, 0));
  const int top_i = (int)(0xffffu - (top & 0xffffu));
  const float e = lane < k ? __expf((tk_value(top) - vmax) * inv_temp) : 0.f;
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
  static const int packed = pt_env_int("PT_TOPK_PACKED", 1);
  if (dtype == PT_F32) { if (V <= 1024) TK(float, 16); else TK(float, 32); }
  else if (dtype == PT_BF16 && V <= 1024 && packed) {
    const int vec = ld % 8 == 0 && (reinterpret_cast<uintptr_t>(logits) & 15u) == 0;
    hipLaunchKernelGGL(topk_bf16_kernel, grid, dim3(256), 0, s, (const bf16_t*)logits, ld, uniforms, out, R, (int)V, (int)k, it, vec);
  }
  else if (dtype == PT_BF16) { if (V <= 1024) TK(bf16_t, 16); else TK(bf16_t, 32); }
  else return PT_ERR_DTYPE;
#undef TK
  PT_LAUNCH_CHECK();
  return PT_OK;
}
