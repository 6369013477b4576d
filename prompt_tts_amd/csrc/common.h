// Device-side common definitions for the gfx950 kernels (wave64, MFMA 16x16, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/prompt_tts_hip.h"

#define PT_WAVE 64

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

struct bf16_t { uint16_t bits; };   // storage type for bf16 activations

// ---- scalar conversions ---------------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;                       // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(uint16_t, h);
}
// two f32 -> one dword of two bf16 (low half = a): ONE v_cvt_pk_bf16_f32 (converting singly costs a shift and an or per pair)
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_v;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_v;
  const bf16x2_v h = __builtin_convertvector((f32x2_v){a, b}, bf16x2_v);
  return __builtin_bit_cast(uint32_t, h);
}
template <typename T> __device__ __forceinline__ float to_f32(T x);
template <> __device__ __forceinline__ float to_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t x) { return bf16_bits_to_f32(x.bits); }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { bf16_t r; r.bits = f32_to_bf16_bits(x); return r; }

// ---- 16-byte vectors of activations: VEC<T> elements per 16 B ------------------------------------------
template <typename T> struct vec_traits;
template <> struct vec_traits<float> { static constexpr int N = 4; };
template <> struct vec_traits<bf16_t> { static constexpr int N = 8; };

template <typename T> struct Vec16 {
  u32x4_t raw;
  static constexpr int N = vec_traits<T>::N;
  __device__ __forceinline__ float get(int i) const;
  __device__ __forceinline__ void set(int i, float v);
};
template <> __device__ __forceinline__ float Vec16<float>::get(int i) const { return __uint_as_float(raw[i]); }
template <> __device__ __forceinline__ void Vec16<float>::set(int i, float v) { raw[i] = __float_as_uint(v); }
template <> __device__ __forceinline__ float Vec16<bf16_t>::get(int i) const {
  uint32_t w = raw[i >> 1];
  return __uint_as_float((i & 1) ? (w & 0xffff0000u) : (w << 16));
}
template <> __device__ __forceinline__ void Vec16<bf16_t>::set(int i, float v) {
  uint32_t b = f32_to_bf16_bits(v);
  uint32_t w = raw[i >> 1];
  raw[i >> 1] = (i & 1) ? ((w & 0x0000ffffu) | (b << 16)) : ((w & 0xffff0000u) | b);
}
template <typename T> __device__ __forceinline__ Vec16<T> load16(const T* p) {
  Vec16<T> v; v.raw = *reinterpret_cast<const u32x4_t*>(p); return v;
}
template <typename T> __device__ __forceinline__ void store16(T* p, const Vec16<T>& v) {
  *reinterpret_cast<u32x4_t*>(p) = v.raw;
}
template <typename T> __device__ __forceinline__ Vec16<T> zero16() { Vec16<T> v; v.raw = (u32x4_t){0, 0, 0, 0}; return v; }


// store 4 consecutive elements of an output row
template <typename T> __device__ __forceinline__ void store4(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<f32x4_t*>(p) = (f32x4_t){a, b, c, d};
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
  u32x2_t o;
  o[0] = pack_bf16x2(a, b);
  o[1] = pack_bf16x2(c, d);
  *reinterpret_cast<u32x2_t*>(p) = o;
}

// ---- reductions ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block sum for blockDim.x == NT (multiple of 64); every thread gets the result. scratch: NT/64 floats.
template <int NT> __device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) scratch[w] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) r += scratch[i];
  return r;
}

// v_rcp_f32 (1 ulp) instead of the IEEE division sequence: the GroupNorm(+SiLU) kernels are as VALU-bound as HBM-bound
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float silu_grad_f(float x) {
  float s = __builtin_amdgcn_rcpf(1.f + __expf(-x));
  return s * (1.f + x * (1.f - s));
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad_f(float x) {
  float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
  float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// erf-GELU and its derivative from ONE exponential: erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7), and
// exp(-(x/sqrt 2)^2) = exp(-x^2/2) is also the Gaussian density the derivative needs.  Used by the GEGLU epilogues of the
// bf16 GEMMs, where a library erff per element would cost more VALU time than the tile's MFMAs.
__device__ __forceinline__ void gelu_erf_fast(float x, float& gelu, float& dgelu) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * z);
  const float e = __expf(-z * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float cdf = 0.5f * (1.f + copysignf(1.f - poly * e, x));
  gelu = x * cdf;
  dgelu = cdf + x * 0.39894228040143267794f * e;
}

// LayerNorm arithmetic shared by ln_fwd_kernel (norms.hip) and the LayerNorm prologue of pt_decode_linear (decode_step.hip), with
// the multiply-add fusion spelled out: under -ffp-contract=fast the backend fuses where it sees fit, and it saw fit differently in
// the two kernels -- the same source expression gave f32 values one ulp apart, i.e. one bf16 ulp whenever the value sits on a
// rounding boundary (1 element in ~30 000; enough to flip a sampler near-tie a few frames later).
__device__ __forceinline__ float pt_ln_sq_acc(float ss, float d) { return __builtin_fmaf(d, d, ss); }
__device__ __forceinline__ float pt_ln_rstd(float ss, float n, float eps) {
#pragma clang fp contract(off)
  const float var = ss / n;
  return rsqrtf(var + eps);
}
__device__ __forceinline__ float pt_ln_apply(float v, float mu, float rs, float g, float b) {
#pragma clang fp contract(off)
  const float t = (v - mu) * rs;
  return __builtin_fmaf(t, g, b);
}

// host-side launch check
extern thread_local int pt_g_last_hip_error;      // capi.hip: the hipError_t behind this thread's most recent PT_ERR_LAUNCH (diagnostics)
#define PT_LAUNCH_CHECK()                                  \
  do {                                                     \
    const hipError_t pt_e_ = hipGetLastError();            \
    if (pt_e_ != hipSuccess) {                             \
      pt_g_last_hip_error = (int)pt_e_;                    \
      return PT_ERR_LAUNCH;                                \
    }                                                      \
  } while (0)

// Diagnostic switches are read from the environment ONCE per process: a function-local `static const int` is initialised
// under the C++11 thread-safe-statics guarantee, so concurrent first calls do not race.
static inline int pt_env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

static inline bool pt_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
