// Per-tensor fp8 quantisation for pt_gemm_fp8 (BASELINE configs[4]: "fp8 MFMA GEMMs").
//   amax = max |x| over the tensor;  scale = amax / FMAX (1 when the tensor is all zero);  q = fp8(clamp(x / scale, +-FMAX)), RNE
// FMAX = 448 (e4m3, OCP "fn": weights and activations) / 57344 (e5m2: output gradients).  `state` = {amax, scale, 256 per-block
// maxima} in device memory: the GEMM reads the scale there, nothing returns to the host, and nothing needs clearing between
// calls (no atomics: the first launch leaves one maximum per workgroup, every workgroup of the second reduces those 256 words).
// Current scaling (the amax of THIS tensor, not a history): the tensor is read twice, the second time out of the infinity cache for every activation of the model sizes here.
// HBM-bound: 2 B read (+ 2 B re-read, mostly cached) + 1 B written per element (+ 1 B for the transposed copy of a weight).
#include "common.h"

namespace {
constexpr float F8_MAX_E4M3 = 448.f, F8_MAX_E5M2 = 57344.f;
static_assert(PT_FP8_AMAX_BLOCKS == 256 && PT_FP8_STATE_FLOATS == 2 + PT_FP8_AMAX_BLOCKS, "one partial maximum per thread of the second launch");

// |x| of bf16 values orders like their 15 low bits as integers: the amax is an integer max over 16-bit fields
__global__ __launch_bounds__(256) void fp8_amax_kernel(const bf16_t* __restrict__ x, int64_t rows, int chunks_per_row, int64_t ld,
                                                      unsigned* __restrict__ part) {
  __shared__ uint32_t wmax[4];
  const int64_t total = rows * chunks_per_row;
  uint32_t m = 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; q + 3 * stride < total; q += 4 * stride) {            // four 16-byte loads in flight per thread
    u32x4_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t qq = q + u * stride, r = qq / chunks_per_row; const int c = (int)(qq - r * chunks_per_row);
      v[u] = *reinterpret_cast<const u32x4_t*>(x + r * ld + 8 * c);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) { m = max(m, v[u][k] & 0x7fffu); m = max(m, (v[u][k] >> 16) & 0x7fffu); }
  }
  for (; q < total; q += stride) {
    const int64_t r = q / chunks_per_row; const int c = (int)(q - r * chunks_per_row);
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(x + r * ld + 8 * c);
#pragma unroll
    for (int k = 0; k < 4; ++k) { m = max(m, v[k] & 0x7fffu); m = max(m, (v[k] >> 16) & 0x7fffu); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])) << 16;   // f32 bits of the bf16 value
}

template <int FMT> __device__ __forceinline__ uint32_t cvt4(float a, float b, float c, float d, float inv) {
  constexpr float FM = FMT == PT_FP8_E4M3 ? F8_MAX_E4M3 : F8_MAX_E5M2;
  a = fminf(fmaxf(a * inv, -FM), FM); b = fminf(fmaxf(b * inv, -FM), FM);
  c = fminf(fmaxf(c * inv, -FM), FM); d = fminf(fmaxf(d * inv, -FM), FM);
  int w = 0;
  if (FMT == PT_FP8_E4M3) { w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false); w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true); }
  else                    { w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, w, false); w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true); }
  return (uint32_t)w;
}
// every workgroup (256 threads) reduces the PT_FP8_AMAX_BLOCKS partial maxima; block 0 publishes {amax, scale}
template <int FMT> __device__ __forceinline__ void scales(float* state, float& scale, float& inv) {
  constexpr float FM = FMT == PT_FP8_E4M3 ? F8_MAX_E4M3 : F8_MAX_E5M2;
  __shared__ uint32_t wmax[4];
  uint32_t m = reinterpret_cast<const uint32_t*>(state)[2 + threadIdx.x];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  const uint32_t abits = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
  const float amax = __uint_as_float(abits);
  scale = amax > 0.f ? amax / FM : 1.f;
  inv = amax > 0.f ? FM / amax : 1.f;
  // a tensor that holds an Inf or a NaN (its |bits| >= 0x7f80 win the integer maximum) must not come back as finite garbage:
  // the published scale is NaN, so the GEMM's dequantisation (accumulator * scale_a * scale_b) poisons every output it feeds and
  // the divergence reaches the loss / gradient norm as it does on the bf16 path
  if (abits >= 0x7f800000u) { scale = __uint_as_float(0x7fc00000u); inv = 0.f; }
  if (blockIdx.x == 0 && threadIdx.x == 0) { state[0] = amax; state[1] = scale; }
}
__device__ __forceinline__ float lo16(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi16(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// 16 elements (two 16-byte chunks in, one out) per thread iteration
template <int FMT>
__global__ __launch_bounds__(256) void fp8_quantize_kernel(const bf16_t* __restrict__ x, int64_t rows, int c16_per_row, int64_t ld,
                                                          uint8_t* __restrict__ out, int64_t ld_out, float* __restrict__ state) {
  float scale, inv;
  scales<FMT>(state, scale, inv);
  const int64_t total = rows * c16_per_row;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
    const int64_t r = q / c16_per_row; const int c = (int)(q - r * c16_per_row);
    const u32x4_t a = *reinterpret_cast<const u32x4_t*>(x + r * ld + 16 * c), b = *reinterpret_cast<const u32x4_t*>(x + r * ld + 16 * c + 8);
    u32x4_t o;
    o[0] = cvt4<FMT>(lo16(a[0]), hi16(a[0]), lo16(a[1]), hi16(a[1]), inv);
    o[1] = cvt4<FMT>(lo16(a[2]), hi16(a[2]), lo16(a[3]), hi16(a[3]), inv);
    o[2] = cvt4<FMT>(lo16(b[0]), hi16(b[0]), lo16(b[1]), hi16(b[1]), inv);
    o[3] = cvt4<FMT>(lo16(b[2]), hi16(b[2]), lo16(b[3]), hi16(b[3]), inv);
    *reinterpret_cast<u32x4_t*>(out + r * ld_out + 16 * c) = o;
  }
}

// weights: the quantised matrix AND its transpose (the dgrad GEMM reads W^T K-contiguous) from one read; 64 x 64 tiles
template <int FMT>
__global__ __launch_bounds__(256) void fp8_quantize_t_kernel(const bf16_t* __restrict__ x, int tiles_c, int64_t ld,
                                                            uint8_t* __restrict__ out, int64_t ld_out, uint8_t* __restrict__ out_t,
                                                            int64_t ld_t, float* __restrict__ state) {
  // [col][row], 68-byte pitch: the byte scatter of one instruction (16 rows x 4 column groups) touches 16 different banks, and
  // the dword reads of the transposed rows (17 row + 4 cq + k mod 64) are conflict-free as well
  __shared__ __attribute__((aligned(16))) uint8_t tile[64][68];
  float scale, inv;
  scales<FMT>(state, scale, inv);
  const int tr = blockIdx.x / tiles_c, tc = blockIdx.x - tr * tiles_c;
  const int64_t r0 = (int64_t)tr * 64, c0 = (int64_t)tc * 64;
  const int row = threadIdx.x >> 2, cq = threadIdx.x & 3;
  const bf16_t* src = x + (r0 + row) * ld + c0 + 16 * cq;
  const u32x4_t a = *reinterpret_cast<const u32x4_t*>(src), b = *reinterpret_cast<const u32x4_t*>(src + 8);
  u32x4_t o;
  o[0] = cvt4<FMT>(lo16(a[0]), hi16(a[0]), lo16(a[1]), hi16(a[1]), inv);
  o[1] = cvt4<FMT>(lo16(a[2]), hi16(a[2]), lo16(a[3]), hi16(a[3]), inv);
  o[2] = cvt4<FMT>(lo16(b[0]), hi16(b[0]), lo16(b[1]), hi16(b[1]), inv);
  o[3] = cvt4<FMT>(lo16(b[2]), hi16(b[2]), lo16(b[3]), hi16(b[3]), inv);
  *reinterpret_cast<u32x4_t*>(out + (r0 + row) * ld_out + c0 + 16 * cq) = o;
#pragma unroll
  for (int e = 0; e < 16; ++e) tile[16 * cq + e][row] = (uint8_t)(o[e >> 2] >> (8 * (e & 3)));
  __syncthreads();
  const uint32_t* trow = reinterpret_cast<const uint32_t*>(&tile[row][16 * cq]);
  *reinterpret_cast<u32x4_t*>(out_t + (c0 + row) * ld_t + r0 + 16 * cq) = (u32x4_t){trow[0], trow[1], trow[2], trow[3]};
}
}  // namespace

extern "C" int pt_fp8_quantize(const void* x, int64_t rows, int64_t cols, int64_t ldx, void* out, int64_t ld_out, void* out_t,
                               int64_t ld_t, float* state, int format, pt_stream stream) {
  if (rows <= 0 || cols <= 0 || cols % 16 != 0 || rows >= (1ll << 31) || cols >= (1ll << 31)) return PT_ERR_SHAPE;
  if (format != PT_FP8_E4M3 && format != PT_FP8_E5M2) return PT_ERR_DTYPE;
  if (!x || !out || !state) return PT_ERR_ARG;
  if (!pt_aligned16(x) || (ldx * 2) % 16 || !pt_aligned16(out) || ld_out % 16 || (reinterpret_cast<uintptr_t>(state) & 7u)) return PT_ERR_ALIGN;
  if (out_t && (rows % 64 != 0 || cols % 64 != 0 || !pt_aligned16(out_t) || ld_t % 16)) return PT_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const bf16_t* xb = (const bf16_t*)x;
  int64_t blocks;
  hipLaunchKernelGGL(fp8_amax_kernel, dim3(PT_FP8_AMAX_BLOCKS), dim3(256), 0, s, xb, rows, (int)(cols / 8), ldx, reinterpret_cast<unsigned*>(state) + 2);
  PT_LAUNCH_CHECK();
  if (out_t) {
    const int tiles_c = (int)(cols / 64);
    const int64_t nblk = (rows / 64) * tiles_c;
    if (nblk >= (1ll << 31)) return PT_ERR_SHAPE;
    if (format == PT_FP8_E4M3) hipLaunchKernelGGL((fp8_quantize_t_kernel<PT_FP8_E4M3>), dim3((unsigned)nblk), dim3(256), 0, s, xb, tiles_c, ldx, (uint8_t*)out, ld_out, (uint8_t*)out_t, ld_t, state);
    else hipLaunchKernelGGL((fp8_quantize_t_kernel<PT_FP8_E5M2>), dim3((unsigned)nblk), dim3(256), 0, s, xb, tiles_c, ldx, (uint8_t*)out, ld_out, (uint8_t*)out_t, ld_t, state);
  } else {
    blocks = (rows * (cols / 16) + 255) / 256; if (blocks > 4096) blocks = 4096;
    if (format == PT_FP8_E4M3) hipLaunchKernelGGL((fp8_quantize_kernel<PT_FP8_E4M3>), dim3((unsigned)blocks), dim3(256), 0, s, xb, rows, (int)(cols / 16), ldx, (uint8_t*)out, ld_out, state);
    else hipLaunchKernelGGL((fp8_quantize_kernel<PT_FP8_E5M2>), dim3((unsigned)blocks), dim3(256), 0, s, xb, rows, (int)(cols / 16), ldx, (uint8_t*)out, ld_out, state);
  }
  PT_LAUNCH_CHECK();
  return PT_OK;
}
