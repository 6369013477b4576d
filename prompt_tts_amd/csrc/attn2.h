// Second-generation flash attention pieces for gfx950 (bf16): 32x32x16 MFMA tiles, LDS-DMA staging, row constants in the
// accumulators.  Shared by attn2_fwd.hip and attn2_bwd.hip.
//
// Why this shape (round-2 counters: a wave issued VALU 27 % of its time, waited 35 %, MFMA pipe ~20 % of peak at D = 64):
//   * at D = 64 a 64-key tile is 512 cycles of MFMA pipe per wave against ~650 cycles of softmax VALU: the loop is bound by
//     vector ISSUE, and a v_mfma_f32_16x16x32 holds the SIMD's issue port for 8 of its 16 cycles, a 32x32x16 for 8 of its 32
//     (MI355X_MICROARCH.md, cycle constants) -- the 32x32 shape leaves 75 % of the issue slots to the softmax;
//   * the query operand is pre-multiplied by scale * log2(e) once, and the running row maximum enters as the INITIAL VALUE of the
//     score accumulators (S' = Q'K^T - m), so a score needs one v_exp_f32 and no multiply / subtract; the maximum is refreshed
//     only when a row outgrows it by 2^8 (lazy rescale: the O-wide multiply leaves the common path);
//   * K / V tiles go HBM -> LDS by global_load_lds (no staging registers, no ds_write); the swizzles live in the SOURCE address.
//
// MFMA 32x32x16 bf16 operand maps (cdna_hip_programming.md 3): lane l, r = l & 31, h = l >> 5
//   A[row r][k = 8h + j], B[k = 8h + j][col r], j = 0..7;  C/D: col = r, row = (reg & 3) + 8 (reg >> 2) + 4h, reg = 0..15.
// An accumulator tile X (rows on registers) is the B operand of a following product over its ROW index with no data movement:
// k-step s takes registers 8s .. 8s+7, whose element j is row 16s + 8 (j >> 2) + 4h + (j & 3) -- the other operand's
// transposed LDS read delivers the same k order (two ds_read_b64_tr_b16: rows +0..3 and +8..11 of the lane half's block).
#pragma once
#include "attn_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ f32x16_t mma32(const bf16x8_t a, const bf16x8_t b, const f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16_t splat16(float v) {
  return (f32x16_t){v, v, v, v, v, v, v, v, v, v, v, v, v, v, v, v};
}

template <int D> struct A2 {
  static constexpr int RB = 2 * D;                          // bytes per image row (one key / query, bf16)
  static constexpr int NCH = RB / 16;                       // 16-byte chunks per row
  static constexpr int RPL = RB >= 256 ? 1 : 256 / RB;      // image rows per 256-byte bank line
  static constexpr int NB = RB / 64;                        // 64-byte blocks per row
  static constexpr int KS = D / 16;                         // k-steps of a product over the head dim
  static constexpr int DT = D / 32;                         // 32-row tiles over the head dim
  static constexpr int TILE = 64 * RB;                      // one 64-row image
  static constexpr int DMA_PER_THREAD = 64 * NCH / 256;     // LDS-DMA instructions per thread per 64-row image (256 threads)
  // ROW-read image (ds_read_b128 of the 32x32 A operand: 16 lanes of a read group = 16 different rows, one chunk column):
  // chunk c of row r lives at chunk c ^ f(r), f(r) = (r / RPL) & (NCH - 1) -- the 16 rows of every read group then cover the
  // 16 chunk slots of a bank line exactly once for D = 32, 64, 128.
  __device__ static __forceinline__ int rsw(int r) { return (r / RPL) & (NCH - 1); }
  __device__ static __forceinline__ int roff(int r, int c) { return r * RB + ((c ^ rsw(r)) << 4); }
  // TRANSPOSED-read image (ds_read_b64_tr_b16: a 32-lane half reads 4 rows x 64 contiguous bytes): the 64-byte block b of row r
  // lives at block b ^ g(r), g(r) = (r / RPL) & (NB - 1) -- the 4 rows then sit on 4 different quarters of the bank line.
  __device__ static __forceinline__ int tsw(int r) { return ((r / RPL) & (NB - 1)) << 2; }
  __device__ static __forceinline__ int toff(int r, int c) { return r * RB + ((c ^ tsw(r)) << 4); }
};

// Per-lane LDS offsets of the fragment reads (loop invariant; tiles 16 / 32 rows further add an immediate).
template <int D> struct A2Offsets {
  using C = A2<D>;
  int rowread[C::KS];     // row-read image: row (lane & 31), chunk 2 ks + h
  int trread[C::DT];      // transposed-read image: row 4h + q, bytes 64 dt + 32 (G & 1) + 8 p   (lane = 16 G + 4 q + p)
  __device__ __forceinline__ void init(int lane) {
    const int r = lane & 31, h = lane >> 5, G = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) rowread[ks] = C::roff(r, 2 * ks + h);
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) trread[dt] = C::toff(4 * h + q, 4 * dt + 2 * (G & 1) + (pp >> 1)) + ((pp & 1) << 3);
  }
};

// A operand fragment by rows: image row row0 + (lane & 31), k-step ks (row0 a multiple of 32)
template <int D> __device__ __forceinline__ bf16x8_t a2_read_rows(const char* img, int off, int row0) {
  return *reinterpret_cast<const bf16x8_t*>(img + off + row0 * A2<D>::RB);
}
// A operand fragment transposed: lane gets column 32 dt + (lane & 31) of image rows k0 + {4h..4h+3, 8+4h..8+4h+3} (k0 % 16 == 0).
// EXEC must be all ones.
template <int D> __device__ __forceinline__ bf16x8_t a2_read_tr(const char* img, int off, int k0) {
  typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + off + k0 * A2<D>::RB));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + off + (k0 + 8) * A2<D>::RB));
  const s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, r);
}

// registers 8s .. 8s+7 of an accumulator tile -> the bf16 B fragment of k-step s
__device__ __forceinline__ bf16x8_t a2_pack(const f32x16_t& x, int s) {
  u32x4_t w;
#pragma unroll
  for (int j = 0; j < 4; ++j) w[j] = pack_bf16x2(x[8 * s + 2 * j], x[8 * s + 2 * j + 1]);
  return __builtin_bit_cast(bf16x8_t, w);
}

// combine a per-lane value with the other 32-lane half's (lane l <-> l ^ 32).  v_permlane32_swap(vdst = x, src = x) swaps lanes
// 32-63 of vdst with lanes 0-31 of src: the two results hold {own | other} and {other | own}, so their max / sum is the
// combination in EVERY lane -- no select, no LDS.
__device__ __forceinline__ float a2_half_max(float v) {
  const int x = __builtin_bit_cast(int, v);
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return fmaxf(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]));
}
__device__ __forceinline__ float a2_half_sum(float v) {
  const int x = __builtin_bit_cast(int, v);
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}

typedef __attribute__((address_space(1))) const void a2_gptr;
typedef __attribute__((address_space(3))) void a2_lptr;

// HBM -> LDS staging of a 64-row image by LDS-DMA.  One wave-instruction fills 1 KiB (lane * 16 bytes from a wave-uniform LDS
// base); image slot q = tid + 256 i is row q / NCH, physical chunk q % NCH, i.e. LOGICAL chunk (q % NCH) ^ swizzle(row): the
// swizzle is applied to the per-lane source address.  Rows past `limit` re-read row limit - 1 (finite data; masked by the caller).
template <int D, bool TR> struct A2Stage {
  using C = A2<D>;
  int row[C::DMA_PER_THREAD], col[C::DMA_PER_THREAD];       // image row, element column of this thread's chunks
  __device__ __forceinline__ void init(int tid) {
#pragma unroll
    for (int i = 0; i < C::DMA_PER_THREAD; ++i) {
      const int q = tid + 256 * i, r = q / C::NCH, cp = q % C::NCH;
      row[i] = r;
      col[i] = (cp ^ (TR ? C::tsw(r) : C::rsw(r))) * 8;
    }
  }
  __device__ __forceinline__ void issue(const bf16_t* base, int64_t ld, int row0, int limit, char* img, int wave) const {
#pragma unroll
    for (int i = 0; i < C::DMA_PER_THREAD; ++i) {
      int r = row0 + row[i]; r = r < limit ? r : limit - 1;
      const bf16_t* src = base + (int64_t)r * ld + col[i];
      __builtin_amdgcn_global_load_lds((a2_gptr*)src, (a2_lptr*)(img + (wave * 64 + 256 * i) * 16), 16, 0, 0);
    }
  }
};
