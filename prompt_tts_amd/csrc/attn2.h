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

// Diagnostic build only (make trace; tools/attn_trace.py): thread 0 of one workgroup in the middle of the grid leaves s_memtime
// stamps per tile phase.  The stamps go to a buffer nothing else reads; no output depends on them.
#ifndef A2_TRACE
#define A2_TRACE 0
#endif
#if A2_TRACE
static __device__ unsigned long long a2_trace_buf[512];
#define A2_STAMP_(k) do { if ((int)blockIdx.x == (int)(gridDim.x / 2) && threadIdx.x == 0 && (k) < 512) a2_trace_buf[k] = __builtin_amdgcn_s_memtime(); } while (0)
#if A2_TRACE == 2          // stamps of the dK/dV kernel instead of the dQ kernel's
#define A2_STAMP(k) do { } while (0)
#define A2_KV_STAMP(k) A2_STAMP_(k)
#else
#define A2_STAMP(k) A2_STAMP_(k)
#define A2_KV_STAMP(k) do { } while (0)
#endif
#else
#define A2_STAMP(k) do { } while (0)
#define A2_KV_STAMP(k) do { } while (0)
#endif

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
  // ONE image serves row reads AND transposed reads (the backward kernels read K, Q and dO both ways): chunk c of row r lives
  // at chunk c ^ sw(r),  sw(r) = (((r / RPL) & (NB - 1)) << 2) | ((r >> 2) & 3).
  //   row read (ds_read_b128 of the 32x32 A operand): the 16 lanes of a read group hold 16 different rows at one chunk column;
  //     with this swizzle they cover the 16 chunk slots of a 256-byte bank line exactly once (D = 32, 64, 128; checked by
  //     enumeration of the four lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32);
  //   transposed read (ds_read_b64_tr_b16): a 32-lane half reads 4 consecutive rows x 64 contiguous bytes; the high swizzle
  //     bits move the 64-byte block of rows r and r + RPL.. to different quarters of the bank line.
  __device__ static __forceinline__ int sw(int r) { return (((r / RPL) & (NB - 1)) << 2) | ((r >> 2) & 3); }
  __device__ static __forceinline__ int off(int r, int c) { return r * RB + (((c ^ sw(r)) & (NCH - 1)) << 4); }
};

// Per-lane LDS offsets of the fragment reads (loop invariant; tiles 16 / 32 rows further add an immediate: sw() of a row
// does not change when a multiple of 16 is added to rows that differ only in such an offset).
template <int D> struct A2Offsets {
  using C = A2<D>;
  int rowread[C::KS];        // row read: row (lane & 31), chunk 2 ks + h
  int trread[2][C::DT];      // transposed read w: row 8 w + 4 h + q, bytes 64 dt + 32 (G & 1) + 8 p   (lane = 16 G + 4 q + p)
  __device__ __forceinline__ void init(int lane) {
    const int r = lane & 31, h = lane >> 5, G = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) rowread[ks] = C::off(r, 2 * ks + h);
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) trread[w][dt] = C::off(8 * w + 4 * h + q, 4 * dt + 2 * (G & 1) + (pp >> 1)) + ((pp & 1) << 3);
  }
};

// A operand fragment by rows: image row row0 + (lane & 31), k-step ks (row0 a multiple of 32)
template <int D> __device__ __forceinline__ bf16x8_t a2_read_rows(const char* img, int off, int row0) {
  return *reinterpret_cast<const bf16x8_t*>(img + off + row0 * A2<D>::RB);
}
// A operand fragment transposed: lane gets column 32 dt + (lane & 31) of image rows k0 + {4h..4h+3, 8+4h..8+4h+3} (k0 % 16 == 0).
// EXEC must be all ones.
template <int D> __device__ __forceinline__ bf16x8_t a2_read_tr(const char* img, int off_lo, int off_hi, int k0) {
  typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + off_lo + k0 * A2<D>::RB));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + off_hi + k0 * A2<D>::RB));
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
// max(a, b, c) as ONE v_max3_f32: written with fmaxf, hipcc emits a canonicalising v_max_f32 v, v beside each of the 31 maxima
// of a 32-score row (48 instructions where 16 do)
__device__ __forceinline__ float a2_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
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

typedef __attribute__((address_space(3))) void a2_lptr;

// One LDS-DMA wave-instruction (64 lanes x 16 bytes -> 1 KiB of LDS at `lds`, which must be wave-uniform), as INLINE ASM.
// Through the builtin, hipcc knows the instruction writes LDS: it then puts an s_waitcnt vmcnt(0) in front of the next ds_read
// that might alias -- with a dynamically indexed ring that is every read of the tile being multiplied, i.e. the load of tile t + 1
// was waited for in the middle of tile t and the double buffering hid nothing.  The asm form is invisible to that pass: the
// kernels wait themselves (a2_dma_wait() in front of the barrier that publishes the tile).  M0 is written inside the statement
// and not used by anything else in these kernels (no movrel, GWS or DMA builtins).
__device__ __forceinline__ void a2_dma16(const void* gsrc, char* lds) {
  const uint32_t a = (uint32_t)(uintptr_t)(a2_lptr*)lds;
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(a) : "memory");
}
__device__ __forceinline__ void a2_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// The same with the tile's base address in SGPRs and a 32-bit per-lane byte offset (the `saddr` form): the per-lane part is
// loop invariant, so a tile costs no vector address arithmetic at all.
__device__ __forceinline__ void a2_dma16s(uint32_t voff, const void* sbase, char* lds) {
  const uint32_t a = (uint32_t)(uintptr_t)(a2_lptr*)lds;
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(a) : "memory");
}

// HBM -> LDS staging of a 64-row image by LDS-DMA.  One wave-instruction fills 1 KiB (lane * 16 bytes from a wave-uniform LDS
// base); image slot q = tid + 256 i is row q / NCH, physical chunk q % NCH, i.e. LOGICAL chunk (q % NCH) ^ sw(row): the
// swizzle is applied to the per-lane source address.  Rows past the tensor's end re-read its last row (finite data; masked by
// the caller): only the last, ragged tile takes that path.
// 64 consecutive floats (one 64-row tile's LSE or delta slice) -> 256 bytes of LDS by ONE wave-instruction (4 bytes per lane);
// rows past `limit` re-read the last row.
__device__ __forceinline__ void a2_dma_f32x64(const float* base, int row0, int limit, char* lds, int lane) {
  const uint64_t ta = (uint64_t)(base + row0);
  const uint32_t ta_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ta >> 32));
  const uint32_t ta_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ta);
  const float* tile = reinterpret_cast<const float*>(((uint64_t)ta_hi << 32) | (uint64_t)ta_lo);
  const int r = row0 + lane < limit ? lane : limit - 1 - row0;
  const uint32_t a = (uint32_t)(uintptr_t)(a2_lptr*)lds;
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dword %0, %1" ::"v"((uint32_t)(r * 4)), "s"(tile), "s"(a) : "memory");
}

// NT = threads of the workgroup (256 or 512): the image's 64 NCH chunks are dealt out over NT threads; when the image has fewer
// chunks than threads (D = 32 with 512 threads) only the first waves issue (wave-uniform predicate).
template <int D, int NT = 256> struct A2Stage {
  using C = A2<D>;
  static constexpr int SLOTS = 64 * C::NCH;
  static constexpr int PER = (SLOTS + NT - 1) / NT;
  uint32_t voff[PER];      // byte offset of this thread's chunks from the tile's first row
  int64_t ld;
  __device__ __forceinline__ void init(int tid, int64_t ld_) {
    ld = ld_;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int q = (tid + NT * i) % SLOTS, r = q / C::NCH, cp = q % C::NCH;
      voff[i] = (uint32_t)(r * ld * 2 + ((cp ^ C::sw(r)) & (C::NCH - 1)) * 16);
    }
  }
  // base: the tensor's (batch, head) slice; row0 (wave-uniform): first row of the tile; limit: rows of the tensor
  __device__ __forceinline__ void issue(const bf16_t* base, int row0, int limit, char* img, int wave, int tid) const {
    // the tile's base address is wave-uniform; say so (an "s" operand the compiler believes divergent is handed over in VGPRs)
    const uint64_t ta = (uint64_t)(base + (int64_t)row0 * ld);
    // (readfirstlane returns a SIGNED int: both halves go through uint32_t, or a low half with bit 31 set sign-extends into
    // the high half -- an address fault that depends on where the allocator put the tensor)
    const uint32_t ta_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ta >> 32));
    const uint32_t ta_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ta);
    const bf16_t* tile = reinterpret_cast<const bf16_t*>(((uint64_t)ta_hi << 32) | (uint64_t)ta_lo);
    if (row0 + 64 <= limit) {
#pragma unroll
      for (int i = 0; i < PER; ++i)
        if (wave * 64 + NT * i < SLOTS) a2_dma16s(voff[i], tile, img + (wave * 64 + NT * i) * 16);
    } else {
#pragma unroll
      for (int i = 0; i < PER; ++i)
        if (wave * 64 + NT * i < SLOTS) {
          const int q = tid + NT * i, cp = q % C::NCH;
          int r = q / C::NCH;
          const int sw = C::sw(r);
          r = row0 + r < limit ? r : limit - 1 - row0;
          a2_dma16s((uint32_t)(r * ld * 2 + ((cp ^ sw) & (C::NCH - 1)) * 16), tile, img + (wave * 64 + NT * i) * 16);
        }
    }
  }
};
