// MFMA 16x16 "atoms" for gfx950, written once for both activation dtypes.
//
// One atom = a 16x16 f32 accumulator tile advanced by a 32-deep reduction chunk:
//   bf16 : one  v_mfma_f32_16x16x32_bf16       (lane l: A[row l&15][k = 8*(l>>4)+j], j = 0..7)
//   f32  : eight v_mfma_f32_16x16x4_f32         (exact f32 FMA chain; instruction j takes, from lane
//          group g = l>>4, the SAME logical k = 8g+j that the bf16 fragment holds in element j, so a
//          fragment is "8 consecutive reduction elements per lane" in both dtypes)
// C/D layout (both): lane l holds D[row = 4*(l>>4) + r][col = l&15] in register r = 0..3.
//
// LDS tiles come in two images, both built from 16-byte chunks with an XOR swizzle:
//   TileK : [rows][128 B]   reduction index contiguous ("N-mode" operand): fragment = chunk read(s)
//   TileT : [k rows][ROWB B] reduction index on the rows ("T-mode" operand): fragment = transposed
//           read (ds_read_b64_tr_b16 for bf16, 8 scalar reads for f32)
#pragma once
#include "common.h"

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8_t v; };
template <> struct Frag<float> { float v[8]; };

__device__ __forceinline__ void mma16(f32x4_t& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4_t& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

// ---- bf16 x 3: f32 operands, f32-class products on the bf16 MFMA ----------------------------------------------------------------
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi);  a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi  (the dropped a_lo b_lo term and the
// rounding of lo are ~2^-17 relative).  Three v_mfma_f32_16x16x32_bf16 (48 cycles) where the exact form takes eight
// v_mfma_f32_16x16x4_f32 (256 cycles).  Used where the reference's fp32 arithmetic has to be matched to 1e-3, not bit for bit
// (Encodec decode at f32: decode_codec.py:12-16); the training parity mode keeps the exact f32 MFMA.
struct FragX3 { bf16x8_t hi, lo; };
__device__ __forceinline__ FragX3 split_x3(const Frag<float>& f) {
  FragX3 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)f.v[j];
    r.hi[j] = h; r.lo[j] = (__bf16)(f.v[j] - (float)h);
  }
  return r;
}
__device__ __forceinline__ void mma16x3(f32x4_t& acc, const FragX3& a, const FragX3& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, acc, 0, 0, 0);     // small terms first
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, acc, 0, 0, 0);
}

// Two accumulators at once, products interleaved so that no MFMA reads the accumulator the one before it writes (a dependent
// 16x16x32 issues ~10 cycles later than an independent one); per accumulator the order of the three terms is mma16x3's.
__device__ __forceinline__ void mma16x3_2a(f32x4_t& acc0, const FragX3& a0, f32x4_t& acc1, const FragX3& a1, const FragX3& b) {   // two A (weights), one B
  acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.lo, b.hi, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.lo, b.hi, acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.hi, b.lo, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.hi, b.lo, acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.hi, b.hi, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.hi, b.hi, acc1, 0, 0, 0);
}
__device__ __forceinline__ void mma16x3_2b(f32x4_t& acc0, const FragX3& b0, f32x4_t& acc1, const FragX3& b1, const FragX3& a) {   // one A, two B
  acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b0.hi, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b1.hi, acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b0.lo, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b1.lo, acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b0.hi, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b1.hi, acc1, 0, 0, 0);
}

// pack 8 f32 (lane-local) into a fragment
__device__ __forceinline__ void frag_from_f32(Frag<bf16_t>& f, const float* x) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)x[j];
}
__device__ __forceinline__ void frag_from_f32(Frag<float>& f, const float* x) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = x[j];
}

// fragment straight from global memory: 8 consecutive elements at p (16-byte aligned for bf16, 32 for f32 pairs)
__device__ __forceinline__ void frag_load_global(Frag<bf16_t>& f, const bf16_t* p) {
  f.v = *reinterpret_cast<const bf16x8_t*>(p);
}
__device__ __forceinline__ void frag_load_global(Frag<float>& f, const float* p) {
  f32x4_t a = *reinterpret_cast<const f32x4_t*>(p), b = *reinterpret_cast<const f32x4_t*>(p + 4);
  f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
  f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}
__device__ __forceinline__ void frag_zero(Frag<bf16_t>& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)0.f;
}
__device__ __forceinline__ void frag_zero(Frag<float>& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = 0.f;
}

// ---------------------------------------------------------------------------------------------------
// TileK: rows of 128 bytes (bf16: 64 elements, f32: 32 elements); chunk c (16 B) of row r lives at
// chunk slot c ^ (r & 7).  ds_read_b128 of 16 rows x one chunk column is then bank-conflict free.
// ---------------------------------------------------------------------------------------------------
template <typename T> struct TileK {
  static constexpr int ROW_BYTES = 128;
  static constexpr int KE = ROW_BYTES / (int)sizeof(T);      // reduction elements per tile row
  static constexpr int CHUNKS = 8;
  __device__ static __forceinline__ int chunk_off(int r, int c) { return r * ROW_BYTES + ((c ^ (r & 7)) << 4); }
  __device__ static __forceinline__ void store_chunk(char* base, int r, int c, const Vec16<T>& v) {
    *reinterpret_cast<u32x4_t*>(base + chunk_off(r, c)) = v.raw;
  }
};
// fragment of row r, reduction elements [k0, k0+8) (k0 multiple of 8)
__device__ __forceinline__ void frag_load_k(Frag<bf16_t>& f, const char* base, int r, int k0) {
  f.v = *reinterpret_cast<const bf16x8_t*>(base + TileK<bf16_t>::chunk_off(r, k0 >> 3));
}
__device__ __forceinline__ void frag_load_k(Frag<float>& f, const char* base, int r, int k0) {
  f32x4_t a = *reinterpret_cast<const f32x4_t*>(base + TileK<float>::chunk_off(r, (k0 >> 2)));
  f32x4_t b = *reinterpret_cast<const f32x4_t*>(base + TileK<float>::chunk_off(r, (k0 >> 2) + 1));
  f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
  f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}

// ---------------------------------------------------------------------------------------------------
// TileT: [k][COLS] with COLS*sizeof(T) = ROWB bytes per k-row; chunk c of k-row k lives at slot
// c ^ swz(k), swz(k) = ((k&3)<<1) | ((((k>>2)^(k>>3))&1)<<3): the 8 k-rows x 2 chunks one 32-lane half
// touches in a ds_read_b64_tr_b16 land on 16 distinct chunk slots for both k patterns used
// ({8g..8g+3, 8g+4..} in GEMMs and {4g.., 16+4g..} when the other operand is an accumulator tile).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tilet_swz(int k) { return ((k & 3) << 1) | ((((k >> 2) ^ (k >> 3)) & 1) << 3); }

template <typename T, int COLS> struct TileT {
  static constexpr int ROWB = COLS * (int)sizeof(T);
  static constexpr int EPC = 16 / (int)sizeof(T);             // elements per chunk
  static_assert(ROWB % 256 == 0, "TileT rows must be a multiple of 256 bytes");
  __device__ static __forceinline__ int chunk_off(int k, int c) { return k * ROWB + ((c ^ tilet_swz(k)) << 4); }
  __device__ static __forceinline__ int elem_off(int k, int col) {
    return chunk_off(k, col / EPC) + (col % EPC) * (int)sizeof(T);
  }
  __device__ static __forceinline__ void store_chunk(char* base, int k, int c, const Vec16<T>& v) {
    *reinterpret_cast<u32x4_t*>(base + chunk_off(k, c)) = v.raw;
  }
};

// Transposed fragment: lane (i = lane&15) gets column col0+i, elements j=0..3 from k-rows kb0..kb0+3 and
// j=4..7 from kb1..kb1+3.  col0 multiple of 16.  EXEC must be all ones (no divergence around this).
template <int COLS>
__device__ __forceinline__ void frag_load_t(Frag<bf16_t>& f, const char* base, int col0, int kb0, int kb1, int lane) {
  using TT = TileT<bf16_t, COLS>;
  const int i = lane & 15, q = i >> 2, p = i & 3;
  const int c = (col0 >> 3) + (p >> 1);                       // chunk holding columns col0+4p .. +3
  const int sub = (p & 1) << 3;                               // 8-byte half of the chunk
  typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + TT::chunk_off(kb0 + q, c) + sub));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + TT::chunk_off(kb1 + q, c) + sub));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  f.v = __builtin_bit_cast(bf16x8_t, r);
}
template <int COLS>
__device__ __forceinline__ void frag_load_t(Frag<float>& f, const char* base, int col0, int kb0, int kb1, int lane) {
  using TT = TileT<float, COLS>;
  const int col = col0 + (lane & 15);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f.v[j] = *reinterpret_cast<const float*>(base + TT::elem_off(kb0 + j, col));
    f.v[4 + j] = *reinterpret_cast<const float*>(base + TT::elem_off(kb1 + j, col));
  }
}
