// Shared pieces of the flash-style attention kernels (forward, dK/dV, dQ).
//
// Orientation used everywhere ("the reduced index never leaves the lane group"):
//   * the operand a wave OWNS for the whole kernel (its q rows, or its keys) sits in registers as MFMA
//     B fragments (column = lane&15), loaded straight from HBM once;
//   * the operand that streams (K/V tiles, or Q/dO tiles) is staged in LDS in ONE image per tensor that
//     serves both row reads (reduction over d) and transposed reads (reduction over the tile's rows);
//   * score tiles are therefore produced with the streamed index on accumulator ROWS and the owned index on
//     the lane, so softmax statistics are per-lane scalars, and P / dS feed the next MFMA as B fragments
//     directly from the accumulator registers (k order 16*(j>>2) + 4*(lane>>4) + (j&3), matched by the
//     transposed LDS read of the other operand) -- P never touches LDS.
#pragma once
#include "mma.h"

struct AttnParams {
  int B, H, Nq, Nk;
  const char* q; int64_t ldq;
  const char* k; int64_t ldk;
  const char* v; int64_t ldv;
  char* o; int64_t ldo;
  float* lse; float scale; int causal; const int32_t* kv_len;
  const char* d_o; int64_t lddo; float* delta;
  char* dq; int64_t lddq; char* dk; int64_t lddk; char* dv; int64_t lddv;
};

// XCD-aware decode of a 1-D grid of nblk x H x B workgroups: ids congruent mod 8 share an XCD (round-robin dispatch), and
// each XCD gets a contiguous run of (b, h, block) triples with the block index fastest, so the blocks of one (b, h) stream
// the operand they all re-read (K/V for query blocks, Q/dO for key blocks) through ONE L2.  With the natural 3-D grid the 8
// blocks of a (b, h) landed on 8 different XCDs and every L2 fetched every K/V: ~3x the fabric bytes (rocprofv3 FETCH_SIZE).
__device__ __forceinline__ void attn_block_ids(int nblk, int H, int& blk, int& h, int& b) {
  const int L = blockIdx.x, nwg = gridDim.x, xcd = L & 7, q = nwg >> 3, r = nwg & 7;
  const int P = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
  blk = P % nblk;
  const int bh = P / nblk;
  h = bh % H; b = bh / H;
}

template <typename T, int D> struct AttnCfg {
  static constexpr int EPC = 16 / (int)sizeof(T);
  static constexpr int COLS = (D * (int)sizeof(T) >= 256) ? D : 256 / (int)sizeof(T);
  using Img = TileT<T, COLS>;
  static constexpr int ROWB = Img::ROWB;
  static constexpr int KS = D / 32;     // reduction steps over the head dim
  static constexpr int DT = D / 16;     // 16-wide tiles over the head dim
  static constexpr int CPR = D / EPC;   // 16-byte chunks per data row
};

// row-read fragment (8 consecutive head-dim elements of image row r)
template <int COLS>
__device__ __forceinline__ void frag_load_n(Frag<bf16_t>& f, const char* base, int r, int k0) {
  f.v = *reinterpret_cast<const bf16x8_t*>(base + TileT<bf16_t, COLS>::chunk_off(r, k0 >> 3));
}
template <int COLS>
__device__ __forceinline__ void frag_load_n(Frag<float>& f, const char* base, int r, int k0) {
  f32x4_t a = *reinterpret_cast<const f32x4_t*>(base + TileT<float, COLS>::chunk_off(r, k0 >> 2));
  f32x4_t b = *reinterpret_cast<const f32x4_t*>(base + TileT<float, COLS>::chunk_off(r, (k0 >> 2) + 1));
  f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
  f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}

// Loop-invariant per-lane LDS byte offsets of the fragment reads (bf16 path).  The swizzle depends only on row bits
// 0..3, so tiles 16 or 32 rows apart differ by an immediate: hoisting these out of the K/V loop removes ~10 VALU of
// address arithmetic per fragment read (the loops were VALU-bound: ~15 VALU per MFMA).
template <typename T, int D> struct FragOffsets {
  using Cfg = AttnCfg<T, D>;
  int rowread[Cfg::KS];        // image row (lane&15), head-dim chunk of k-step ks
  int trread[Cfg::DT];         // transposed read: image row 4*(lane>>4) + ((lane&15)>>2), columns 16*dt + 4*((lane&15)&3)
  __device__ __forceinline__ void init(int lane) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
#pragma unroll
    for (int ks = 0; ks < Cfg::KS; ++ks) rowread[ks] = Cfg::Img::chunk_off(li, (ks * 32 + 8 * g) / Cfg::EPC);
#pragma unroll
    for (int dt = 0; dt < Cfg::DT; ++dt) trread[dt] = Cfg::Img::chunk_off(4 * g + q, 2 * dt + (pp >> 1)) + ((pp & 1) << 3);
  }
};
// bf16 fragment reads from precomputed offsets: tile_rows = first image row of the 16-row tile (multiple of 16)
template <int ROWB>
__device__ __forceinline__ void frag_read_rows(Frag<bf16_t>& f, const char* img, int off, int tile_row0) {
  f.v = *reinterpret_cast<const bf16x8_t*>(img + off + tile_row0 * ROWB);
}
// transposed: elements 0..3 from image rows kb0 + 4g + q.., 4..7 from kb0 + 16 + ...  (kb0 multiple of 32)
template <int ROWB>
__device__ __forceinline__ void frag_read_tr(Frag<bf16_t>& f, const char* img, int off, int kb0) {
  typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + off + kb0 * ROWB));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + off + (kb0 + 16) * ROWB));
  s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  f.v = __builtin_bit_cast(bf16x8_t, r);
}

// Stage ROWS x D elements (rows row0.., clipped to `limit` rows -> zero fill) HBM -> registers -> LDS image.
template <typename T, int D, int ROWS> struct TileStage {
  using Cfg = AttnCfg<T, D>;
  static constexpr int NCHUNK = ROWS * Cfg::CPR;
  static constexpr int PER = (NCHUNK + 255) / 256;
  Vec16<T> r[PER];
  __device__ __forceinline__ void load(const T* base, int64_t ld, int row0, int limit, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int q = tid + 256 * i;
      const int row = q / Cfg::CPR, c = q - row * Cfg::CPR;
      if (q < NCHUNK && row0 + row < limit) r[i] = load16(base + (int64_t)(row0 + row) * ld + c * Cfg::EPC);
      else r[i] = zero16<T>();
    }
  }
  __device__ __forceinline__ void store(char* img, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int q = tid + 256 * i;
      const int row = q / Cfg::CPR, c = q - row * Cfg::CPR;
      if (q < NCHUNK) Cfg::Img::store_chunk(img, row, c, r[i]);
    }
  }
};

// accumulator pair -> B fragment: elements 0..3 from tile lo, 4..7 from tile hi
template <typename T>
__device__ __forceinline__ void frag_from_acc(Frag<T>& f, const f32x4_t& lo, const f32x4_t& hi) {
  float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  frag_from_f32(f, x);
}


#define PT_LOG2E 1.4426950408889634f
