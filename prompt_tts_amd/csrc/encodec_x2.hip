// Encodec decoder, fused stage kernels at the REFERENCE's precision (decode_codec.py:12-16 decodes in fp32): the f32-class forms
// of encodec_res1 / encodec_stage2 / encodec_tail (encodec.hip), dtype PT_BF16X2.
//
// f32-class = every product carried as a bf16 x 3 split on the bf16 MFMA (mma.h: a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi, error
// ~2^-16 per product), f32 accumulation, f32 bias / ELU.  What round 3 learned about that arithmetic (profiles/r03_decode_f32_sq.csv):
// splitting an f32 operand into hi / lo INSIDE the product loop costs a quarter of the wave's cycles in conversions.  Here an
// activation is split ONCE, by the phase that produces it, and lives -- in HBM between launches and in LDS inside them -- as two
// bf16 planes: a row of C elements is [C hi | C lo].  Consumer loops are then LDS reads and MFMAs only.  Weights arrive as f32
// and are split once per workgroup into register- (or LDS-) resident hi / lo B fragments.
//
// Launch structure = the bf16 kernels' (one workgroup per tile of input rows + recomputed halo, intermediates in LDS, next tile's
// input rows prefetched into registers under the current tile's phases), re-cut for twice the bytes per element:
//   res1   (128 ch, 600 Hz -> 3 kHz stage's residual block): 8 waves, 62 + 2 rows, k3 / 1x1 weights in registers (144 per lane)
//   stage2 (3 kHz -> 12 kHz: transposed conv k8 s4 128 -> 64 + residual block): 8 waves, 30 + 2 input rows; wave = (phase, half
//          of the output channels) of the transposed conv with its 32 hi / lo fragments in registers, k3 and 1x1 weights as
//          hi / lo fragments in LDS
//   tail   (12 -> 24 kHz: transposed conv k4 s2 64 -> 32 + residual block + final conv k7 32 -> 1): 4 waves, 56 + 8 input rows,
//          two workgroups per CU; the final conv is ONE MFMA per 16 samples (P[j][tap] = w[tap] . oute[j], then seven adds along
//          the diagonal) instead of seven with fifteen idle output rows each.
#include <type_traits>
#include "mma.h"

namespace {

// Diagnostic build (-DX2_TRACE=1, `make exp`): lane 0 of waves 0 and 5 of workgroup 0 stamps s_memtime at the phase boundaries of
// its first 8 tiles (tools/x2_trace.py prints the differences)
#ifndef X2_TRACE
#define X2_TRACE 0
#endif
#if X2_TRACE
__device__ unsigned long long x2_trace_buf[3 * 2 * 8 * 16];
#define XT_STAMP(k, w2, i) do { if (blockIdx.x == 0 && xt_it < 8 && lane == 0 && (wave == 0 || wave == (w2))) \
    x2_trace_buf[(((k) * 2 + (wave != 0)) * 8 + xt_it) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define XT_NEXT ++xt_it
#define XT_DECL int xt_it = 0
#else
#define XT_STAMP(k, w2, i) do { } while (0)
#define XT_NEXT do { } while (0)
#define XT_DECL do { } while (0)
#endif

__device__ __forceinline__ float x2_elu(float v) { return v < 0.f ? (__expf(v) - 1.f) : v; }

// 8 consecutive f32 weights -> register-resident hi / lo B fragment
__device__ __forceinline__ FragX3 x2_wfrag(const float* w) {
  Frag<float> f;
  frag_load_global(f, w);
  return split_x3(f);
}
__device__ __forceinline__ FragX3 x2_zero() {
  FragX3 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) { r.hi[j] = (__bf16)0.f; r.lo[j] = (__bf16)0.f; }
  return r;
}
// four f32 results -> 8 bytes of the hi plane and 8 bytes of the lo plane
__device__ __forceinline__ void x2_store4(char* hi, char* lo, float a, float b, float c, float d) {
  u32x2_t h, l;
  h[0] = pack_bf16x2(a, b); h[1] = pack_bf16x2(c, d);
  l[0] = pack_bf16x2(a - __uint_as_float(h[0] << 16), b - __uint_as_float(h[0] & 0xffff0000u));
  l[1] = pack_bf16x2(c - __uint_as_float(h[1] << 16), d - __uint_as_float(h[1] & 0xffff0000u));
  *reinterpret_cast<u32x2_t*>(hi) = h;
  *reinterpret_cast<u32x2_t*>(lo) = l;
}
// fragment from a row whose stride is only 8-byte aligned (two ds_read_b64)
__device__ __forceinline__ bf16x8_t x2_frag8(const char* p) {
  const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(p), hi = *reinterpret_cast<const u32x2_t*>(p + 8);
  const u32x4_t v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t x2_frag16(const char* p) { return *reinterpret_cast<const bf16x8_t*>(p); }
// LDS images of plane rows without padding: the 16-byte chunk c of row r is stored at chunk c ^ f(r).  Rows are read as B fragments
// (ds_read_b128, lane (li, g) -> row li + const, chunk g + const) and written four channels (8 bytes) at a time from product
// layout (lane (li, g) -> row li or 4 li + const, byte 8 g + const); tools/lds_model.py counts, for each candidate stride / f, the
// LDS cycles of those instructions under the bank rules of MI355X_MICROARCH.md: the 136- / 72-byte strides of this file's first
// version (8-byte aligned rows, hence ds_read2_b64 at half the read rate) cost 8 cycles a read and 16 a write (4-way conflict
// between rows 4 apart), these 4 - 7 and 8.
__device__ __forceinline__ int x2_sw128(int row, int chunk) { return row * 128 + ((chunk ^ ((row ^ (row >> 2)) & 7)) << 4); }   // 64-channel rows
__device__ __forceinline__ int x2_sw64(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 3)) << 4); }             // 32-channel rows
__device__ __forceinline__ bf16x8_t x2_bzero() {
  const u32x4_t z = {0u, 0u, 0u, 0u};
  return __builtin_bit_cast(bf16x8_t, z);
}
// ELU of 8 split elements (hi + lo), split again
__device__ __forceinline__ void x2_elu8(const u32x4_t h, const u32x4_t l, u32x4_t& eh, u32x4_t& el) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float a = x2_elu(__uint_as_float(h[k] << 16) + __uint_as_float(l[k] << 16));
    const float b = x2_elu(__uint_as_float(h[k] & 0xffff0000u) + __uint_as_float(l[k] & 0xffff0000u));
    eh[k] = pack_bf16x2(a, b);
    el[k] = pack_bf16x2(a - __uint_as_float(eh[k] << 16), b - __uint_as_float(eh[k] & 0xffff0000u));
  }
}

// =====================================================================================================================
// residual block, 128 channels:  x1 (raw planes) -> ELU -> causal conv k3 (128 -> 64) -> ELU -> 1x1 ([64 | 128] -> 128) -> ELU
// =====================================================================================================================
constexpr int XR_C = 128, XR_ROWS = 64, XR_OWN = 62, XR_XS = 288, XR_S3 = 128;   // x1 rows padded to 16 x 18 bytes, c3e rows swizzled (x2_sw128)
constexpr int XR_XPLANE = XR_ROWS * XR_XS, XR_3PLANE = XR_ROWS * XR_S3;
struct X2ResParams {
  int B, n;
  const bf16_t* x; int64_t ldx;            // [B*n][ldx]: hi at columns 0..127, lo at 128..255
  const float* w3; const float* b3;        // [64][384], [64]
  const float* wf; const float* bf;        // [128][192], [128]
  bf16_t* y; int64_t ldy;
  int tiles_per_item;
};

__global__ __launch_bounds__(512, 1) void encodec_res1_x2_kernel(const X2ResParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * XR_XPLANE + 2 * XR_3PLANE];
  char* X1r = smem;                           // [2 planes][64 rows][288 B]
  char* X1e = X1r + 2 * XR_XPLANE;            // ELU(x1); reused for the output tile
  char* C3e = X1e + 2 * XR_XPLANE;            // [2 planes][64][128 B], chunks swizzled
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
  // k3 conv: wave = (column tile nt3 = wave & 3, row half wave >> 2); 1x1: wave = column tile of the 128 outputs, all four row tiles
  const int nt3 = wave & 3, rh = wave >> 2;
  FragX3 w3[12], wf[6];
#pragma unroll
  for (int ks = 0; ks < 12; ++ks) w3[ks] = x2_wfrag(p.w3 + (16 * nt3 + li) * 384 + 32 * ks + 8 * g);
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) wf[ks] = x2_wfrag(p.wf + (16 * wave + li) * 192 + 32 * ks + 8 * g);
  float b34[4], bf4[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { b34[r] = p.b3[16 * nt3 + 4 * g + r]; bf4[r] = p.bf[16 * wave + 4 * g + r]; }

  // next tile's rows in registers: 64 rows x 16 (hi chunk, lo chunk) pairs = 1024 pairs, two per thread
  u32x4_t pfh[2], pfl[2];
  auto fetch = [&](int tile_) {
    const int b_ = tile_ / p.tiles_per_item, n0_ = (tile_ - b_ * p.tiles_per_item) * XR_OWN;
    const int tb_ = n0_ >= 2 ? n0_ - 2 : 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = tid + 512 * k, i = q >> 4, ch = q & 15, row = tb_ + i;
      pfh[k] = (u32x4_t){0u, 0u, 0u, 0u}; pfl[k] = (u32x4_t){0u, 0u, 0u, 0u};
      if (row < p.n) {
        const bf16_t* src = p.x + ((int64_t)b_ * p.n + row) * p.ldx + 8 * ch;
        pfh[k] = *reinterpret_cast<const u32x4_t*>(src); pfl[k] = *reinterpret_cast<const u32x4_t*>(src + XR_C);
      }
    }
  };
  const int n_tiles = p.B * p.tiles_per_item;
  __builtin_amdgcn_s_waitcnt(0x0F70);                            // the weight loads: not inside the loop (see encodec_tail_kernel)
  if ((int)blockIdx.x < n_tiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int b = tile / p.tiles_per_item, n0 = (tile - b * p.tiles_per_item) * XR_OWN;
    const int t_base = n0 >= 2 ? n0 - 2 : 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = tid + 512 * k, i = q >> 4, ch = q & 15;
      *reinterpret_cast<u32x4_t*>(X1r + i * XR_XS + 16 * ch) = pfh[k];
      *reinterpret_cast<u32x4_t*>(X1r + XR_XPLANE + i * XR_XS + 16 * ch) = pfl[k];
      u32x4_t eh, el;
      x2_elu8(pfh[k], pfl[k], eh, el);
      *reinterpret_cast<u32x4_t*>(X1e + i * XR_XS + 16 * ch) = eh;
      *reinterpret_cast<u32x4_t*>(X1e + XR_XPLANE + i * XR_XS + 16 * ch) = el;
    }
    __syncthreads();
    if (tile + (int)gridDim.x < n_tiles) fetch(tile + gridDim.x);
    // ---- c3e[j][16 nt3 ..] = ELU(b3 + conv k3 over ELU(x1), causal with reflect at the item start); rows 32 rh .. 32 rh + 31 ----
    // Product loops as in encodec_stage2_x2_kernel: the reads of step s + 1 are issued before the MFMAs of step s (pinned), two
    // accumulators interleaved, accumulators start at the bias.
    {
      int jj[2], tt[2];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) { jj[rr] = 16 * (2 * rh + rr) + li; tt[rr] = t_base + jj[rr]; }
      auto ld = [&](int s_, int rr) {
        const int tap = s_ >> 2, kk = s_ & 3;
        // rows before the tile belong to the halo rows of a tile that does not start an item: their results are never used, so
        // the row index is clamped instead of the load being predicated (a predicated load became a branch around every product)
        const int v = tt[rr] + tap - 2, sj0 = (v < 0 ? -v : v) - t_base, sj = sj0 < 0 ? 0 : sj0;
        FragX3 f;
        f.hi = x2_frag16(X1e + sj * XR_XS + 64 * kk + 16 * g); f.lo = x2_frag16(X1e + XR_XPLANE + sj * XR_XS + 64 * kk + 16 * g);
        return f;
      };
      f32x4_t acc[2] = {(f32x4_t){b34[0], b34[1], b34[2], b34[3]}, (f32x4_t){b34[0], b34[1], b34[2], b34[3]}};
      FragX3 f0 = ld(0, 0), f1 = ld(0, 1);
#pragma unroll
      for (int s_ = 0; s_ < 12; ++s_) {
        FragX3 n0 = f0, n1 = f1;
        if (s_ + 1 < 12) { n0 = ld(s_ + 1, 0); n1 = ld(s_ + 1, 1); }
        __builtin_amdgcn_sched_barrier(0);
        mma16x3_2b(acc[0], f0, acc[1], f1, w3[s_]);             // D[row = channel 16 nt3 + 4 g + r][col = row j]
        __builtin_amdgcn_sched_barrier(0);
        f0 = n0; f1 = n1;
      }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        char* dst = C3e + x2_sw128(jj[rr], 2 * nt3 + (g >> 1)) + 8 * (g & 1);
        x2_store4(dst, dst + XR_3PLANE, x2_elu(acc[rr][0]), x2_elu(acc[rr][1]), x2_elu(acc[rr][2]), x2_elu(acc[rr][3]));
      }
    }
    __syncthreads();
    // ---- out[j][16 wave ..] = ELU(bf + Wf [c3e[j] (64) | x1[j] (128)]) -> LDS (over X1e) ----
    f32x4_t acc[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[rt] = (f32x4_t){bf4[0], bf4[1], bf4[2], bf4[3]};
    {
      auto ld = [&](int ks, int rt) {
        const int j = 16 * rt + li;
        FragX3 f;
        if (ks < 2) { const int off = x2_sw128(j, 4 * ks + g); f.hi = x2_frag16(C3e + off); f.lo = x2_frag16(C3e + XR_3PLANE + off); }
        else { f.hi = x2_frag16(X1r + j * XR_XS + 64 * (ks - 2) + 16 * g); f.lo = x2_frag16(X1r + XR_XPLANE + j * XR_XS + 64 * (ks - 2) + 16 * g); }
        return f;
      };
      FragX3 f0 = ld(0, 0), f1 = ld(0, 1);
#pragma unroll
      for (int s_ = 0; s_ < 12; ++s_) {                            // step = (row tiles 2 (s_ / 6), + 1; ks = s_ % 6)
        const int rp = s_ / 6, ks = s_ % 6;
        FragX3 n0 = f0, n1 = f1;
        if (s_ + 1 < 12) { n0 = ld((s_ + 1) % 6, 2 * ((s_ + 1) / 6)); n1 = ld((s_ + 1) % 6, 2 * ((s_ + 1) / 6) + 1); }
        __builtin_amdgcn_sched_barrier(0);
        mma16x3_2b(acc[2 * rp], f0, acc[2 * rp + 1], f1, wf[ks]);
        __builtin_amdgcn_sched_barrier(0);
        f0 = n0; f1 = n1;
      }
    }
    // X1e was last read by the k3 conv, which every wave has left (barrier above): the output tile is staged there
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int j = 16 * rt + li;
      char* dst = X1e + j * XR_XS + (16 * wave + 4 * g) * 2;
      x2_store4(dst, dst + XR_XPLANE, x2_elu(acc[rt][0]), x2_elu(acc[rt][1]), x2_elu(acc[rt][2]), x2_elu(acc[rt][3]));
    }
    __syncthreads();
    const int j_lo = n0 - t_base;
    {                                                             // 62 rows x (16 hi + 16 lo chunks): four reads, then four stores
      const int i0 = tid >> 5, ch = tid & 31, pl = ch >> 4, c = ch & 15;
      bf16_t* yb = p.y + ((int64_t)b * p.n + n0 + i0) * p.ldy + pl * XR_C + 8 * c;
      u32x4_t o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + 16 * k < XR_OWN ? i0 + 16 * k : XR_OWN - 1;
        o[k] = *reinterpret_cast<const u32x4_t*>(X1e + pl * XR_XPLANE + (j_lo + i) * XR_XS + 16 * c);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + 16 * k;
        if (i < XR_OWN && n0 + i < p.n) *reinterpret_cast<u32x4_t*>(yb + (int64_t)(16 * k) * p.ldy) = o[k];
      }
    }
  }
}

// =====================================================================================================================
// stage 2:  xe (ELU'd planes, 128 ch) -> transposed conv k8 s4 (128 -> 64) -> x1 -> ELU -> conv k3 (64 -> 32) -> ELU
//           -> 1x1 ([32 | 64] -> 64) -> ELU -> out (planes, 64 ch, 4 n rows)
// =====================================================================================================================
constexpr int X2S_CIN = 128, X2S_C = 64, X2S_RIN = 30, X2S_HALO = 2, X2S_RI = 32, X2S_RO = 128;
// every fragment is ONE ds_read_b128 (ds_read2_b64, what two 8-byte reads become, runs at half its rate, and the k3 / 1x1 phases
// were bound by the LDS: tools/x2_trace.py); input / weight rows are padded to 16 bytes x (2 mod 4), the conflict-free strides for
// the fragment read, x1 / c3e rows are unpadded and chunk-swizzled (x2_sw128 / x2_sw64)
constexpr int X2S_XS = 288, X2S_S1 = 128, X2S_S3 = 64, X2S_W3S = 416, X2S_W3PLANE = 32 * X2S_W3S, X2S_WFS = 224, X2S_WFPLANE = 64 * X2S_WFS;
constexpr int X2S_XPLANE = X2S_RI * X2S_XS, X2S_1PLANE = X2S_RO * X2S_S1, X2S_3PLANE = X2S_RO * X2S_S3;
struct X2StageParams {
  int B, n;
  const bf16_t* x; int64_t ldx;            // [B*n][ldx]: hi 0..127, lo 128..255
  const float* wt; const float* bt;        // [256][256], [256]  (row rho*64+co, col tap*128+ci; bias repeated per phase)
  const float* w3; const float* b3;        // [32][192],  [32]
  const float* wf; const float* bf;        // [64][96],   [64]
  bf16_t* y; int64_t ldy;                  // [B*4n][ldy]: hi 0..63, lo 64..127
  int tiles_per_item;
};

__global__ __launch_bounds__(512, 1) void encodec_stage2_x2_kernel(const X2StageParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * X2S_XPLANE + 4 * X2S_1PLANE + 2 * X2S_3PLANE + 2 * X2S_W3PLANE + 2 * X2S_WFPLANE + (256 + 32 + 64) * 4 + 64];
  char* Xin = smem;                           // [2][32][288]
  char* X1r = Xin + 2 * X2S_XPLANE;           // [2][128][128], chunks swizzled
  char* X1e = X1r + 2 * X2S_1PLANE;           // the same; reused for the output tile
  char* C3e = X1e + 2 * X2S_1PLANE;           // [2][128][64], chunks swizzled
  char* W3s = C3e + 2 * X2S_3PLANE;           // [2][32 rows][416 B]: hi / lo of the k3 weights
  char* Wfs = W3s + 2 * X2S_W3PLANE;          // [2][64 rows][224 B]: hi / lo of the 1x1 weights
  float* Bts = reinterpret_cast<float*>(Wfs + 2 * X2S_WFPLANE);   // biases (read where they are added: 16 registers a lane)
  float* B3s = Bts + 256;
  float* Bfs = B3s + 32;
  const char* Zr = reinterpret_cast<const char*>(Bfs + 64);     // 64 zero bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
  // transposed conv: wave = (phase rho, half h of the 64 output channels); rows 64 rho + 32 h + 16 nt + li of Wt
  const int rho = wave >> 1, hh = wave & 1;
  FragX3 wt[2][8];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) wt[nt][ks] = x2_wfrag(p.wt + (64 * rho + 32 * hh + 16 * nt + li) * 256 + 32 * ks + 8 * g);
  for (int q = tid; q < 256 + 32 + 64; q += 512) Bts[q] = q < 256 ? p.bt[q] : (q < 288 ? p.b3[q - 256] : p.bf[q - 288]);
  if (tid < 16) Bfs[64 + tid] = 0.f;
  // k3 conv: wave = (column tile nt3 = wave & 1, row tiles 2 (wave >> 1), +1); weights hi / lo in LDS (in registers they left the product loops no room to read a step ahead)
  const int nt3 = wave & 1, rg3 = wave >> 1;
  for (int q = tid; q < 32 * 24; q += 512) {
    const int row = q / 24, ch = q - row * 24;
    Frag<float> f;
    frag_load_global(f, p.w3 + row * 192 + 8 * ch);
    const FragX3 s = split_x3(f);
    *reinterpret_cast<bf16x8_t*>(W3s + row * X2S_W3S + 16 * ch) = s.hi;
    *reinterpret_cast<bf16x8_t*>(W3s + X2S_W3PLANE + row * X2S_W3S + 16 * ch) = s.lo;
  }
  // 1x1: wave = (column tile ntf = wave & 3, row tiles 4 (wave >> 2) .. + 3); weights hi / lo in LDS
  const int ntf = wave & 3, rgf = wave >> 2;
  for (int q = tid; q < 64 * 12; q += 512) {
    const int row = q / 12, ch = q - row * 12;
    Frag<float> f;
    frag_load_global(f, p.wf + row * 96 + 8 * ch);
    const FragX3 s = split_x3(f);
    *reinterpret_cast<bf16x8_t*>(Wfs + row * X2S_WFS + 16 * ch) = s.hi;
    *reinterpret_cast<bf16x8_t*>(Wfs + X2S_WFPLANE + row * X2S_WFS + 16 * ch) = s.lo;
  }
  const int n_out = 4 * p.n;

  // next tile's input rows: 32 rows x 16 (hi, lo) chunk pairs = 512 pairs, one per thread
  u32x4_t pfh, pfl;
  auto fetch = [&](int tile_) {
    const int b_ = tile_ / p.tiles_per_item, n0_ = (tile_ - b_ * p.tiles_per_item) * X2S_RIN;
    const int ni0_ = n0_ >= X2S_HALO ? n0_ - X2S_HALO : 0;
    const int i = tid >> 4, ch = tid & 15, nrow = ni0_ + i;
    pfh = (u32x4_t){0u, 0u, 0u, 0u}; pfl = (u32x4_t){0u, 0u, 0u, 0u};
    if (nrow < p.n) {
      const bf16_t* src = p.x + ((int64_t)b_ * p.n + nrow) * p.ldx + 8 * ch;
      pfh = *reinterpret_cast<const u32x4_t*>(src); pfl = *reinterpret_cast<const u32x4_t*>(src + X2S_CIN);
    }
  };
  const int n_tiles = p.B * p.tiles_per_item;
  XT_DECL;
  __builtin_amdgcn_s_waitcnt(0x0F70);
  // The prefetched rows go to LDS at the END of the previous tile, between its last barrier and its output stores: loads and
  // stores share one vmcnt, the store count of a tile is not a compile-time constant, so a copy placed after the stores waited
  // for every store's acknowledgement (tools/x2_trace.py: ~500 cycles a tile).  Xin is dead from the transposed conv on.
  auto fill = [&]() {
    *reinterpret_cast<u32x4_t*>(Xin + (tid >> 4) * X2S_XS + 16 * (tid & 15)) = pfh;
    *reinterpret_cast<u32x4_t*>(Xin + X2S_XPLANE + (tid >> 4) * X2S_XS + 16 * (tid & 15)) = pfl;
  };
  if ((int)blockIdx.x < n_tiles) { fetch(blockIdx.x); fill(); }
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int b = tile / p.tiles_per_item, n0 = (tile - b * p.tiles_per_item) * X2S_RIN;
    const int ni0 = n0 >= X2S_HALO ? n0 - X2S_HALO : 0;
    const int t_base = 4 * ni0;
    XT_STAMP(1, 5, 0);
    __syncthreads();                                              // Xin is filled; the previous tile's staged output has been read
    XT_STAMP(1, 5, 1);
    if (tile + (int)gridDim.x < n_tiles) fetch(tile + gridDim.x);
    // Every product loop below is an explicit two-stage pipeline -- the LDS reads of step s + 1 are issued, THEN the six MFMAs of
    // step s (two independent accumulators, interleaved) -- pinned with sched_barrier: left alone, hipcc (at 254 registers) emitted
    // read -> s_waitcnt 0 -> three dependent MFMAs per step, and tools/x2_trace.py measured 36 % MFMA occupancy in the k3 and 1x1
    // phases.
    // ---- transposed conv: x1[4 i + rho][32 h + co] = bt + sum_tap sum_ci xe[i - tap][ci] Wt[rho*64 + 32 h + co][tap*128 + ci] ----
    {
      auto ld = [&](int s_) {
        const int rt_ = s_ >> 3, ks_ = s_ & 7, src = 16 * rt_ + li - (ks_ >> 2);
        // xe[-1] = 0 at the start of an item: the address moves to 64 zero bytes (a predicated load became a branch per product)
        const char* ph = src >= 0 ? Xin + src * X2S_XS + (ks_ & 3) * 64 + 16 * g : Zr + 16 * g;
        FragX3 f;
        f.hi = x2_frag16(ph); f.lo = x2_frag16(src >= 0 ? ph + X2S_XPLANE : ph);
        return f;
      };
      FragX3 fa = ld(0);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        // accumulators start at the bias: read from LDS here, under the first fragments' latency (read in the epilogue, each bias
        // fetch was an exposed LDS round trip: tools/x2_trace.py)
        f32x4_t acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[nt] = *reinterpret_cast<const f32x4_t*>(Bts + 64 * rho + 32 * hh + 16 * nt + 4 * g);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          FragX3 fn = fa;
          if (8 * rt + ks + 1 < 16) fn = ld(8 * rt + ks + 1);
          __builtin_amdgcn_sched_barrier(0);
          mma16x3_2a(acc[0], wt[0][ks], acc[1], wt[1][ks], fa);
          __builtin_amdgcn_sched_barrier(0);
          fa = fn;
        }
#if X2_TRACE
        asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][0]));     // the stamp waits for the block's last MFMAs
        XT_STAMP(1, 5, 9 + 2 * rt);
#endif
        const int orow = 4 * (16 * rt + li) + rho;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[nt][r];
          const int off = x2_sw128(orow, 4 * hh + 2 * nt + (g >> 1)) + 8 * (g & 1);           // channels 32 hh + 16 nt + 4 g ..
          char* dr = X1r + off;
          x2_store4(dr, dr + X2S_1PLANE, v[0], v[1], v[2], v[3]);
          char* de = X1e + off;
          x2_store4(de, de + X2S_1PLANE, x2_elu(v[0]), x2_elu(v[1]), x2_elu(v[2]), x2_elu(v[3]));
        }
#if X2_TRACE
        if (rt == 0) XT_STAMP(1, 5, 10);
#endif
      }
    }
    XT_STAMP(1, 5, 2);
    __syncthreads();
    XT_STAMP(1, 5, 3);
    // ---- c3e[j][16 nt3 ..] = ELU(b3 + conv k3 over ELU(x1)); the wave's two row tiles are the two accumulators ----
    {
      int jj[2], tt[2];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) { jj[rr] = 16 * (2 * rg3 + rr) + li; tt[rr] = t_base + jj[rr]; }
      auto ld = [&](int s_, int rr) {
        const int tap = s_ >> 1, kk = s_ & 1;
        // rows before the tile belong to the halo rows of a tile that does not start an item: their results are never used, so
        // the row index is clamped instead of the load being predicated
        const int v = tt[rr] + tap - 2, sj0 = (v < 0 ? -v : v) - t_base, sj = sj0 < 0 ? 0 : sj0;
        FragX3 f;
        const int off = x2_sw128(sj, 4 * kk + g);
        f.hi = x2_frag16(X1e + off); f.lo = x2_frag16(X1e + X2S_1PLANE + off);
        return f;
      };
      auto ldw = [&](int s_) {
        FragX3 w;
        w.hi = x2_frag16(W3s + (16 * nt3 + li) * X2S_W3S + (32 * s_ + 8 * g) * 2);
        w.lo = x2_frag16(W3s + X2S_W3PLANE + (16 * nt3 + li) * X2S_W3S + (32 * s_ + 8 * g) * 2);
        return w;
      };
      f32x4_t acc[2];                                             // start at the bias (see the transposed conv)
      acc[0] = *reinterpret_cast<const f32x4_t*>(B3s + 16 * nt3 + 4 * g); acc[1] = acc[0];
      FragX3 wb = ldw(0), f0 = ld(0, 0), f1 = ld(0, 1);
#pragma unroll
      for (int s_ = 0; s_ < 6; ++s_) {
        FragX3 n0 = f0, n1 = f1, wn = wb;
        if (s_ + 1 < 6) { wn = ldw(s_ + 1); n0 = ld(s_ + 1, 0); n1 = ld(s_ + 1, 1); }
        __builtin_amdgcn_sched_barrier(0);
        mma16x3_2b(acc[0], f0, acc[1], f1, wb);
        __builtin_amdgcn_sched_barrier(0);
        f0 = n0; f1 = n1; wb = wn;
      }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        char* dst = C3e + x2_sw64(jj[rr], 2 * nt3 + (g >> 1)) + 8 * (g & 1);
        x2_store4(dst, dst + X2S_3PLANE, x2_elu(acc[rr][0]), x2_elu(acc[rr][1]), x2_elu(acc[rr][2]), x2_elu(acc[rr][3]));
      }
    }
    XT_STAMP(1, 5, 4);
    __syncthreads();
    XT_STAMP(1, 5, 5);
    // ---- out[j][16 ntf ..] = ELU(bf + Wf [c3e[j] (32) | x1[j] (64)]); rows 64 rgf .. 64 rgf + 63; staged over X1e ----
    f32x4_t acc[4];                                               // start at the bias (see the transposed conv)
    acc[0] = *reinterpret_cast<const f32x4_t*>(Bfs + 16 * ntf + 4 * g);
#pragma unroll
    for (int rt = 1; rt < 4; ++rt) acc[rt] = acc[0];
    {
      auto ld = [&](int ks, int rt) {
        const int j = 16 * (4 * rgf + rt) + li;
        FragX3 f;
        if (ks == 0) { const int off = x2_sw64(j, g); f.hi = x2_frag16(C3e + off); f.lo = x2_frag16(C3e + X2S_3PLANE + off); }
        else { const int off = x2_sw128(j, 4 * (ks - 1) + g); f.hi = x2_frag16(X1r + off); f.lo = x2_frag16(X1r + X2S_1PLANE + off); }
        return f;
      };
      auto ldw = [&](int ks) {
        FragX3 w;
        w.hi = x2_frag16(Wfs + (16 * ntf + li) * X2S_WFS + 64 * ks + 16 * g);
        w.lo = x2_frag16(Wfs + X2S_WFPLANE + (16 * ntf + li) * X2S_WFS + 64 * ks + 16 * g);
        return w;
      };
      FragX3 wb = ldw(0), f0 = ld(0, 0), f1 = ld(0, 1);
#pragma unroll
      for (int s_ = 0; s_ < 6; ++s_) {                             // step = (ks = s_ >> 1, row tiles 2 (s_ & 1), + 1)
        const int ks = s_ >> 1, rp = s_ & 1;
        FragX3 n0 = f0, n1 = f1, wn = wb;
        if (s_ + 1 < 6) {
          n0 = ld((s_ + 1) >> 1, 2 * ((s_ + 1) & 1)); n1 = ld((s_ + 1) >> 1, 2 * ((s_ + 1) & 1) + 1);
          if (((s_ + 1) >> 1) != ks) wn = ldw((s_ + 1) >> 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma16x3_2b(acc[2 * rp], f0, acc[2 * rp + 1], f1, wb);
        __builtin_amdgcn_sched_barrier(0);
        f0 = n0; f1 = n1; wb = wn;
      }
    }
    // X1e was last read by the k3 conv, which every wave has left (barrier above); the output tile is staged there so that HBM
    // sees whole 128-byte plane rows
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int j = 16 * (4 * rgf + rt) + li;
      char* dst = X1e + x2_sw128(j, 2 * ntf + (g >> 1)) + 8 * (g & 1);
      x2_store4(dst, dst + X2S_1PLANE, x2_elu(acc[rt][0]), x2_elu(acc[rt][1]), x2_elu(acc[rt][2]), x2_elu(acc[rt][3]));
    }
    XT_STAMP(1, 5, 6);
    __syncthreads();
    XT_STAMP(1, 5, 7);
    const int j_lo = 4 * (n0 - ni0);
#if X2_TRACE
    XT_STAMP(1, 5, 12);
#endif
    if (tile + (int)gridDim.x < n_tiles) fill();                 // the next tile's rows (fetched at the top of this one)
#if X2_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    XT_STAMP(1, 5, 13);
#endif
    {                                                             // 120 rows x (8 hi + 8 lo chunks): four reads, then four stores
      const int i0 = tid >> 4, ch = tid & 15, pl = ch >> 3, c = ch & 7;
      bf16_t* yb = p.y + ((int64_t)b * n_out + 4 * n0 + i0) * p.ldy + pl * X2S_C + 8 * c;
      bf16x8_t o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + 32 * k < 4 * X2S_RIN ? i0 + 32 * k : 4 * X2S_RIN - 1;
        o[k] = x2_frag16(X1e + pl * X2S_1PLANE + x2_sw128(j_lo + i, c));
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + 32 * k;
        if (i < 4 * X2S_RIN && 4 * n0 + i < n_out) *reinterpret_cast<bf16x8_t*>(yb + (int64_t)(32 * k) * p.ldy) = o[k];
      }
    }
    XT_STAMP(1, 5, 8);
    XT_NEXT;
  }
}

// =====================================================================================================================
// tail:  xe (ELU'd planes, 64 ch, 12 kHz) -> transposed conv k4 s2 (64 -> 32) -> x1 -> ELU -> conv k3 (32 -> 16) -> ELU
//        -> 1x1 ([16 | 32] -> 32) -> ELU -> conv k7 (32 -> 1) -> waveform f32
// =====================================================================================================================
constexpr int X2T_CIN = 64, X2T_C = 32, X2T_RIN = 56, X2T_HALO = 8, X2T_RI = 64, X2T_RO = 128;
constexpr int X2T_XS = 144, X2T_S64 = 72, X2T_S32 = 40, X2T_W3S = 208;
constexpr int X2T_XPLANE = X2T_RI * X2T_XS, X2T_1PLANE = X2T_RO * X2T_S64, X2T_3PLANE = X2T_RO * X2T_S32, X2T_WPLANE = 16 * X2T_W3S;
static_assert(X2T_XPLANE == X2T_1PLANE, "Oute aliases Xin");
static_assert(2 * X2T_3PLANE >= X2T_RO * 8 * 4, "the final conv's tap products alias C3e");
struct X2TailParams {
  int B, n;
  const bf16_t* x; int64_t ldx;            // [B*n][ldx]: hi 0..63, lo 64..127
  const float* wt; const float* bt;        // [64][128], [64]
  const float* w3; const float* b3;        // [16][96],  [16]
  const float* wf; const float* bf;        // [32][64] (48 used), [32]
  const float* wfin; const float* bfin;    // [1][224], [1]
  float* wav;
  int tiles_per_item;
};

__global__ __launch_bounds__(256, 2) void encodec_tail_x2_kernel(const X2TailParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * X2T_XPLANE + 4 * X2T_1PLANE + 2 * X2T_3PLANE + 2 * X2T_WPLANE + 256 + 64 + 2048];
  char* Xin = smem;                           // [2][64][144]
  char* X1r = Xin + 2 * X2T_XPLANE;           // [2][128][72]
  char* X1e = X1r + 2 * X2T_1PLANE;           // [2][128][72]
  char* C3e = X1e + 2 * X2T_1PLANE;           // [2][128][40]
  char* W3s = C3e + 2 * X2T_3PLANE;           // [2][16][208]
  float* Bts = reinterpret_cast<float*>(W3s + 2 * X2T_WPLANE);
  char* Wfin = reinterpret_cast<char*>(Bts + 80);   // [hi | lo][64 lanes][16 B]: the final conv's A fragment (after 64 biases + 64 zero bytes)
  char* Oute = Xin;                           // [2][128][72], written in phase D (Xin is dead after phase B)
  float* Ptap = reinterpret_cast<float*>(C3e);   // [128][8] f32, written in phase E (C3e is dead after phase D)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
  FragX3 wt[4][4], wf[2][2];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wt[nt][ks] = x2_wfrag(p.wt + (16 * nt + li) * 128 + 32 * ks + 8 * g);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wf[nt][ks] = x2_wfrag(p.wf + (16 * nt + li) * 64 + 32 * ks + 8 * g);
  // final conv as ONE product per row tile: A row = tap (7 of 16 rows), K = the 32 channels
  // (its weights, used once a tile, wait in LDS as hi / lo fragments in lane order: 8 registers fewer held through every phase)
  if (tid < 64) {
    FragX3 w = x2_zero();
    if (li < 7) w = x2_wfrag(p.wfin + 32 * li + 8 * g);
    *reinterpret_cast<bf16x8_t*>(Wfin + 16 * tid) = w.hi;
    *reinterpret_cast<bf16x8_t*>(Wfin + 1024 + 16 * tid) = w.lo;
  }
  for (int q = tid; q < 16 * 12; q += 256) {
    const int row = q / 12, ch = q - row * 12;
    Frag<float> f;
    frag_load_global(f, p.w3 + row * 96 + 8 * ch);
    const FragX3 s = split_x3(f);
    *reinterpret_cast<bf16x8_t*>(W3s + row * X2T_W3S + 16 * ch) = s.hi;
    *reinterpret_cast<bf16x8_t*>(W3s + X2T_WPLANE + row * X2T_W3S + 16 * ch) = s.lo;
  }
  if (tid < 64) Bts[tid] = p.bt[tid];
  const char* Zr = reinterpret_cast<const char*>(Bts + 64);     // 64 zero bytes
  if (tid < 16) Bts[64 + tid] = 0.f;
  float b34[4], bf4[2][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { b34[r] = p.b3[4 * g + r]; bf4[0][r] = p.bf[4 * g + r]; bf4[1][r] = p.bf[16 + 4 * g + r]; }
  const float bfin = p.bfin[0];
  const int n_out = 2 * p.n;

  // next tile's input rows: 64 rows x 8 (hi, lo) chunk pairs = 512 pairs, two per thread
  u32x4_t pfh[2], pfl[2];
  auto fetch = [&](int tile_) {
    const int b_ = tile_ / p.tiles_per_item, n0_ = (tile_ - b_ * p.tiles_per_item) * X2T_RIN;
    const int ni0_ = n0_ >= X2T_HALO ? n0_ - X2T_HALO : 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = tid + 256 * k, i = q >> 3, ch = q & 7, nrow = ni0_ + i;
      pfh[k] = (u32x4_t){0u, 0u, 0u, 0u}; pfl[k] = (u32x4_t){0u, 0u, 0u, 0u};
      if (nrow < p.n) {
        const bf16_t* src = p.x + ((int64_t)b_ * p.n + nrow) * p.ldx + 8 * ch;
        pfh[k] = *reinterpret_cast<const u32x4_t*>(src); pfl[k] = *reinterpret_cast<const u32x4_t*>(src + X2T_CIN);
      }
    }
  };
  const int n_tiles = p.B * p.tiles_per_item;
  XT_DECL;
  __builtin_amdgcn_s_waitcnt(0x0F70);
  // the prefetched rows go to LDS before the previous tile's waveform stores (see encodec_stage2_x2_kernel): Xin = Oute is dead
  // once phase E has been left
  auto fill = [&]() {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = tid + 256 * k;
      *reinterpret_cast<u32x4_t*>(Xin + (q >> 3) * X2T_XS + 16 * (q & 7)) = pfh[k];
      *reinterpret_cast<u32x4_t*>(Xin + X2T_XPLANE + (q >> 3) * X2T_XS + 16 * (q & 7)) = pfl[k];
    }
  };
  if ((int)blockIdx.x < n_tiles) { fetch(blockIdx.x); fill(); }
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int b = tile / p.tiles_per_item, n0 = (tile - b * p.tiles_per_item) * X2T_RIN;
    const int ni0 = n0 >= X2T_HALO ? n0 - X2T_HALO : 0;          // first input row held in LDS
    const int t_base = 2 * ni0;                                   // output row of LDS row 0 of X1 / C3e / Oute
    XT_STAMP(2, 3, 0);
    __syncthreads();                                              // Xin is filled; the previous tile's tap products have been read
    XT_STAMP(2, 3, 1);
    if (tile + (int)gridDim.x < n_tiles) fetch(tile + gridDim.x);
    // ---- B: transposed conv: x1[2 i + rho][co] = bt + sum_tap sum_ci xe[i - tap][ci] Wt[rho*32 + co][tap*64 + ci]; row tile = wave ----
    {
      f32x4_t acc[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = *reinterpret_cast<const f32x4_t*>(Bts + 16 * nt + 4 * g);     // start at the bias
      const int i = 16 * wave + li;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int src = i - (ks >> 1);
        FragX3 fa;
        const char* ph = src >= 0 ? Xin + src * X2T_XS + (ks & 1) * 64 + 16 * g : Zr + 16 * g;     // xe[-1] = 0: 64 zero bytes
        fa.hi = x2_frag16(ph); fa.lo = x2_frag16(src >= 0 ? ph + X2T_XPLANE : ph);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) mma16x3(acc[nt], wt[nt][ks], fa);          // D[row = out column][col = input row]
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {                                            // columns 16 nt + 4 g + r: rho = nt >> 1
        const int orow = 2 * i + (nt >> 1), co = 16 * (nt & 1) + 4 * g;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[nt][r];
        char* dr = X1r + orow * X2T_S64 + co * 2;
        x2_store4(dr, dr + X2T_1PLANE, v[0], v[1], v[2], v[3]);
        char* de = X1e + orow * X2T_S64 + co * 2;
        x2_store4(de, de + X2T_1PLANE, x2_elu(v[0]), x2_elu(v[1]), x2_elu(v[2]), x2_elu(v[3]));
      }
    }
    XT_STAMP(2, 3, 2);
    __syncthreads();
    XT_STAMP(2, 3, 3);
    // ---- C: c3e[j] = ELU(b3 + conv k3 over ELU(x1), causal with reflect at the item start) ----
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      f32x4_t acc = (f32x4_t){b34[0], b34[1], b34[2], b34[3]};
      const int j = 16 * (wave + 4 * rr) + li, t = t_base + j;
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) {
        const int v = t + tap - 2, sj0 = (v < 0 ? -v : v) - t_base, sj = sj0 < 0 ? 0 : sj0;     // clamped: see encodec_res1_x2_kernel
        FragX3 fa, wb;
        fa.hi = x2_frag8(X1e + sj * X2T_S64 + 16 * g); fa.lo = x2_frag8(X1e + X2T_1PLANE + sj * X2T_S64 + 16 * g);
        wb.hi = x2_frag16(W3s + li * X2T_W3S + 64 * tap + 16 * g); wb.lo = x2_frag16(W3s + X2T_WPLANE + li * X2T_W3S + 64 * tap + 16 * g);
        mma16x3(acc, wb, fa);
      }
      char* dst = C3e + j * X2T_S32 + 8 * g;
      x2_store4(dst, dst + X2T_3PLANE, x2_elu(acc[0]), x2_elu(acc[1]), x2_elu(acc[2]), x2_elu(acc[3]));
    }
    XT_STAMP(2, 3, 4);
    __syncthreads();
    XT_STAMP(2, 3, 5);
    // ---- D: oute[j] = ELU(bf + Wf [c3e[j] (16) | x1[j] (32)]) ----
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      f32x4_t acc[2] = {(f32x4_t){bf4[0][0], bf4[0][1], bf4[0][2], bf4[0][3]}, (f32x4_t){bf4[1][0], bf4[1][1], bf4[1][2], bf4[1][3]}};
      const int j = 16 * (wave + 4 * rr) + li;
      FragX3 f0, f1;
      if (g < 2) {                                                                // k 0..15 = c3e | 16..31 = x1[0..15]
        f0.hi = x2_frag8(C3e + j * X2T_S32 + 16 * g); f0.lo = x2_frag8(C3e + X2T_3PLANE + j * X2T_S32 + 16 * g);
        f1.hi = x2_frag8(X1r + j * X2T_S64 + 32 + 16 * g); f1.lo = x2_frag8(X1r + X2T_1PLANE + j * X2T_S64 + 32 + 16 * g);   // k 32..47 = x1[16..31]
      } else {
        f0.hi = x2_frag8(X1r + j * X2T_S64 + 16 * (g - 2)); f0.lo = x2_frag8(X1r + X2T_1PLANE + j * X2T_S64 + 16 * (g - 2));
        f1 = x2_zero();
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) { mma16x3(acc[nt], wf[nt][0], f0); mma16x3(acc[nt], wf[nt][1], f1); }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        char* dst = Oute + j * X2T_S64 + (16 * nt + 4 * g) * 2;
        x2_store4(dst, dst + X2T_1PLANE, x2_elu(acc[nt][0]), x2_elu(acc[nt][1]), x2_elu(acc[nt][2]), x2_elu(acc[nt][3]));
      }
    }
    XT_STAMP(2, 3, 6);
    __syncthreads();
    XT_STAMP(2, 3, 7);
    // ---- E: P[j][tap] = sum_c wfin[tap][c] oute[j][c]  (one product per row tile: D[row = tap][col = row j]) ----
    FragX3 wfe;
    wfe.hi = x2_frag16(Wfin + 16 * lane); wfe.lo = x2_frag16(Wfin + 1024 + 16 * lane);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      const int j = 16 * (wave + 4 * rr) + li;
      FragX3 fa;
      fa.hi = x2_frag8(Oute + j * X2T_S64 + 16 * g); fa.lo = x2_frag8(Oute + X2T_1PLANE + j * X2T_S64 + 16 * g);
      mma16x3(acc, wfe, fa);
      if (g < 2) *reinterpret_cast<f32x4_t*>(Ptap + j * 8 + 4 * g) = acc;        // taps 4 g + r
    }
    XT_STAMP(2, 3, 8);
    __syncthreads();
    XT_STAMP(2, 3, 9);
    // ---- F: wav[t] = bfin + sum_tap P[reflect(t + tap - 6)][tap]; only this tile's own 112 samples are written ----
    const int j_lo = 2 * (n0 - ni0);
    if (tile + (int)gridDim.x < n_tiles) fill();
    if (tid < 2 * X2T_RIN) {
      const int j = j_lo + tid, t = t_base + j;
      if (t < n_out) {
        float s = bfin;
#pragma unroll
        for (int tap = 0; tap < 7; ++tap) {
          const int v = t + tap - 6, sj = (v < 0 ? -v : v) - t_base;
          s += Ptap[sj * 8 + tap];
        }
        p.wav[(int64_t)b * n_out + t] = s;
      }
    }
    XT_STAMP(2, 3, 10);
    XT_NEXT;
  }
}

}  // namespace

// entry points: called by pt_encodec_res / pt_encodec_stage / pt_encodec_tail (encodec.hip) for dtype PT_BF16X2; the descriptors'
// weight pointers are then f32 matrices in the same [rows][K] forms, x / y are plane rows
int pt_x2_encodec_res(const pt_encodec_stage_desc* d, hipStream_t s) {
  if (d->B <= 0 || d->n < 3 || d->cin != XR_C || d->cout != XR_C || d->r != 1 || d->B * d->n >= (1ll << 30)) return PT_ERR_SHAPE;
  if (!d->x || !d->w3 || !d->b3 || !d->wf || !d->bf || !d->y) return PT_ERR_ARG;
  if (!pt_aligned16(d->x) || (d->ldx * 2) % 16 || d->ldx < 2 * XR_C || !pt_aligned16(d->w3) || !pt_aligned16(d->wf) || !pt_aligned16(d->y) ||
      (d->ldy * 2) % 16 || d->ldy < 2 * XR_C) return PT_ERR_ALIGN;
  X2ResParams p;
  p.B = (int)d->B; p.n = (int)d->n; p.x = (const bf16_t*)d->x; p.ldx = d->ldx;
  p.w3 = (const float*)d->w3; p.b3 = d->b3; p.wf = (const float*)d->wf; p.bf = d->bf;
  p.y = (bf16_t*)d->y; p.ldy = d->ldy;
  p.tiles_per_item = (int)((d->n + XR_OWN - 1) / XR_OWN);
  const int64_t tiles = (int64_t)p.B * p.tiles_per_item;
  const unsigned grid = (unsigned)(tiles < 256 * 8 ? tiles : 256 * 8);
  hipLaunchKernelGGL(encodec_res1_x2_kernel, dim3(grid), dim3(512), 0, s, p);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

int pt_x2_encodec_stage(const pt_encodec_stage_desc* d, hipStream_t s) {
  if (d->B <= 0 || d->n < 4 || d->cin != X2S_CIN || d->cout != X2S_C || d->r != 4 || d->B * d->n >= (1ll << 29)) return PT_ERR_SHAPE;
  if (!d->x || !d->wt || !d->bt || !d->w3 || !d->b3 || !d->wf || !d->bf || !d->y) return PT_ERR_ARG;
  if (!pt_aligned16(d->x) || (d->ldx * 2) % 16 || d->ldx < 2 * X2S_CIN || !pt_aligned16(d->wt) || !pt_aligned16(d->w3) || !pt_aligned16(d->wf) ||
      !pt_aligned16(d->y) || d->ldy % 8 || d->ldy < 2 * X2S_C) return PT_ERR_ALIGN;
  X2StageParams p;
  p.B = (int)d->B; p.n = (int)d->n; p.x = (const bf16_t*)d->x; p.ldx = d->ldx;
  p.wt = (const float*)d->wt; p.bt = d->bt; p.w3 = (const float*)d->w3; p.b3 = d->b3; p.wf = (const float*)d->wf; p.bf = d->bf;
  p.y = (bf16_t*)d->y; p.ldy = d->ldy;
  p.tiles_per_item = (int)((d->n + X2S_RIN - 1) / X2S_RIN);
  const int64_t tiles = (int64_t)p.B * p.tiles_per_item;
  const unsigned grid = (unsigned)(tiles < 256 * 8 ? tiles : 256 * 8);
  hipLaunchKernelGGL(encodec_stage2_x2_kernel, dim3(grid), dim3(512), 0, s, p);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

int pt_x2_encodec_tail(const pt_encodec_tail_desc* d, hipStream_t s) {
  if (d->B <= 0 || d->n < 8 || d->cin != X2T_CIN || d->cout != X2T_C || d->r != 2 || d->B * d->n >= (1ll << 30)) return PT_ERR_SHAPE;
  if (!d->x || !d->wt || !d->bt || !d->w3 || !d->b3 || !d->wf || !d->bf || !d->wfin || !d->bfin || !d->wav) return PT_ERR_ARG;
  if (!pt_aligned16(d->x) || (d->ldx * 2) % 16 || d->ldx < 2 * X2T_CIN || !pt_aligned16(d->wt) || !pt_aligned16(d->w3) || !pt_aligned16(d->wf) ||
      !pt_aligned16(d->wfin)) return PT_ERR_ALIGN;
  X2TailParams p;
  p.B = (int)d->B; p.n = (int)d->n; p.x = (const bf16_t*)d->x; p.ldx = d->ldx;
  p.wt = (const float*)d->wt; p.bt = d->bt; p.w3 = (const float*)d->w3; p.b3 = d->b3; p.wf = (const float*)d->wf; p.bf = d->bf;
  p.wfin = (const float*)d->wfin; p.bfin = d->bfin; p.wav = d->wav;
  p.tiles_per_item = (int)((d->n + X2T_RIN - 1) / X2T_RIN);
  const int64_t tiles = (int64_t)p.B * p.tiles_per_item;
  const unsigned grid = (unsigned)(tiles < 256 * 2 * 8 ? tiles : 256 * 2 * 8);
  hipLaunchKernelGGL(encodec_tail_x2_kernel, dim3(grid), dim3(256), 0, s, p);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

#if X2_TRACE
extern "C" int pt_debug_x2_trace(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(x2_trace_buf), sizeof(unsigned long long) * (n < 768 ? n : 768)) == hipSuccess ? 0 : -3;
}
#endif
