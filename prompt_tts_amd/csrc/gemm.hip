// GEMM family for gfx950: C[m][n] (+)= sum_k VA(m,k) VB(n,k), 128x128 workgroup tile, 4 waves (2x2) of
// 64x64, MFMA 16x16 atoms (mma.h).  Operand tiles go HBM -> LDS directly (global_load_lds_dwordx4, no VGPR
// staging): the LDS image is lane-linear per wave-instruction, so the XOR swizzle is applied to the per-lane
// SOURCE address (and again on the fragment read); chunks that must read as zero (conv padding, M/N/K tails)
// point at a zero page.  Two LDS stages: the loads of k-tile t+1 are in flight under the MFMAs of k-tile t,
// one barrier per k-tile; 72 KiB LDS and <= 256 registers keep two workgroups resident per CU, so one workgroup's
// prologue/epilogue overlaps the other's main loop (most GEMMs here have K = 512 = 8 k-tiles).
// Measured with tools/gemm_probe.py: address generation (not memory) bounded the staging path and 8-byte scattered
// stores the epilogue; hence per-chunk source pointers advanced by a byte step (recomputed only at segment
// boundaries: new conv tap, concat half, K tail) and an LDS-transposed epilogue that stores whole 128-byte rows.
//
// Operands are "virtual matrices" (include/prompt_tts_hip.h): plain, channel-concat, conv-gather (implicit
// GEMM for Conv1d k=3 stride 1/2, upsample+conv, and their dgrad/wgrad) and flipped conv weights; each is
// consumed either with the reduction index along its columns (TileK image) or along its rows (TileT image,
// transposed fragment reads), so forward, dgrad and wgrad all read the tensors where they lie in HBM:
// no transposes, no im2col, no concat copies.
//
// Roofline: MFMA-bound for K >= 512 (2*128*128*K flops per 2*128*K*sizeof(T) operand bytes per tile).
#include <stdlib.h>
#include <type_traits>
#include "mma.h"

#ifndef PT_EPI_NT
#define PT_EPI_NT -1       // tools/gemm_probe.py variant 7 / 8: force non-temporal epilogue stores on / off (-1: GemmParams::nt_store)
#endif
#ifndef PT_GEMM_ASM_DMA
#define PT_GEMM_ASM_DMA 0       // experiment (make exp EXP_FLAGS=-DPT_GEMM_ASM_DMA=1): LDS-DMA of the two-stage kernels as inline asm;
                                // measured neutral (round 3): unlike the attention rings, hipcc puts no alias wait into this loop
#endif
#ifndef PT_GEMM_ABLATE
#define PT_GEMM_ABLATE 0   // tools/gemm_probe.py: 1 no MFMA/LDS reads, 2 no staging loads, 3 no epilogue stores, 4 all three (launch floor); 11: plane conv operands without the A loads of every second k-tile
#endif

namespace {

struct VOp {
  const char* p; const char* p2;
  int64_t ld, ld2, c_split;
  int kind, taps, cin, rowmap, stride;
  int n_out, n_in;
  int cin_shift, nout_shift;   // log2 when a power of two, else -1
  int seg_kt;                  // k-tiles over which a chunk's address advances by `step` bytes per k-tile (1: recompute always)
  int64_t step;                // bytes per k-tile inside a segment
  int edge_slow;               // also recompute at positions 1 and seg_kt-1 of a segment (conv rows at batch-item edges)
  int64_t rows, cols;   // logical extent of the virtual matrix
  // PT_BF16X2 (split storage: a row of C f32-class elements is [C hi | C lo] bf16): the virtual matrix has 2 K columns, k-tile t
  // (64 columns = one 128-byte row of the LDS image) = [hi of logical columns 32 t .. 32 t + 31 | lo of the same columns], so that
  // one k-tile carries both planes of 32 logical columns and the loop issues the bf16 x 3 product (mma.h: a_hi w_lo + a_lo w_hi +
  // a_hi w_hi) from four fragment reads, with no conversion.  Round 4's first form ran a plain bf16 GEMM over THREE plane copies
  // of K ({hi, hi, lo} against {hi, lo, hi}): the same products, but a_hi and w_hi were staged twice -- and staging, not the
  // MFMAs, bounds these GEMMs (tools/decode_probe.py on the ablation builds: N = 640, K = 512: 1.69 ms, 0.93 without the
  // staging loads, 1.16 without the MFMAs).  plane1 / plane2 = element offset of the lo plane in rows of p / p2.  krep = K > 0
  // marks a plane operand, 0 plain storage.
  int krep, plane1, plane2, pmap;      // (32-bit: eight GemmParams must fit the 4 KiB kernel-argument segment of pt_wgrad_group)
};

struct GemmParams {
  VOp A, B;
  int64_t M, N, K;
  char* C; int64_t ldc;
  int out_kind, split_k;
  const float* bias; const float* row_bias; int64_t row_bias_rows, row_bias_ld;
  const char* residual; int64_t ldr;
  const char* residual2; int64_t ldr2;
  int conv_wgrad_cin, conv_wgrad_cin_store;
  float alpha;
  int act, act2; char* C2; int64_t ldc2;
  float* arow_sum; int64_t arow_n, arow_stride; int arow_rep;
  int tiles_m, tiles_n;
  int64_t geglu_rows;          // pt_wgrad_group: > 0 = the GEMM's rows are GEGLU-interleaved weight rows (F = geglu_rows)
  const float* scale_a; const float* scale_b;   // pt_gemm_fp8: device-resident dequantisation factors of the two operands
  int nt_store;                // epilogue rows as non-temporal stores (store_out)
  int x3;                      // f32 forward GEMMs: bf16 x 3 products (mma.h) instead of the exact f32 MFMA
  int planes_c, planes_c2;     // PT_BF16X2 outputs: P > 0 = C / C2 rows are [hi | lo] planes in blocks of P logical columns: column n
                               // sits at (n / P) 2 P + n % P, its lo part P elements further (P = N: one block; P = cout for a
                               // transposed conv whose N = r cout columns are r output rows of cout channels each)
};
struct f8_t { uint8_t bits; };   // one fp8 operand element (e4m3 or e5m2): addressing only

__device__ __attribute__((aligned(256))) const uint32_t pt_zero_page[64] = {0};
#ifndef PT_GEMM_TRACE
#define PT_GEMM_TRACE 0       // tools/gemm_probe.py --trace: wave 0 of workgroup 0 leaves s_memtime stamps in pt_trace
#endif
#if PT_GEMM_TRACE
__device__ unsigned long long pt_trace[8 + 32 * 4];     // [8 ..]: per k-tile stamps of the two-stage loop (tools/gemm_probe.py --ktrace)
#endif
#if PT_GEMM_TRACE
#define PT_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) pt_trace[k] = __builtin_amdgcn_s_memtime(); } while (0)
// the clock the kernel runs at: shader cycles (s_memtime) per 100 MHz tick (s_memrealtime) between entry and exit of workgroup 0
#define PT_STAMP_RT(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) pt_trace[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PT_STAMP(k) do { } while (0)
#define PT_STAMP_RT(k) do { } while (0)
#endif

typedef __attribute__((address_space(1))) const void pt_gptr;
typedef __attribute__((address_space(3))) void pt_lptr;

// one LDS-DMA wave-instruction (1 KiB at the wave-uniform LDS address `lds`), builtin or asm form (see attn2.h: a2_dma16)
__device__ __forceinline__ void pt_dma16(const char* src, char* lds) {
#if PT_GEMM_ASM_DMA
  const uint32_t a = (uint32_t)(uintptr_t)(pt_lptr*)lds;
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(a) : "memory");
#else
  __builtin_amdgcn_global_load_lds((pt_gptr*)src, (pt_lptr*)lds, 16, 0, 0);
#endif
}

__device__ __forceinline__ void divmod(int x, int d, int shift, int& q, int& r) {
  if (shift >= 0) { q = x >> shift; r = x & (d - 1); }
  else { q = x / d; r = x - q * d; }
}

// address of the 16-byte chunk V[row][col .. col+EPC) of a virtual matrix, or the zero page
// KC = kind class, fixed at compile time so that only one addressing scheme's invariants occupy registers:
// 0 plain / channel-concat, 1 conv gather, 2 flipped conv weights
// X2 (compile time): the operand may be PT_BF16X2 split storage.  A separate instantiation, not a run-time branch in the common
// kernels: with the plane fields live in every kernel the register allocation of ALL of them moved (eight-phase forward kernel:
// 28 -> 74 spilled registers, grouped wgrad 15 -> 30) and the training step lost 5 % (39.96 -> 42.0 ms on one box)
template <typename T, int KC, bool X2 = false>
__device__ __forceinline__ const char* vaddr(const VOp& op, int64_t row, int64_t col) {
  const char* zero = reinterpret_cast<const char*>(pt_zero_page);
  if (row >= op.rows || col >= op.cols) return zero;
  const T* ptr;
  int64_t plane = 0;                         // PT_BF16X2: 1 = this copy of the K columns reads the lo plane
  if (X2 && KC != 2 && op.krep > 0) {       // virtual column -> (plane, logical column)
    plane = (col >> 5) & 1;
    col = ((col >> 6) << 5) | (col & 31);
  }
  if (KC == 0) {
    ptr = (op.kind == PT_V_PLAIN || col < op.c_split)
              ? reinterpret_cast<const T*>(op.p) + row * op.ld + col + plane * (int64_t)op.plane1
              : reinterpret_cast<const T*>(op.p2) + row * op.ld2 + (col - op.c_split) + plane * (int64_t)op.plane2;
  } else if (KC == 1) {
    int tap, ci, b, n;
    divmod((int)col, op.cin, op.cin_shift, tap, ci);
    divmod((int)row, op.n_out, op.nout_shift, b, n);
    int ns; bool ok;
    const int u = n + tap - 1;
    switch (op.rowmap) {
      case PT_MAP_S1: ns = u; ok = (u >= 0) && (u < op.n_in); break;
      case PT_MAP_S2: ns = 2 * n + tap - 1; ok = (ns >= 0) && (ns < op.n_in); break;
      case PT_MAP_UP2: ns = u >> 1; ok = (u >= 0) && (u < 2 * op.n_in); break;
      case PT_MAP_S2_DGRAD: ns = u >> 1; ok = (u >= 0) && ((u & 1) == 0) && (ns < op.n_in); break;
      case PT_MAP_CAUSAL_REFLECT: { const int v = n + tap - (op.taps - 1); ns = v < 0 ? -v : v; ok = ns < op.n_in; } break;
      case PT_MAP_STRIDED_REFLECT: { const int v = n * op.stride + tap - (op.taps - op.stride); ns = v < 0 ? -v : v; ok = ns < op.n_in; } break;
      default: /* PT_MAP_BACK */ ns = n - tap; ok = ns >= 0; break;
    }
    if (!ok) return zero;
    ptr = reinterpret_cast<const T*>(op.p) + ((int64_t)b * op.n_in + ns) * op.ld + ci + plane * op.plane1;
  } else {  // PT_V_WFLIP: row = tap*cout + co
    int tap, co;
    divmod((int)row, op.cin, op.cin_shift, tap, co);
    ptr = reinterpret_cast<const T*>(op.p) + ((int64_t)co * 3 + (2 - tap)) * op.ld + col;
  }
  return reinterpret_cast<const char*>(ptr);
}

// GEGLU interleave of a [2F]-row weight (value rows 0..F-1, gate rows F..2F-1): interleaved row 64 q + t is value row
// 32 q + t for t < 32 and gate row F + 32 q + (t - 32) otherwise, so that one lane's accumulator blocks j and j + 2 of a
// 64-column wave tile hold the value and the gate of the same activation column (fused GEGLU epilogues).
__host__ __device__ __forceinline__ int64_t geglu_orig_row(int64_t mi, int64_t F) {
  const int64_t q = mi >> 6; const int t = (int)(mi & 63);
  return t < 32 ? 32 * q + t : F + 32 * q + (t - 32);
}

constexpr int SCRATCH_PER_WAVE = 2048;  // epilogue transpose scratch: 16 rows x 128 B per wave

// PT_GEMM_NT=1: output rows leave as NON-TEMPORAL 16-byte stores.  A tile is 128 KiB per CU, i.e. a round of tiles dirties an
// XCD's whole 4 MiB L2, and the epilogue of a 256 x 256 bf16 tile takes 16.7 us (2 TB/s chip-wide, half of what a plain fill
// kernel reaches; 63 % of a K = 512 GEMM: tools/gemm_probe.py 0 6).  Streaming past L2 makes the K = 512 GEMMs 7-12 % faster
// ALONE (tools/gemm_probe.py 0 7) but changes nothing in the training step (43.6 ms either way: the consumer then misses L2),
// so it is off by default.  Also measured: starting half of the first round's workgroups half a tile period late (to take the
// CUs out of lock-step) -- no gain, the slow write phase is a per-CU matter, not chip-wide HBM contention.
__device__ __forceinline__ void store_out(u32x4_t* dst, const u32x4_t v, int nt) {
  if (PT_GEMM_ABLATE == 9) { asm volatile("" ::"v"(v)); return; }      // probe: the whole epilogue but its global stores
  if (PT_EPI_NT >= 0 ? PT_EPI_NT : nt) __builtin_nontemporal_store(v, dst);
  else *dst = v;
}

// ---- epilogue (shared by both kernels): acc[i][j] is the 16x16 tile at rows 16 i, columns 16 j of the wave's
// (16 MI) x 64 sub-tile whose origin is (m0 + wm * 16 MI, n0 + wn * 64); `scratch` = 2 KiB of wave-private LDS ----------
template <typename T, bool ATOMIC, int MI, int BM, int BN, bool X2 = false>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4_t (&acc)[MI][4], const int64_t m0, const int64_t n0,
                                              const int wm, const int wn, const int lane, char* scratch0,
                                              const int scr_stride = 0) {
  // scr_stride: bytes between the transposition scratch of consecutive 16-row blocks (0: one 2 KiB scratch reused by all of
  // them).  The callers pass the dead operand stages (2 KiB per row block per wave): with private scratch per block there is no
  // write-after-read wait between blocks and the compiler overlaps one block's LDS round trip with the next block's conversions.
  char* scratch = scratch0;
  constexpr int WM = 16 * MI;
  const int g = lane >> 4, li = lane & 15;
  if (ATOMIC) {
    float* C = reinterpret_cast<float*>(p.C);
    if (PT_GEMM_ABLATE == 3 || PT_GEMM_ABLATE == 4) {      // probe: keep the accumulators live, issue no atomics
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
      return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t n = n0 + wn * 64 + 16 * j + li;
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t m = m0 + wm * WM + 16 * i + 4 * g + r;
          if (m >= p.M) continue;
          int64_t idx;
          if (p.conv_wgrad_cin > 0) {           // padded-Cin conv (conv_in): drop the pad channels, keep [Cout][3][Cin]
            const int tap = (int)n / p.conv_wgrad_cin, ci = (int)n - tap * p.conv_wgrad_cin;
            if (ci >= p.conv_wgrad_cin_store) continue;
            idx = (m * 3 + tap) * p.conv_wgrad_cin_store + ci;
          } else {
            idx = m * p.ldc + n;
          }
          unsafeAtomicAdd(C + idx, p.alpha * acc[i][j][r]);
        }
      }
    return;
  }
  // Phase 1 (accumulator layout: lane = row li, 4 consecutive columns per register group): alpha, bias, row bias,
  // residuals (vector loads).  Phase 2: activation + conversion, then the wave's 16 x 64 (bf16) / 16 x 32 (f32)
  // sub-tile goes through a wave-private XOR-swizzled LDS scratch and leaves as whole row segments, 16 B per lane.
  const bool out32 = sizeof(T) == 4 || p.out_kind == PT_OUT_F32;
  // Fast path (interior tile, bf16 output, 16-byte aligned rows): no per-element predicates at all.  The generic
  // path below costs ~2500 executed instructions per wave -- as much as the MFMA work of a K = 512 tile.
  if (sizeof(T) == 2 && !out32 && PT_GEMM_ABLATE != 3 && PT_GEMM_ABLATE != 4 && m0 + BM <= p.M && n0 + BN <= p.N &&
      (p.ldc & 7) == 0 && (!p.C2 || (p.ldc2 & 7) == 0) && (reinterpret_cast<uintptr_t>(p.C) & 15u) == 0 &&
      (reinterpret_cast<uintptr_t>(p.C2) & 15u) == 0) {
    const int64_t mbase = m0 + wm * WM, nbase = n0 + wn * 64;
    const int wr_off = li * 128 + ((g & 1) << 3);            // scratch write: row li, 8-byte half (g&1) of chunk 2j + (g>>1)
    const int rd_row = lane >> 3, rd_c = lane & 7;           // scratch read: rows rd_row, rd_row + 8; chunk rd_c
    const bool has_alpha = p.alpha != 1.0f;
    // LOAD PASS, in place on the accumulators, BEFORE the first store of the tile: alpha, bias (loaded once, it does not depend
    // on the row block), row bias and residuals.  gfx950 has ONE counter for loads and stores (vmcnt), so a load issued after a
    // store can only be waited for together with that store's acknowledgement from memory: with the loads inside the store
    // loop every 16-row block stalled on the previous block's write round trip (the epilogue of a 256 x 256 tile took 16.7 us).
    {
      f32x4_t bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bv[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
          // act 2: output columns are GEGLU-interleaved (64 q + t: t < 32 value row 32 q + t, else gate row F + 32 q + t - 32)
          const int64_t bcol = p.act == 2 ? (j < 2 ? (nbase >> 1) + 16 * j : (p.N >> 1) + (nbase >> 1) + 16 * (j - 2)) : nbase + 16 * j;
          bv[j] = *reinterpret_cast<const f32x4_t*>(p.bias + bcol + 4 * g);
        }
      }
      const bool res1 = p.residual != nullptr && p.act != 3;      // act 3 reads its "residual" (the saved projection) itself
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (has_alpha) acc[i][j] *= p.alpha;
          acc[i][j] += bv[j];
        }
      // each optional operand in ONE branch, its loads issued in batches of four row blocks and only then consumed (inside a
      // per-row-block `if` the compiler waits for every block's loads before it issues the next block's: 8 memory round trips)
      auto add_bf16_rows = [&](const char* base, int64_t ld) {
#pragma unroll
        for (int i0 = 0; i0 < MI; i0 += 4) {
          u32x2_t t[4][4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bf16_t* rp = reinterpret_cast<const bf16_t*>(base) + (mbase + 16 * (i0 + i) + li) * ld + nbase + 4 * g;
#pragma unroll
            for (int j = 0; j < 4; ++j) t[i][j] = *reinterpret_cast<const u32x2_t*>(rp + 16 * j);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[i0 + i][j][0] += __uint_as_float(t[i][j][0] << 16); acc[i0 + i][j][1] += __uint_as_float(t[i][j][0] & 0xffff0000u);
              acc[i0 + i][j][2] += __uint_as_float(t[i][j][1] << 16); acc[i0 + i][j][3] += __uint_as_float(t[i][j][1] & 0xffff0000u);
            }
        }
      };
      if (res1) add_bf16_rows(p.residual, p.ldr);
      if (p.residual2) add_bf16_rows(p.residual2, p.ldr2);
      if (p.row_bias) {
#pragma unroll
        for (int i0 = 0; i0 < MI; i0 += 2) {
          f32x4_t t[2][4];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int64_t m = mbase + 16 * (i0 + i) + li;
            const float* rb = p.row_bias + (m / p.row_bias_rows) * p.row_bias_ld + nbase + 4 * g;
#pragma unroll
            for (int j = 0; j < 4; ++j) t[i][j] = *reinterpret_cast<const f32x4_t*>(rb + 16 * j);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i0 + i][j] += t[i][j];
        }
      }
    }
    PT_STAMP(4);
    // STORE PASS.  The unrolled epilogue is straight-line code every wave executes ONCE per tile, so its size is its cost: with
    // every variant (ELU, second output, GEGLU forward / backward) inlined in one body the pass was ~45 KB of instructions and
    // ran at instruction-fetch speed -- 11 us per 256 x 256 tile even with the global stores compiled out (tools/gemm_probe.py
    // 0 9 6), 40 % of a K = 512 GEMM.  The plain case (one output, no activation: most launches) therefore has its own compact
    // body; ELU gets a copy; the GEGLU / two-output variants share the general body below.
    if (p.act <= 1 && !p.C2 && !(X2 && p.planes_c > 0)) {
      // (per-row-block scratch: sc = scratch0 + i * scr_stride)
      bf16_t* Cb0 = reinterpret_cast<bf16_t*>(p.C) + (mbase + rd_row) * p.ldc + nbase + 8 * rd_c;
      const int64_t step8 = 8 * p.ldc;
      if (PT_GEMM_ABLATE == 10) {                      // probe: no LDS transposition, 8-byte stores straight from the accumulators
        bf16_t* Cd = reinterpret_cast<bf16_t*>(p.C) + (mbase + li) * p.ldc + nbase + 4 * g;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            u32x2_t o;
            o[0] = pack_bf16x2(acc[i][j][0], acc[i][j][1]);
            o[1] = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
            *reinterpret_cast<u32x2_t*>(Cd + (int64_t)16 * i * p.ldc + 16 * j) = o;
          }
        return;
      }
      if (p.act == 0) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          char* sc = scratch0 + i * scr_stride;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            u32x2_t o;
            o[0] = pack_bf16x2(acc[i][j][0], acc[i][j][1]);
            o[1] = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
            *reinterpret_cast<u32x2_t*>(sc + wr_off + (((2 * j + (g >> 1)) ^ (li & 7)) << 4)) = o;
          }
          const u32x4_t o0 = *reinterpret_cast<const u32x4_t*>(sc + rd_row * 128 + ((rd_c ^ (rd_row & 7)) << 4));
          const u32x4_t o1 = *reinterpret_cast<const u32x4_t*>(sc + (8 + rd_row) * 128 + ((rd_c ^ (rd_row & 7)) << 4));
          store_out(reinterpret_cast<u32x4_t*>(Cb0), o0, p.nt_store);
          store_out(reinterpret_cast<u32x4_t*>(Cb0 + step8), o1, p.nt_store);
          Cb0 += 2 * step8;
        }
      } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          char* sc = scratch0 + i * scr_stride;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float w[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = acc[i][j][r] < 0.f ? (__expf(acc[i][j][r]) - 1.f) : acc[i][j][r];
            u32x2_t o;
            o[0] = pack_bf16x2(w[0], w[1]);
            o[1] = pack_bf16x2(w[2], w[3]);
            *reinterpret_cast<u32x2_t*>(sc + wr_off + (((2 * j + (g >> 1)) ^ (li & 7)) << 4)) = o;
          }
          const u32x4_t o0 = *reinterpret_cast<const u32x4_t*>(sc + rd_row * 128 + ((rd_c ^ (rd_row & 7)) << 4));
          const u32x4_t o1 = *reinterpret_cast<const u32x4_t*>(sc + (8 + rd_row) * 128 + ((rd_c ^ (rd_row & 7)) << 4));
          store_out(reinterpret_cast<u32x4_t*>(Cb0), o0, p.nt_store);
          store_out(reinterpret_cast<u32x4_t*>(Cb0 + step8), o1, p.nt_store);
          Cb0 += 2 * step8;
        }
      }
      return;
    }
if (p.act == 3) {
      // GEGLU backward fused into the ff2 dgrad: the tile holds d(act)[m][jc]; with value / gate read from the saved
      // (interleaved) projection, C receives d(proj) in the same interleaved layout: per 32-column block q of act,
      // 64 output columns [d value (32) | d gate (32)] -- one whole 128-byte row segment per block.  The projection values of
      // row block i + 1 are requested BEFORE row block i is stored (loads behind stores wait for the stores: one vmcnt).
      u32x2_t hv[2][2][2], gv[2][2][2];                    // [parity][h][jj]
      auto fetch = [&](int i, int par) {
        const bf16_t* pr = reinterpret_cast<const bf16_t*>(p.residual) + (mbase + 16 * i + li) * p.ldr;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            hv[par][h][jj] = *reinterpret_cast<const u32x2_t*>(pr + 2 * nbase + 64 * h + 16 * jj + 4 * g);
            gv[par][h][jj] = *reinterpret_cast<const u32x2_t*>(pr + 2 * nbase + 64 * h + 32 + 16 * jj + 4 * g);
          }
      };
      fetch(0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        scratch = scratch0 + i * scr_stride;
        if (i + 1 < MI) fetch(i + 1, (i + 1) & 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int64_t q64 = 2 * nbase + 64 * h;                  // first interleaved column of this 32-column block
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const u32x2_t hw = hv[i & 1][h][jj], gw = gv[i & 1][h][jj];
            const float val[4] = {__uint_as_float(hw[0] << 16), __uint_as_float(hw[0] & 0xffff0000u),
                                  __uint_as_float(hw[1] << 16), __uint_as_float(hw[1] & 0xffff0000u)};
            const float gat[4] = {__uint_as_float(gw[0] << 16), __uint_as_float(gw[0] & 0xffff0000u),
                                  __uint_as_float(gw[1] << 16), __uint_as_float(gw[1] & 0xffff0000u)};
            float dv[4], dg[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float ge, dge;
              gelu_erf_fast(gat[r], ge, dge);
              const float d = acc[i][2 * h + jj][r];
              dv[r] = d * ge; dg[r] = d * val[r] * dge;
            }
            u32x2_t o;
            o[0] = pack_bf16x2(dv[0], dv[1]);
            o[1] = pack_bf16x2(dv[2], dv[3]);
            *reinterpret_cast<u32x2_t*>(scratch + wr_off + (((2 * jj + (g >> 1)) ^ (li & 7)) << 4)) = o;
            o[0] = pack_bf16x2(dg[0], dg[1]);
            o[1] = pack_bf16x2(dg[2], dg[3]);
            *reinterpret_cast<u32x2_t*>(scratch + wr_off + (((4 + 2 * jj + (g >> 1)) ^ (li & 7)) << 4)) = o;
          }
          bf16_t* Cb = reinterpret_cast<bf16_t*>(p.C) + (mbase + 16 * i) * p.ldc + q64 + 8 * rd_c;
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int r = 8 * it + rd_row;
            store_out(reinterpret_cast<u32x4_t*>(Cb + r * p.ldc), *reinterpret_cast<const u32x4_t*>(scratch + r * 128 + ((rd_c ^ (r & 7)) << 4)), p.nt_store);
          }
        }
      }
      return;
    }
    if (p.act == 2) {
      // GEGLU forward fused into the ff1 GEMM: columns are interleaved so that register block j (value) and j + 2 (gate)
      // of one lane belong to the same act column: C2[m][nbase / 2 + 16 j + 4 g + r] = value * gelu(gate); the raw
      // projection tile goes to C as well (the backward needs it)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        scratch = scratch0 + i * scr_stride;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          float w[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float ge, dge;
            gelu_erf_fast(acc[i][jj + 2][r], ge, dge);
            w[r] = acc[i][jj][r] * ge;
          }
          u32x2_t o;
          o[0] = pack_bf16x2(w[0], w[1]);
          o[1] = pack_bf16x2(w[2], w[3]);
          *reinterpret_cast<u32x2_t*>(scratch + li * 64 + (((2 * jj + (g >> 1)) ^ (li & 3)) << 4) + ((g & 1) << 3)) = o;
        }
        {
          const int r = lane >> 2, c = lane & 3;               // 16 rows x 64 bytes: one pass, 16 bytes per lane
          bf16_t* Cb = reinterpret_cast<bf16_t*>(p.C2) + (mbase + 16 * i + r) * p.ldc2 + (nbase >> 1) + 8 * c;
          store_out(reinterpret_cast<u32x4_t*>(Cb), *reinterpret_cast<const u32x4_t*>(scratch + r * 64 + ((c ^ (r & 3)) << 4)), p.nt_store);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          u32x2_t o;
          o[0] = pack_bf16x2(acc[i][j][0], acc[i][j][1]);
          o[1] = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
          *reinterpret_cast<u32x2_t*>(scratch + wr_off + (((2 * j + (g >> 1)) ^ (li & 7)) << 4)) = o;
        }
        bf16_t* Cb = reinterpret_cast<bf16_t*>(p.C) + (mbase + 16 * i) * p.ldc + nbase + 8 * rd_c;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int r = 8 * it + rd_row;
          store_out(reinterpret_cast<u32x4_t*>(Cb + r * p.ldc), *reinterpret_cast<const u32x4_t*>(scratch + r * 128 + ((rd_c ^ (r & 7)) << 4)), p.nt_store);
        }
      }
      return;
    }
    // two outputs (C, C2) with their own activations (Encodec decoder: raw + ELU copy); PT_BF16X2: each output as [hi | lo] planes
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float v[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[j][r] = acc[i][j][r];
      for (int op = 0; op < (p.C2 ? 2 : 1); ++op) {
        const int act = op == 0 ? p.act : p.act2;
        const int64_t plane_off = X2 ? (op == 0 ? p.planes_c : p.planes_c2) : 0;
        const int64_t ld = op == 0 ? p.ldc : p.ldc2;
        for (int pl = 0; pl < (X2 && plane_off > 0 ? 2 : 1); ++pl) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float w[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = (act == 1 && v[j][r] < 0.f) ? (__expf(v[j][r]) - 1.f) : v[j][r];
            u32x2_t o;
            o[0] = pack_bf16x2(w[0], w[1]);
            o[1] = pack_bf16x2(w[2], w[3]);
            if (X2 && pl == 1) {                               // lo plane: what the hi plane's rounding left over
              o[0] = pack_bf16x2(w[0] - __uint_as_float(o[0] << 16), w[1] - __uint_as_float(o[0] & 0xffff0000u));
              o[1] = pack_bf16x2(w[2] - __uint_as_float(o[1] << 16), w[3] - __uint_as_float(o[1] & 0xffff0000u));
            }
            *reinterpret_cast<u32x2_t*>(scratch + wr_off + (((2 * j + (g >> 1)) ^ (li & 7)) << 4)) = o;
          }
          const int64_t ccol = nbase + 8 * rd_c;
          bf16_t* Cb = reinterpret_cast<bf16_t*>(op == 0 ? p.C : p.C2) + (mbase + 16 * i) * ld +
                       (X2 && plane_off > 0 ? (ccol / plane_off) * 2 * plane_off + ccol % plane_off + pl * plane_off : ccol);
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int r = 8 * it + rd_row;
            const u32x4_t ov = *reinterpret_cast<const u32x4_t*>(scratch + r * 128 + ((rd_c ^ (r & 7)) << 4));
            store_out(reinterpret_cast<u32x4_t*>(Cb + r * ld), ov, p.nt_store);
          }
        }
      }
    }
    return;
  }
  const int n_outs = p.C2 ? 2 : 1;
  const int64_t mrow0 = (PT_GEMM_ABLATE == 3 || PT_GEMM_ABLATE == 4) ? ((int64_t)1 << 40) : m0 + wm * WM;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int64_t m = mrow0 + 16 * i + li;
    const bool mok = m < p.M;
    const float* rbias = (p.row_bias && mok) ? p.row_bias + (m / p.row_bias_rows) * p.row_bias_ld : nullptr;
    float v[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + 16 * j + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[j][r] = p.alpha * acc[i][j][r];
      if (mok && n + 3 < p.N) {
        if (p.bias) { const f32x4_t t = *reinterpret_cast<const f32x4_t*>(p.bias + n); v[j][0] += t[0]; v[j][1] += t[1]; v[j][2] += t[2]; v[j][3] += t[3]; }
        if (rbias) { const f32x4_t t = *reinterpret_cast<const f32x4_t*>(rbias + n); v[j][0] += t[0]; v[j][1] += t[1]; v[j][2] += t[2]; v[j][3] += t[3]; }
        if (p.residual) {
          const T* rp = reinterpret_cast<const T*>(p.residual) + m * p.ldr + n;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[j][r] += to_f32<T>(rp[r]);
        }
        if (p.residual2) {
          const T* rp = reinterpret_cast<const T*>(p.residual2) + m * p.ldr2 + n;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[j][r] += to_f32<T>(rp[r]);
        }
      } else if (mok) {
        for (int r = 0; r < 4 && n + r < p.N; ++r) {
          if (p.bias) v[j][r] += p.bias[n + r];
          if (rbias) v[j][r] += rbias[n + r];
          if (p.residual) v[j][r] += to_f32<T>(reinterpret_cast<const T*>(p.residual)[m * p.ldr + n + r]);
          if (p.residual2) v[j][r] += to_f32<T>(reinterpret_cast<const T*>(p.residual2)[m * p.ldr2 + n + r]);
        }
      }
    }
    for (int op = 0; op < n_outs; ++op) {
      char* Cb = op == 0 ? p.C : p.C2;
      const int64_t ld = op == 0 ? p.ldc : p.ldc2;
      const int act = op == 0 ? p.act : p.act2;
      if (!out32) {
        const int64_t plane_off = X2 ? (op == 0 ? p.planes_c : p.planes_c2) : 0;          // PT_BF16X2: [hi | lo] planes
        for (int pl = 0; pl < (X2 && plane_off > 0 ? 2 : 1); ++pl) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float w[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = (act == 1 && v[j][r] < 0.f) ? (__expf(v[j][r]) - 1.f) : v[j][r];
            u32x2_t o;
            o[0] = pack_bf16x2(w[0], w[1]);
            o[1] = pack_bf16x2(w[2], w[3]);
            if (X2 && pl == 1) {
              o[0] = pack_bf16x2(w[0] - __uint_as_float(o[0] << 16), w[1] - __uint_as_float(o[0] & 0xffff0000u));
              o[1] = pack_bf16x2(w[2] - __uint_as_float(o[1] << 16), w[3] - __uint_as_float(o[1] & 0xffff0000u));
            }
            const int chunk = 2 * j + (g >> 1);
            *reinterpret_cast<u32x2_t*>(scratch + li * 128 + ((chunk ^ (li & 7)) << 4) + ((g & 1) << 3)) = o;
          }
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int r = 8 * it + (lane >> 3), c = lane & 7;
            const u32x4_t val = *reinterpret_cast<const u32x4_t*>(scratch + r * 128 + ((c ^ (r & 7)) << 4));
            const int64_t mm = mrow0 + 16 * i + r, nn = n0 + wn * 64 + 8 * c;
            if (mm < p.M && nn < p.N) {
              bf16_t* dst = reinterpret_cast<bf16_t*>(Cb) + mm * ld + (X2 && plane_off > 0 ? (nn / plane_off) * 2 * plane_off + nn % plane_off + pl * plane_off : nn);
              if (nn + 7 < p.N && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) {
                *reinterpret_cast<u32x4_t*>(dst) = val;
              } else {
                for (int e = 0; e < 8 && nn + e < p.N; ++e) {
                  const uint32_t wv = val[e >> 1];
                  bf16_t h; h.bits = (uint16_t)((e & 1) ? (wv >> 16) : (wv & 0xffffu));
                  dst[e] = h;
                }
              }
            }
          }
        }
      } else {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * half + jj;
            f32x4_t w;
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = (act == 1 && v[j][r] < 0.f) ? (__expf(v[j][r]) - 1.f) : v[j][r];
            const int chunk = 4 * jj + g;
            *reinterpret_cast<f32x4_t*>(scratch + li * 128 + ((chunk ^ (li & 7)) << 4)) = w;
          }
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int r = 8 * it + (lane >> 3), c = lane & 7;
            const f32x4_t val = *reinterpret_cast<const f32x4_t*>(scratch + r * 128 + ((c ^ (r & 7)) << 4));
            const int64_t mm = mrow0 + 16 * i + r, nn = n0 + wn * 64 + 32 * half + 4 * c;
            if (mm < p.M && nn < p.N) {
              float* dst = reinterpret_cast<float*>(Cb) + mm * ld + nn;
              if (nn + 3 < p.N && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) *reinterpret_cast<f32x4_t*>(dst) = val;
              else for (int e = 0; e < 4 && nn + e < p.N; ++e) dst[e] = val[e];
            }
          }
        }
      }
    }
  }
}

// Tile configurations (tools/gemm_probe.py: the kernel is bound by L2 -> LDS operand traffic, ~15 TB/s with 64 KB in
// flight per CU, while the MFMA-only loop runs at ~1.2 PF; operand bytes per flop scale with 1/BM + 1/BN):
//   128 x 128: 4 waves (2x2) of 64x64,  2 LDS stages,  72 KiB -> two workgroups per CU (small / few-tile problems)
//   256 x 128: 8 waves (4x2) of 64x64,  3 LDS stages, 160 KiB -> counted s_waitcnt vmcnt + raw s_barrier, 2 tiles in flight
//   256 x 256: 8 waves (2x4) of 128x64, 2 LDS stages, 144 KiB -> half the operand bytes per flop, 25 % fewer LDS
//              fragment bytes per flop (128 accumulator registers per lane)
template <int BM_, int BN_> struct TileCfg {
  static constexpr int BM = BM_, BN = BN_;
  static constexpr int NWAVES = (BM_ == 128) ? 4 : 8, NTHREADS = 64 * NWAVES;
  static constexpr int WAVES_N = BN_ / 64, WAVES_M = NWAVES / WAVES_N;
  static constexpr int MI = BM_ / WAVES_M / 16, NJ = 4, WM = 16 * MI;      // per-wave tile WM x 64
  static constexpr int NSTAGE = (BM_ == 256 && BN_ == 128) ? 3 : 2;
  static constexpr int A_BYTES = BM_ * 128, B_BYTES = BN_ * 128, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int A_CHUNKS = BM_ * 8 / NTHREADS, B_CHUNKS = BN_ * 8 / NTHREADS;   // 16-byte chunks per thread per k-tile
  static constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES + NWAVES * SCRATCH_PER_WAVE;
  static constexpr int MIN_WAVES_PER_SIMD = (BM_ == 128) ? 2 : 1;
};

// F8 (T = bf16_t only): fp8 operands as in gemm8p_body -- 128 one-byte elements per k-tile row, the same images and fragment reads,
// one v_mfma_f32_16x16x128_f8f6f4 per pair of 16-byte fragment reads; 1: activations e4m3, 2: e5m2 (weights always e4m3).
template <typename T, bool TA, bool TB, bool ATOMIC, int KA, int KB, int BM, int BN, int F8 = 0, bool X2 = false>
__global__ __launch_bounds__((TileCfg<BM, BN>::NTHREADS), (TileCfg<BM, BN>::MIN_WAVES_PER_SIMD)) void gemm_kernel(const GemmParams p) {
  using Cfg = TileCfg<BM, BN>;
  using TO = typename std::conditional<F8 != 0, f8_t, T>::type;     // operand element (addressing)
  static_assert(F8 == 0 || (sizeof(T) == 2 && !TA && !TB && !ATOMIC), "fp8 operands: K-contiguous, bf16 store epilogue");
  constexpr int NTHREADS = Cfg::NTHREADS, NSTAGE = Cfg::NSTAGE, A_BYTES = Cfg::A_BYTES, STAGE_BYTES = Cfg::STAGE_BYTES;
  constexpr int MI = Cfg::MI, WM = Cfg::WM;
  constexpr int BK = F8 ? 128 : TileK<T>::KE;      // 64 (bf16) / 32 (f32) / 128 (fp8)
  constexpr int EPC = F8 ? 16 : 16 / (int)sizeof(T);         // elements per 16-byte chunk
  constexpr int TCHA = BM / EPC, TCHB = BN / EPC;  // chunks per TileT row of the A / B image
  __shared__ __attribute__((aligned(16))) char smem[Cfg::LDS_BYTES];

  if (PT_GEMM_ABLATE == 5) return;      // probe: pure dispatch cost
  PT_STAMP(0); PT_STAMP_RT(6);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;     // WAVES_M x WAVES_N waves of WM x 64
  const int g = lane >> 4, li = lane & 15;

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous
  // run of tiles so neighbours that share an A row-panel hit the same L2.  Bijective for any grid size.
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // grid.x = tiles x split_k, K-slice major: the run of ids an XCD owns is (nearly) one K-slice of every tile, so a split-K
  // wgrad streams each operand through one L2 once (tile-major slices made every XCD re-fetch whole panels: 3x the bytes)
  const int ntile = p.tiles_m * p.tiles_n;
  const int kslice = bid / ntile, tix = bid - kslice * ntile;
  const int tm = tix / p.tiles_n, tn = tix - tm * p.tiles_n;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  const int nkt_total = (int)((p.K + BK - 1) / BK);
  const int per = (nkt_total + p.split_k - 1) / p.split_k;
  const int kt_begin = kslice * per;
  const int kt_end = min(nkt_total, kt_begin + per);
  if (kt_begin >= kt_end) return;

  f32x4_t acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // bias gradient on the wgrad GEMM: sum_k VA(m,k) as one extra MFMA column against an all-ones fragment, computed once per
  // output row panel (first tile column, first wave column) -- replaces a separate column-sum pass over dy
  const bool do_sum = ATOMIC && p.arow_sum != nullptr && tn == 0 && wn == 0;
  f32x4_t accb[MI];
  Frag<T> fones;
  {
    const float one8[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    frag_from_f32(fones, one8);
#pragma unroll
    for (int i = 0; i < MI; ++i) accb[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }

  // One wave-instruction of global_load_lds writes 64 x 16 B = 1 KiB of LDS linearly (base + lane*16).  Chunk slot
  // q of a tile image therefore holds, for TileK, row q>>3 / data chunk (q&7)^(row&7); for TileT, k-row q/TCH /
  // data chunk (q%TCH)^swz(k): the swizzle lives in the SOURCE address.
  const int wbase = __builtin_amdgcn_readfirstlane(wave) * 64;
  // Per-chunk source pointers live in registers and advance by a per-chunk byte step (0 for chunks parked on the zero
  // page): 2 VALU per chunk per k-tile.  They are recomputed from scratch (vaddr) only when a countdown reaches zero:
  // first k-tile, new conv tap / concat half, batch-item edges of a conv-wgrad operand, the K-tail tile.
  // (PMC: the loop was instruction-issue bound -- MFMA pipes 17 % busy, ~225 instructions per k-tile per wave.)
  const char* pa[Cfg::A_CHUNKS]; const char* pb[Cfg::B_CHUNKS];
  uint32_t sta[Cfg::A_CHUNKS], stb[Cfg::B_CHUNKS];
  const char* zero_page = reinterpret_cast<const char*>(pt_zero_page);
  const int last_kt = nkt_total - 1;
  const bool ktail = (p.K % BK) != 0;
  int until_slow = 0;                           // k-tiles that may still take the fast path
  auto fast_tiles_after = [&](int kt) {         // evaluated only on the slow path
    auto dist = [&](const VOp& o) {
      const int pos = kt % o.seg_kt;
      int d = o.seg_kt - pos;
      if (o.edge_slow) { if (pos == 0) d = 1; else if (pos < o.seg_kt - 1) d = min(d, o.seg_kt - 1 - pos); }
      return d;
    };
    int d = min(dist(p.A), dist(p.B));
    if (ktail && kt < last_kt) d = min(d, last_kt - kt);
    return d - 1;
  };
  auto stage = [&](int kt, int stg) {
    char* sa = smem + stg * STAGE_BYTES;
    char* sb = sa + A_BYTES;
    if (until_slow == 0) {
      const int64_t k0 = (int64_t)kt * BK;
#pragma unroll
      for (int i = 0; i < Cfg::A_CHUNKS; ++i) {
        const int q = tid + NTHREADS * i;
        if (!TA) { const int r = q >> 3; pa[i] = vaddr<TO, KA, X2>(p.A, m0 + r, k0 + ((q & 7) ^ (r & 7)) * EPC); }
        else     { const int k = q / TCHA; pa[i] = vaddr<TO, KA, X2>(p.A, k0 + k, m0 + ((q % TCHA) ^ tilet_swz(k)) * EPC); }
        sta[i] = pa[i] == zero_page ? 0u : (uint32_t)p.A.step;
      }
#pragma unroll
      for (int i = 0; i < Cfg::B_CHUNKS; ++i) {
        const int q = tid + NTHREADS * i;
        if (!TB) { const int r = q >> 3; pb[i] = vaddr<TO, KB, X2>(p.B, n0 + r, k0 + ((q & 7) ^ (r & 7)) * EPC); }
        else     { const int k = q / TCHB; pb[i] = vaddr<TO, KB, X2>(p.B, k0 + k, n0 + ((q % TCHB) ^ tilet_swz(k)) * EPC); }
        stb[i] = pb[i] == zero_page ? 0u : (uint32_t)p.B.step;
      }
      until_slow = fast_tiles_after(kt);
    } else {
      --until_slow;
#pragma unroll
      for (int i = 0; i < Cfg::A_CHUNKS; ++i) pa[i] += sta[i];
#pragma unroll
      for (int i = 0; i < Cfg::B_CHUNKS; ++i) pb[i] += stb[i];
    }
    if (PT_GEMM_ABLATE == 2 || PT_GEMM_ABLATE == 4) return;
    if (!(PT_GEMM_ABLATE == 11 && X2 && KA == 1 && (kt & 1))) {      // probe 11: every second k-tile without its A loads (the byte
#pragma unroll                                                          // count of a conv operand staged once per two taps; wrong results)
      for (int i = 0; i < Cfg::A_CHUNKS; ++i) pt_dma16(pa[i], sa + (wbase + NTHREADS * i) * 16);
    }
#pragma unroll
    for (int i = 0; i < Cfg::B_CHUNKS; ++i) pt_dma16(pb[i], sb + (wbase + NTHREADS * i) * 16);
  };

  auto compute = [&](int stg) {
    const char* sa = smem + stg * STAGE_BYTES;
    const char* sb = sa + A_BYTES;
    if constexpr (F8 != 0) {
      typedef __attribute__((ext_vector_type(8))) int i32x8_t;
      auto cat = [](const Frag<bf16_t>& lo, const Frag<bf16_t>& hi) {
        const u32x4_t a = __builtin_bit_cast(u32x4_t, lo.v), b = __builtin_bit_cast(u32x4_t, hi.v);
        return (i32x8_t){(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
      };
      Frag<bf16_t> fb[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) frag_load_k(fb[j][ks], sb, wn * 64 + 16 * j + li, ks * 32 + 8 * g);
#pragma unroll
      for (int h = 0; h < MI; h += 4) {
        Frag<bf16_t> fa[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) frag_load_k(fa[i][ks], sa, wm * WM + 16 * (h + i) + li, ks * 32 + 8 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const i32x8_t a8 = cat(fa[i][0], fa[i][1]);
#pragma unroll
          for (int j = 0; j < 4; ++j)      // D[row = n][col = m]: srcA = weights (e4m3), srcB = activations (blgp 0 e4m3 / 1 e5m2)
            acc[h + i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat(fb[j][0], fb[j][1]), a8, acc[h + i][j], 0,
                                                                             F8 == 2 ? 1 : 0, 0, 0, 0, 0);
        }
      }
      return;
    }
    constexpr int NKS = (PT_GEMM_ABLATE == 1 || PT_GEMM_ABLATE == 4) ? 0 : BK / 32;
    if constexpr (std::is_same<T, float>::value && !TA && !TB && !ATOMIC && F8 == 0) {
      if (p.x3) {                                            // wave-uniform: f32 operands, bf16 x 3 products
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const int kb = ks * 32 + 8 * g;
          FragX3 xa[MI], xb[4];
#pragma unroll
          for (int i = 0; i < MI; ++i) { Frag<T> f; frag_load_k(f, sa, wm * WM + 16 * i + li, kb); xa[i] = split_x3(f); }
#pragma unroll
          for (int j = 0; j < 4; ++j) { Frag<T> f; frag_load_k(f, sb, wn * 64 + 16 * j + li, kb); xb[j] = split_x3(f); }
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mma16x3(acc[i][j], xb[j], xa[i]);   // D[row = n][col = m]
        }
        return;
      }
    }
    if constexpr (X2) {                                      // k-tile = [32 hi | 32 lo] of both operands: one bf16 x 3 step
      if (NKS > 0) {
        FragX3 xb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          Frag<T> h, l;
          frag_load_k(h, sb, wn * 64 + 16 * j + li, 8 * g); frag_load_k(l, sb, wn * 64 + 16 * j + li, 32 + 8 * g);
          xb[j].hi = h.v; xb[j].lo = l.v;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          Frag<T> h, l;
          frag_load_k(h, sa, wm * WM + 16 * i + li, 8 * g); frag_load_k(l, sa, wm * WM + 16 * i + li, 32 + 8 * g);
          FragX3 xa; xa.hi = h.v; xa.lo = l.v;
#pragma unroll
          for (int j = 0; j < 4; ++j) mma16x3(acc[i][j], xb[j], xa);      // D[row = n][col = m]
        }
      }
      return;
    }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      Frag<T> fa[MI], fb[4];
      const int kb = ks * 32 + 8 * g;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if (!TA) frag_load_k(fa[i], sa, wm * WM + 16 * i + li, kb);
        else     frag_load_t<BM>(fa[i], sa, wm * WM + 16 * i, kb, kb + 4, lane);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!TB) frag_load_k(fb[j], sb, wn * 64 + 16 * j + li, kb);
        else     frag_load_t<BN>(fb[j], sb, wn * 64 + 16 * j, kb, kb + 4, lane);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (ATOMIC) mma16(acc[i][j], fa[i], fb[j]);   // D[row = m][col = n]
          else        mma16(acc[i][j], fb[j], fa[i]);   // D[row = n][col = m]: 4 consecutive n per lane
        }
      if (ATOMIC && do_sum) {
#pragma unroll
        for (int i = 0; i < MI; ++i) mma16(accb[i], fa[i], fones);
      }
    }
  };
  if (NSTAGE == 2) {
    stage(kt_begin, 0);
    PT_STAMP(1);
    if (PT_GEMM_ASM_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();          // hipcc drains vmcnt(0) before the barrier while LDS-DMA is outstanding
    PT_STAMP(2);
    int cur = 0;
#if PT_GEMM_TRACE
    __shared__ unsigned long long ktr[32 * 4];      // per k-tile: loop top, loads issued, MFMAs issued, loads landed (then the barrier)
#define PT_KSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && kt - kt_begin < 32) ktr[(kt - kt_begin) * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PT_KSTAMP(i) do { } while (0)
#endif
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      PT_KSTAMP(0);
      if (kt + 1 < kt_end) stage(kt + 1, cur ^ 1);
      PT_KSTAMP(1);
      compute(cur);
      PT_KSTAMP(2);
      if (PT_GEMM_ASM_DMA || PT_GEMM_TRACE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PT_KSTAMP(3);
      __syncthreads();
      cur ^= 1;
    }
#if PT_GEMM_TRACE
    if (blockIdx.x == 0 && threadIdx.x < 128) pt_trace[8 + threadIdx.x] = ktr[threadIdx.x];
#endif
  } else {
    constexpr int PER_TILE = Cfg::A_CHUNKS + Cfg::B_CHUNKS;     // LDS-DMA instructions a thread issues per k-tile
    stage(kt_begin, 0);
    if (kt_begin + 1 < kt_end) stage(kt_begin + 1, 1);
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      // retire k-tile kt (the younger one stays in flight), then rendezvous: every wave's pieces of kt have landed
      // AND every wave is done reading the buffer that stage(kt + 2) overwrites
      if (kt + 1 < kt_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 2 < kt_end) stage(kt + 2, cur == 0 ? 2 : cur - 1);
      compute(cur);
      cur = cur == 2 ? 0 : cur + 1;
    }
    __syncthreads();
  }

  if (PT_GEMM_ABLATE == 6) {             // probe: everything but the epilogue code (accumulators kept live: no dead MFMAs)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  if (ATOMIC && do_sum && li == 0) {     // column 0 of the ones product: rows 4g + r of each 16-row block
    float* dst = p.arow_sum + (int64_t)(blockIdx.x % p.arow_rep) * p.arow_stride;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + wm * WM + 16 * i + 4 * g + r;
        if (m < p.arow_n) unsafeAtomicAdd(dst + m, p.alpha * accb[i][r]);
      }
  }
  if (F8) {                               // per-tensor dequantisation
    const float sc = p.scale_a[0] * p.scale_b[0];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] *= sc;
  }
  PT_STAMP(3);
  // every wave is past the loop's last barrier: the operand stages are dead and serve as per-row-block transposition scratch
  static_assert(Cfg::NWAVES * MI * SCRATCH_PER_WAVE <= NSTAGE * STAGE_BYTES, "stages hold one scratch per row block per wave");
  gemm_epilogue<T, ATOMIC, MI, BM, BN, X2>(p, acc, m0, n0, wm, wn, lane, smem + wave * (MI * SCRATCH_PER_WAVE), SCRATCH_PER_WAVE);
  PT_STAMP(5); PT_STAMP_RT(7);
}

// =====================================================================================================================
// 256 x 256 "eight-phase" kernel (bf16): the deep-pipelined variant for problems with at least a round of 256x256 tiles.
// The 2-stage loop above has every load of k-tile t+1 issued at once and drained (vmcnt(0)) one MFMA block later: bytes in
// flight collapse to zero around every barrier, and the loop runs at load latency + transfer time.  Here a k-tile is
// staged as four 16 KiB UNITS -- a0/a1 = the first/second 64 rows of each wave-row's 128-row A panel, b0/b1 = the first /
// second 32 columns of each wave-column's 64-column B panel -- consumed one output quadrant per phase:
//     phase 0: (a0,b0)   phase 1: (a0,b1)   phase 2: (a1,b1)   phase 3: (a1,b0; b0 stays in registers)
// so a unit's LDS is dead two phases after its last read and is restaged with the k-tile two ahead while the current one
// is still being multiplied: 4-5 units (64-80 KiB) are in flight at all times, retired by COUNTED s_waitcnt vmcnt(N)
// (never 0 in the steady state) in front of raw s_barriers.
//     issue order:  tile t phase 0: a1(t+1)   1: b1(t+1)   2: a0(t+2)   3: b0(t+2)
//     waits:        end of phase 3: vmcnt(8) retires a0,b0(t+1);   end of phase 0: vmcnt(6) retires a1,b1(t)
// Each phase = {fragment reads + one unit's LDS-DMA} barrier {16 MFMA} barrier, and the two wave-rows run one barrier
// apart (the wr = 1 waves take one extra barrier up front, wr = 0 one at the end): on every SIMD one wave multiplies while
// the other reads LDS / issues loads.  Hazards with that stagger: a unit is restaged >= 2 phases after its last ds_read,
// and read >= 1 phase after the wait that retires it.
// Unit images are TileK ([128 rows][128 B]) or TileT<128> ([64 k][256 B]) exactly as in the kernel above, so every
// operand kind / transposition goes through the same vaddr() and fragment loaders.
// =====================================================================================================================
constexpr int P8_UNIT = 16384, P8_BUF = 4 * P8_UNIT, P8_THREADS = 512;

// OUT: 0 = store epilogue (bias / residual / activation), 1 = f32 atomics into C, 2 = SLAB: the workgroup's raw 256 x 256 f32
// partial tile goes to `slab` with whole-wave 1 KiB stores (pt_wgrad_group: split-K partials summed by wgrad_fold_kernel).
// `bid` is the workgroup's index inside its problem: tiles x split_k, K-slice major.
// F8: 0 = bf16 operands; 1 / 2 = fp8 operands, K-contiguous (no transposed images), B (the weights) e4m3 and A (activations
// / output gradients) e4m3 (1) or e5m2 (2): a k-tile is 128 elements -- the same 128-byte rows, unit images, swizzle and
// staging schedule -- and the two 16-byte fragment reads of a row feed ONE v_mfma_f32_16x16x128_f8f6f4 instead of two
// 16x16x32 bf16 MFMAs.  Which k a byte of a fragment stands for is irrelevant as long as both operands are read alike.
template <bool TA, bool TB, int OUT, int KA, int KB, int F8 = 0, bool X2 = false>
__device__ __forceinline__ void gemm8p_body(const GemmParams& p, const int bid, float* __restrict__ slab, char* smem) {
  using T = bf16_t;
  using TO = typename std::conditional<F8 != 0, f8_t, bf16_t>::type;     // operand element (addressing)
  static_assert(F8 == 0 || (!TA && !TB && OUT == 0), "fp8 operands: K-contiguous, store epilogue");
  constexpr bool ATOMIC = OUT == 1;            // accumulator orientation D[row = m][col = n] (contiguous n per atomic)
  constexpr int BM = 256, BN = 256, BK = F8 ? 128 : 64, EPC = F8 ? 16 : 8;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int g = lane >> 4, li = lane & 15;

  // grid.x = tiles x split_k, K-slice major: the run of ids an XCD owns is (nearly) one K-slice of every tile, so a split-K
  // wgrad streams each operand through one L2 once (tile-major slices made every XCD re-fetch whole panels: 3x the bytes)
  const int ntile = p.tiles_m * p.tiles_n;
  const int kslice = bid / ntile, tix = bid - kslice * ntile;
  const int tm = tix / p.tiles_n, tn = tix - tm * p.tiles_n;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  const int nkt_total = (int)((p.K + BK - 1) / BK);
  const int per = (nkt_total + p.split_k - 1) / p.split_k;
  const int kt_begin = kslice * per;
  const int kt_end = min(nkt_total, kt_begin + per);
  if (kt_begin >= kt_end) return;                    // whole workgroup: no barrier has been executed yet
  const int nt = kt_end - kt_begin;

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // SLAB mode: the bias gradient sum_k VA(m,k) as an all-ones MFMA product, in the first tile column's workgroups; the 128
  // rows of a wave-row are shared out over its four waves (wave-column wc takes the 16-row block wc of each 64-row half)
  const bool do_sum = OUT == 2 && p.arow_sum != nullptr && tn == 0;
  f32x4_t accb[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
  Frag<T> fones;
  {
    const float one8[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    frag_from_f32(fones, one8);
  }

  // ---- staging state: unit U = 2 * operand + half; two 16-byte chunks per thread per unit ----
  const char* ptr[4][2]; uint32_t stp[4][2]; int cnt[4] = {0, 0, 0, 0};
  const char* zero_page = reinterpret_cast<const char*>(pt_zero_page);
  const int last_kt = nkt_total - 1;
  const bool ktail = (p.K % BK) != 0;
  auto fast_tiles_after = [&](const VOp& o, int kt) {
    const int pos = kt % o.seg_kt;
    int d = o.seg_kt - pos;
    if (o.edge_slow) { if (pos == 0) d = 1; else if (pos < o.seg_kt - 1) d = min(d, o.seg_kt - 1 - pos); }
    if (ktail && kt < last_kt) d = min(d, last_kt - kt);
    return d - 1;
  };
  const int wbase = wave * 64;
  // unit-local index -> tile row (A) / tile column (B): u = (wave-row or wave-col) * (64 | 32) + offset
  auto stage = [&](auto oper_c, auto half_c, int kt, char* buf) {
    constexpr int OPER = decltype(oper_c)::value, S = decltype(half_c)::value, U = 2 * OPER + S;
    constexpr bool TR = OPER == 0 ? TA : TB;
    constexpr int KC = OPER == 0 ? KA : KB;
    const VOp& op = OPER == 0 ? p.A : p.B;
    const int64_t t0 = OPER == 0 ? m0 : n0;
    char* dst = buf + U * P8_UNIT;
    if (cnt[U] == 0) {
      const int64_t k0 = (int64_t)kt * BK;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = tid + P8_THREADS * i;
        if (!TR) {
          const int u = q >> 3, c = (q & 7) ^ (u & 7);
          const int trow = OPER == 0 ? ((u >> 6) * 128 + S * 64 + (u & 63)) : ((u >> 5) * 64 + S * 32 + (u & 31));
          ptr[U][i] = vaddr<TO, KC, X2>(op, t0 + trow, k0 + c * EPC);
        } else {
          const int k = q >> 4, uc = ((q & 15) ^ tilet_swz(k)) * EPC;
          const int tcol = OPER == 0 ? ((uc >> 6) * 128 + S * 64 + (uc & 63)) : ((uc >> 5) * 64 + S * 32 + (uc & 31));
          ptr[U][i] = vaddr<TO, KC, X2>(op, k0 + k, t0 + tcol);
        }
        stp[U][i] = ptr[U][i] == zero_page ? 0u : (uint32_t)op.step;
      }
      cnt[U] = fast_tiles_after(op, kt);
    } else {
      --cnt[U];
#pragma unroll
      for (int i = 0; i < 2; ++i) ptr[U][i] += stp[U][i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((pt_gptr*)ptr[U][i], (pt_lptr*)(dst + (wbase + P8_THREADS * i) * 16), 16, 0, 0);
  };
  using std::integral_constant;
  integral_constant<int, 0> OA, H0; integral_constant<int, 1> OB, H1;

  Frag<T> fa[4][2], fb0[2][2], fb1[2][2];
  // Transposed (TileT<128>) fragment addresses from ONE lane-dependent base per operand.  For k = 32 ks + 8 g + 4 hi + q
  // (q = (lane & 15) >> 2) the image swizzle is tilet_swz(k) = (q << 1) | ((hi ^ g) & 1) << 3, independent of ks, and a
  // fragment's chunk index c = (16-column block) * 2 + (p >> 1) has its block number in bits the lane part never carries
  // into, so   chunk_off(k, c) = [lane base] ^ (((2 blk) ^ (8 hi)) << 4)  +  (4 hi + 32 ks) * 256.
  // The XOR is issued right at the read (one VALU each); written as plain chunk_off() calls, hipcc hoists all 2 buffers x
  // 2 halves x 24 loop-invariant addresses out of the k loop and then spills the staging pointers INTO the loop, where every
  // reload is followed by s_waitcnt vmcnt(0) and drains the LDS-DMA pipeline.
  const int tq = li >> 2, tp = li & 3;
  const int tswz = (tq << 1) | ((g & 1) << 3);
  const int t_lane_a = (8 * g + tq) * 256 + ((((wr * 8) + (tp >> 1)) ^ tswz) << 4) + ((tp & 1) << 3);
  const int t_lane_b = (8 * g + tq) * 256 + ((((wc * 4) + (tp >> 1)) ^ tswz) << 4) + ((tp & 1) << 3);
  auto tr_frag = [&](Frag<T>& f, const char* img, int lane_base, int blk, int ks) {
    typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
    int a_lo, a_hi;
    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a_lo) : "v"(lane_base), "v"((2 * blk) << 4));
    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a_hi) : "v"(lane_base), "v"(((2 * blk) ^ 8) << 4));
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + a_lo + ks * 32 * 256));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + a_hi + (ks * 32 + 4) * 256));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    f.v = __builtin_bit_cast(bf16x8_t, r);
  };
  auto read_a = [&](const char* buf, int s) {
    const char* img = buf + s * P8_UNIT;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int kb = ks * 32 + 8 * g;
        if (!TA) frag_load_k(fa[i][ks], img, wr * 64 + 16 * i + li, kb);
        else     tr_frag(fa[i][ks], img, t_lane_a, i, ks);
      }
  };
  auto read_b = [&](Frag<T> (&fb)[2][2], const char* buf, int s) {
    const char* img = buf + (2 + s) * P8_UNIT;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int kb = ks * 32 + 8 * g;
        if (!TB) frag_load_k(fb[j][ks], img, wc * 32 + 16 * j + li, kb);
        else     tr_frag(fb[j][ks], img, t_lane_b, j, ks);
      }
  };
  // one output quadrant: rows 64 sa .. +63, columns 32 sb .. +31 of the wave's 128 x 64 tile, over the whole k-tile
  auto quadrant = [&](auto sa_c, auto sb_c, const Frag<T> (&fb)[2][2]) {
    constexpr int SA = decltype(sa_c)::value, SB = decltype(sb_c)::value;
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if (F8) {
      typedef __attribute__((ext_vector_type(8))) int i32x8_t;
      auto cat = [](const Frag<T>& lo, const Frag<T>& hi) {
        const u32x4_t a = __builtin_bit_cast(u32x4_t, lo.v), b = __builtin_bit_cast(u32x4_t, hi.v);
        return (i32x8_t){(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
      };
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const i32x8_t a8 = cat(fa[i][0], fa[i][1]);
#pragma unroll
        for (int j = 0; j < 2; ++j)      // D[row = n][col = m]: srcA = weights (e4m3), srcB = activations (blgp: 0 e4m3, 1 e5m2)
          acc[4 * SA + i][2 * SB + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
              cat(fb[j][0], fb[j][1]), a8, acc[4 * SA + i][2 * SB + j], 0, F8 == 2 ? 1 : 0, 0, 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (ATOMIC) mma16(acc[4 * SA + i][2 * SB + j], fa[i][ks], fb[j][ks]);   // D[row = m][col = n]
          else        mma16(acc[4 * SA + i][2 * SB + j], fb[j][ks], fa[i][ks]);   // D[row = n][col = m]
        }
    }
    if (OUT == 2 && SB == 0 && do_sum) {       // phases 0 and 3: every column of the product is sum_k A[m][k]
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (wc == i) { mma16(accb[SA], fones, fa[i][0]); mma16(accb[SA], fones, fa[i][1]); }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
#define P8_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
  // HAS1 / HAS2: k-tiles t+1 / t+2 exist (the last two tiles of the range issue fewer units, so their counts differ)
  auto tile = [&](auto has1_c, auto has2_c, int kt, char* cur, char* nxt) {
    constexpr bool HAS1 = decltype(has1_c)::value, HAS2 = decltype(has2_c)::value;
    // phase 0
    read_b(fb0, cur, 0); __builtin_amdgcn_sched_barrier(0); read_a(cur, 0);
    if (HAS1) { stage(OA, H1, kt + 1, nxt); P8_VMCNT(6); } else { P8_VMCNT(0); }
    quadrant(H0, H0, fb0);
    // phase 1
    read_b(fb1, cur, 1);
    if (HAS1) stage(OB, H1, kt + 1, nxt);
    quadrant(H0, H1, fb1);
    // phase 2
    read_a(cur, 1);
    if (HAS2) stage(OA, H0, kt + 2, cur);
    quadrant(H1, H1, fb1);
    // phase 3
    if (HAS2) { stage(OB, H0, kt + 2, cur); P8_VMCNT(8); } else if (HAS1) { P8_VMCNT(4); }
    quadrant(H1, H0, fb0);
  };
  integral_constant<bool, true> YES; integral_constant<bool, false> NO;

  // ---- prologue: a0,b0,a1,b1 of the first k-tile, a0,b0 of the second ----
  stage(OA, H0, kt_begin, smem); stage(OB, H0, kt_begin, smem);
  stage(OA, H1, kt_begin, smem); stage(OB, H1, kt_begin, smem);
  if (nt > 1) { stage(OA, H0, kt_begin + 1, smem + P8_BUF); stage(OB, H0, kt_begin + 1, smem + P8_BUF); P8_VMCNT(8); }
  else { P8_VMCNT(4); }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();         // stagger the wave-rows by one barrier

  int lt = 0;
  for (; lt + 2 < nt; ++lt) {
    char* cur = smem + (lt & 1) * P8_BUF; char* nxt = smem + ((lt & 1) ^ 1) * P8_BUF;
    tile(YES, YES, kt_begin + lt, cur, nxt);
  }
  if (lt + 1 < nt) {
    char* cur = smem + (lt & 1) * P8_BUF; char* nxt = smem + ((lt & 1) ^ 1) * P8_BUF;
    tile(YES, NO, kt_begin + lt, cur, nxt);
    ++lt;
  }
  {
    char* cur = smem + (lt & 1) * P8_BUF; char* nxt = smem + ((lt & 1) ^ 1) * P8_BUF;
    tile(NO, NO, kt_begin + lt, cur, nxt);
  }
#undef P8_VMCNT
  if (wr == 0) __builtin_amdgcn_s_barrier();         // re-align: every wave has finished its LDS reads past this point
  __builtin_amdgcn_s_barrier();
  if (PT_GEMM_ABLATE == 6) {              // probe: everything but the epilogue (accumulators kept live)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  if (OUT == 2) {
    // raw partial tile: register (i, j) of every wave is one whole-wave 1 KiB store; element r of lane l in it is
    // C[m0 + wr*128 + 16 i + (l & 15)][n0 + wc*64 + 16 j + 4 (l >> 4) + r]  (decoded again by wgrad_fold_kernel)
    float* dst = slab + ((size_t)wave * 32 * 64 + lane) * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4_t*>(dst + (i * 4 + j) * 256) = acc[i][j];
    if (do_sum && g == 0) {
      float* bdst = p.arow_sum + (int64_t)(blockIdx.x % p.arow_rep) * p.arow_stride;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int64_t m = m0 + wr * 128 + 64 * h + 16 * wc + li;
        if (m < p.arow_n) unsafeAtomicAdd(bdst + (p.geglu_rows > 0 ? geglu_orig_row(m, p.geglu_rows) : m), p.alpha * accb[h][0]);
      }
    }
    return;
  }
  if (F8) {                               // per-tensor dequantisation: the product of the operands' device-resident factors
    const float sc = p.scale_a[0] * p.scale_b[0];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] *= sc;
  }
  gemm_epilogue<T, ATOMIC, 8, BM, BN, X2>(p, acc, m0, n0, wr, wc, lane, smem + wave * (8 * SCRATCH_PER_WAVE), SCRATCH_PER_WAVE);
}

__device__ __forceinline__ int xcd_contiguous_id(int bid, int nwg) {
  // XCD-aware order: blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous run of ids.  Bijective.
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <bool TA, bool TB, bool ATOMIC, int KA, int KB, bool X2 = false>
__global__ __launch_bounds__(P8_THREADS, 1) void gemm8p_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * P8_BUF];     // the ONLY LDS object (epilogue scratch aliases it)
  gemm8p_body<TA, TB, ATOMIC ? 1 : 0, KA, KB, 0, X2>(p, xcd_contiguous_id(blockIdx.x, gridDim.x), nullptr, smem);
}

template <int F8>
__global__ __launch_bounds__(P8_THREADS, 1) void gemm8p_f8_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * P8_BUF];
  gemm8p_body<false, false, 0, 0, 0, F8>(p, xcd_contiguous_id(blockIdx.x, gridDim.x), nullptr, smem);
}

// =====================================================================================================================
// Grouped weight gradients (pt_wgrad_group): ONE launch computes up to PT_WG_MAX weight-gradient GEMMs dW_p = dY_p^T X_p
// (both operands read where they lie, reduction over tokens on the operand rows) on the eight-phase body above.
// A weight gradient is a small output (4..32 tiles of 256 x 256) under a 8 192..32 768-row reduction, so filling 256 CUs
// needs split-K, and every workgroup ends with a 256 KiB f32 partial tile whatever the problem: launched one GEMM at a time
// that is 64 MiB of partials per launch (as memory-side float atomics at ~1.3 TB/s: 49 us per launch, more than the MFMA
// work of most of them).  Grouping the weight gradients of one transformer block / resnet into one launch keeps the 256
// workgroups busy with 3-10 splits per problem, i.e. ONE set of partials per group; the partials leave as plain 16-byte
// stores (whole-wave 1 KiB lines, ~5 TB/s) into a slab workspace and wgrad_fold_kernel adds the splits into the flat
// gradient buffer.  Workgroups are dealt out problem-major / K-slice-major in XCD-contiguous order.
// =====================================================================================================================
constexpr int PT_WG_MAX = 8;
struct WgradGroup {
  GemmParams p[PT_WG_MAX];
  int wg_end[PT_WG_MAX];         // exclusive end of problem i's run of workgroup ids
  int nprob;
  float* slabs;                  // [total workgroups][256 * 256] f32
};
static_assert(sizeof(WgradGroup) <= 4096, "kernel argument segment is 4 KiB");

template <int KB>
__global__ __launch_bounds__(P8_THREADS, 1) void wgrad8p_group_kernel(const WgradGroup g) {
  __shared__ __attribute__((aligned(16))) char smem[2 * P8_BUF];
  const int bid = xcd_contiguous_id(blockIdx.x, gridDim.x);
  int pid = 0;
#pragma unroll
  for (int i = 0; i < PT_WG_MAX - 1; ++i) pid += (i < g.nprob - 1 && bid >= g.wg_end[i]) ? 1 : 0;
  const int base = pid ? g.wg_end[pid - 1] : 0;
  gemm8p_body<true, true, 2, 0, KB>(g.p[pid], bid - base, g.slabs + (size_t)bid * (256 * 256), smem);
}

struct FoldProb { float* C; int64_t ldc, M, N, geglu_rows; int tiles_n, ntile, split_k, wg_base; float alpha; int blk_end; };
struct FoldGroup { FoldProb f[PT_WG_MAX]; int nprob; const float* slabs; };

// C[m][n .. n+3] += alpha * sum over the K-slices of the tile's partials; one thread per 16-byte quad of a tile,
// 64 blocks of 256 threads per tile; slab reads are fully coalesced (quad-major), gradient rows get 64-byte segments.
__global__ __launch_bounds__(256) void wgrad_fold_kernel(const FoldGroup g) {
  int pid = 0;
#pragma unroll
  for (int i = 0; i < PT_WG_MAX - 1; ++i) pid += (i < g.nprob - 1 && (int)blockIdx.x >= g.f[i].blk_end) ? 1 : 0;
  const FoldProb& f = g.f[pid];
  const int lb = blockIdx.x - (pid ? g.f[pid - 1].blk_end : 0);
  const int tix = lb >> 6, q = ((lb & 63) << 8) + threadIdx.x;          // tile, quad inside the tile (0 .. 16383)
  const float* src = g.slabs + ((size_t)f.wg_base + tix) * (256 * 256) + (size_t)q * 4;
  const size_t sstride = (size_t)f.ntile * (256 * 256);
  f32x4_t s0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, s1 = s0;
  int s = 0;
  for (; s + 1 < f.split_k; s += 2) {
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(src + (size_t)s * sstride);
    const f32x4_t b = *reinterpret_cast<const f32x4_t*>(src + (size_t)(s + 1) * sstride);
    s0 += a; s1 += b;
  }
  if (s < f.split_k) s0 += *reinterpret_cast<const f32x4_t*>(src + (size_t)s * sstride);
  s0 += s1;
  const int lane = q & 63, reg = (q >> 6) & 31, wave = q >> 11;
  const int tm = tix / f.tiles_n, tn = tix - tm * f.tiles_n;
  const int64_t m = (int64_t)tm * 256 + (wave >> 2) * 128 + 16 * (reg >> 2) + (lane & 15);
  const int64_t n = (int64_t)tn * 256 + (wave & 3) * 64 + 16 * (reg & 3) + 4 * (lane >> 4);
  if (m >= f.M || n >= f.N) return;
  float* dst = f.C + (f.geglu_rows > 0 ? geglu_orig_row(m, f.geglu_rows) : m) * f.ldc + n;
  if (n + 3 < f.N && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
    f32x4_t c = *reinterpret_cast<const f32x4_t*>(dst);
    c[0] += f.alpha * s0[0]; c[1] += f.alpha * s0[1]; c[2] += f.alpha * s0[2]; c[3] += f.alpha * s0[3];
    *reinterpret_cast<f32x4_t*>(dst) = c;
  } else {
    for (int r = 0; r < 4 && n + r < f.N; ++r) dst[r] += f.alpha * s0[r];
  }
}

// x2_map: 0 = plain storage; else PT_BF16X2 plane rows, cols = 2 K virtual columns (see VOp)
VOp make_vop(const pt_operand& o, int64_t rows, int64_t cols, int es, int x2_map = 0) {
  VOp v;
  v.krep = 0; v.plane1 = v.plane2 = 0; v.pmap = 0;
  const int64_t K1 = x2_map ? cols / 2 : cols;                 // logical columns
  if (x2_map) {
    v.krep = (int)K1; v.pmap = 0;
    if (o.kind == PT_V_CONV) v.plane1 = o.cin;
    else if (o.kind == PT_V_CONCAT) { v.plane1 = (int)o.c_split; v.plane2 = (int)(K1 - o.c_split); }
    else v.plane1 = (int)K1;
  }
  v.p = reinterpret_cast<const char*>(o.p); v.p2 = reinterpret_cast<const char*>(o.p2);
  v.ld = o.ld; v.ld2 = o.ld2; v.c_split = o.c_split;
  v.kind = o.kind; v.taps = o.taps; v.cin = o.cin > 0 ? o.cin : 1; v.rowmap = o.rowmap; v.stride = o.stride > 0 ? o.stride : 1;
  v.n_out = (int)(o.n_out > 0 ? o.n_out : 1); v.n_in = (int)o.n_in;
  auto lg = [](int x) { return (x > 0 && (x & (x - 1)) == 0) ? __builtin_ctz(x) : -1; };
  v.cin_shift = lg(v.cin); v.nout_shift = lg(v.n_out);
  v.rows = rows; v.cols = cols;
  // segment / step of the incremental addressing (see stage()): bk elements per k-tile
  const int bk = 128 / es;
  v.seg_kt = 1; v.step = 0; v.edge_slow = 0;
  if (!o.trans) {                               // reduction along columns
    v.step = (int64_t)bk * es;
    if (o.kind == PT_V_PLAIN) v.seg_kt = 1 << 30;
    else if (o.kind == PT_V_CONCAT) v.seg_kt = (o.c_split % bk == 0) ? (int)(o.c_split / bk) : 1;
    else if (o.kind == PT_V_CONV) v.seg_kt = (v.cin % bk == 0) ? v.cin / bk : 1;
    if (x2_map) {       // a k-tile advances 32 logical columns in both planes (build_params: K, cin, c_split are multiples of 32)
      v.step = 32 * es;
      if (o.kind == PT_V_CONCAT) v.seg_kt = (int)(o.c_split / 32);
      else if (o.kind == PT_V_CONV) v.seg_kt = v.cin / 32;
    }
  } else {                                      // reduction along rows
    if (o.kind == PT_V_PLAIN) { v.seg_kt = 1 << 30; v.step = (int64_t)bk * o.ld * es; }
    else if (o.kind == PT_V_CONCAT && o.ld == o.ld2) { v.seg_kt = 1 << 30; v.step = (int64_t)bk * o.ld * es; }
    else if (o.kind == PT_V_WFLIP && v.cin % bk == 0) { v.seg_kt = v.cin / bk; v.step = (int64_t)bk * 3 * o.ld * es; }
    else if (o.kind == PT_V_CONV && v.n_out % bk == 0 && v.n_out / bk >= 4 && o.rowmap <= PT_MAP_UP2) {
      // wgrad operand: rows walk one batch item (n_out rows = seg_kt k-tiles) linearly; only the first / last rows of
      // the item can be padding, so the first two and the last k-tile of each item are recomputed
      v.seg_kt = v.n_out / bk; v.edge_slow = 1;
      const int64_t rows_per_kt = o.rowmap == PT_MAP_S1 ? bk : (o.rowmap == PT_MAP_S2 ? 2 * bk : bk / 2);
      v.step = rows_per_kt * o.ld * es;
    }
  }
  return v;
}

int check_operand(const pt_operand& o, int esize) {
  if (!o.p || !pt_aligned16(o.p)) return PT_ERR_ALIGN;
  if ((o.ld * esize) % 16 != 0) return PT_ERR_ALIGN;
  if (o.kind == PT_V_CONCAT) {
    if (!o.p2 || !pt_aligned16(o.p2) || (o.ld2 * esize) % 16 != 0 || (o.c_split * esize) % 16 != 0) return PT_ERR_ALIGN;
  } else if (o.kind == PT_V_CONV) {
    if (o.taps < 1 || o.taps > 16) return PT_ERR_ARG;
    if (o.cin <= 0 || (o.cin * esize) % 16 != 0 || o.n_out <= 0 || o.n_in <= 0) return PT_ERR_SHAPE;
    if (o.rowmap < PT_MAP_S1 || o.rowmap > PT_MAP_STRIDED_REFLECT) return PT_ERR_ARG;
    if (o.rowmap == PT_MAP_STRIDED_REFLECT && (o.stride < 1 || o.taps < o.stride || o.n_in < o.taps)) return PT_ERR_SHAPE;
    if (o.rowmap == PT_MAP_CAUSAL_REFLECT && o.n_in < o.taps) return PT_ERR_SHAPE;   // reflect needs n_in > pad
  } else if (o.kind == PT_V_WFLIP) {
    if (o.cin <= 0) return PT_ERR_SHAPE;
  } else if (o.kind != PT_V_PLAIN) {
    return PT_ERR_ARG;
  }
  return PT_OK;
}

template <typename T, bool TA, bool TB, bool ATOMIC, int KA, int KB, int BM, int BN, bool X2 = false>
int launch_cfg(GemmParams p, hipStream_t s) {
  p.tiles_m = (int)((p.M + BM - 1) / BM); p.tiles_n = (int)((p.N + BN - 1) / BN);
  if ((int64_t)p.tiles_m * p.tiles_n * p.split_k >= (1ll << 31)) return PT_ERR_SHAPE;
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n * p.split_k), 1, 1);
  hipLaunchKernelGGL((gemm_kernel<T, TA, TB, ATOMIC, KA, KB, BM, BN, 0, X2>), grid, dim3(TileCfg<BM, BN>::NTHREADS), 0, s, p);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

// Short reductions (K < PT_GEMM_8P_MIN_K, default 1536: the K = 512 linears of config B, the K = 1024 ones of config E) take the two-stage 256 x 256 kernel instead of the
// eight-phase one: with 8 k-tiles per tile the eight-phase prologue / staggered drain is a third of the tile's life, and the
// simpler kernel is 8 - 15 % faster alone (tools/tile_probe.py), 1.2 ms per step in the benchmark and 1 % of a config E step.
// Tile choice.  PT_GEMM_TILE=128|256|512|8 (= 128x128 | 256x128 | 256x256 two-stage | 256x256 eight-phase) forces one
// configuration for A/B probing (tools/gemm_probe.py); PT_GEMM_8P_MASK selects which GEMM classes may use the eight-phase
// kernel (bit 0 forward plain, 1 dgrad plain, 2 conv forward, 3 conv dgrad, 4 wgrad).
inline int pick_tile(const GemmParams& p, int cls, bool bf16, bool x2 = false) {
  static const int forced = pt_env_int("PT_GEMM_TILE", 0), mask = pt_env_int("PT_GEMM_8P_MASK", 0x0f),
                   min8pk = pt_env_int("PT_GEMM_8P_MIN_K", 1536);
  if (forced == 128 || forced == 256 || forced == 512 || forced == 8) return forced;
  const int64_t tiles256 = ((p.M + 255) / 256) * ((p.N + 255) / 256) * p.split_k;
  if (bf16 && ((mask >> cls) & 1) && tiles256 >= 192) {
    // a last column tile that is at most half full and a sixth or more of the tile columns (the Encodec decoder's N = 128 and
    // N = 640 transposed-conv GEMMs): 256 x 128 tiles waste nothing (0.353 -> 0.204 ms and 0.650 -> 0.606 ms, tools/decode_probe.py)
    const int64_t rem = p.N % 256, cols = (p.N + 255) / 256 * 256;
    // (plane operands: 128 x 128, two workgroups per CU whose staging overlaps -- N = 640: 1.40 -> 1.32 ms, N = 128: 0.44 -> 0.41)
    if (rem > 0 && rem <= 128 && 6 * (256 - rem) >= cols) return x2 ? 128 : 256;
    return p.K >= min8pk ? 8 : 512;
  }
  return 128;
}

template <bool TA, bool TB, bool ATOMIC, int KA, int KB, bool X2 = false>
int launch_8p(GemmParams p, hipStream_t s) {
  p.tiles_m = (int)((p.M + 255) / 256); p.tiles_n = (int)((p.N + 255) / 256);
  if ((int64_t)p.tiles_m * p.tiles_n * p.split_k >= (1ll << 31)) return PT_ERR_SHAPE;
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n * p.split_k), 1, 1);
  hipLaunchKernelGGL((gemm8p_kernel<TA, TB, ATOMIC, KA, KB, X2>), grid, dim3(P8_THREADS), 0, s, p);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

template <typename T, bool TA, bool TB, bool ATOMIC, int KA, int KB, bool X2 = false>
int launch(const GemmParams& p, hipStream_t s) {
  constexpr int cls = ATOMIC ? 4 : (!TB ? (KA == 0 ? 0 : 2) : (KA == 0 ? 1 : 3));
  int tile = pick_tile(p, cls, sizeof(T) == 2, X2);
  if (p.arow_sum && tile != 128) tile = 128;            // the fused bias gradient lives in the two-stage 128 x 128 kernel
  if constexpr (X2) { if (tile == 8) tile = 512; }      // plane operands: the two-stage kernels carry the bf16 x 3 k-tile
  if constexpr (sizeof(T) == 2 && !X2) { if (tile == 8) return launch_8p<TA, TB, ATOMIC, KA, KB, X2>(p, s); }
  switch (tile) {
    case 512: return launch_cfg<T, TA, TB, ATOMIC, KA, KB, 256, 256, X2>(p, s);
    case 256: return launch_cfg<T, TA, TB, ATOMIC, KA, KB, 256, 128, X2>(p, s);
    default: return launch_cfg<T, TA, TB, ATOMIC, KA, KB, 128, 128, X2>(p, s);
  }
}

inline int kind_class(int kind) { return kind == PT_V_CONV ? 1 : (kind == PT_V_WFLIP ? 2 : 0); }

// Only the operand combinations the training step issues are instantiated (forward, dgrad, wgrad of linear / conv).
template <typename T>
int dispatch(const GemmParams& p, bool ta, bool tb, hipStream_t s) {
  const bool atomic = p.out_kind == PT_OUT_F32_ATOMIC;
  const int ka = kind_class(p.A.kind), kb = kind_class(p.B.kind);
  if (!ta && !tb && !atomic && kb == 0) {                        // forward: linear / conv1x1 / conv k3
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (p.A.krep > 0) {                                        // PT_BF16X2 plane operands: their own instantiations
        if (ka == 0) return launch<T, false, false, false, 0, 0, true>(p, s);
        if (ka == 1) return launch<T, false, false, false, 1, 0, true>(p, s);
      }
    }
    if (ka == 0) return launch<T, false, false, false, 0, 0>(p, s);
    if (ka == 1) return launch<T, false, false, false, 1, 0>(p, s);
  }
  if constexpr (std::is_same<T, float>::value) {                 // skinny f32 forward under split-K (time-embedding projections)
    if (!ta && !tb && atomic && ka == 0 && kb == 0) return launch_cfg<T, false, false, true, 0, 0, 128, 128>(p, s);
  }
  if (!ta && tb && !atomic) {                                    // dgrad
    if (ka == 0 && kb == 0) return launch<T, false, true, false, 0, 0>(p, s);
    if (ka == 1 && kb == 2) return launch<T, false, true, false, 1, 2>(p, s);
  }
  if (ta && tb && atomic && ka == 0) {                           // wgrad (split-K, f32 atomics)
    if (kb == 0) return launch<T, true, true, true, 0, 0>(p, s);
    if (kb == 1) return launch<T, true, true, true, 0, 1>(p, s);
  }
  return PT_ERR_ARG;
}

// descriptor checks + conversion shared by pt_gemm and pt_wgrad_group
static int build_params(const pt_gemm_desc* d, int dtype, GemmParams& p, int operand_es = 0) {
  if (!d) return PT_ERR_ARG;
  if (dtype != PT_F32 && dtype != PT_BF16 && dtype != PT_BF16X2) return PT_ERR_DTYPE;
  const bool x2 = dtype == PT_BF16X2;
  if (x2) {       // forward GEMMs of f32-class inference only: K-contiguous operands, store epilogue, bias / ELU, one or two outputs
    if (operand_es != 0 || d->A.trans || d->B.trans || d->out_kind == PT_OUT_F32_ATOMIC || d->split_k != 1 || d->residual || d->residual2 ||
        d->row_bias || d->act > 1 || d->B.kind != PT_V_PLAIN || d->A.kind == PT_V_WFLIP || d->K % 32 != 0 || d->N % 8 != 0 || 2 * d->K >= (1ll << 31))
      return PT_ERR_ARG;
    if (d->A.kind == PT_V_CONCAT && (d->A.c_split <= 0 || d->A.c_split >= d->K || d->A.c_split % 32 != 0)) return PT_ERR_ARG;
    if (d->A.kind == PT_V_CONV && d->A.cin % 32 != 0) return PT_ERR_ARG;
  }
  const int oes_t = dtype == PT_F32 ? 4 : 2;                 // output / residual element size
  const int es = operand_es > 0 ? operand_es : oes_t;        // operand element size (1: fp8 operands, bf16 output)
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return PT_ERR_SHAPE;
  if (d->M >= (1ll << 31) || d->N >= (1ll << 31) || d->K >= (1ll << 31)) return PT_ERR_SHAPE;
  int st;
  if ((st = check_operand(d->A, es)) != PT_OK) return st;
  if ((st = check_operand(d->B, es)) != PT_OK) return st;
  if (!d->C) return PT_ERR_ARG;
  if (d->out_kind < PT_OUT_T || d->out_kind > PT_OUT_F32_ATOMIC) return PT_ERR_ARG;
  if (d->split_k < 1 || (d->split_k > 1 && d->out_kind != PT_OUT_F32_ATOMIC)) return PT_ERR_ARG;
  if (d->out_kind == PT_OUT_F32_ATOMIC && (d->bias || d->row_bias || d->residual || d->residual2 || d->C2 || d->act)) return PT_ERR_ARG;
  if (d->out_kind != PT_OUT_F32_ATOMIC) {
    const int oes = d->out_kind == PT_OUT_F32 ? 4 : oes_t;
    if ((reinterpret_cast<uintptr_t>(d->C) & 15u) || (d->ldc * oes) % (4 * oes) != 0) return PT_ERR_ALIGN;
    if (d->bias && (reinterpret_cast<uintptr_t>(d->bias) & 15u)) return PT_ERR_ALIGN;
    if (d->row_bias && ((reinterpret_cast<uintptr_t>(d->row_bias) & 15u) || d->row_bias_rows <= 0 || (d->N % 4) || (d->row_bias_ld % 4))) return PT_ERR_ALIGN;
    if (d->conv_wgrad_cin > 0) return PT_ERR_ARG;
    if (d->C2 && ((reinterpret_cast<uintptr_t>(d->C2) & 15u) || d->ldc2 % 4 != 0)) return PT_ERR_ALIGN;
  }
  // reduction extent must be whole 16-byte chunks when it lies along operand columns
  p.M = d->M; p.N = d->N; p.K = x2 ? 2 * d->K : d->K;
  if (x2) {       // virtual columns: k-tile t = [hi | lo] of logical columns 32 t .. 32 t + 31 (VOp)
    p.A = make_vop(d->A, d->M, 2 * d->K, es, 1);
    p.B = make_vop(d->B, d->N, 2 * d->K, es, 1);
  } else {
    p.A = d->A.trans ? make_vop(d->A, d->K, d->M, es) : make_vop(d->A, d->M, d->K, es);
    p.B = d->B.trans ? make_vop(d->B, d->K, d->N, es) : make_vop(d->B, d->N, d->K, es);
  }
  if (p.A.step >= (1ll << 32) || p.B.step >= (1ll << 32)) { p.A.seg_kt = 1; p.B.seg_kt = 1; p.A.edge_slow = p.B.edge_slow = 0; }   // absurd strides: always recompute
  p.C = reinterpret_cast<char*>(d->C); p.ldc = d->ldc;
  p.out_kind = d->out_kind; p.split_k = d->split_k;
  p.bias = d->bias; p.row_bias = d->row_bias; p.row_bias_rows = d->row_bias_rows; p.row_bias_ld = d->row_bias_ld > 0 ? d->row_bias_ld : d->N;
  p.residual = reinterpret_cast<const char*>(d->residual); p.ldr = d->ldr;
  p.residual2 = reinterpret_cast<const char*>(d->residual2); p.ldr2 = d->ldr2;
  p.conv_wgrad_cin = d->conv_wgrad_cin;
  p.conv_wgrad_cin_store = d->conv_wgrad_cin_store > 0 ? d->conv_wgrad_cin_store : d->conv_wgrad_cin;
  p.alpha = d->alpha;
  p.act = d->act; p.act2 = d->act2; p.C2 = reinterpret_cast<char*>(d->C2); p.ldc2 = d->ldc2;
  p.arow_sum = d->arow_sum; p.arow_n = d->arow_n; p.arow_stride = d->arow_stride; p.arow_rep = d->arow_rep > 0 ? d->arow_rep : 1;
  if (p.arow_sum && (d->out_kind != PT_OUT_F32_ATOMIC || d->arow_n <= 0 || d->arow_n > d->M || (p.arow_rep > 1 && d->arow_stride < d->arow_n))) return PT_ERR_ARG;
  p.tiles_m = p.tiles_n = 0;   // set per tile configuration at launch
  p.geglu_rows = d->geglu_rows;
  p.scale_a = p.scale_b = nullptr;
  { static const int nt = pt_env_int("PT_GEMM_NT", 0); p.nt_store = nt; }
  p.x3 = (dtype == PT_F32 && d->f32_x3) ? 1 : 0;
  if (x2 && (d->x2_block < 0 || (d->x2_block > 0 && (d->x2_block % 8 != 0 || d->N % d->x2_block != 0)))) return PT_ERR_ARG;
  const int64_t pblock = d->x2_block > 0 ? d->x2_block : d->N;
  p.planes_c = (x2 && d->out_kind == PT_OUT_T) ? (int)pblock : 0; p.planes_c2 = (x2 && d->C2 && d->out_kind == PT_OUT_T) ? (int)pblock : 0;
  if (x2 && d->out_kind == PT_OUT_T && (d->ldc < 2 * d->N || d->ldc % 8 != 0 || (d->C2 && (d->ldc2 < 2 * d->N || d->ldc2 % 8 != 0)))) return PT_ERR_ARG;
  if (d->act < 0 || d->act > 3 || d->act2 < 0 || d->act2 > 1 || d->geglu_rows < 0) return PT_ERR_ARG;
  if (d->geglu_rows > 0 && (d->out_kind != PT_OUT_F32_ATOMIC || d->M != 2 * d->geglu_rows || d->geglu_rows % 32 != 0)) return PT_ERR_ARG;
  if (d->act >= 2) {
    // fused GEGLU epilogues exist on the predicate-free bf16 path only: whole tiles, 16-byte aligned rows
    if (dtype != PT_BF16 || d->out_kind != PT_OUT_T || d->M % 256 != 0 || d->N % 256 != 0 || d->ldc % 8 != 0) return PT_ERR_ARG;
    if (d->act == 2 && (!d->C2 || d->ldc2 % 8 != 0 || d->residual || d->residual2 || d->row_bias)) return PT_ERR_ARG;
    if (d->act == 3 && (!d->residual || d->ldr % 8 != 0 || (reinterpret_cast<uintptr_t>(d->residual) & 15u) || d->bias ||
                        d->row_bias || d->residual2 || d->C2 || d->ldc < 2 * d->N)) return PT_ERR_ARG;
  }
  return PT_OK;
}

}  // namespace

extern "C" int pt_gemm(const pt_gemm_desc* d, int dtype, pt_stream stream) {
  GemmParams p;
  const int st = build_params(d, dtype, p);
  if (st != PT_OK) return st;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PT_F32) return dispatch<float>(p, d->A.trans != 0, d->B.trans != 0, s);
  return dispatch<bf16_t>(p, d->A.trans != 0, d->B.trans != 0, s);      // PT_BF16 and PT_BF16X2 (plane operands: bf16 x 3 k-tiles)
}

// fp8 operands (e4m3 weights; e4m3 or e5m2 activations / gradients), bf16 output, f32 accumulation, every epilogue of pt_gemm.
extern "C" int pt_gemm_fp8(const pt_gemm_desc* d, int a_format, const float* scale_a, const float* scale_b, pt_stream stream) {
  GemmParams p;
  const int st = build_params(d, PT_BF16, p, 1);
  if (st != PT_OK) return st;
  if (a_format != PT_FP8_E4M3 && a_format != PT_FP8_E5M2) return PT_ERR_DTYPE;
  if (!scale_a || !scale_b) return PT_ERR_ARG;
  if (d->A.trans || d->B.trans || d->A.kind != PT_V_PLAIN || d->B.kind != PT_V_PLAIN) return PT_ERR_ARG;   // K-contiguous plain operands
  if (d->out_kind != PT_OUT_T || d->split_k != 1 || d->K % 16 != 0) return PT_ERR_ARG;
  p.scale_a = scale_a; p.scale_b = scale_b;
  p.tiles_m = (int)((p.M + 255) / 256); p.tiles_n = (int)((p.N + 255) / 256);
  if ((int64_t)p.tiles_m * p.tiles_n >= (1ll << 31)) return PT_ERR_SHAPE;
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n), 1, 1);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // short reductions (K < PT_GEMM_F8_8P_MIN_K bytes, default 3072 = 24 k-tiles) on the two-stage 256 x 256 kernel, like bf16
  static const int min8pk = pt_env_int("PT_GEMM_F8_8P_MIN_K", 3072);
  if (p.K < min8pk) {
    constexpr int NT256 = TileCfg<256, 256>::NTHREADS;
    if (a_format == PT_FP8_E4M3) hipLaunchKernelGGL((gemm_kernel<bf16_t, false, false, false, 0, 0, 256, 256, 1>), grid, dim3(NT256), 0, s, p);
    else                         hipLaunchKernelGGL((gemm_kernel<bf16_t, false, false, false, 0, 0, 256, 256, 2>), grid, dim3(NT256), 0, s, p);
    PT_LAUNCH_CHECK();
    return PT_OK;
  }
  if (a_format == PT_FP8_E4M3) hipLaunchKernelGGL((gemm8p_f8_kernel<1>), grid, dim3(P8_THREADS), 0, s, p);
  else                         hipLaunchKernelGGL((gemm8p_f8_kernel<2>), grid, dim3(P8_THREADS), 0, s, p);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

#if PT_GEMM_TRACE
extern "C" int pt_debug_gemm_trace(unsigned long long* out8) {       // diagnostic builds only (tools/gemm_probe.py --trace)
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(pt_trace), sizeof(unsigned long long) * 8) == hipSuccess ? PT_OK : PT_ERR_LAUNCH;
}
extern "C" int pt_debug_gemm_ktrace(unsigned long long* out136) {
  return hipMemcpyFromSymbol(out136, HIP_SYMBOL(pt_trace), sizeof(unsigned long long) * 136) == hipSuccess ? PT_OK : PT_ERR_LAUNCH;
}
#endif

extern "C" int64_t pt_wgrad_group_ws_floats(int target_wgs) {
  return (int64_t)(target_wgs > 0 ? target_wgs : 256) * 256 * 256;
}

extern "C" int pt_wgrad_group(const pt_gemm_desc* descs, int n, float* ws, int64_t ws_floats, int target_wgs, pt_stream stream) {
  if (!descs || n < 1 || n > PT_WG_MAX || !ws || (reinterpret_cast<uintptr_t>(ws) & 15u)) return PT_ERR_ARG;
  if (target_wgs <= 0) target_wgs = 256;
  WgradGroup g; FoldGroup fg;
  int kb = -1;
  int64_t tiles_total = 0;
  int nkt[PT_WG_MAX], tiles[PT_WG_MAX];
  for (int i = 0; i < n; ++i) {
    const pt_gemm_desc* d = descs + i;
    const int st = build_params(d, PT_BF16, g.p[i]);
    if (st != PT_OK) return st;
    if (d->out_kind != PT_OUT_F32_ATOMIC || !d->A.trans || !d->B.trans || d->conv_wgrad_cin > 0) return PT_ERR_ARG;
    if (kind_class(d->A.kind) != 0) return PT_ERR_ARG;
    const int k = kind_class(d->B.kind);
    if (k == 2 || (kb >= 0 && k != kb)) return PT_ERR_ARG;          // one B operand class per group (plain/concat or conv gather)
    kb = k;
    if ((reinterpret_cast<uintptr_t>(d->C) & 3u) != 0) return PT_ERR_ALIGN;
    g.p[i].tiles_m = (int)((d->M + 255) / 256); g.p[i].tiles_n = (int)((d->N + 255) / 256);
    tiles[i] = g.p[i].tiles_m * g.p[i].tiles_n;
    nkt[i] = (int)((d->K + 63) / 64);
    tiles_total += tiles[i];
  }
  if (tiles_total > target_wgs) return PT_ERR_SHAPE;
  // smallest per-workgroup share w (k-tiles) whose split counts fit the target: every workgroup then has <= w k-tiles
  auto wgs_for = [&](int w) { int64_t t = 0; for (int i = 0; i < n; ++i) t += (int64_t)tiles[i] * ((nkt[i] + w - 1) / w); return t; };
  int lo = 1, hi = 1;
  for (int i = 0; i < n; ++i) hi = nkt[i] > hi ? nkt[i] : hi;
  while (lo < hi) { const int mid = (lo + hi) / 2; if (wgs_for(mid) <= target_wgs) hi = mid; else lo = mid + 1; }
  int base = 0, blk = 0;
  for (int i = 0; i < n; ++i) {
    int split = (nkt[i] + lo - 1) / lo;
    const int per = (nkt[i] + split - 1) / split;
    split = (nkt[i] + per - 1) / per;                                // no empty K-slice: every workgroup writes its slab
    g.p[i].split_k = split;
    FoldProb& f = fg.f[i];
    f.C = reinterpret_cast<float*>(g.p[i].C); f.ldc = g.p[i].ldc; f.M = g.p[i].M; f.N = g.p[i].N;
    f.geglu_rows = g.p[i].geglu_rows; f.tiles_n = g.p[i].tiles_n; f.ntile = tiles[i]; f.split_k = split; f.wg_base = base; f.alpha = g.p[i].alpha;
    base += tiles[i] * split; blk += tiles[i] * 64;
    g.wg_end[i] = base; f.blk_end = blk;
  }
  for (int i = n; i < PT_WG_MAX; ++i) { g.wg_end[i] = base; fg.f[i] = fg.f[n - 1]; g.p[i] = g.p[n - 1]; }
  if ((int64_t)base * 256 * 256 > ws_floats) return PT_ERR_ARG;
  g.nprob = fg.nprob = n; g.slabs = ws; fg.slabs = ws;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (kb == 0) hipLaunchKernelGGL((wgrad8p_group_kernel<0>), dim3((unsigned)base), dim3(P8_THREADS), 0, s, g);
  else         hipLaunchKernelGGL((wgrad8p_group_kernel<1>), dim3((unsigned)base), dim3(P8_THREADS), 0, s, g);
  PT_LAUNCH_CHECK();
  hipLaunchKernelGGL(wgrad_fold_kernel, dim3((unsigned)blk), dim3(256), 0, s, fg);
  PT_LAUNCH_CHECK();
  return PT_OK;
}
