// GEMM family for gfx950: C[m][n] (+)= sum_k VA(m,k) VB(n,k), 128x128 workgroup tile, 4 waves (2x2) of
// 64x64, MFMA 16x16 atoms (mma.h).  Operand tiles go HBM -> LDS directly (global_load_lds_dwordx4, no VGPR
// staging): the LDS image is lane-linear per wave-instruction, so the XOR swizzle is applied to the per-lane
// SOURCE address (and again on the fragment read); chunks that must read as zero (conv padding, M/N/K tails)
// point at a zero page.  Two LDS stages: the loads of k-tile t+1 are in flight under the MFMAs of k-tile t,
// one barrier per k-tile; 64 KiB LDS and <= 256 registers keep two workgroups resident per CU.
//
// Operands are "virtual matrices" (include/prompt_tts_hip.h): plain, channel-concat, conv-gather (implicit
// GEMM for Conv1d k=3 stride 1/2, upsample+conv, and their dgrad/wgrad) and flipped conv weights; each is
// consumed either with the reduction index along its columns (TileK image) or along its rows (TileT image,
// transposed fragment reads), so forward, dgrad and wgrad all read the tensors where they lie in HBM:
// no transposes, no im2col, no concat copies.
//
// Roofline: MFMA-bound for K >= 512 (2*128*128*K flops per 2*128*K*sizeof(T) operand bytes per tile).
#include "mma.h"

namespace {

struct VOp {
  const char* p; const char* p2;
  int64_t ld, ld2, c_split;
  int kind, taps, cin, rowmap;
  int n_out, n_in;
  int cin_shift, nout_shift;   // log2 when a power of two, else -1
  int64_t rows, cols;   // logical extent of the virtual matrix
};

struct GemmParams {
  VOp A, B;
  int64_t M, N, K;
  char* C; int64_t ldc;
  int out_kind, split_k;
  const float* bias; const float* row_bias; int64_t row_bias_rows;
  const char* residual; int64_t ldr;
  const char* residual2; int64_t ldr2;
  int conv_wgrad_cin, conv_wgrad_cin_store;
  float alpha;
  int act, act2; char* C2; int64_t ldc2;
  int tiles_m, tiles_n;
};

__device__ __attribute__((aligned(256))) const uint32_t pt_zero_page[64] = {0};

typedef __attribute__((address_space(1))) const void pt_gptr;
typedef __attribute__((address_space(3))) void pt_lptr;

__device__ __forceinline__ void divmod(int x, int d, int shift, int& q, int& r) {
  if (shift >= 0) { q = x >> shift; r = x & (d - 1); }
  else { q = x / d; r = x - q * d; }
}

// address of the 16-byte chunk V[row][col .. col+EPC) of a virtual matrix, or the zero page
// KC = kind class, fixed at compile time so that only one addressing scheme's invariants occupy registers:
// 0 plain / channel-concat, 1 conv gather, 2 flipped conv weights
template <typename T, int KC>
__device__ __forceinline__ const char* vaddr(const VOp& op, int64_t row, int64_t col) {
  const char* zero = reinterpret_cast<const char*>(pt_zero_page);
  if (row >= op.rows || col >= op.cols) return zero;
  const T* ptr;
  if (KC == 0) {
    ptr = (op.kind == PT_V_PLAIN || col < op.c_split)
              ? reinterpret_cast<const T*>(op.p) + row * op.ld + col
              : reinterpret_cast<const T*>(op.p2) + row * op.ld2 + (col - op.c_split);
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
      default: /* PT_MAP_BACK */ ns = n - tap; ok = ns >= 0; break;
    }
    if (!ok) return zero;
    ptr = reinterpret_cast<const T*>(op.p) + ((int64_t)b * op.n_in + ns) * op.ld + ci;
  } else {  // PT_V_WFLIP: row = tap*cout + co
    int tap, co;
    divmod((int)row, op.cin, op.cin_shift, tap, co);
    ptr = reinterpret_cast<const T*>(op.p) + ((int64_t)co * 3 + (2 - tap)) * op.ld + col;
  }
  return reinterpret_cast<const char*>(ptr);
}

constexpr int BM = 128, BN = 128;
constexpr int STAGE_BYTES = 16384;   // one operand tile image

template <typename T, bool TA, bool TB, bool ATOMIC, int KA, int KB>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
  constexpr int BK = TileK<T>::KE;                 // 64 (bf16) / 32 (f32)
  constexpr int EPC = 16 / (int)sizeof(T);         // elements per 16-byte chunk
  constexpr int TCH = 128 / EPC;                   // chunks per TileT row (128 columns)
  __shared__ __attribute__((aligned(16))) char smem[4 * STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, li = lane & 15;

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous
  // run of tiles so neighbours that share an A row-panel hit the same L2.  Bijective for any grid size.
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  const int nkt_total = (int)((p.K + BK - 1) / BK);
  const int per = (nkt_total + p.split_k - 1) / p.split_k;
  const int kt_begin = blockIdx.z * per;
  const int kt_end = min(nkt_total, kt_begin + per);
  if (kt_begin >= kt_end) return;

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // One wave-instruction of global_load_lds writes 64 x 16 B = 1 KiB of LDS linearly (base + lane*16).  Chunk slot
  // q of a tile image therefore holds, for TileK, row q>>3 / data chunk (q&7)^(row&7); for TileT, k-row q/TCH /
  // data chunk (q%TCH)^swz(k): the swizzle lives in the SOURCE address.
  const int wbase = __builtin_amdgcn_readfirstlane(wave) * 64;
  auto stage = [&](int kt, int stg) {
    const int64_t k0 = (int64_t)kt * BK;
    char* sa = smem + stg * 2 * STAGE_BYTES;
    char* sb = sa + STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 256 * i;
      const char* ga; const char* gb;
      if (!TA) { const int r = q >> 3; ga = vaddr<T, KA>(p.A, m0 + r, k0 + ((q & 7) ^ (r & 7)) * EPC); }
      else     { const int k = q / TCH; ga = vaddr<T, KA>(p.A, k0 + k, m0 + ((q % TCH) ^ tilet_swz(k)) * EPC); }
      if (!TB) { const int r = q >> 3; gb = vaddr<T, KB>(p.B, n0 + r, k0 + ((q & 7) ^ (r & 7)) * EPC); }
      else     { const int k = q / TCH; gb = vaddr<T, KB>(p.B, k0 + k, n0 + ((q % TCH) ^ tilet_swz(k)) * EPC); }
      const int lds_off = (wbase + 256 * i) * 16;
      __builtin_amdgcn_global_load_lds((pt_gptr*)ga, (pt_lptr*)(sa + lds_off), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((pt_gptr*)gb, (pt_lptr*)(sb + lds_off), 16, 0, 0);
    }
  };

  stage(kt_begin, 0);
  __syncthreads();          // hipcc drains vmcnt(0) before the barrier while LDS-DMA is outstanding

  int cur = 0;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const bool more = kt + 1 < kt_end;
    if (more) stage(kt + 1, cur ^ 1);
    const char* sa = smem + cur * 2 * STAGE_BYTES;
    const char* sb = sa + STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      Frag<T> fa[4], fb[4];
      const int kb = ks * 32 + 8 * g;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!TA) frag_load_k(fa[i], sa, wm * 64 + 16 * i + li, kb);
        else     frag_load_t<128>(fa[i], sa, wm * 64 + 16 * i, kb, kb + 4, lane);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!TB) frag_load_k(fb[j], sb, wn * 64 + 16 * j + li, kb);
        else     frag_load_t<128>(fb[j], sb, wn * 64 + 16 * j, kb, kb + 4, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (ATOMIC) mma16(acc[i][j], fa[i], fb[j]);   // D[row = m][col = n]
          else        mma16(acc[i][j], fb[j], fa[i]);   // D[row = n][col = m]: 4 consecutive n per lane
        }
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue ------------------------------------------------------------------------------------
  if (ATOMIC) {
    float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t n = n0 + wn * 64 + 16 * j + li;
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t m = m0 + wm * 64 + 16 * i + 4 * g + r;
          if (m >= p.M) continue;
          int64_t idx;
          if (p.conv_wgrad_cin > 0) {
            const int tap = (int)n / p.conv_wgrad_cin, ci = (int)n - tap * p.conv_wgrad_cin;
            if (ci >= p.conv_wgrad_cin_store) continue;
            idx = (m * p.conv_wgrad_cin_store + ci) * 3 + tap;
          } else {
            idx = m * p.ldc + n;
          }
          unsafeAtomicAdd(C + idx, p.alpha * acc[i][j][r]);
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + wm * 64 + 16 * i + li;
    if (m >= p.M) continue;
    const float* rbias = p.row_bias ? p.row_bias + (m / p.row_bias_rows) * p.N : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + 16 * j + 4 * g;
      if (n >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = p.alpha * acc[i][j][r];
      const bool full = n + 3 < p.N;
      if (full) {
        if (p.bias) { f32x4_t b = *reinterpret_cast<const f32x4_t*>(p.bias + n); v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3]; }
        if (rbias) { f32x4_t b = *reinterpret_cast<const f32x4_t*>(rbias + n); v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3]; }
        if (p.residual) {
          const T* rp = reinterpret_cast<const T*>(p.residual) + m * p.ldr + n;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += to_f32<T>(rp[r]);
        }
        if (p.residual2) {
          const T* rp = reinterpret_cast<const T*>(p.residual2) + m * p.ldr2 + n;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += to_f32<T>(rp[r]);
        }
        auto put = [&](char* base, int64_t ld, int act) {
          float w[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) w[r] = (act == 1 && v[r] < 0.f) ? (__expf(v[r]) - 1.f) : v[r];
          if (p.out_kind == PT_OUT_F32) {
            *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(base) + m * ld + n) = (f32x4_t){w[0], w[1], w[2], w[3]};
          } else if (sizeof(T) == 4) {
            *reinterpret_cast<f32x4_t*>(reinterpret_cast<T*>(base) + m * ld + n) = (f32x4_t){w[0], w[1], w[2], w[3]};
          } else {
            u32x2_t o;
            o[0] = (uint32_t)f32_to_bf16_bits(w[0]) | ((uint32_t)f32_to_bf16_bits(w[1]) << 16);
            o[1] = (uint32_t)f32_to_bf16_bits(w[2]) | ((uint32_t)f32_to_bf16_bits(w[3]) << 16);
            *reinterpret_cast<u32x2_t*>(reinterpret_cast<T*>(base) + m * ld + n) = o;
          }
        };
        put(p.C, p.ldc, p.act);
        if (p.C2) put(p.C2, p.ldc2, p.act2);
      } else {
        for (int r = 0; r < 4 && n + r < p.N; ++r) {
          float x = v[r];
          if (p.bias) x += p.bias[n + r];
          if (rbias) x += rbias[n + r];
          if (p.residual) x += to_f32<T>(reinterpret_cast<const T*>(p.residual)[m * p.ldr + n + r]);
          if (p.residual2) x += to_f32<T>(reinterpret_cast<const T*>(p.residual2)[m * p.ldr2 + n + r]);
          const float x1 = (p.act == 1 && x < 0.f) ? (__expf(x) - 1.f) : x;
          if (p.out_kind == PT_OUT_F32) reinterpret_cast<float*>(p.C)[m * p.ldc + n + r] = x1;
          else reinterpret_cast<T*>(p.C)[m * p.ldc + n + r] = from_f32<T>(x1);
          if (p.C2) {
            const float x2 = (p.act2 == 1 && x < 0.f) ? (__expf(x) - 1.f) : x;
            if (p.out_kind == PT_OUT_F32) reinterpret_cast<float*>(p.C2)[m * p.ldc2 + n + r] = x2;
            else reinterpret_cast<T*>(p.C2)[m * p.ldc2 + n + r] = from_f32<T>(x2);
          }
        }
      }
    }
  }
}

VOp make_vop(const pt_operand& o, int64_t rows, int64_t cols) {
  VOp v;
  v.p = reinterpret_cast<const char*>(o.p); v.p2 = reinterpret_cast<const char*>(o.p2);
  v.ld = o.ld; v.ld2 = o.ld2; v.c_split = o.c_split;
  v.kind = o.kind; v.taps = o.taps; v.cin = o.cin > 0 ? o.cin : 1; v.rowmap = o.rowmap;
  v.n_out = (int)(o.n_out > 0 ? o.n_out : 1); v.n_in = (int)o.n_in;
  auto lg = [](int x) { return (x > 0 && (x & (x - 1)) == 0) ? __builtin_ctz(x) : -1; };
  v.cin_shift = lg(v.cin); v.nout_shift = lg(v.n_out);
  v.rows = rows; v.cols = cols;
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
    if (o.rowmap < PT_MAP_S1 || o.rowmap > PT_MAP_BACK) return PT_ERR_ARG;
    if (o.rowmap == PT_MAP_CAUSAL_REFLECT && o.n_in < o.taps) return PT_ERR_SHAPE;   // reflect needs n_in > pad
  } else if (o.kind == PT_V_WFLIP) {
    if (o.cin <= 0) return PT_ERR_SHAPE;
  } else if (o.kind != PT_V_PLAIN) {
    return PT_ERR_ARG;
  }
  return PT_OK;
}

template <typename T, bool TA, bool TB, bool ATOMIC, int KA, int KB>
int launch(const GemmParams& p, hipStream_t s) {
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n), 1, (unsigned)p.split_k);
  hipLaunchKernelGGL((gemm_kernel<T, TA, TB, ATOMIC, KA, KB>), grid, dim3(256), 0, s, p);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

inline int kind_class(int kind) { return kind == PT_V_CONV ? 1 : (kind == PT_V_WFLIP ? 2 : 0); }

// Only the operand combinations the training step issues are instantiated (forward, dgrad, wgrad of linear / conv).
template <typename T>
int dispatch(const GemmParams& p, bool ta, bool tb, hipStream_t s) {
  const bool atomic = p.out_kind == PT_OUT_F32_ATOMIC;
  const int ka = kind_class(p.A.kind), kb = kind_class(p.B.kind);
  if (!ta && !tb && !atomic && kb == 0) {                        // forward: linear / conv1x1 / conv k3
    if (ka == 0) return launch<T, false, false, false, 0, 0>(p, s);
    if (ka == 1) return launch<T, false, false, false, 1, 0>(p, s);
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

}  // namespace

extern "C" int pt_gemm(const pt_gemm_desc* d, int dtype, pt_stream stream) {
  if (!d) return PT_ERR_ARG;
  if (dtype != PT_F32 && dtype != PT_BF16) return PT_ERR_DTYPE;
  const int es = dtype == PT_F32 ? 4 : 2;
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
    const int oes = d->out_kind == PT_OUT_F32 ? 4 : es;
    if ((reinterpret_cast<uintptr_t>(d->C) & 15u) || (d->ldc * oes) % (4 * oes) != 0) return PT_ERR_ALIGN;
    if (d->bias && (reinterpret_cast<uintptr_t>(d->bias) & 15u)) return PT_ERR_ALIGN;
    if (d->row_bias && ((reinterpret_cast<uintptr_t>(d->row_bias) & 15u) || d->row_bias_rows <= 0 || (d->N % 4))) return PT_ERR_ALIGN;
    if (d->conv_wgrad_cin > 0) return PT_ERR_ARG;
    if (d->C2 && ((reinterpret_cast<uintptr_t>(d->C2) & 15u) || d->ldc2 % 4 != 0)) return PT_ERR_ALIGN;
  }
  // reduction extent must be whole 16-byte chunks when it lies along operand columns
  GemmParams p;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.A = d->A.trans ? make_vop(d->A, d->K, d->M) : make_vop(d->A, d->M, d->K);
  p.B = d->B.trans ? make_vop(d->B, d->K, d->N) : make_vop(d->B, d->N, d->K);
  p.C = reinterpret_cast<char*>(d->C); p.ldc = d->ldc;
  p.out_kind = d->out_kind; p.split_k = d->split_k;
  p.bias = d->bias; p.row_bias = d->row_bias; p.row_bias_rows = d->row_bias_rows;
  p.residual = reinterpret_cast<const char*>(d->residual); p.ldr = d->ldr;
  p.residual2 = reinterpret_cast<const char*>(d->residual2); p.ldr2 = d->ldr2;
  p.conv_wgrad_cin = d->conv_wgrad_cin;
  p.conv_wgrad_cin_store = d->conv_wgrad_cin_store > 0 ? d->conv_wgrad_cin_store : d->conv_wgrad_cin;
  p.alpha = d->alpha;
  p.act = d->act; p.act2 = d->act2; p.C2 = reinterpret_cast<char*>(d->C2); p.ldc2 = d->ldc2;
  p.tiles_m = (int)((d->M + BM - 1) / BM); p.tiles_n = (int)((d->N + BN - 1) / BN);
  if ((int64_t)p.tiles_m * p.tiles_n >= (1ll << 31)) return PT_ERR_SHAPE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PT_F32) return dispatch<float>(p, d->A.trans != 0, d->B.trans != 0, s);
  return dispatch<bf16_t>(p, d->A.trans != 0, d->B.trans != 0, s);
}
