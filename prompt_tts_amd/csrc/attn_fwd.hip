// Flash attention forward for gfx950.  Workgroup = 4 waves x 32 query rows; K/V stream through a
// double-buffered LDS ring in 64-key tiles (register-staged prefetch, one barrier per tile).
// MFMA-bound: 4*Nq*Nk*D flops per (batch, head) against (Nq + 2 Nk) * D * sizeof(T) bytes.
#include "attn_common.h"

namespace {

template <typename T, int D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p) {
  using Cfg = AttnCfg<T, D>;
  constexpr int KS = Cfg::KS, DT = Cfg::DT, COLS = Cfg::COLS;
  constexpr int IMG = 64 * Cfg::ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][K image | V image]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  int b, h, blk;
  attn_block_ids((p.Nq + 127) / 128, p.H, blk, h, b);
  const int qblk = blk * 128, q0 = qblk + wave * 32;
  int nk = p.Nk;
  if (p.kv_len) { nk = p.kv_len[b]; nk = nk < 1 ? 1 : (nk > p.Nk ? p.Nk : nk); }
  int klimit = nk;                                   // keys this workgroup has to visit
  if (p.causal) klimit = min(nk, qblk + 128);
  const int ntiles = (klimit + 63) / 64;

  const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.Nq * p.ldq + h * D;
  const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.Nk * p.ldk + h * D;
  const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * p.Nk * p.ldv + h * D;

  Frag<T> fq[2][KS];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    int row = q0 + 16 * qt + li; row = row < p.Nq ? row : p.Nq - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) frag_load_global(fq[qt][ks], Q + (int64_t)row * p.ldq + ks * 32 + 8 * g);
  }

  constexpr bool FAST = sizeof(T) == 2;          // bf16: hoisted LDS offsets; f32 (parity mode): generic addressing
  FragOffsets<T, D> fo; fo.init(lane);
  f32x4_t o[DT][2];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { o[dt][0] = (f32x4_t){0, 0, 0, 0}; o[dt][1] = (f32x4_t){0, 0, 0, 0}; }
  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
  const float sl2 = p.scale * PT_LOG2E;

  TileStage<T, D, 64> sk, sv;
  sk.load(K, p.ldk, 0, nk, tid); sv.load(V, p.ldv, 0, nk, tid);
  sk.store(smem, tid); sv.store(smem + IMG, tid);
  __syncthreads();

  int cur = 0;
  for (int t = 0; t < ntiles; ++t) {
    const bool more = t + 1 < ntiles;
    if (more) { sk.load(K, p.ldk, (t + 1) * 64, nk, tid); sv.load(V, p.ldv, (t + 1) * 64, nk, tid); }
    const char* kimg = smem + cur * 2 * IMG;
    const char* vimg = kimg + IMG;

    // S^T[key][q] = K Q^T
    f32x4_t s[4][2];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt][0] = (f32x4_t){0, 0, 0, 0}; s[kt][1] = (f32x4_t){0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        Frag<T> fk;
        if constexpr (FAST) frag_read_rows<Cfg::ROWB>(fk, kimg, fo.rowread[ks], 16 * kt);
        else frag_load_n<COLS>(fk, kimg, 16 * kt + li, ks * 32 + 8 * g);
        mma16(s[kt][0], fk, fq[0][ks]);
        mma16(s[kt][1], fk, fq[1][ks]);
      }
    }
    const int key0 = t * 64;
    if (key0 + 64 > nk || p.causal) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = key0 + 16 * kt + 4 * g + r, q = q0 + 16 * qt + li;
            if (key >= nk || (p.causal && key > q)) s[kt][qt][r] = -INFINITY;
          }
    }
    // online softmax; statistics are per lane (q = lane&15), partial row sums stay per lane group
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][qt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m[qt], mx);
      const float alpha = __builtin_amdgcn_exp2f((m[qt] - mn) * sl2);
      m[qt] = mn;
      float rs = 0.f;
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      const f32x2_t sl2v = {sl2, sl2}, mn2v = {mn * sl2, mn * sl2};
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {          // s * sl2 - mn * sl2 as one packed FMA per pair (the loop is VALU-bound)
          const f32x2_t sv = {s[kt][qt][r], s[kt][qt][r + 1]};
          const f32x2_t tv = __builtin_elementwise_fma(sv, sl2v, -mn2v);
          const float p0 = __builtin_amdgcn_exp2f(tv[0]), p1 = __builtin_amdgcn_exp2f(tv[1]);
          s[kt][qt][r] = p0; s[kt][qt][r + 1] = p1; rs += p0 + p1;
        }
      l[qt] = l[qt] * alpha + rs;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) o[dt][qt] *= alpha;
    }
    // O^T[d][q] += V^T P^T
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      Frag<T> fp[2];
      frag_from_acc(fp[0], s[2 * st][0], s[2 * st + 1][0]);
      frag_from_acc(fp[1], s[2 * st][1], s[2 * st + 1][1]);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        Frag<T> fv;
        if constexpr (FAST) frag_read_tr<Cfg::ROWB>(fv, vimg, fo.trread[dt], 32 * st);
        else frag_load_t<COLS>(fv, vimg, 16 * dt, 32 * st + 4 * g, 32 * st + 16 + 4 * g, lane);
        mma16(o[dt][0], fv, fp[0]);
        mma16(o[dt][1], fv, fp[1]);
      }
    }
    if (more) { sk.store(smem + (cur ^ 1) * 2 * IMG, tid); sv.store(smem + (cur ^ 1) * 2 * IMG + IMG, tid); }
    __syncthreads();
    cur ^= 1;
  }

  T* O = reinterpret_cast<T*>(p.o) + (int64_t)b * p.Nq * p.ldo + h * D;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float lt = l[qt];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const float inv = 1.f / lt;
    const int q = q0 + 16 * qt + li;
    if (q < p.Nq) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        store4<T>(O + (int64_t)q * p.ldo + 16 * dt + 4 * g, o[dt][qt][0] * inv, o[dt][qt][1] * inv, o[dt][qt][2] * inv, o[dt][qt][3] * inv);
      if (g == 0 && p.lse) p.lse[((int64_t)b * p.H + h) * p.Nq + q] = m[qt] * p.scale + logf(lt);
    }
  }
}

template <typename T, int D> int launch_fwd(const AttnParams& p, hipStream_t s) {
  using Cfg = AttnCfg<T, D>;
  const size_t lds = 2 * 2 * 64 * (size_t)Cfg::ROWB;
  static bool attr_set = false;   // idempotent; racing setters write the same value
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<T, D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PT_ERR_LAUNCH;
    attr_set = true;
  }
  const int64_t nwg = (int64_t)((p.Nq + 127) / 128) * p.H * p.B;
  if (nwg >= (1ll << 31)) return PT_ERR_SHAPE;
  dim3 grid((unsigned)nwg);
  hipLaunchKernelGGL((attn_fwd_kernel<T, D>), grid, dim3(256), lds, s, p);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

}  // namespace

int pt_attn_fill_params(const pt_attn_desc* d, int dtype, bool bwd, AttnParams& p);   // attn_bwd.hip
int pt_attn2_fwd(const AttnParams& p, int D, hipStream_t s);                           // attn2_fwd.hip

extern "C" int pt_attn_fwd(const pt_attn_desc* d, int dtype, pt_stream stream) {
  AttnParams p;
  int st = pt_attn_fill_params(d, dtype, false, p);
  if (st) return st;
  hipStream_t s = (hipStream_t)stream;
  // bf16: the second-generation kernel (32x32x16 MFMA, LDS-DMA staging, lazy maximum); PT_ATTN_V2=0 keeps the round-2 kernel
  static const int v2 = pt_env_int("PT_ATTN_V2", 1);
  if (dtype == PT_BF16 && v2) return pt_attn2_fwd(p, (int)d->D, s);
#define FWD(TT) \
  switch (d->D) { case 32: return launch_fwd<TT, 32>(p, s); case 64: return launch_fwd<TT, 64>(p, s); \
                  case 128: return launch_fwd<TT, 128>(p, s); default: return PT_ERR_SHAPE; }
  if (dtype == PT_F32) { FWD(float) }
  FWD(bf16_t)
#undef FWD
}
