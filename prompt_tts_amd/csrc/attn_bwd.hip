// Flash attention backward for gfx950: P is recomputed from Q, K and the forward's LSE.
//   delta[q] = rowsum(dO * O) is formed inside the dQ kernel (which runs first) and read by the dK/dV kernel
//   dK/dV kernel : a wave owns 32 keys (K, V as register B fragments, dK^T/dV^T accumulators in registers)
//                  and sweeps the queries in 32-row Q/dO tiles staged in LDS   -> no cross-workgroup sums
//   dQ kernel    : a wave owns 32 queries (Q, dO as register B fragments) and sweeps 64-key K/V tiles
// Seven MFMA products per (q,k) tile pair instead of the minimal five: S and dP are formed in both kernels so
// that dQ needs neither atomics nor an ordered hand-off and every gradient is bitwise reproducible.
#include "attn_common.h"

namespace {

// ---- dK, dV -----------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const AttnParams p) {
  using Cfg = AttnCfg<T, D>;
  constexpr int KS = Cfg::KS, DT = Cfg::DT, COLS = Cfg::COLS;
  constexpr int QT = 64;                          // queries per staged tile: two 32-query compute blocks per barrier
  constexpr int IMG = QT * Cfg::ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][Q image | dO image]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  int b, h, blk;
  attn_block_ids((p.Nk + 127) / 128, p.H, blk, h, b);
  const int kblk = blk * 128, k0 = kblk + wave * 32;
  int nk = p.Nk;
  if (p.kv_len) { nk = p.kv_len[b]; nk = nk < 1 ? 1 : (nk > p.Nk ? p.Nk : nk); }

  const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.Nq * p.ldq + h * D;
  const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.Nk * p.ldk + h * D;
  const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * p.Nk * p.ldv + h * D;
  const T* DO = reinterpret_cast<const T*>(p.d_o) + (int64_t)b * p.Nq * p.lddo + h * D;
  const float* LSE = p.lse + ((int64_t)b * p.H + h) * p.Nq;
  const float* DELTA = p.delta + ((int64_t)b * p.H + h) * p.Nq;

  Frag<T> fk[2][KS], fv[2][KS];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    int row = k0 + 16 * kt + li; row = row < p.Nk ? row : p.Nk - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      frag_load_global(fk[kt][ks], K + (int64_t)row * p.ldk + ks * 32 + 8 * g);
      frag_load_global(fv[kt][ks], V + (int64_t)row * p.ldv + ks * 32 + 8 * g);
    }
  }
  f32x4_t dk[DT][2], dv[DT][2];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) { dk[dt][kt] = (f32x4_t){0, 0, 0, 0}; dv[dt][kt] = (f32x4_t){0, 0, 0, 0}; }
  const float sl2 = p.scale * PT_LOG2E;

  // causal: queries before this workgroup's first key see none of its keys
  const int qstart = p.causal ? (kblk / 32) * 32 : 0;
  const int nsteps = (p.Nq - qstart + QT - 1) / QT;

  TileStage<T, D, QT> sq, sdo;
  if (nsteps > 0) {
    sq.load(Q, p.ldq, qstart, p.Nq, tid); sdo.load(DO, p.lddo, qstart, p.Nq, tid);
    sq.store(smem, tid); sdo.store(smem + IMG, tid);
  }
  __syncthreads();

  int cur = 0;
  for (int st = 0; st < nsteps; ++st) {
    const int qs0 = qstart + st * QT;
    const bool more = st + 1 < nsteps;
    if (more) { sq.load(Q, p.ldq, qs0 + QT, p.Nq, tid); sdo.load(DO, p.lddo, qs0 + QT, p.Nq, tid); }
    const char* qimg = smem + cur * 2 * IMG;
    const char* doimg = qimg + IMG;
#pragma unroll
    for (int hf = 0; hf < QT / 32; ++hf) {
    const int qs = qs0 + 32 * hf, ro = 32 * hf;       // this block's first query / its row offset inside the images
    if (qs >= p.Nq) break;

    f32x4_t s[2][2], dp[2][2];     // [q tile][key tile]; lane: key = lane&15, q = 4g + r
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) { s[qt][kt] = (f32x4_t){0, 0, 0, 0}; dp[qt][kt] = (f32x4_t){0, 0, 0, 0}; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        Frag<T> a, d;
        frag_load_n<COLS>(a, qimg, ro + 16 * qt + li, ks * 32 + 8 * g);
        frag_load_n<COLS>(d, doimg, ro + 16 * qt + li, ks * 32 + 8 * g);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) { mma16(s[qt][kt], a, fk[kt][ks]); mma16(dp[qt][kt], d, fv[kt][ks]); }
      }
    }
    // masking is needed only on ragged / causal tiles: keep the common path free of per-element predicates
    const bool need_mask = p.causal || (qs + 32 > p.Nq) || (kblk + 128 > nk);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qb = qs + 16 * qt + 4 * g;
      float ls[4], de[4];
      if (!need_mask) {
        const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(LSE + qb), d4 = *reinterpret_cast<const f32x4_t*>(DELTA + qb);
#pragma unroll
        for (int r = 0; r < 4; ++r) { ls[r] = l4[r] * PT_LOG2E; de[r] = d4[r]; }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = __builtin_amdgcn_exp2f(s[qt][kt][r] * sl2 - ls[r]);
            s[qt][kt][r] = pv;
            dp[qt][kt][r] = pv * (dp[qt][kt][r] - de[r]);
          }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = qb + r; const bool ok = q < p.Nq;
          ls[r] = ok ? LSE[q] * PT_LOG2E : 0.f; de[r] = ok ? DELTA[q] : 0.f;
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const int key = k0 + 16 * kt + li;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = qb + r;
            const bool ok = (q < p.Nq) && (key < nk) && !(p.causal && key > q);
            const float pv = ok ? __builtin_amdgcn_exp2f(s[qt][kt][r] * sl2 - ls[r]) : 0.f;
            s[qt][kt][r] = pv;
            dp[qt][kt][r] = pv * (dp[qt][kt][r] - de[r]);
          }
        }
      }
    }
    Frag<T> fp[2], fds[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) { frag_from_acc(fp[kt], s[0][kt], s[1][kt]); frag_from_acc(fds[kt], dp[0][kt], dp[1][kt]); }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      Frag<T> fdoT, fqT;
      frag_load_t<COLS>(fdoT, doimg, 16 * dt, ro + 4 * g, ro + 16 + 4 * g, lane);
      frag_load_t<COLS>(fqT, qimg, 16 * dt, ro + 4 * g, ro + 16 + 4 * g, lane);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) { mma16(dv[dt][kt], fdoT, fp[kt]); mma16(dk[dt][kt], fqT, fds[kt]); }
    }
    }   // hf
    if (more) { sq.store(smem + (cur ^ 1) * 2 * IMG, tid); sdo.store(smem + (cur ^ 1) * 2 * IMG + IMG, tid); }
    __syncthreads();
    cur ^= 1;
  }

  T* DK = reinterpret_cast<T*>(p.dk) + (int64_t)b * p.Nk * p.lddk + h * D;
  T* DV = reinterpret_cast<T*>(p.dv) + (int64_t)b * p.Nk * p.lddv + h * D;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int key = k0 + 16 * kt + li;
    if (key < p.Nk) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        store4<T>(DK + (int64_t)key * p.lddk + 16 * dt + 4 * g, dk[dt][kt][0] * p.scale, dk[dt][kt][1] * p.scale,
                  dk[dt][kt][2] * p.scale, dk[dt][kt][3] * p.scale);
        store4<T>(DV + (int64_t)key * p.lddv + 16 * dt + 4 * g, dv[dt][kt][0], dv[dt][kt][1], dv[dt][kt][2], dv[dt][kt][3]);
      }
    }
  }
}

// ---- dQ ----------------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const AttnParams p) {
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
  int klimit = nk;
  if (p.causal) klimit = min(nk, qblk + 128);
  const int ntiles = (klimit + 63) / 64;

  const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.Nq * p.ldq + h * D;
  const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.Nk * p.ldk + h * D;
  const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * p.Nk * p.ldv + h * D;
  const T* DO = reinterpret_cast<const T*>(p.d_o) + (int64_t)b * p.Nq * p.lddo + h * D;
  const T* O = reinterpret_cast<const T*>(p.o) + (int64_t)b * p.Nq * p.ldo + h * D;
  const float* LSE = p.lse + ((int64_t)b * p.H + h) * p.Nq;
  float* DELTA = p.delta + ((int64_t)b * p.H + h) * p.Nq;

  // delta[q] = rowsum(dO * O) is formed HERE from the dO fragments this wave owns anyway (lane (li, g) holds 8 consecutive
  // columns per 32-column step of row li: the four g lanes of a row cover all D columns) and published for the dK/dV kernel,
  // which is launched after this one: no separate pass over dO and O.
  Frag<T> fq[2][KS], fdo[2][KS];
  float ls[2], de[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    int row = q0 + 16 * qt + li; row = row < p.Nq ? row : p.Nq - 1;
    ls[qt] = LSE[row] * PT_LOG2E;
    float acc = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      frag_load_global(fq[qt][ks], Q + (int64_t)row * p.ldq + ks * 32 + 8 * g);
      frag_load_global(fdo[qt][ks], DO + (int64_t)row * p.lddo + ks * 32 + 8 * g);
      Frag<T> fo;
      frag_load_global(fo, O + (int64_t)row * p.ldo + ks * 32 + 8 * g);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += (float)fdo[qt][ks].v[e] * (float)fo.v[e];
    }
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    de[qt] = acc;
    if (g == 0 && q0 + 16 * qt + li < p.Nq) DELTA[q0 + 16 * qt + li] = acc;
  }
  f32x4_t dq[DT][2];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { dq[dt][0] = (f32x4_t){0, 0, 0, 0}; dq[dt][1] = (f32x4_t){0, 0, 0, 0}; }
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

    f32x4_t s[4][2], dp[4][2];     // [key tile][q tile]; lane: q = lane&15, key = 4g + r
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) { s[kt][qt] = (f32x4_t){0, 0, 0, 0}; dp[kt][qt] = (f32x4_t){0, 0, 0, 0}; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        Frag<T> a, vv;
        frag_load_n<COLS>(a, kimg, 16 * kt + li, ks * 32 + 8 * g);
        frag_load_n<COLS>(vv, vimg, 16 * kt + li, ks * 32 + 8 * g);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) { mma16(s[kt][qt], a, fq[qt][ks]); mma16(dp[kt][qt], vv, fdo[qt][ks]); }
      }
    }
    const int key0 = t * 64;
    if (!(p.causal || key0 + 64 > nk)) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = __builtin_amdgcn_exp2f(s[kt][qt][r] * sl2 - ls[qt]);
            dp[kt][qt][r] = pv * (dp[kt][qt][r] - de[qt]);
          }
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = key0 + 16 * kt + 4 * g + r, q = q0 + 16 * qt + li;
            const bool ok = (key < nk) && !(p.causal && key > q);
            const float pv = ok ? __builtin_amdgcn_exp2f(s[kt][qt][r] * sl2 - ls[qt]) : 0.f;
            dp[kt][qt][r] = pv * (dp[kt][qt][r] - de[qt]);
          }
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      Frag<T> fds[2];
      frag_from_acc(fds[0], dp[2 * st][0], dp[2 * st + 1][0]);
      frag_from_acc(fds[1], dp[2 * st][1], dp[2 * st + 1][1]);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        Frag<T> fkT;
        frag_load_t<COLS>(fkT, kimg, 16 * dt, 32 * st + 4 * g, 32 * st + 16 + 4 * g, lane);
        mma16(dq[dt][0], fkT, fds[0]);
        mma16(dq[dt][1], fkT, fds[1]);
      }
    }
    if (more) { sk.store(smem + (cur ^ 1) * 2 * IMG, tid); sv.store(smem + (cur ^ 1) * 2 * IMG + IMG, tid); }
    __syncthreads();
    cur ^= 1;
  }

  T* DQ = reinterpret_cast<T*>(p.dq) + (int64_t)b * p.Nq * p.lddq + h * D;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = q0 + 16 * qt + li;
    if (q < p.Nq) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        store4<T>(DQ + (int64_t)q * p.lddq + 16 * dt + 4 * g, dq[dt][qt][0] * p.scale, dq[dt][qt][1] * p.scale,
                  dq[dt][qt][2] * p.scale, dq[dt][qt][3] * p.scale);
    }
  }
}

template <typename K> int set_lds(K kernel, size_t lds, bool& done) {
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PT_ERR_LAUNCH;
    done = true;
  }
  return PT_OK;
}

template <typename T, int D> int launch_bwd(const AttnParams& p, hipStream_t s) {
  using Cfg = AttnCfg<T, D>;
  static bool a1 = false, a2 = false;
  const size_t lds_kv = 2 * 2 * 64 * (size_t)Cfg::ROWB, lds_q = 2 * 2 * 64 * (size_t)Cfg::ROWB;
  int st;
  if ((st = set_lds(&attn_bwd_kv_kernel<T, D>, lds_kv, a1))) return st;
  if ((st = set_lds(&attn_bwd_q_kernel<T, D>, lds_q, a2))) return st;
  // dQ first: it also produces delta (rowsum(dO * O)) for the dK/dV kernel that follows on the same stream
  hipLaunchKernelGGL((attn_bwd_q_kernel<T, D>), dim3((unsigned)(((p.Nq + 127) / 128) * p.H * p.B)), dim3(256), lds_q, s, p);
  hipLaunchKernelGGL((attn_bwd_kv_kernel<T, D>), dim3((unsigned)(((p.Nk + 127) / 128) * p.H * p.B)), dim3(256), lds_kv, s, p);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

}  // namespace

int pt_attn_fill_params(const pt_attn_desc* d, int dtype, bool bwd, AttnParams& p) {
  if (!d) return PT_ERR_ARG;
  if (dtype != PT_F32 && dtype != PT_BF16) return PT_ERR_DTYPE;
  const int es = dtype == PT_F32 ? 4 : 2;
  if (d->B <= 0 || d->H <= 0 || d->Nq <= 0 || d->Nk <= 0) return PT_ERR_SHAPE;
  if (d->D != 32 && d->D != 64 && d->D != 128) return PT_ERR_SHAPE;
  if (d->B > 65535 || d->H > 65535 || d->Nq >= (1 << 30) || d->Nk >= (1 << 30)) return PT_ERR_SHAPE;
  if (!d->q || !d->k || !d->v || !d->o || !d->lse) return PT_ERR_ARG;
  const int64_t need = d->H * d->D;
  if (d->ldq < need || d->ldk < need || d->ldv < need || d->ldo < need) return PT_ERR_SHAPE;
  auto al = [&](const void* ptr, int64_t ld) { return pt_aligned16(ptr) && (ld * es) % 16 == 0; };
  if (!al(d->q, d->ldq) || !al(d->k, d->ldk) || !al(d->v, d->ldv) || !al(d->o, d->ldo)) return PT_ERR_ALIGN;
  p.B = (int)d->B; p.H = (int)d->H; p.Nq = (int)d->Nq; p.Nk = (int)d->Nk;
  p.q = (const char*)d->q; p.ldq = d->ldq; p.k = (const char*)d->k; p.ldk = d->ldk; p.v = (const char*)d->v; p.ldv = d->ldv;
  p.o = (char*)d->o; p.ldo = d->ldo; p.lse = d->lse; p.scale = d->scale; p.causal = d->causal; p.kv_len = d->kv_len;
  p.d_o = nullptr; p.lddo = 0; p.delta = nullptr; p.dq = p.dk = p.dv = nullptr; p.lddq = p.lddk = p.lddv = 0;
  if (bwd) {
    if (!d->d_o || !d->delta || !d->dq || !d->dk || !d->dv) return PT_ERR_ARG;
    if (d->lddo < need || d->lddq < need || d->lddk < need || d->lddv < need) return PT_ERR_SHAPE;
    if (!al(d->d_o, d->lddo) || !al(d->dq, d->lddq) || !al(d->dk, d->lddk) || !al(d->dv, d->lddv)) return PT_ERR_ALIGN;
    p.d_o = (const char*)d->d_o; p.lddo = d->lddo; p.delta = d->delta;
    p.dq = (char*)d->dq; p.lddq = d->lddq; p.dk = (char*)d->dk; p.lddk = d->lddk; p.dv = (char*)d->dv; p.lddv = d->lddv;
  }
  return PT_OK;
}

int pt_attn2_bwd(const AttnParams& p, int D, hipStream_t s);        // attn2_bwd.hip

extern "C" int pt_attn_bwd(const pt_attn_desc* d, int dtype, pt_stream stream) {
  AttnParams p;
  int st = pt_attn_fill_params(d, dtype, true, p);
  if (st) return st;
  hipStream_t s = (hipStream_t)stream;
  // bf16: the second-generation kernels (attn2_bwd.hip); PT_ATTN_V2=0 / PT_ATTN_BWD_V2=0 keep the round-2 kernels
  // (self-attention N = 1024, B H = 256, D = 64: 305 vs 370 us, profiles/r03_attn_probe.log)
  static const int v2 = pt_env_int("PT_ATTN_V2", 1) && pt_env_int("PT_ATTN_BWD_V2", 1);
  if (dtype == PT_BF16 && v2) return pt_attn2_bwd(p, (int)d->D, s);
#define BWD(TT) \
  switch (d->D) { case 32: return launch_bwd<TT, 32>(p, s); case 64: return launch_bwd<TT, 64>(p, s); \
                  case 128: return launch_bwd<TT, 128>(p, s); default: return PT_ERR_SHAPE; }
  if (dtype == PT_F32) { BWD(float) }
  BWD(bf16_t)
#undef BWD
}
