// Flash attention backward, second generation (bf16; attn2.h explains the shapes).  Same plan as attn_bwd.hip -- P is recomputed
// from Q, K and the forward's LSE; a dQ kernel (a wave owns 32 queries, sweeps 64-key tiles; it also forms delta = rowsum(dO o O))
// followed by a dK/dV kernel (a wave owns 32 keys, sweeps 64-query tiles); seven products, no atomics, bitwise reproducible --
// on 32x32x16 MFMA tiles with the row constants in the accumulators:
//   S' = (Q sl2) K^T - lse2     starts from -lse2 (sl2 = scale log2 e folded into the register-resident operand)   P = exp2(S')
//   dP' = dO V^T - delta        starts from -delta                                                                 dS = P dP'
// so a score costs one v_exp_f32, one v_mul_f32 and its share of two bf16 packs.  Q / dO (resp. K / V) tiles arrive by LDS-DMA
// in ONE image each that serves the row reads (S, dP) and the transposed reads (dV, dK, resp. dQ).
#include "attn2.h"

namespace {

// 16 row constants of a 32-row accumulator tile: rows (e & 3) + 8 (e >> 2) + 4h  ->  registers e = 0..15, as -src[row] * mul
__device__ __forceinline__ f32x16_t a2_row_consts(const float* src, int row0, int h, int limit, float mul, bool guard) {
  f32x16_t v;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = row0 + 8 * i + 4 * h;
    if (!guard) {
      const f32x4_t x = *reinterpret_cast<const f32x4_t*>(src + q);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * i + j] = -x[j] * mul;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int qq = q + j < limit ? q + j : limit - 1; v[4 * i + j] = -src[qq] * mul; }
    }
  }
  return v;
}

// the same from the LDS copy of a 64-row tile's slice (row0 = 0 or 32): four broadcast ds_read_b128
__device__ __forceinline__ f32x16_t a2_row_consts_lds(const char* lds, int row0, int h, float mul) {
  f32x16_t v;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4_t x = *reinterpret_cast<const f32x4_t*>(lds + (row0 + 8 * i + 4 * h) * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[4 * i + j] = -x[j] * mul;
  }
  return v;
}

// ---- dQ (and delta) ---------------------------------------------------------------------------------------------------
template <int D, int WPS, int NW>
__global__ __launch_bounds__(64 * NW, WPS) void attn2_bwd_q_kernel(const AttnParams p) {
  using C = A2<D>;
  constexpr int KS = C::KS, DT = C::DT, TILE = C::TILE;
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // [2 stages][K image | V image]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  int b, hd, blk;
  constexpr int QB = 32 * NW;                           // queries per workgroup
  attn_block_ids((p.Nq + QB - 1) / QB, p.H, blk, hd, b);
  const int qblk = blk * QB, q0 = qblk + wave * 32;
  int nk = p.Nk;
  if (p.kv_len) { nk = p.kv_len[b]; nk = nk < 1 ? 1 : (nk > p.Nk ? p.Nk : nk); }
  int klimit = nk;
  if (p.causal) klimit = min(nk, qblk + QB);
  const int ntiles = (klimit + 63) / 64;

  const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + (int64_t)b * p.Nq * p.ldq + hd * D;
  const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + (int64_t)b * p.Nk * p.ldk + hd * D;
  const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + (int64_t)b * p.Nk * p.ldv + hd * D;
  const bf16_t* DO = reinterpret_cast<const bf16_t*>(p.d_o) + (int64_t)b * p.Nq * p.lddo + hd * D;
  const bf16_t* O = reinterpret_cast<const bf16_t*>(p.o) + (int64_t)b * p.Nq * p.ldo + hd * D;
  const float* LSE = p.lse + ((int64_t)b * p.H + hd) * p.Nq;
  float* DELTA = p.delta + ((int64_t)b * p.H + hd) * p.Nq;

  A2Stage<D, 64 * NW> stK, stV;
  stK.init(tid, p.ldk); stV.init(tid, p.ldv);
  stK.issue(K, 0, p.Nk, smem, wave, tid);
  stV.issue(V, 0, p.Nk, smem + TILE, wave, tid);

  // register-resident B operands: column = query q0 + r, k = 16 ks + 8 h + j.  delta = rowsum(dO o O) is formed here from the dO
  // fragments this wave owns anyway (each lane half covers half of the columns) and published for the dK/dV kernel
  const float sl2 = p.scale * PT_LOG2E;
  const int qrow = q0 + r, qld = qrow < p.Nq ? qrow : p.Nq - 1;
  bf16x8_t qf[KS], dof[KS];
  float de = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const bf16x8_t raw = *reinterpret_cast<const bf16x8_t*>(Q + (int64_t)qld * p.ldq + 16 * ks + 8 * h);
    dof[ks] = *reinterpret_cast<const bf16x8_t*>(DO + (int64_t)qld * p.lddo + 16 * ks + 8 * h);
    const bf16x8_t of = *reinterpret_cast<const bf16x8_t*>(O + (int64_t)qld * p.ldo + 16 * ks + 8 * h);
#pragma unroll
    for (int j = 0; j < 8; ++j) { qf[ks][j] = (__bf16)((float)raw[j] * sl2); de += (float)dof[ks][j] * (float)of[j]; }
  }
  de = a2_half_sum(de);
  if (h == 0 && qrow < p.Nq) DELTA[qrow] = de;
  const float nlse2 = -LSE[qld] * PT_LOG2E, nde = -de;
  A2Offsets<D> fo; fo.init(lane);

  f32x16_t dq[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) dq[dt] = splat16(0.f);
  a2_dma_wait();
  __syncthreads();

  A2_STAMP(0);
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    A2_STAMP(8 * t + 1);
    const char* kimg = smem + cur * 2 * TILE;
    const char* vimg = kimg + TILE;

    f32x16_t s[2], dp[2];     // [key tile]: rows = keys, column = this lane's query
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      // all fragment reads of the key tile before its first MFMA: counted lgkmcnt waits instead of one full wait per MFMA
      bf16x8_t ka[KS], va[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) { ka[ks] = a2_read_rows<D>(kimg, fo.rowread[ks], 32 * kt); va[ks] = a2_read_rows<D>(vimg, fo.rowread[ks], 32 * kt); }
      __builtin_amdgcn_sched_barrier(0);
      s[kt] = splat16(nlse2); dp[kt] = splat16(nde);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) { s[kt] = mma32(ka[ks], qf[ks], s[kt]); dp[kt] = mma32(va[ks], dof[ks], dp[kt]); }
      if (kt == 0) {
        // the next tile's LDS-DMA goes out in the shadow of the MFMAs just issued (each piece costs the wave 60 - 180 cycles of
        // issue; at the top of the loop nothing overlapped them)
        __builtin_amdgcn_sched_barrier(0);
        A2_STAMP(8 * t + 2);
        if (t + 1 < ntiles) {
          stK.issue(K, (t + 1) * 64, p.Nk, smem + (cur ^ 1) * 2 * TILE, wave, tid);
          stV.issue(V, (t + 1) * 64, p.Nk, smem + (cur ^ 1) * 2 * TILE + TILE, wave, tid);
        }
        A2_STAMP(8 * t + 7);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const int key0 = t * 64;
    if (key0 + 64 > nk || p.causal) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = key0 + 32 * kt + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= nk || (p.causal && key > qrow)) s[kt][e] = -INFINITY;
        }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) dp[kt][e] *= __builtin_amdgcn_exp2f(s[kt][e]);       // dS = P (dP - delta)
#if A2_TRACE
    asm volatile("" ::"v"(dp[1][15]));      // the stamp below sits behind the last score's arithmetic
#endif
    A2_STAMP(8 * t + 3);
    // dQ^T[d][q] += K^T dS^T
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {
        const bf16x8_t dsf = a2_pack(dp[kt], sk);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          dq[dt] = mma32(a2_read_tr<D>(kimg, fo.trread[0][dt], fo.trread[1][dt], 32 * kt + 16 * sk), dsf, dq[dt]);
      }
    A2_STAMP(8 * t + 4);
    a2_dma_wait();
    A2_STAMP(8 * t + 5);
    __syncthreads();
    A2_STAMP(8 * t + 6);
  }

  bf16_t* DQ = reinterpret_cast<bf16_t*>(p.dq) + (int64_t)b * p.Nq * p.lddq + hd * D;
  if (qrow < p.Nq) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
        store4<bf16_t>(DQ + (int64_t)qrow * p.lddq + 32 * dt + 8 * rg + 4 * h, dq[dt][4 * rg] * p.scale, dq[dt][4 * rg + 1] * p.scale,
                       dq[dt][4 * rg + 2] * p.scale, dq[dt][4 * rg + 3] * p.scale);
  }
}

// ---- dK, dV -------------------------------------------------------------------------------------------------------------
template <int D, int WPS, int NW>
__global__ __launch_bounds__(64 * NW, WPS) void attn2_bwd_kv_kernel(const AttnParams p) {
  using C = A2<D>;
  constexpr int KS = C::KS, DT = C::DT, TILE = C::TILE;
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // [2 stages][Q image | dO image | LSE (64 f32) | delta (64 f32)]
  constexpr int STAGE = 2 * TILE + 512;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  int b, hd, blk;
  constexpr int KB = 32 * NW;                           // keys per workgroup
  attn_block_ids((p.Nk + KB - 1) / KB, p.H, blk, hd, b);
  const int kblk = blk * KB, k0 = kblk + wave * 32;
  int nk = p.Nk;
  if (p.kv_len) { nk = p.kv_len[b]; nk = nk < 1 ? 1 : (nk > p.Nk ? p.Nk : nk); }

  const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + (int64_t)b * p.Nq * p.ldq + hd * D;
  const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + (int64_t)b * p.Nk * p.ldk + hd * D;
  const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + (int64_t)b * p.Nk * p.ldv + hd * D;
  const bf16_t* DO = reinterpret_cast<const bf16_t*>(p.d_o) + (int64_t)b * p.Nq * p.lddo + hd * D;
  const float* LSE = p.lse + ((int64_t)b * p.H + hd) * p.Nq;
  const float* DELTA = p.delta + ((int64_t)b * p.H + hd) * p.Nq;

  // causal: queries before this workgroup's first key see none of its keys (tiles of 64 queries)
  const int qstart = p.causal ? (kblk / 64) * 64 : 0;
  const int nsteps = (p.Nq - qstart + 63) / 64;

  A2Stage<D, 64 * NW> stQ, stDO;
  stQ.init(tid, p.ldq); stDO.init(tid, p.lddo);
  // LSE / delta of a step's 64 queries travel with the tile (one 256-byte DMA piece each, waves 0 and 1): inside the loop there is
  // then NO compiler-visible vector-memory load -- a global load of the row constants was waited for with vmcnt(0), i.e. together
  // with the next tile's LDS-DMA issued just before it (in-order counter): 2 500 - 3 500 stalled cycles per 32-query block.
  auto issue_step = [&](int q0_, char* stage) {
    stQ.issue(Q, q0_, p.Nq, stage, wave, tid);
    stDO.issue(DO, q0_, p.Nq, stage + TILE, wave, tid);
    if (wave == 0) a2_dma_f32x64(LSE, q0_, p.Nq, stage + 2 * TILE, lane);
    if (wave == 1) a2_dma_f32x64(DELTA, q0_, p.Nq, stage + 2 * TILE + 256, lane);
  };
  if (nsteps > 0) issue_step(qstart, smem);

  // register-resident B operands: column = key k0 + r, k = 16 ks + 8 h + j; K pre-multiplied by scale * log2(e)
  const float sl2 = p.scale * PT_LOG2E;
  const int krow = k0 + r, kld = krow < p.Nk ? krow : p.Nk - 1;
  bf16x8_t kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const bf16x8_t raw = *reinterpret_cast<const bf16x8_t*>(K + (int64_t)kld * p.ldk + 16 * ks + 8 * h);
    vf[ks] = *reinterpret_cast<const bf16x8_t*>(V + (int64_t)kld * p.ldv + 16 * ks + 8 * h);
#pragma unroll
    for (int j = 0; j < 8; ++j) kf[ks][j] = (__bf16)((float)raw[j] * sl2);
  }
  A2Offsets<D> fo; fo.init(lane);

  f32x16_t dk[DT], dv[DT];      // dK^T / dV^T: rows = head dim, column = this lane's key
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { dk[dt] = splat16(0.f); dv[dt] = splat16(0.f); }
  a2_dma_wait();
  __syncthreads();

  const bool kmask = p.causal || (kblk + KB > nk);
  for (int st = 0; st < nsteps; ++st) {
    const int cur = st & 1, qs0 = qstart + st * 64;
    const char* qimg = smem + cur * STAGE;
    const char* doimg = qimg + TILE;
    const char* lseimg = qimg + 2 * TILE;
    A2_KV_STAMP(8 * st + 1);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int qs = qs0 + 32 * hf;                       // this block's first query; image rows 32 hf ..
      if (qs < p.Nq) {                                    // wave-uniform
        const bool need_mask = kmask || (qs + 32 > p.Nq);
        // rows = queries, column = this lane's key; the accumulators start from the row constants -lse2 / -delta
        f32x16_t s = a2_row_consts_lds(lseimg, 32 * hf, h, PT_LOG2E), dp = a2_row_consts_lds(lseimg + 256, 32 * hf, h, 1.f);
        {
          bf16x8_t qa[KS], da[KS];
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) { qa[ks] = a2_read_rows<D>(qimg, fo.rowread[ks], 32 * hf); da[ks] = a2_read_rows<D>(doimg, fo.rowread[ks], 32 * hf); }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) { s = mma32(qa[ks], kf[ks], s); dp = mma32(da[ks], vf[ks], dp); }
        }
        if (hf == 0) {          // next step's LDS-DMA in the shadow of the MFMAs just issued (block 0 of a step always runs)
          __builtin_amdgcn_sched_barrier(0);
          if (st + 1 < nsteps) issue_step(qs0 + 64, smem + (cur ^ 1) * STAGE);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (need_mask) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int q = qs + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (q >= p.Nq || krow >= nk || (p.causal && krow > q)) s[e] = -INFINITY;
          }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float pv = __builtin_amdgcn_exp2f(s[e]); s[e] = pv; dp[e] *= pv; }
#if A2_TRACE
        asm volatile("" ::"v"(dp[15]));
#endif
        A2_KV_STAMP(8 * st + 2 + 2 * hf);
        // dV^T[d][key] += dO^T P,  dK^T[d][key] += Q^T dS   (reduction over the block's 32 queries: two k-steps)
#pragma unroll
        for (int sk = 0; sk < 2; ++sk) {
          const bf16x8_t pf = a2_pack(s, sk), dsf = a2_pack(dp, sk);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            dv[dt] = mma32(a2_read_tr<D>(doimg, fo.trread[0][dt], fo.trread[1][dt], 32 * hf + 16 * sk), pf, dv[dt]);
            dk[dt] = mma32(a2_read_tr<D>(qimg, fo.trread[0][dt], fo.trread[1][dt], 32 * hf + 16 * sk), dsf, dk[dt]);
          }
        }
        A2_KV_STAMP(8 * st + 3 + 2 * hf);
      }
    }
    A2_KV_STAMP(8 * st + 6);
    a2_dma_wait();
    A2_KV_STAMP(8 * st + 7);
    __syncthreads();
    A2_KV_STAMP(8 * st + 8);
  }

  bf16_t* DK = reinterpret_cast<bf16_t*>(p.dk) + (int64_t)b * p.Nk * p.lddk + hd * D;
  bf16_t* DV = reinterpret_cast<bf16_t*>(p.dv) + (int64_t)b * p.Nk * p.lddv + hd * D;
  if (krow < p.Nk) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        store4<bf16_t>(DK + (int64_t)krow * p.lddk + 32 * dt + 8 * rg + 4 * h, dk[dt][4 * rg] * p.scale, dk[dt][4 * rg + 1] * p.scale,
                       dk[dt][4 * rg + 2] * p.scale, dk[dt][4 * rg + 3] * p.scale);
        store4<bf16_t>(DV + (int64_t)krow * p.lddv + 32 * dt + 8 * rg + 4 * h, dv[dt][4 * rg], dv[dt][4 * rg + 1], dv[dt][4 * rg + 2],
                       dv[dt][4 * rg + 3]);
      }
  }
}

// WPS = waves per SIMD the register allocation aims for (3 caps the kernels at 168 VGPRs: 8 - 33 spilled registers at D = 64;
// 2 leaves them 256).  NW = waves per workgroup: 8 waves share each K / V (Q / dO) tile, which halves the LDS-DMA pieces a wave
// issues per tile (a piece costs 60 - 180 cycles of issue) and the L2 -> LDS traffic.  PT_ATTN_BWD_WPS / PT_ATTN_BWD_NW pick.
template <int D, int WPS, int NW> int launch_bwd2(const AttnParams& p, hipStream_t s) {
  const size_t lds = 2 * 2 * (size_t)A2<D>::TILE, lds_kv = 2 * (2 * (size_t)A2<D>::TILE + 512);
  static const int a1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_bwd_q_kernel<D, WPS, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  static const int a2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_bwd_kv_kernel<D, WPS, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
  if (a1 != hipSuccess || a2 != hipSuccess) return PT_ERR_LAUNCH;
  constexpr int BLK = 32 * NW;
  const int64_t nq = (int64_t)((p.Nq + BLK - 1) / BLK) * p.H * p.B, nkv = (int64_t)((p.Nk + BLK - 1) / BLK) * p.H * p.B;
  if (nq >= (1ll << 31) || nkv >= (1ll << 31)) return PT_ERR_SHAPE;
  // dQ first: it also produces delta (rowsum(dO o O)) for the dK/dV kernel that follows on the same stream
  hipLaunchKernelGGL((attn2_bwd_q_kernel<D, WPS, NW>), dim3((unsigned)nq), dim3(64 * NW), lds, s, p);
  hipLaunchKernelGGL((attn2_bwd_kv_kernel<D, WPS, NW>), dim3((unsigned)nkv), dim3(64 * NW), lds_kv, s, p);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

}  // namespace

#if A2_TRACE
extern "C" int pt_debug_attn2_trace(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(a2_trace_buf), sizeof(unsigned long long) * (n < 512 ? n : 512)) == hipSuccess ? 0 : -3;
}
#endif

int pt_attn2_bwd(const AttnParams& p, int D, hipStream_t s) {
  static const int nw = pt_env_int("PT_ATTN_BWD_NW", 4);
  switch (D) {
    case 32: return nw == 8 ? launch_bwd2<32, 2, 8>(p, s) : launch_bwd2<32, 3, 4>(p, s);
    case 64: return nw == 8 ? launch_bwd2<64, 2, 8>(p, s) : launch_bwd2<64, 2, 4>(p, s);
    case 128: return launch_bwd2<128, 1, 4>(p, s);      // one wave per SIMD: 512 registers (at 256 the kernels spill 6 - 30 of them)
    default: return PT_ERR_SHAPE;
  }
}
