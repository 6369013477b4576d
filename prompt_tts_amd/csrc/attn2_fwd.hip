// Flash attention forward, second generation (bf16; attn2.h explains the design).  Replaces F.scaled_dot_product_attention at
// diffusers' AttnProcessor2_0 (reference call sites: tts/models.py:93-103, tts/ldm/transformer_1d.py:258-265).
// Workgroup = 4 waves x 32 query rows; K / V stream through a double-buffered LDS ring in 64-key tiles filled by LDS-DMA.
//   S'^T[key][q] = K Q'^T - m[q]      (32x32x16 MFMA; Q' = Q * scale * log2 e lives in registers as B fragments; the running
//                                      maximum m is the accumulator's initial value)
//   P = exp2(S')                      (one v_exp_f32 per score; row sums per lane; the maximum is refreshed lazily)
//   O^T[d][q] += V^T P^T              (P straight from the accumulator registers as the B operand, V by transposed LDS reads)
// MFMA-bound in the roofline sense: 4 Nq Nk D flops per (batch, head) against (Nq + 2 Nk) D 2 bytes.
#include "attn2.h"

namespace {

constexpr float A2_RESCALE_THR = 8.f;    // log2 units: probabilities stay below 2^8 between two refreshes of the maximum

template <int D>
__global__ __launch_bounds__(256, (D == 128 ? 1 : 3)) void attn2_fwd_kernel(const AttnParams p) {
  using C = A2<D>;
  constexpr int KS = C::KS, DT = C::DT, TILE = C::TILE;
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // [2 stages][K image | V image]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  int b, hd, blk;
  attn_block_ids((p.Nq + 127) / 128, p.H, blk, hd, b);
  const int qblk = blk * 128, q0 = qblk + wave * 32;
  int nk = p.Nk;
  if (p.kv_len) { nk = p.kv_len[b]; nk = nk < 1 ? 1 : (nk > p.Nk ? p.Nk : nk); }
  int klimit = nk;                                   // keys this workgroup has to visit
  if (p.causal) klimit = min(nk, qblk + 128);
  const int ntiles = (klimit + 63) / 64;

  const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + (int64_t)b * p.Nq * p.ldq + hd * D;
  const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + (int64_t)b * p.Nk * p.ldk + hd * D;
  const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + (int64_t)b * p.Nk * p.ldv + hd * D;

  A2Stage<D> stK, stV;
  stK.init(tid, p.ldk); stV.init(tid, p.ldv);
  stK.issue(K, 0, p.Nk, smem, wave, tid);
  stV.issue(V, 0, p.Nk, smem + TILE, wave, tid);

  // Q' fragments: B operand, column = query q0 + r, k = 16 ks + 8 h + j; pre-multiplied by scale * log2(e)
  const float sl2 = p.scale * PT_LOG2E;
  bf16x8_t qf[KS];
  {
    int row = q0 + r; row = row < p.Nq ? row : p.Nq - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8_t raw = *reinterpret_cast<const bf16x8_t*>(Q + (int64_t)row * p.ldq + 16 * ks + 8 * h);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[ks][j] = (__bf16)((float)raw[j] * sl2);
    }
  }
  A2Offsets<D> fo; fo.init(lane);

  f32x16_t o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = splat16(0.f);
  float m = 0.f, l = 0.f;                 // m: reference maximum of this lane's query (log2 units); l: this lane half's partial row sum

  const int qrow = q0 + r;
  a2_dma_wait();
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    const char* kimg = smem + cur * 2 * TILE;
    const char* vimg = kimg + TILE;

    // S'^T = K Q'^T - m  (the first tile starts from 0: its maximum becomes the reference)
    // (all fragment reads of the phase are issued before its first MFMA: the waits in front of the MFMAs are then COUNTED,
    // lgkmcnt(7), (6), ... -- left alone, hipcc reads two fragments, waits for both, multiplies, and exposes the LDS latency
    // once per MFMA)
    f32x16_t s[2];
    {
      bf16x8_t ka[2][KS];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) ka[kt][ks] = a2_read_rows<D>(kimg, fo.rowread[ks], 32 * kt);
      __builtin_amdgcn_sched_barrier(0);
      s[0] = splat16(-m);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s[0] = mma32(ka[0][ks], qf[ks], s[0]);
      // the next tile's LDS-DMA is issued HERE, in the shadow of the MFMAs just issued (a DMA piece costs the wave 60 - 180
      // cycles of issue: at the top of the loop nothing overlapped them)
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < ntiles) {
        stK.issue(K, (t + 1) * 64, p.Nk, smem + (cur ^ 1) * 2 * TILE, wave, tid);
        stV.issue(V, (t + 1) * 64, p.Nk, smem + (cur ^ 1) * 2 * TILE + TILE, wave, tid);
      }
      __builtin_amdgcn_sched_barrier(0);
      s[1] = splat16(-m);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s[1] = mma32(ka[1][ks], qf[ks], s[1]);
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
    // row maximum over this tile (32 values per lane, then the other lane half)
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
#ifdef A2_NO_MAX3
    for (int e = 1; e < 16; ++e) mx = fmaxf(mx, fmaxf(s[0][e], s[1][e]));
#else
    for (int e = 1; e < 16; ++e) mx = a2_max3(mx, s[0][e], s[1][e]);
#endif
    mx = a2_half_max(mx);
    // first tile: adopt the maximum; later: refresh only if some row outgrew the reference by 2^THR (wave-uniform branch)
    const bool refresh = t == 0 || __any(mx > A2_RESCALE_THR);
    if (refresh) {
      const float dlt = t == 0 ? mx : fmaxf(mx, 0.f);          // finite: every row has at least one visible key in tile 0
      const float alpha = t == 0 ? 1.f : __builtin_amdgcn_exp2f(-dlt);       // tile 0: nothing accumulated yet (and 2^-dlt may overflow)
      m += dlt; l *= alpha;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) o[dt] *= alpha;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) s[kt] -= dlt;
    }
    float rs = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) { const float pv = __builtin_amdgcn_exp2f(s[kt][e]); s[kt][e] = pv; rs += pv; }
    l += rs;
    // O^T += V^T P^T
    {
      bf16x8_t va[4][DT];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) va[kk][dt] = a2_read_tr<D>(vimg, fo.trread[0][dt], fo.trread[1][dt], 16 * kk);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8_t pf = a2_pack(s[kk >> 1], kk & 1);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = mma32(va[kk][dt], pf, o[dt]);

      }
    }
    a2_dma_wait();            // this wave's pieces of the next tile have landed ...
    __syncthreads();          // ... and so have everyone's; the tile just multiplied is free to be overwritten
  }

  // epilogue: lane holds O^T[d = 32 dt + (e & 3) + 8 (e >> 2) + 4 h][q = q0 + r]
  const float lt = a2_half_sum(l);
  const float inv = 1.f / lt;
  bf16_t* O = reinterpret_cast<bf16_t*>(p.o) + (int64_t)b * p.Nq * p.ldo + hd * D;
  if (qrow < p.Nq) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
        store4<bf16_t>(O + (int64_t)qrow * p.ldo + 32 * dt + 8 * rg + 4 * h, o[dt][4 * rg] * inv, o[dt][4 * rg + 1] * inv,
                       o[dt][4 * rg + 2] * inv, o[dt][4 * rg + 3] * inv);
    if (h == 0 && p.lse) p.lse[((int64_t)b * p.H + hd) * p.Nq + qrow] = (m + __builtin_amdgcn_logf(lt)) * 0.69314718055994530942f;
  }
}

template <int D> int launch_fwd2(const AttnParams& p, hipStream_t s) {
  const size_t lds = 2 * 2 * (size_t)A2<D>::TILE;
  static const int attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_fwd_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr != hipSuccess) return PT_ERR_LAUNCH;
  const int64_t nwg = (int64_t)((p.Nq + 127) / 128) * p.H * p.B;
  if (nwg >= (1ll << 31)) return PT_ERR_SHAPE;
  hipLaunchKernelGGL((attn2_fwd_kernel<D>), dim3((unsigned)nwg), dim3(256), lds, s, p);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

}  // namespace

// bf16 forward on the second-generation kernel; PT_ERR_SHAPE for head dims it does not cover (the caller falls back)
int pt_attn2_fwd(const AttnParams& p, int D, hipStream_t s) {
  switch (D) {
    case 32: return launch_fwd2<32>(p, s);
    case 64: return launch_fwd2<64>(p, s);
    case 128: return launch_fwd2<128>(p, s);
    default: return PT_ERR_SHAPE;
  }
}
