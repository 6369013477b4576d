// LayerNorm / GroupNorm(+SiLU) forward and backward on token-major activations.  HBM-bound: every kernel
// streams its tensors once with 16-byte loads/stores; a thread keeps a FIXED column chunk while it walks rows,
// so per-channel quantities (gamma/beta, scale/shift, dgamma/dbeta partials) live in registers.
//   algorithmic bytes: LN fwd 2*M*C*s; LN bwd 3*M*C*s (+dres); GN stats M*C*s; GN apply 2*M*C*s;
//   GN bwd 2*M*C*s (sums) + 3*M*C*s (apply)            (s = sizeof(T))
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int NT = 256;

// ---------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, C <= 64*MAXC*EPC.
// ---------------------------------------------------------------------------------------------------
template <typename T, int MAXC>
__global__ __launch_bounds__(NT) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, T* __restrict__ y,
                                                    float* __restrict__ mean, float* __restrict__ rstd,
                                                    int64_t M, int C, float eps) {
  constexpr int EPC = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= M) return;
  const T* xr = x + row * C;
  Vec16<T> v[MAXC];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    const int col = (lane + 64 * j) * EPC;
    if (col < C) {
      v[j] = load16(xr + col);
#pragma unroll
      for (int e = 0; e < EPC; ++e) s += v[j].get(e);
    }
  }
  const float mu = wave_sum(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    const int col = (lane + 64 * j) * EPC;
    if (col < C) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) { float d = v[j].get(e) - mu; ss = pt_ln_sq_acc(ss, d); }
    }
  }
  const float rs = pt_ln_rstd(wave_sum(ss), (float)C, eps);      // (explicit fusion choices: common.h, shared with pt_decode_linear)
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  T* yr = y + row * C;
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    const int col = (lane + 64 * j) * EPC;
    if (col < C) {
      Vec16<T> o;
#pragma unroll
      for (int e = 0; e < EPC; ++e) o.set(e, pt_ln_apply(v[j].get(e), mu, rs, gamma[col + e], beta[col + e]));
      store16(yr + col, o);
    }
  }
}

// One wave per row, RW rows per wave in flight (their loads are issued together: with a single row per wave the kernel ran
// at 2.4 TB/s, latency-bound).  dgamma / dbeta go to replica (block % n_rep) of the destination (see colsum_kernel).
template <typename T, int MAXC>
__global__ __launch_bounds__(NT) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ gamma, const T* __restrict__ dres,
                                                    T* __restrict__ dx, float* __restrict__ dgamma,
                                                    float* __restrict__ dbeta, int64_t M, int C, int n_rep, int64_t rep_stride) {
  constexpr int EPC = Vec16<T>::N;
  constexpr int RW = MAXC == 1 ? 4 : (MAXC == 2 ? 2 : 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t nwaves = (int64_t)gridDim.x * (NT / 64);
  float gam[MAXC][EPC], dg[MAXC][EPC], db[MAXC][EPC];
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    const int col = (lane + 64 * j) * EPC;
#pragma unroll
    for (int e = 0; e < EPC; ++e) { gam[j][e] = col < C ? gamma[col + e] : 0.f; dg[j][e] = 0.f; db[j][e] = 0.f; }
  }
  for (int64_t row0 = ((int64_t)blockIdx.x * (NT / 64) + wave) * RW; row0 < M; row0 += nwaves * RW) {
    Vec16<T> vx[RW][MAXC], vd[RW][MAXC], vr[RW][MAXC];
    float mu[RW], rs[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int64_t row = row0 + r < M ? row0 + r : M - 1;       // clamped: the duplicate row is computed, never stored
      mu[r] = mean[row]; rs[r] = rstd[row];
#pragma unroll
      for (int j = 0; j < MAXC; ++j) {
        const int col = (lane + 64 * j) * EPC;
        if (col < C) {
          vx[r][j] = load16(x + row * C + col);
          vd[r][j] = load16(dy + row * C + col);
          if (dres) vr[r][j] = load16(dres + row * C + col);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const bool live = row0 + r < M;
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int j = 0; j < MAXC; ++j) {
        const int col = (lane + 64 * j) * EPC;
        if (col < C) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            const float xh = (vx[r][j].get(e) - mu[r]) * rs[r], d = vd[r][j].get(e), dxh = d * gam[j][e];
            c1 += dxh; c2 += dxh * xh;
            if (live) { dg[j][e] += d * xh; db[j][e] += d; }
          }
        }
      }
      c1 = wave_sum(c1) / (float)C; c2 = wave_sum(c2) / (float)C;
      if (live) {
#pragma unroll
        for (int j = 0; j < MAXC; ++j) {
          const int col = (lane + 64 * j) * EPC;
          if (col < C) {
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
              const float xh = (vx[r][j].get(e) - mu[r]) * rs[r];
              float g = rs[r] * (vd[r][j].get(e) * gam[j][e] - c1 - xh * c2);
              if (dres) g += vr[r][j].get(e);
              o.set(e, g);
            }
            store16(dx + (row0 + r) * C + col, o);
          }
        }
      }
    }
  }
  // block-level reduce of the per-wave dgamma/dbeta partials through LDS, then one atomic per column per block
  __shared__ float sh[2][64 * MAXC * EPC];
  for (int w = 0; w < NT / 64; ++w) {
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < MAXC; ++j)
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const int idx = (lane + 64 * j) * EPC + e;
          if (w == 0) { sh[0][idx] = dg[j][e]; sh[1][idx] = db[j][e]; }
          else { sh[0][idx] += dg[j][e]; sh[1][idx] += db[j][e]; }
        }
    }
    __syncthreads();
  }
  const int64_t roff = (int64_t)(blockIdx.x % n_rep) * rep_stride;
  for (int c = threadIdx.x; c < C; c += NT) {
#ifndef PT_DIAG_NO_COLATOMICS
    unsafeAtomicAdd(dgamma + roff + c, sh[0][c]);
    unsafeAtomicAdd(dbeta + roff + c, sh[1][c]);
#endif
  }
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm on x = concat(x1[B][N][C1], x2[B][N][C2]) (token-major).  A block owns (batch b, a run of rows);
// thread (cw, rr) owns column chunks cw + j*CW and rows rr, rr+RP, ...
// ---------------------------------------------------------------------------------------------------
struct GnGeom { int C1, C2, C, CC, CW, RP, J, G, cpg, N, rows_per_block; float raw_cnt, eps; };

// per-(batch, group) statistics: either finalized (mean, rstd) or -- raw_cnt > 0 -- the raw (sum, sum of squares) the
// stats kernel accumulated, finalized here by every consumer (same arithmetic as gn_finalize_kernel; saves a launch)
__device__ __forceinline__ void gn_stat(const float* __restrict__ mean, const float* __restrict__ rstd, int idx, const GnGeom& g,
                                        float& mu, float& rs) {
  mu = mean[idx]; rs = rstd[idx];
  if (g.raw_cnt > 0.f) {
    mu = mu / g.raw_cnt;
    const float var = fmaxf(rs / g.raw_cnt - mu * mu, 0.f);
    rs = rsqrtf(var + g.eps);
  }
}

template <typename T> __device__ __forceinline__ const T* gn_src(const T* x1, const T* x2, const GnGeom& g, int64_t row, int col) {
  return col < g.C1 ? x1 + row * g.C1 + col : x2 + row * g.C2 + (col - g.C1);
}

template <typename T>
__global__ __launch_bounds__(NT) void gn_stats_kernel(const T* __restrict__ x1, const T* __restrict__ x2,
                                                      float* __restrict__ sum, float* __restrict__ sumsq, GnGeom g) {
  constexpr int EPC = Vec16<T>::N;
  const int b = blockIdx.y, r0 = blockIdx.x * g.rows_per_block, r1 = min(g.N, r0 + g.rows_per_block);
  const int cw = threadIdx.x % g.CW, rr = threadIdx.x / g.CW;
  float s[2][EPC], ss[2][EPC];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < EPC; ++e) { s[j][e] = 0.f; ss[j][e] = 0.f; }
  // one branch-free streaming loop per owned column chunk, unrolled so several 16-byte loads are in flight per thread
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = cw + j * g.CW;
    if (rr < g.RP && j < g.J && c < g.CC) {
      const int col = c * EPC;
      const T* src = col < g.C1 ? x1 + col : x2 + (col - g.C1);
      const int64_t ldx = col < g.C1 ? g.C1 : g.C2;
#pragma unroll 8
      for (int r = r0 + rr; r < r1; r += g.RP) {
        Vec16<T> v = load16(src + ((int64_t)b * g.N + r) * ldx);
#pragma unroll
        for (int e = 0; e < EPC; ++e) { float f = v.get(e); s[j][e] += f; ss[j][e] += f * f; }
      }
    }
  }
  __shared__ float sh[2 * 256];
  for (int i = threadIdx.x; i < 2 * g.G; i += NT) sh[i] = 0.f;
  __syncthreads();
  if (rr < g.RP) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = cw + j * g.CW;
      if (j < g.J && c < g.CC) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const int grp = (c * EPC + e) / g.cpg;
          atomicAdd(&sh[2 * grp], s[j][e]);
          atomicAdd(&sh[2 * grp + 1], ss[j][e]);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < g.G; i += NT) {
    unsafeAtomicAdd(sum + b * g.G + i, sh[2 * i]);
    unsafeAtomicAdd(sumsq + b * g.G + i, sh[2 * i + 1]);
  }
}

__global__ void gn_finalize_kernel(float* mean, float* rstd, int n, float cnt, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float mu = mean[i] / cnt;
  const float var = fmaxf(rstd[i] / cnt - mu * mu, 0.f);
  mean[i] = mu;
  rstd[i] = rsqrtf(var + eps);
}

template <typename T>
__global__ __launch_bounds__(NT) void gn_apply_kernel(const T* __restrict__ x1, const T* __restrict__ x2,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      T* __restrict__ y, T* __restrict__ xcat, GnGeom g, int silu) {
  constexpr int EPC = Vec16<T>::N;
  const int b = blockIdx.y, r0 = blockIdx.x * g.rows_per_block, r1 = min(g.N, r0 + g.rows_per_block);
  const int cw = threadIdx.x % g.CW, rr = threadIdx.x / g.CW;
  if (rr >= g.RP) return;
  float sc[2][EPC], sf[2][EPC];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = cw + j * g.CW;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      sc[j][e] = 0.f; sf[j][e] = 0.f;
      if (j < g.J && c < g.CC) {
        const int col = c * EPC + e, grp = col / g.cpg;
        float mu, rs; gn_stat(mean, rstd, b * g.G + grp, g, mu, rs);
        sc[j][e] = rs * gamma[col];
        sf[j][e] = beta[col] - mu * rs * gamma[col];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = cw + j * g.CW;
    if (j < g.J && c < g.CC) {
      const int col = c * EPC;
      const T* src = col < g.C1 ? x1 + col : x2 + (col - g.C1);
      const int64_t ldx = col < g.C1 ? g.C1 : g.C2;
#pragma unroll 8
      for (int r = r0 + rr; r < r1; r += g.RP) {
        const int64_t row = (int64_t)b * g.N + r;
        Vec16<T> v = load16(src + row * ldx), o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float z = v.get(e) * sc[j][e] + sf[j][e];
          o.set(e, silu ? silu_f(z) : z);
        }
        store16(y + row * g.C + col, o);
        if (xcat) store16(xcat + row * g.C + col, v);
      }
    }
  }
}

// backward pass 1: per-column sums of dz and dz*xhat over this block's rows -> dgamma/dbeta (global, all b)
// and per-(b,group) A = sum gamma*dz, Bq = sum gamma*dz*xhat into ws[b][G][2].
template <typename T>
__global__ __launch_bounds__(NT) void gn_bwd_sums_kernel(const T* __restrict__ dy, const T* __restrict__ x1,
                                                         const T* __restrict__ x2, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, float* __restrict__ ws, GnGeom g,
                                                         int silu, int n_rep, int64_t rep_stride) {
  constexpr int EPC = Vec16<T>::N;
  const int b = blockIdx.y, r0 = blockIdx.x * g.rows_per_block, r1 = min(g.N, r0 + g.rows_per_block);
  const int cw = threadIdx.x % g.CW, rr = threadIdx.x / g.CW;
  float mu[2][EPC], rs[2][EPC], ga[2][EPC], be[2][EPC], sg[2][EPC], sb[2][EPC];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = cw + j * g.CW;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      mu[j][e] = 0.f; rs[j][e] = 0.f; ga[j][e] = 0.f; be[j][e] = 0.f; sg[j][e] = 0.f; sb[j][e] = 0.f;
      if (j < g.J && c < g.CC) {
        const int col = c * EPC + e, grp = col / g.cpg;
        gn_stat(mean, rstd, b * g.G + grp, g, mu[j][e], rs[j][e]);
        ga[j][e] = gamma[col]; be[j][e] = beta[col];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = cw + j * g.CW;
    if (rr < g.RP && j < g.J && c < g.CC) {
      const int col = c * EPC;
      const T* src = col < g.C1 ? x1 + col : x2 + (col - g.C1);
      const int64_t ldx = col < g.C1 ? g.C1 : g.C2;
#pragma unroll 4
      for (int r = r0 + rr; r < r1; r += g.RP) {
        const int64_t row = (int64_t)b * g.N + r;
        Vec16<T> vx = load16(src + row * ldx);
        Vec16<T> vd = load16(dy + row * g.C + col);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const float xh = (vx.get(e) - mu[j][e]) * rs[j][e];
          float dz = vd.get(e);
          if (silu) dz *= silu_grad_f(xh * ga[j][e] + be[j][e]);
          sg[j][e] += dz * xh; sb[j][e] += dz;
        }
      }
    }
  }
  // combine the RP row-lanes of each column through LDS
  extern __shared__ float dyn[];                 // [2][C] column sums, then [2][G] group sums
  float* colg = dyn; float* colb = dyn + g.C; float* grp2 = dyn + 2 * g.C;
  for (int i = threadIdx.x; i < 2 * g.C + 2 * g.G; i += NT) dyn[i] = 0.f;
  __syncthreads();
  if (rr < g.RP) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = cw + j * g.CW;
      if (j < g.J && c < g.CC) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const int col = c * EPC + e;
          atomicAdd(&colg[col], sg[j][e]);
          atomicAdd(&colb[col], sb[j][e]);
        }
      }
    }
  }
  __syncthreads();
  const int64_t roff = (int64_t)((blockIdx.x + gridDim.x * blockIdx.y) % n_rep) * rep_stride;
  for (int col = threadIdx.x; col < g.C; col += NT) {
    const float a = colg[col], bb = colb[col], gm = gamma[col];
#ifndef PT_DIAG_NO_COLATOMICS
    unsafeAtomicAdd(dgamma + roff + col, a);
    unsafeAtomicAdd(dbeta + roff + col, bb);
#endif
    atomicAdd(&grp2[2 * (col / g.cpg)], gm * bb);        // A  = sum gamma*dz
    atomicAdd(&grp2[2 * (col / g.cpg) + 1], gm * a);     // Bq = sum gamma*dz*xhat
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * g.G; i += NT) unsafeAtomicAdd(ws + (int64_t)b * 2 * g.G + i, grp2[i]);
}

template <typename T>
__global__ __launch_bounds__(NT) void gn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x1,
                                                          const T* __restrict__ x2, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ ws,
                                                          const T* __restrict__ dres, T* __restrict__ dx1,
                                                          T* __restrict__ dx2, GnGeom g, int silu, int acc_dx2) {
  constexpr int EPC = Vec16<T>::N;
  const int b = blockIdx.y, r0 = blockIdx.x * g.rows_per_block, r1 = min(g.N, r0 + g.rows_per_block);
  const int cw = threadIdx.x % g.CW, rr = threadIdx.x / g.CW;
  if (rr >= g.RP) return;
  const float inv_cnt = 1.f / ((float)g.N * (float)g.cpg);
  float mu[2][EPC], rs[2][EPC], ga[2][EPC], be[2][EPC], mA[2][EPC], mB[2][EPC];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = cw + j * g.CW;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      mu[j][e] = rs[j][e] = ga[j][e] = be[j][e] = mA[j][e] = mB[j][e] = 0.f;
      if (j < g.J && c < g.CC) {
        const int col = c * EPC + e, grp = col / g.cpg;
        gn_stat(mean, rstd, b * g.G + grp, g, mu[j][e], rs[j][e]);
        ga[j][e] = gamma[col]; be[j][e] = beta[col];
        mA[j][e] = ws[((int64_t)b * g.G + grp) * 2] * inv_cnt;
        mB[j][e] = ws[((int64_t)b * g.G + grp) * 2 + 1] * inv_cnt;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = cw + j * g.CW;
    if (j < g.J && c < g.CC) {
      const int col0 = c * EPC;
      const bool first = col0 < g.C1;
      const T* src = first ? x1 + col0 : x2 + (col0 - g.C1);
      T* dbase = first ? dx1 + col0 : dx2 + (col0 - g.C1);
      const int64_t ldx = first ? g.C1 : g.C2;
      const bool accum = acc_dx2 && !first;
#pragma unroll 4
      for (int r = r0 + rr; r < r1; r += g.RP) {
        const int64_t row = (int64_t)b * g.N + r;
        Vec16<T> vx = load16(src + row * ldx);
        Vec16<T> vd = load16(dy + row * g.C + col0);
        Vec16<T> vr, o, old;
        if (dres) vr = load16(dres + row * g.C + col0);
        T* dst = dbase + row * ldx;
        if (accum) old = load16(dst);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const float xh = (vx.get(e) - mu[j][e]) * rs[j][e];
          float dz = vd.get(e);
          if (silu) dz *= silu_grad_f(xh * ga[j][e] + be[j][e]);
          float gr = rs[j][e] * (dz * ga[j][e] - mA[j][e] - xh * mB[j][e]);
          if (dres) gr += vr.get(e);
          if (accum) gr += old.get(e);
          o.set(e, gr);
        }
        store16(dst, o);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm "slab" kernels (bf16, N <= 1024 tokens per item; N <= 2048 for the 32-channel slab): ONE launch per direction.  A workgroup of 64 CCH threads owns a
// slab of 8 CCH channels (whole groups; CCH = 4: 32 channels, 256 threads) of one batch item and keeps it in REGISTERS (x:
// 16 B chunks, NCH per thread; backward: x and dy), so statistics / group sums and the normalisation are one pass: each
// activation byte is read once (the two-pass kernels read x twice forward, x and dy twice backward) and all of a thread's
// loads are in flight together.  The grid is (C / 8 CCH) x B workgroups at EVERY UNet level, where the row-block grids of
// the two-pass kernels shrink with N.  Thread t holds column chunk t % CCH of rows t / CCH + 64 i.
// 32-channel slabs read half cache lines: the slab index is decoded XCD-aware so that the two slabs of a line run on the
// same XCD (second toucher hits L2).  (A 512-thread / 64-channel version measured 3x slower INSIDE the training step than
// alone: a 200-register 8-wave workgroup has to wait for a whole CU while the wgrad GEMMs hold half of each.)
// ---------------------------------------------------------------------------------------------------
struct GnSlab {
  const bf16_t* x1; const bf16_t* x2; int C1, C2, C, N, G, cpg;
  const float* gamma; const float* beta; float* mean; float* rstd; float eps; int silu;
  bf16_t* y;                                                       // forward
  const bf16_t* dy; const bf16_t* dres; bf16_t* dx1; bf16_t* dx2;     // backward
  float* dgamma; float* dbeta; int acc_dx2, n_rep; int64_t rep_stride; float raw_cnt;
  float* item_sum; int64_t item_ld;                                 // backward: item_sum[b][c] += sum_n dx[(b, n)][c] (or NULL)
};

// sum over the lanes of this wave that hold the same column chunk (rows differ in the lane bits above log2(CCH))
template <int CCH> __device__ __forceinline__ float rows_sum(float v) {
  if (CCH <= 4) v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ int slab_index() {       // blocks b, b+8 share an XCD: give each XCD a run of adjacent slabs
  const int n = gridDim.x, x = blockIdx.x;
  return (n & 7) == 0 ? (x & 7) * (n >> 3) + (x >> 3) : x;
}

template <int NCH, int CCH>
__global__ __launch_bounds__(64 * CCH) void gn_slab_fwd_kernel(const GnSlab p) {
  using V = Vec16<bf16_t>;
  constexpr int NW = CCH, RS = 64;                                // waves; row stride between a thread's chunks
  __shared__ float red[NW][CCH][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cch = tid % CCH, r0 = tid / CCH;
  const int b = blockIdx.y, col0 = slab_index() * 8 * CCH, col = col0 + 8 * cch;
  const bool first = col0 < p.C1;
  const bf16_t* src = first ? p.x1 + col : p.x2 + (col - p.C1);
  const int64_t ld = first ? p.C1 : p.C2;
  V vx[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int r = r0 + RS * i; r = r < p.N ? r : p.N - 1;
    vx[i] = load16(src + ((int64_t)b * p.N + r) * ld);
  }
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (r0 + RS * i < p.N) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float f = vx[i].get(e); s += f; ss += f * f; }
    }
  s = rows_sum<CCH>(s); ss = rows_sum<CCH>(ss);
  if (lane < CCH) { red[wave][cch][0] = s; red[wave][cch][1] = ss; }
  __syncthreads();
  const int cpgc = p.cpg >> 3, c_lo = cch / cpgc * cpgc;          // chunks of this thread's group inside the slab
  float S = 0.f, SS = 0.f;
  for (int w = 0; w < NW; ++w)
    for (int c = c_lo; c < c_lo + cpgc; ++c) { S += red[w][c][0]; SS += red[w][c][1]; }
  const float cnt = (float)p.N * (float)p.cpg;
  const float mu = S / cnt, rs = rsqrtf(fmaxf(SS / cnt - mu * mu, 0.f) + p.eps);
  if (r0 == 0 && cch == c_lo) { p.mean[b * p.G + col / p.cpg] = mu; p.rstd[b * p.G + col / p.cpg] = rs; }
  float sc[8], sf[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = rs * p.gamma[col + e]; sf[e] = p.beta[col + e] - mu * sc[e]; }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int r = r0 + RS * i;
    if (r < p.N) {
      V o;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float z = vx[i].get(e) * sc[e] + sf[e]; o.set(e, p.silu ? silu_f(z) : z); }
      store16(p.y + ((int64_t)b * p.N + r) * p.C + col, o);
    }
  }
}

template <int NCH, int CCH>
__global__ __launch_bounds__(64 * CCH) void gn_slab_bwd_kernel(const GnSlab p) {
  using V = Vec16<bf16_t>;
  constexpr int NW = CCH, RS = 64, SC = 8 * CCH;                  // waves; row stride; slab channels
  __shared__ float red[NW][SC][2];           // per-wave column sums (dz*xhat, dz)
  __shared__ float tot[2][SC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cch = tid % CCH, r0 = tid / CCH;
  const int b = blockIdx.y, col0 = slab_index() * SC, col = col0 + 8 * cch;
  const bool first = col0 < p.C1;
  const bf16_t* src = first ? p.x1 + col : p.x2 + (col - p.C1);
  const int64_t ld = first ? p.C1 : p.C2;
  V vx[NCH], vd[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int r = r0 + RS * i; r = r < p.N ? r : p.N - 1;
    const int64_t row = (int64_t)b * p.N + r;
    vx[i] = load16(src + row * ld);
    vd[i] = load16(p.dy + row * p.C + col);
  }
  float mu = p.mean[b * p.G + col / p.cpg], rs = p.rstd[b * p.G + col / p.cpg];
  const float cnt = (float)p.N * (float)p.cpg;
  if (p.raw_cnt > 0.f) { mu = mu / cnt; rs = rsqrtf(fmaxf(rs / cnt - mu * mu, 0.f) + p.eps); }
  float ga[8], be[8], dg[8], db[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { ga[e] = p.gamma[col + e]; be[e] = p.beta[col + e]; dg[e] = 0.f; db[e] = 0.f; }
  // pass 1 also REPLACES the register copies of x / dy by xhat / dz (bf16, the precision the result is stored in anyway): the
  // apply pass then needs no second sigmoid -- the kernel is as VALU-heavy as it is HBM-heavy (load / math / store do not overlap
  // inside a workgroup), so the exp + divide per element it saves is ~20 % of its time
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (r0 + RS * i < p.N) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (vx[i].get(e) - mu) * rs;
        float dz = vd[i].get(e);
        if (p.silu) dz *= silu_grad_f(xh * ga[e] + be[e]);
        vx[i].set(e, xh); vd[i].set(e, dz);
        dg[e] += dz * xh; db[e] += dz;
      }
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) { dg[e] = rows_sum<CCH>(dg[e]); db[e] = rows_sum<CCH>(db[e]); }
  if (lane < CCH) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[wave][8 * cch + e][0] = dg[e]; red[wave][8 * cch + e][1] = db[e]; }
  }
  __syncthreads();
  if (tid < 2 * SC) {
    const int c = tid % SC, k = tid / SC;
    float a = 0.f;
    for (int w = 0; w < NW; ++w) a += red[w][c][k];
    tot[k][c] = a;
    float* dst = (k == 0 ? p.dgamma : p.dbeta) + (int64_t)((blockIdx.x + gridDim.x * blockIdx.y) % p.n_rep) * p.rep_stride;
    unsafeAtomicAdd(dst + col0 + c, a);
  }
  __syncthreads();
  const int g_lo = (8 * cch) / p.cpg * p.cpg;                      // first slab column of this thread's group
  float A = 0.f, Bq = 0.f;
  for (int c = g_lo; c < g_lo + p.cpg; ++c) { const float gm = p.gamma[col0 + c]; A += gm * tot[1][c]; Bq += gm * tot[0][c]; }
  A /= cnt; Bq /= cnt;
  bf16_t* dbase = first ? p.dx1 + col : p.dx2 + (col - p.C1);
  const bool accum = p.acc_dx2 && !first;
  float cs[8];                                   // per-item column sums of dx (the time-embedding gradient of a resnet)
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = 0.f;
  // The residual gradient / the accumulated destination are fetched in groups of GRP rows AHEAD of the stores: gfx950 counts
  // loads and stores in one counter (vmcnt), so a load issued behind a store is only complete when that store is; fetched row by
  // row inside the store loop every row waited for the previous row's write round trip.
  constexpr int GRP = NCH < 4 ? NCH : 4;
#pragma unroll
  for (int i0 = 0; i0 < NCH; i0 += GRP) {
    V vr[GRP], old[GRP];
#pragma unroll
    for (int k = 0; k < GRP; ++k) {
      const int r = r0 + RS * (i0 + k);
      if (r < p.N) {
        const int64_t row = (int64_t)b * p.N + r;
        if (p.dres) vr[k] = load16(p.dres + row * p.C + col);
        if (accum) old[k] = load16(dbase + row * ld);
      }
    }
#pragma unroll
    for (int k = 0; k < GRP; ++k) {
      const int i = i0 + k, r = r0 + RS * i;
      if (r < p.N) {
        const int64_t row = (int64_t)b * p.N + r;
        V o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = vx[i].get(e), dz = vd[i].get(e);
          float gr = rs * (dz * ga[e] - A - xh * Bq);
          if (p.dres) gr += vr[k].get(e);
          if (accum) gr += old[k].get(e);
          o.set(e, gr);
          cs[e] += gr;
        }
        store16(dbase + row * ld, o);
      }
    }
  }
  if (p.item_sum) {                              // this workgroup is the only producer of its item's 8 CCH columns
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = rows_sum<CCH>(cs[e]);
    __syncthreads();                             // `red` is reused
    if (lane < CCH) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wave][8 * cch + e][0] = cs[e];
    }
    __syncthreads();
    if (tid < SC) {
      float a = 0.f;
      for (int w = 0; w < NW; ++w) a += red[w][tid][0];
      unsafeAtomicAdd(p.item_sum + (int64_t)b * p.item_ld + col0 + tid, a);
    }
  }
}

// slab width in 16-byte chunks (4: 32 channels / 256 threads, 8: 64 channels / 512 threads), or 0: use the two-pass kernels
inline int gn_slab_cch(int64_t N, int64_t C1, int64_t C2, int64_t G) {
  static const int enabled = pt_env_int("PT_GN_SLAB", 1);
  const int64_t C = C1 + C2;
  if (!enabled || G <= 0 || C % G != 0 || N < 1 || N > 2048) return 0;
  const int64_t cpg = C / G;
  // N <= 2048 (config E's 2048-token items): 32 chunks per thread, for the 256-thread slab only -- one wave per SIMD, so the
  // backward's 256 registers of x / dy plus its temporaries fit the 512 (arch + acc) registers a lone wave may use
  if ((cpg == 8 || cpg == 16 || cpg == 32) && C1 % 32 == 0 && C2 % 32 == 0) return 4;
  if (cpg == 64 && C1 % 64 == 0 && C2 % 64 == 0 && N <= 1024) return 8;
  return 0;
}
#define GN_SLAB_LAUNCH(KERNEL, CCH, N, B, s, p)                                                              \
  do {                                                                                                       \
    dim3 grid((unsigned)((p).C / (8 * CCH)), (unsigned)(B)), blk(64 * CCH);                                  \
    if ((N) <= 64) hipLaunchKernelGGL((KERNEL<1, CCH>), grid, blk, 0, s, p);                                 \
    else if ((N) <= 128) hipLaunchKernelGGL((KERNEL<2, CCH>), grid, blk, 0, s, p);                           \
    else if ((N) <= 256) hipLaunchKernelGGL((KERNEL<4, CCH>), grid, blk, 0, s, p);                           \
    else if ((N) <= 512) hipLaunchKernelGGL((KERNEL<8, CCH>), grid, blk, 0, s, p);                           \
    else if ((N) <= 1024 || CCH != 4) hipLaunchKernelGGL((KERNEL<16, CCH>), grid, blk, 0, s, p);             \
    else hipLaunchKernelGGL((KERNEL<(CCH == 4 ? 32 : 16), CCH>), grid, blk, 0, s, p);                        \
  } while (0)

template <typename T> int gn_geom(GnGeom& g, int64_t N, int64_t C1, int64_t C2, int64_t G) {
  constexpr int EPC = Vec16<T>::N;
  const int64_t C = C1 + C2;
  if (C <= 0 || G <= 0 || G > 256 || C % G != 0 || C1 % EPC != 0 || C2 % EPC != 0 || N <= 0) return PT_ERR_SHAPE;
  g.C1 = (int)C1; g.C2 = (int)C2; g.C = (int)C; g.G = (int)G; g.cpg = (int)(C / G); g.N = (int)N;
  g.CC = g.C / EPC;
  g.CW = g.CC < NT ? g.CC : NT;
  g.RP = NT / g.CW;
  g.J = (g.CC + g.CW - 1) / g.CW;
  if (g.J > 2) return PT_ERR_SHAPE;
  g.rows_per_block = 64;
  g.raw_cnt = 0.f; g.eps = 0.f;
  return PT_OK;
}

template <typename T>
int ln_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t M,
           int64_t C, float eps, hipStream_t s) {
  constexpr int EPC = Vec16<T>::N;
  const int chunks = (int)((C / EPC + 63) / 64);
  dim3 grid((unsigned)((M + 3) / 4));
#define LN_F(MC) hipLaunchKernelGGL((ln_fwd_kernel<T, MC>), grid, dim3(NT), 0, s, (const T*)x, gamma, beta, (T*)y, mean, rstd, M, (int)C, eps)
  if (chunks <= 1) LN_F(1); else if (chunks <= 2) LN_F(2); else if (chunks <= 4) LN_F(4); else if (chunks <= 8) LN_F(8); else return PT_ERR_SHAPE;
#undef LN_F
  PT_LAUNCH_CHECK();
  return PT_OK;
}

template <typename T>
int ln_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const void* dres,
           void* dx, float* dgamma, float* dbeta, int64_t M, int64_t C, int n_rep, int64_t rep_stride, hipStream_t s) {
  constexpr int EPC = Vec16<T>::N;
  const int chunks = (int)((C / EPC + 63) / 64);
  const int rw = chunks <= 1 ? 4 : (chunks <= 2 ? 2 : 1);
  const int64_t want = (M + 4 * rw - 1) / (4 * rw);
  // <= 512 workgroups (16 rows per wave at M = 32768: amortises the dgamma / dbeta atomics; against 1024 workgroups: 27.2 -> 26.6 us
  // there, 18.4 -> 16.2 us at M = 16384, 25.9 -> 19.2 us at M = 8192, C = 1024 -- tools/norm_probe.py)
  static const int cap = pt_env_int("PT_LN_BWD_GRID", 512);
  dim3 grid((unsigned)(want < cap ? want : cap));
#define LN_B(MC) hipLaunchKernelGGL((ln_bwd_kernel<T, MC>), grid, dim3(NT), 0, s, (const T*)dy, (const T*)x, mean, rstd, gamma, (const T*)dres, (T*)dx, dgamma, dbeta, M, (int)C, n_rep, rep_stride)
  if (chunks <= 1) LN_B(1); else if (chunks <= 2) LN_B(2); else if (chunks <= 4) LN_B(4); else return PT_ERR_SHAPE;
#undef LN_B
  PT_LAUNCH_CHECK();
  return PT_OK;
}

}  // namespace

extern "C" int pt_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                int64_t M, int64_t C, float eps, int dtype, pt_stream stream) {
  if (M <= 0 || C <= 0 || C % 8 != 0) return PT_ERR_SHAPE;
  if (!pt_aligned16(x) || !pt_aligned16(y)) return PT_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == PT_F32) return ln_fwd<float>(x, gamma, beta, y, mean, rstd, M, C, eps, s);
  if (dtype == PT_BF16) return ln_fwd<bf16_t>(x, gamma, beta, y, mean, rstd, M, C, eps, s);
  return PT_ERR_DTYPE;
}

extern "C" int pt_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                const void* dres, void* dx, float* dgamma, float* dbeta, int64_t M, int64_t C, int n_rep,
                                int64_t rep_stride, int dtype, pt_stream stream) {
  if (M <= 0 || C <= 0 || C % 8 != 0) return PT_ERR_SHAPE;
  if (n_rep < 1 || (n_rep > 1 && rep_stride <= 0)) return PT_ERR_ARG;
  if (!pt_aligned16(x) || !pt_aligned16(dy) || !pt_aligned16(dx) || (dres && !pt_aligned16(dres))) return PT_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == PT_F32) return ln_bwd<float>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, M, C, n_rep, rep_stride, s);
  if (dtype == PT_BF16) return ln_bwd<bf16_t>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, M, C, n_rep, rep_stride, s);
  return PT_ERR_DTYPE;
}

template <typename T>
static int gn_stats_t(const void* x1, const void* x2, float* mean, float* rstd, int64_t B, int64_t N, int64_t C1,
                      int64_t C2, int64_t G, float eps, hipStream_t s) {
  GnGeom g; int st = gn_geom<T>(g, N, C1, C2, G); if (st) return st;
  const bool raw = eps < 0.f;               // raw mode: accumulate (sum, sumsq) into caller-zeroed arrays, no finalize
  if (!raw) {
    if (hipMemsetAsync(mean, 0, sizeof(float) * B * G, s) != hipSuccess) return PT_ERR_LAUNCH;
    if (hipMemsetAsync(rstd, 0, sizeof(float) * B * G, s) != hipSuccess) return PT_ERR_LAUNCH;
  }
  dim3 grid((unsigned)((N + g.rows_per_block - 1) / g.rows_per_block), (unsigned)B);
  hipLaunchKernelGGL((gn_stats_kernel<T>), grid, dim3(NT), 0, s, (const T*)x1, (const T*)x2, mean, rstd, g);
  if (!raw) {
    const int n = (int)(B * G);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, s, mean, rstd, n, (float)N * (float)g.cpg, eps);
  }
  PT_LAUNCH_CHECK();
  return PT_OK;
}

extern "C" int pt_groupnorm_stats(const void* x1, const void* x2, float* mean, float* rstd, int64_t B, int64_t N,
                                  int64_t C1, int64_t C2, int64_t G, float eps, int dtype, pt_stream stream) {
  if (B <= 0 || (C2 > 0 && !x2)) return PT_ERR_SHAPE;
  if (!pt_aligned16(x1) || (x2 && !pt_aligned16(x2))) return PT_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == PT_F32) return gn_stats_t<float>(x1, x2, mean, rstd, B, N, C1, C2, G, eps, s);
  if (dtype == PT_BF16) return gn_stats_t<bf16_t>(x1, x2, mean, rstd, B, N, C1, C2, G, eps, s);
  return PT_ERR_DTYPE;
}

template <typename T>
static int gn_apply_t(const void* x1, const void* x2, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, void* y, void* xcat, int64_t B, int64_t N, int64_t C1, int64_t C2, int64_t G,
                      int silu, float raw_eps, hipStream_t s) {
  GnGeom g; int st = gn_geom<T>(g, N, C1, C2, G); if (st) return st;
  if (raw_eps >= 0.f) { g.raw_cnt = (float)N * (float)g.cpg; g.eps = raw_eps; }
  dim3 grid((unsigned)((N + g.rows_per_block - 1) / g.rows_per_block), (unsigned)B);
  hipLaunchKernelGGL((gn_apply_kernel<T>), grid, dim3(NT), 0, s, (const T*)x1, (const T*)x2, mean, rstd, gamma, beta,
                     (T*)y, (T*)xcat, g, silu);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

extern "C" int pt_groupnorm_apply(const void* x1, const void* x2, const float* mean, const float* rstd,
                                  const float* gamma, const float* beta, void* y, void* xcat, int64_t B, int64_t N,
                                  int64_t C1, int64_t C2, int64_t G, int silu, float raw_eps, int dtype, pt_stream stream) {
  if (B <= 0 || (C2 > 0 && !x2)) return PT_ERR_SHAPE;
  if (!pt_aligned16(x1) || (x2 && !pt_aligned16(x2)) || !pt_aligned16(y) || (xcat && !pt_aligned16(xcat))) return PT_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == PT_F32) return gn_apply_t<float>(x1, x2, mean, rstd, gamma, beta, y, xcat, B, N, C1, C2, G, silu, raw_eps, s);
  if (dtype == PT_BF16) return gn_apply_t<bf16_t>(x1, x2, mean, rstd, gamma, beta, y, xcat, B, N, C1, C2, G, silu, raw_eps, s);
  return PT_ERR_DTYPE;
}

/* statistics + normalisation [+ SiLU] in one call: the slab kernel when the shape allows it, else stats / finalize / apply */
extern "C" int pt_groupnorm_fwd(const void* x1, const void* x2, const float* gamma, const float* beta, void* y, float* mean,
                                float* rstd, int64_t B, int64_t N, int64_t C1, int64_t C2, int64_t G, float eps, int silu,
                                int dtype, pt_stream stream) {
  if (B <= 0 || (C2 > 0 && !x2) || eps < 0.f) return PT_ERR_SHAPE;
  if (!pt_aligned16(x1) || (x2 && !pt_aligned16(x2)) || !pt_aligned16(y)) return PT_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int cch = dtype == PT_BF16 && B <= 65535 ? gn_slab_cch(N, C1, C2, G) : 0;
  if (cch) {
    GnSlab p{};
    p.x1 = (const bf16_t*)x1; p.x2 = (const bf16_t*)x2; p.C1 = (int)C1; p.C2 = (int)C2; p.C = (int)(C1 + C2); p.N = (int)N;
    p.G = (int)G; p.cpg = p.C / p.G; p.gamma = gamma; p.beta = beta; p.mean = mean; p.rstd = rstd; p.eps = eps; p.silu = silu;
    p.y = (bf16_t*)y;
    if (cch == 4) GN_SLAB_LAUNCH(gn_slab_fwd_kernel, 4, N, B, s, p); else GN_SLAB_LAUNCH(gn_slab_fwd_kernel, 8, N, B, s, p);
    PT_LAUNCH_CHECK();
    return PT_OK;
  }
  int st;
  if (dtype == PT_F32) {
    if ((st = gn_stats_t<float>(x1, x2, mean, rstd, B, N, C1, C2, G, eps, s))) return st;
    return gn_apply_t<float>(x1, x2, mean, rstd, gamma, beta, y, nullptr, B, N, C1, C2, G, silu, -1.f, s);
  }
  if (dtype != PT_BF16) return PT_ERR_DTYPE;
  if ((st = gn_stats_t<bf16_t>(x1, x2, mean, rstd, B, N, C1, C2, G, eps, s))) return st;
  return gn_apply_t<bf16_t>(x1, x2, mean, rstd, gamma, beta, y, nullptr, B, N, C1, C2, G, silu, -1.f, s);
}

template <typename T>
static int gn_bwd_t(const void* dy, const void* x1, const void* x2, const float* mean, const float* rstd,
                    const float* gamma, const float* beta, const void* dres, void* dx1, void* dx2, float* dgamma,
                    float* dbeta, float* ws, int64_t B, int64_t N, int64_t C1, int64_t C2, int64_t G, int silu,
                    int acc_dx2, float raw_eps, int ws_zeroed, int n_rep, int64_t rep_stride, hipStream_t s) {
  GnGeom g; int st = gn_geom<T>(g, N, C1, C2, G); if (st) return st;
  if (raw_eps >= 0.f) { g.raw_cnt = (float)N * (float)g.cpg; g.eps = raw_eps; }
  if (!ws_zeroed && hipMemsetAsync(ws, 0, sizeof(float) * B * G * 2, s) != hipSuccess) return PT_ERR_LAUNCH;
  dim3 grid((unsigned)((N + g.rows_per_block - 1) / g.rows_per_block), (unsigned)B);
  const size_t dyn = sizeof(float) * (2 * (size_t)g.C + 2 * (size_t)g.G);
  hipLaunchKernelGGL((gn_bwd_sums_kernel<T>), grid, dim3(NT), dyn, s, (const T*)dy, (const T*)x1, (const T*)x2, mean,
                     rstd, gamma, beta, dgamma, dbeta, ws, g, silu, n_rep, rep_stride);
  hipLaunchKernelGGL((gn_bwd_apply_kernel<T>), grid, dim3(NT), 0, s, (const T*)dy, (const T*)x1, (const T*)x2, mean,
                     rstd, gamma, beta, ws, (const T*)dres, (T*)dx1, (T*)dx2, g, silu, acc_dx2);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

extern "C" int pt_groupnorm_bwd(const void* dy, const void* x1, const void* x2, const float* mean, const float* rstd,
                                const float* gamma, const float* beta, const void* dres, void* dx1, void* dx2,
                                float* dgamma, float* dbeta, float* ws, int64_t B, int64_t N, int64_t C1, int64_t C2,
                                int64_t G, int silu, int accumulate_dx2, float raw_eps, int ws_zeroed, int n_rep,
                                int64_t rep_stride, float* dx_item_sum, int64_t item_ld, int dtype, pt_stream stream) {
  if (B <= 0 || (C2 > 0 && (!x2 || !dx2))) return PT_ERR_SHAPE;
  if (dx_item_sum && (C2 > 0 || item_ld < C1)) return PT_ERR_ARG;
  if (!pt_aligned16(dy) || !pt_aligned16(x1) || !pt_aligned16(dx1) || (x2 && !pt_aligned16(x2)) ||
      (dx2 && !pt_aligned16(dx2)) || (dres && !pt_aligned16(dres))) return PT_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (n_rep < 1 || (n_rep > 1 && rep_stride <= 0)) return PT_ERR_ARG;
  const int cch = dtype == PT_BF16 && B <= 65535 ? gn_slab_cch(N, C1, C2, G) : 0;
  if (cch) {
    GnSlab p{};
    p.x1 = (const bf16_t*)x1; p.x2 = (const bf16_t*)x2; p.C1 = (int)C1; p.C2 = (int)C2; p.C = (int)(C1 + C2); p.N = (int)N;
    p.G = (int)G; p.cpg = p.C / p.G; p.gamma = gamma; p.beta = beta; p.mean = const_cast<float*>(mean); p.rstd = const_cast<float*>(rstd);
    p.eps = raw_eps; p.raw_cnt = raw_eps >= 0.f ? 1.f : 0.f; p.silu = silu;
    p.dy = (const bf16_t*)dy; p.dres = (const bf16_t*)dres; p.dx1 = (bf16_t*)dx1; p.dx2 = (bf16_t*)dx2;
    p.dgamma = dgamma; p.dbeta = dbeta; p.acc_dx2 = accumulate_dx2; p.n_rep = n_rep; p.rep_stride = rep_stride;
    p.item_sum = dx_item_sum; p.item_ld = item_ld;
    if (cch == 4) GN_SLAB_LAUNCH(gn_slab_bwd_kernel, 4, N, B, s, p); else GN_SLAB_LAUNCH(gn_slab_bwd_kernel, 8, N, B, s, p);
    PT_LAUNCH_CHECK();
    return PT_OK;
  }
  int st;
  if (dtype == PT_F32) st = gn_bwd_t<float>(dy, x1, x2, mean, rstd, gamma, beta, dres, dx1, dx2, dgamma, dbeta, ws, B, N, C1, C2, G, silu, accumulate_dx2, raw_eps, ws_zeroed, n_rep, rep_stride, s);
  else if (dtype == PT_BF16) st = gn_bwd_t<bf16_t>(dy, x1, x2, mean, rstd, gamma, beta, dres, dx1, dx2, dgamma, dbeta, ws, B, N, C1, C2, G, silu, accumulate_dx2, raw_eps, ws_zeroed, n_rep, rep_stride, s);
  else return PT_ERR_DTYPE;
  if (st != PT_OK || !dx_item_sum) return st;
  return pt_colsum(dx1, C1, dx_item_sum, item_ld, B * N, C1, N, 1, 0, dtype, stream);   // two-pass path: a separate segmented sum
}
