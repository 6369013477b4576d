// Fused AdamW over ONE flat f32 master buffer (all parameters of the model), with global-norm clipping read
// from device memory (no host sync) and the activation-dtype weight-shadow refresh fused in.
// HBM-bound: 16 B/param read (p, g, m, v) + 12 B/param written (p, m, v) + sizeof(T) shadow write.
// Semantics = torch.optim.AdamW (decoupled weight decay, bias-corrected), train.py:41-47,116-120.
#include "common.h"

namespace {
constexpr int NT = 256;
constexpr int CHUNK = 4096;   // elements of the flat buffer per block-iteration

__device__ __forceinline__ int find_seg(const pt_param_seg* seg, int n_seg, int64_t pos) {
  int lo = 0, hi = n_seg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (seg[mid].offset <= pos) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// layout 2 (GEGLU projection weight [2F][cin], F = numel / cin / 2): shadow rows in the interleaved order the fused GEGLU
// epilogues of pt_gemm expect -- value row 32 q + t at 64 q + t, gate row F + 32 q + t at 64 q + 32 + t.
__device__ __forceinline__ int64_t geglu_shadow_index(const pt_param_seg& sg, int64_t local) {
  const int64_t row = local / sg.cin, col = local - row * sg.cin, F = sg.numel / sg.cin / 2;
  const int64_t r = row < F ? row : row - F;
  return (64 * (r >> 5) + (r & 31) + (row < F ? 0 : 32)) * sg.cin + col;
}

template <typename T>
__device__ __forceinline__ void shadow_store(T* shadow, const pt_param_seg& sg, int64_t local, float val) {
  if (sg.layout == 0) {
    shadow[sg.shadow_offset + local] = from_f32<T>(val);
  } else if (sg.layout == 2) {
    shadow[sg.shadow_offset + geglu_shadow_index(sg, local)] = from_f32<T>(val);
  } else {  // Conv1d weight (Cout, Cin, 3) -> [Cout][3][cin_pad]
    const int64_t per_co = (int64_t)sg.cin * 3;
    const int64_t co = local / per_co; const int rem = (int)(local - co * per_co);
    const int ci = rem / 3, tap = rem - ci * 3;
    shadow[sg.shadow_offset + (co * 3 + tap) * sg.cin_pad + ci] = from_f32<T>(val);
  }
}

// MODE 0: pack the shadow from the master (no update).  MODE 1: clip + AdamW + shadow refresh.  MODE 2: IMPORT -- the new
// parameter values are read from `g` (a peer rank computed them: sharded optimizer, prompt_tts_amd/parallel.py) and written to
// the master and the shadow.  The kernel walks flat positions [lo, hi) in GRADIENT order: for every tensor but a Conv1d k = 3
// weight that is the master order; a conv weight's gradient (and shadow) is tap-major [Cout][3][Cin] while its master is
// [Cout][Cin][3], so position i of the walk addresses g[i] and the master element at the transposed place of the same output
// row.  Walking in gradient order makes a contiguous range of the flat GRADIENT buffer (what a reduce-scatter hands a rank) a
// self-contained unit of work.  MODE 1 with `publish`: the new value also overwrites g[i] (the gradient is dead then), so an
// all-gather of the gradient buffer distributes the updated parameters.
template <typename T, int MODE>
__global__ __launch_bounds__(NT) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, T* __restrict__ shadow,
                                                   const pt_param_seg* __restrict__ seg, int n_seg, int64_t lo, int64_t hi,
                                                   const float* __restrict__ gnorm_sq, float max_norm, float lr, float b1,
                                                   float b2, float eps, float wd, float bc1, float bc2, int publish) {
  constexpr bool UPDATE = MODE == 1, IMPORT = MODE == 2;
  float clip = 1.f;
  if (UPDATE && gnorm_sq) {
    const float nrm = sqrtf(*gnorm_sq);
    clip = fminf(1.f, max_norm / (nrm + 1e-6f));       // torch.nn.utils.clip_grad_norm_
  }
  __shared__ int s_first;
  const float decay = 1.f - lr * wd, rbc2 = 1.f / sqrtf(bc2), step = lr / bc1;
  for (int64_t base = lo + (int64_t)blockIdx.x * CHUNK; base < hi; base += (int64_t)gridDim.x * CHUNK) {
    __syncthreads();
    if (threadIdx.x == 0) s_first = find_seg(seg, n_seg, base);     // one binary search per 4096-element chunk
    __syncthreads();
    // Quads: every tensor starts on a multiple of 4 elements (and lo is one), so 4 consecutive elements never straddle two
    // tensors.  Plain (layout 0) quads that lie wholly inside their tensor move as 16-byte loads / stores of p, g, m, v (and an
    // 8-byte shadow store); conv k=3 tensors, tails and padding take the element path.
    for (int k = threadIdx.x * 4; k < CHUNK; k += NT * 4) {
      const int64_t i = base + k;
      if (i >= hi) break;
      int si = s_first;
      while (si + 1 < n_seg && seg[si + 1].offset <= i) ++si;       // tensors are mostly larger than a chunk: 0-1 steps
      const pt_param_seg sg = seg[si];
      const int64_t local = i - sg.offset;
      if (local >= sg.numel) continue;                 // alignment padding between tensors
      if (sg.layout != 1 && local + 3 < sg.numel && i + 3 < hi && (sg.layout == 0 || (sg.cin & 3) == 0)) {
        f32x4_t pv;
        if (IMPORT) {
          pv = *reinterpret_cast<const f32x4_t*>(g + i);
          if (!sg.frozen) *reinterpret_cast<f32x4_t*>(p + i) = pv; else pv = *reinterpret_cast<const f32x4_t*>(p + i);
        } else {
          pv = *reinterpret_cast<const f32x4_t*>(p + i);
        }
        if (UPDATE && !sg.frozen) {
          const f32x4_t gv = *reinterpret_cast<const f32x4_t*>(g + i);
          f32x4_t mv = *reinterpret_cast<const f32x4_t*>(m + i), vv = *reinterpret_cast<const f32x4_t*>(v + i);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float ge = gv[e] * clip;
            mv[e] = b1 * mv[e] + (1.f - b1) * ge;
            vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
            pv[e] = pv[e] * decay - step * (mv[e] / (sqrtf(vv[e]) * rbc2 + eps));
          }
          *reinterpret_cast<f32x4_t*>(m + i) = mv; *reinterpret_cast<f32x4_t*>(v + i) = vv;
          *reinterpret_cast<f32x4_t*>(p + i) = pv;
        }
        if (UPDATE && publish) *reinterpret_cast<f32x4_t*>(g + i) = pv;
        store4<T>(shadow + sg.shadow_offset + (sg.layout == 2 ? geglu_shadow_index(sg, local) : local), pv[0], pv[1], pv[2], pv[3]);
        continue;
      }
      if (sg.layout == 1 && (sg.cin & 3) == 0 && local + 3 < sg.numel && i + 3 < hi) {
        // conv k=3 quad: 4 consecutive input channels of one (output channel, tap).  g, m, v (tap-major, like the shadow) move as
        // 16-byte vectors; only the master, which keeps the reference's (Cout, Cin, 3) order, is touched element-wise (stride 3)
        const int64_t per_co = (int64_t)sg.cin * 3;
        const int64_t co = local / per_co; const int rem = (int)(local - co * per_co);
        const int tap = rem / sg.cin, ci = rem - tap * sg.cin;
        float* pp = p + sg.offset + co * per_co + (int64_t)ci * 3 + tap;
        f32x4_t pv = {pp[0], pp[3], pp[6], pp[9]};
        if (IMPORT && !sg.frozen) {
          pv = *reinterpret_cast<const f32x4_t*>(g + i);
          pp[0] = pv[0]; pp[3] = pv[1]; pp[6] = pv[2]; pp[9] = pv[3];
        }
        if (UPDATE && !sg.frozen) {
          const f32x4_t gv = *reinterpret_cast<const f32x4_t*>(g + i);
          f32x4_t mv = *reinterpret_cast<const f32x4_t*>(m + i), vv = *reinterpret_cast<const f32x4_t*>(v + i);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float ge = gv[e] * clip;
            mv[e] = b1 * mv[e] + (1.f - b1) * ge;
            vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
            pv[e] = pv[e] * decay - step * (mv[e] / (sqrtf(vv[e]) * rbc2 + eps));
          }
          *reinterpret_cast<f32x4_t*>(m + i) = mv; *reinterpret_cast<f32x4_t*>(v + i) = vv;
          pp[0] = pv[0]; pp[3] = pv[1]; pp[6] = pv[2]; pp[9] = pv[3];
        }
        if (UPDATE && publish) *reinterpret_cast<f32x4_t*>(g + i) = pv;
        store4<T>(shadow + sg.shadow_offset + (co * 3 + tap) * sg.cin_pad + ci, pv[0], pv[1], pv[2], pv[3]);
        continue;
      }
      for (int e = 0; e < 4 && local + e < sg.numel && i + e < hi; ++e) {
        const int64_t ge_i = i + e;                  // position in the gradient buffer
        int64_t le = local + e;                      // element of the tensor in MASTER order
        if (sg.layout == 1) {                        // conv k=3: gradient position (co, tap, ci) <-> master element (co, ci, tap)
          const int64_t per_co = (int64_t)sg.cin * 3;
          const int64_t co = le / per_co; const int rem = (int)(le - co * per_co);
          const int tap = rem / sg.cin, ci = rem - tap * sg.cin;
          le = co * per_co + (int64_t)ci * 3 + tap;
        }
        const int64_t ie = sg.offset + le;
        float pv = p[ie];
        if (IMPORT && !sg.frozen) { pv = g[ge_i]; p[ie] = pv; }
        if (UPDATE && !sg.frozen) {
          // the Adam moments live at the GRADIENT position (for a conv weight: tap-major, like g): whoever owns a range of
          // the gradient buffer owns exactly the moments at the same positions
          const float gv = g[ge_i] * clip;
          const float mv = b1 * m[ge_i] + (1.f - b1) * gv;
          const float vv = b2 * v[ge_i] + (1.f - b2) * gv * gv;
          m[ge_i] = mv; v[ge_i] = vv;
          pv = pv * decay - step * (mv / (sqrtf(vv) * rbc2 + eps));
          p[ie] = pv;
        }
        if (UPDATE && publish) g[ge_i] = pv;
        shadow_store<T>(shadow, sg, le, pv);
      }
    }
  }
}
// Transposed bf16 copies of weight matrices (the data-gradient GEMMs then read W^T with the reduction index contiguous, the
// plain K-contiguous operand form: 6 - 23 % faster than transposed fragment reads of W in place, tools/dgrad_probe.py).
// One launch for every registered matrix; workgroup = one 64 x 64 tile through LDS (130-byte pitch: the 2-byte column reads
// of one instruction fall on 32 different banks).
__global__ __launch_bounds__(256) void transpose_batch_kernel(const pt_transpose_seg* __restrict__ segs, int n_seg) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[64][65];
  int lo = 0, hi = n_seg - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (segs[mid].tile_begin <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1; }
  const pt_transpose_seg sg = segs[lo];
  const int t = (int)(blockIdx.x - sg.tile_begin), tiles_c = (int)(sg.cols / 64);
  const int tr = t / tiles_c, tc = t - tr * tiles_c;
  const uint16_t* src = reinterpret_cast<const uint16_t*>(sg.src) + (int64_t)tr * 64 * sg.src_ld + tc * 64;
  uint16_t* dst = reinterpret_cast<uint16_t*>(sg.dst) + (int64_t)tc * 64 * sg.dst_ld + tr * 64;
  const int r = threadIdx.x >> 2, cq = threadIdx.x & 3;
  const u32x4_t a = *reinterpret_cast<const u32x4_t*>(src + (int64_t)r * sg.src_ld + 16 * cq);
  const u32x4_t b = *reinterpret_cast<const u32x4_t*>(src + (int64_t)r * sg.src_ld + 16 * cq + 8);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    tile[r][16 * cq + 2 * e] = (uint16_t)(a[e] & 0xffffu); tile[r][16 * cq + 2 * e + 1] = (uint16_t)(a[e] >> 16);
    tile[r][16 * cq + 8 + 2 * e] = (uint16_t)(b[e] & 0xffffu); tile[r][16 * cq + 8 + 2 * e + 1] = (uint16_t)(b[e] >> 16);
  }
  __syncthreads();
  u32x4_t o0, o1;                                   // output row r = source column r; 16 source rows 16 cq .. 16 cq + 15
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o0[e] = (uint32_t)tile[16 * cq + 2 * e][r] | ((uint32_t)tile[16 * cq + 2 * e + 1][r] << 16);
    o1[e] = (uint32_t)tile[16 * cq + 8 + 2 * e][r] | ((uint32_t)tile[16 * cq + 8 + 2 * e + 1][r] << 16);
  }
  *reinterpret_cast<u32x4_t*>(dst + (int64_t)r * sg.dst_ld + 16 * cq) = o0;
  *reinterpret_cast<u32x4_t*>(dst + (int64_t)r * sg.dst_ld + 16 * cq + 8) = o1;
}
}  // namespace

extern "C" int pt_transpose_batch(const pt_transpose_seg* segs_dev, int64_t n_seg, int64_t n_tiles, int dtype, pt_stream stream) {
  if (!segs_dev || n_seg <= 0 || n_seg > (1 << 20) || n_tiles <= 0 || n_tiles >= (1ll << 31)) return PT_ERR_ARG;
  if (dtype != PT_BF16) return PT_ERR_DTYPE;
  hipLaunchKernelGGL(transpose_batch_kernel, dim3((unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream, segs_dev, (int)n_seg);
  PT_LAUNCH_CHECK();
  return PT_OK;
}

static int adamw_launch(int mode, float* p, float* g, float* m, float* v, void* shadow, const pt_param_seg* seg_dev, int64_t n_seg,
                        int64_t lo, int64_t hi, const float* gnorm_sq, float max_norm, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int64_t step, int publish, int dtype, hipStream_t s) {
  const float bc1 = mode == 1 ? 1.f - powf(beta1, (float)step) : 1.f, bc2 = mode == 1 ? 1.f - powf(beta2, (float)step) : 1.f;
  int64_t blocks = (hi - lo + CHUNK - 1) / CHUNK;
  if (blocks > 4096) blocks = 4096;
#define ADAMW_GO(TT, MODE) hipLaunchKernelGGL((adamw_kernel<TT, MODE>), dim3((unsigned)blocks), dim3(NT), 0, s, p, g, m, v, (TT*)shadow, \
                                              seg_dev, (int)n_seg, lo, hi, gnorm_sq, max_norm, lr, beta1, beta2, eps, weight_decay, bc1, bc2, publish)
  if (dtype == PT_F32) { if (mode == 0) ADAMW_GO(float, 0); else if (mode == 1) ADAMW_GO(float, 1); else ADAMW_GO(float, 2); }
  else if (dtype == PT_BF16) { if (mode == 0) ADAMW_GO(bf16_t, 0); else if (mode == 1) ADAMW_GO(bf16_t, 1); else ADAMW_GO(bf16_t, 2); }
  else return PT_ERR_DTYPE;
#undef ADAMW_GO
  PT_LAUNCH_CHECK();
  return PT_OK;
}

extern "C" int pt_adamw_step(float* p, const float* g, float* m, float* v, void* shadow, const pt_param_seg* seg_dev,
                             int64_t n_seg, int64_t n_total, const float* gnorm_sq, float max_norm, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int64_t step, int dtype, pt_stream stream) {
  if (n_seg <= 0 || n_total <= 0 || step <= 0) return PT_ERR_SHAPE;
  if (!p || !g || !m || !v || !shadow || !seg_dev) return PT_ERR_ARG;
  return adamw_launch(1, p, const_cast<float*>(g), m, v, shadow, seg_dev, n_seg, 0, n_total, gnorm_sq, max_norm, lr, beta1, beta2, eps,
                      weight_decay, step, 0, dtype, (hipStream_t)stream);
}

extern "C" int pt_adamw_step_range(float* p, float* g, float* m, float* v, void* shadow, const pt_param_seg* seg_dev, int64_t n_seg,
                                   int64_t n_total, int64_t lo, int64_t hi, const float* gnorm_sq, float max_norm, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, int64_t step, int publish, int dtype,
                                   pt_stream stream) {
  if (n_seg <= 0 || n_total <= 0 || step <= 0 || lo < 0 || hi > n_total || lo >= hi || (lo & 3)) return PT_ERR_SHAPE;
  if (!p || !g || !m || !v || !shadow || !seg_dev) return PT_ERR_ARG;
  return adamw_launch(1, p, g, m, v, shadow, seg_dev, n_seg, lo, hi, gnorm_sq, max_norm, lr, beta1, beta2, eps, weight_decay, step,
                      publish ? 1 : 0, dtype, (hipStream_t)stream);
}

extern "C" int pt_import_params_range(float* p, const float* values, void* shadow, const pt_param_seg* seg_dev, int64_t n_seg,
                                      int64_t n_total, int64_t lo, int64_t hi, int dtype, pt_stream stream) {
  if (n_seg <= 0 || n_total <= 0 || lo < 0 || hi > n_total || lo >= hi || (lo & 3)) return PT_ERR_SHAPE;
  if (!p || !values || !shadow || !seg_dev) return PT_ERR_ARG;
  return adamw_launch(2, p, const_cast<float*>(values), nullptr, nullptr, shadow, seg_dev, n_seg, lo, hi, nullptr, 0.f, 0.f, 0.f, 0.f,
                      0.f, 0.f, 1, 0, dtype, (hipStream_t)stream);
}

extern "C" int pt_pack_shadow(const float* p, void* shadow, const pt_param_seg* seg_dev, int64_t n_seg, int64_t n_total,
                              int dtype, pt_stream stream) {
  if (n_seg <= 0 || n_total <= 0) return PT_ERR_SHAPE;
  if (!p || !shadow || !seg_dev) return PT_ERR_ARG;
  return adamw_launch(0, const_cast<float*>(p), nullptr, nullptr, nullptr, shadow, seg_dev, n_seg, 0, n_total, nullptr, 0.f, 0.f, 0.f,
                      0.f, 0.f, 0.f, 1, 0, dtype, (hipStream_t)stream);
}
