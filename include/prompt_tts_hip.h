/*
 * prompt_tts_hip.h -- C-ABI of the MI355X (gfx950) hot-path library for khaidoan25/prompt-tts.
 *
 * The reference has no FFI of its own (it is pure Python on ATen); each entry point below
 * replaces the ATen ops that one reference call site dispatches, cited as file:line relative to
 * the reference root.  Conventions (all entry points):
 *   - plain pointers and sizes only; the CALLER owns every buffer; nothing here allocates,
 *     frees or synchronises; every launch goes to the `stream` argument only;
 *   - activations are token-major ("channels-last"): a (B, N, C) tensor is a row-major
 *     [B*N][C] matrix; `dtype` selects the storage/compute element type of activations and
 *     weight shadows: PT_F32 (parity mode, exact-f32 MFMA) or PT_BF16 (bf16 MFMA, f32 accumulate);
 *   - statistics, biases, LSE, losses, master weights and ALL weight gradients are f32;
 *   - return value 0 = launched; <0 = refused before any launch (see pt_status); never throws.
 *   - thread-safe: any entry point may be called from several host threads at once.  Process-wide state is limited to
 *     diagnostic switches read from the environment (the defaults are the product), the per-THREAD word behind
 *     pt_last_hip_error(), and ONE mutex-guarded event per device that orders the persistent LSTM launches (below).
 *   - concurrency on the DEVICE: every entry point may run beside any other on another stream, as long as the two calls
 *     share no output / scratch buffer -- with ONE exception, pt_lstm2_forward in its persistent forms: such a launch needs
 *     all of its workgroups resident at once (one per CU), so two of them must not overlap on one device.  The library
 *     enforces that itself: a call's persistent launches wait on the device (hipStreamWaitEvent) for the previous call's,
 *     whatever stream that was on (see pt_lstm2_forward).  Other kernels that hold CUs for long (another library's
 *     persistent kernels, a CU-masked stream) can still starve it: every wait inside is bounded and ends in the status word.
 */
#ifndef PROMPT_TTS_HIP_H
#define PROMPT_TTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pt_stream;            /* a hipStream_t */

enum pt_status { PT_OK = 0, PT_ERR_SHAPE = -1, PT_ERR_DTYPE = -2, PT_ERR_LAUNCH = -3, PT_ERR_ALIGN = -4,
                 PT_ERR_ARG = -5 };
enum pt_dtype { PT_F32 = 0, PT_BF16 = 1,
                PT_BF16X2 = 2 };    /* f32-class SPLIT storage for inference at the reference's fp32 precision (decode_codec.py:12-16):
                                       an element x is the bf16 pair hi = bf16(x), lo = bf16(x - hi); a row of C elements is stored
                                       [C hi | C lo] (2 C bf16, the bytes of C f32).  Products are bf16 x 3 splits on the bf16 MFMA
                                       (a_hi b_hi + a_hi b_lo + a_lo b_hi, error ~2^-16), accumulation / bias / ELU in f32.  The
                                       producer splits once; consumer loops hold no conversion.  Accepted by pt_gemm (forward),
                                       pt_rvq_decode, pt_lstm2_forward, pt_encodec_res / _stage / _tail only. */
enum pt_fp8_format { PT_FP8_E4M3 = 0, PT_FP8_E5M2 = 1 };   /* OCP fp8: e4m3 "fn" (max 448), e5m2 (max 57344) */

int pt_abi_version(void);                       /* bumps on any signature change */
const char* pt_status_string(int status);
const char* pt_last_hip_error(void);            /* hipGetErrorString of the HIP error behind the most recent PT_ERR_LAUNCH */
int pt_struct_size(int which);                  /* sizeof: 0 pt_operand, 1 pt_gemm_desc, 2 pt_attn_desc, 3 pt_param_seg, 4 pt_rowconv_desc, 5 pt_lstm2_desc, 6 pt_fold_seg, 7 pt_encodec_tail_desc, 8 pt_encodec_stage_desc, 9 pt_transpose_seg, 10 pt_decode_linear_desc */

/* ------------------------------------------------------------------------------------------------
 * GEMM family.  C[m][n] (+)= sum_k VA(m,k) * VB(n,k) with f32 accumulation on MFMA.
 * An operand is a "virtual matrix" V[row][col] (pt_operand) read either with the reduction index
 * on its columns (trans = 0) or on its rows (trans = 1; LDS-transposed fragment reads).
 * Replaces: nn.Linear / Conv1d(k=1) / Conv1d(k=3, stride 1|2) / nearest-x2-upsample+Conv1d and
 * their autograd backward (tts/ldm/resnet.py:34,73,171,193,226-228; transformer_1d.py:134;
 * unet_1d_condition.py:193-195,410-412; diffusers Attention.to_q/k/v/to_out, FeedForward).
 * ---------------------------------------------------------------------------------------------- */
enum pt_vkind {
  PT_V_PLAIN = 0,   /* V[r][c] = p[r*ld + c]                                     (c < cols)          */
  PT_V_CONCAT = 1,  /* c < c_split ? p[r*ld + c] : p2[r*ld2 + c - c_split]        (channel concat)    */
  PT_V_CONV = 2,    /* c = tap*cin + ci ; r = b*n_out + n ; V = p[(b*n_in + src(n,tap))*ld + ci] or 0  */
  PT_V_WFLIP = 3    /* r = tap*cout + co ; V = p[(co*3 + (2-tap))*ld_tap + c]  (conv dgrad weights)    */
};
enum pt_rowmap {    /* src(n, tap) for PT_V_CONV; "invalid" rows read as zero */
  PT_MAP_S1 = 0,        /* n + tap - 1                    (conv k3 stride 1, pad 1)                    */
  PT_MAP_S2 = 1,        /* 2n + tap - 1                   (conv k3 stride 2, pad 1)                    */
  PT_MAP_UP2 = 2,       /* (n + tap - 1) >> 1 over 2*n_in (nearest x2 upsample, then conv k3)          */
  PT_MAP_S2_DGRAD = 3,  /* u = n + tap - 1 ; u even ? u/2 : invalid   (dgrad of the stride-2 conv)     */
  PT_MAP_CAUSAL_REFLECT = 4, /* u = n + tap - (taps-1) ; u < 0 ? -u : u   (Encodec causal conv, reflect left pad) */
  PT_MAP_BACK = 5,      /* u = n - tap ; u < 0 invalid                (the two taps of a stride-r transposed conv) */
  PT_MAP_STRIDED_REFLECT = 6 /* u = n*stride + tap - (taps-stride) ; u < 0 ? -u : u  (Encodec ENCODER causal conv k = taps,
                                stride r, reflect left pad k - r; n_in = stride * n_out)                              */
};

typedef struct pt_operand {
  const void* p;  int64_t ld;
  const void* p2; int64_t ld2; int64_t c_split;     /* PT_V_CONCAT */
  int32_t kind;   int32_t trans;
  int32_t taps;   int32_t cin;  int32_t rowmap;     /* PT_V_CONV / PT_V_WFLIP (cin = cout there) */
  int32_t stride;                                   /* PT_MAP_STRIDED_REFLECT only */
  int64_t n_out;  int64_t n_in;                     /* rows per batch item of the output / source */
} pt_operand;

enum pt_out_kind {
  PT_OUT_T = 0,          /* store in `dtype`                                                      */
  PT_OUT_F32 = 1,        /* store f32                                                             */
  PT_OUT_F32_ATOMIC = 2  /* f32 atomicAdd (split-K wgrad into the flat grad buffer)               */
};

typedef struct pt_gemm_desc {
  int64_t M, N, K;
  pt_operand A, B;
  void* C; int64_t ldc;
  int32_t out_kind;
  int32_t split_k;               /* >= 1; > 1 requires PT_OUT_F32_ATOMIC                          */
  const float* bias;             /* [N] or NULL                                                   */
  const float* row_bias;         /* [M / row_bias_rows][N] f32 (time-embedding add) or NULL       */
  int64_t row_bias_rows;
  int64_t row_bias_ld;           /* stride between row_bias rows (0: N) -- lets it be a column slice of a wider matrix */
  const void* residual; int64_t ldr;   /* same dtype as activations, or NULL                      */
  const void* residual2; int64_t ldr2; /* second addend (may alias C: in-place accumulation)      */
  int32_t conv_wgrad_cin;        /* > 0 (padded-Cin conv wgrad): C index (m, n=tap*cin+ci) -> (m*3 + tap)*cin_store + ci, ci >= cin_store dropped */
  int32_t conv_wgrad_cin_store;  /* real Cin of the stored [Cout][3][Cin] gradient                 */
  float alpha;                   /* scales the accumulator before the epilogue adds               */
  int32_t act;                   /* 0 none, 1 ELU(alpha=1) applied to what is stored in C;
                                    2 GEGLU forward (bf16; diffusers FeedForward GEGLU): the N = 2F output columns are in the
                                      INTERLEAVED order (column 64q+t: t < 32 value 32q+t, else gate F+32q+t-32 -- the order of
                                      the weight shadow rows); C gets the raw projection, C2[m][32q+t] = value * gelu_erf(gate)
                                      (ldc2 >= F); bias is indexed in the ORIGINAL order;
                                    3 GEGLU backward fused into the following Linear's dgrad: the tile is d(act)[M][N = F],
                                      `residual` = the saved interleaved projection [M][>= 2F], C = d(projection) [M][ldc >= 2F],
                                      interleaved.  2 and 3 need M, N multiples of 256 and 16-byte aligned rows              */
  int32_t act2;                  /* same for the optional second output                            */
  void* C2; int64_t ldc2;        /* optional second store of the same tile (e.g. raw + ELU), or NULL */
  float* arow_sum;               /* PT_OUT_F32_ATOMIC only, or NULL: arow_sum[m] += alpha * sum_k VA(m,k) for m < arow_n -- the bias
                                    gradient (column sums of dy) rides on the wgrad GEMM as one extra all-ones MFMA column, in
                                    the workgroups of the first tile column; replicated destination like pt_colsum           */
  int64_t arow_n; int64_t arow_stride; int32_t arow_rep;
  int32_t f32_x3;                /* PT_F32 forward GEMMs (plain / conv operands, store epilogue): != 0 = products as a bf16 x 3 split
                                    (3 bf16 MFMAs, error ~2^-16 per product) instead of the exact f32 MFMA (8 instructions at 1/16
                                    of the bf16 rate).  For f32 INFERENCE that has to match fp32 to 1e-3 (Encodec decode,
                                    decode_codec.py:12-16); the training parity mode leaves it 0.                              */
  int64_t geglu_rows;            /* pt_wgrad_group only, or 0: F > 0 = the M = 2F rows of this weight gradient are in the
                                    interleaved order of act = 2 (A = d(projection) as written by act = 3); C and arow_sum are
                                    written in the ORIGINAL row order                                                       */
  int64_t x2_block;              /* PT_BF16X2 only, or 0 (= N): the output's plane blocking P -- column n of C / C2 is stored at
                                    (n / P) 2P + n % P (hi) and P further (lo).  P = cout for a transposed conv whose N = r cout
                                    columns are viewed as r rows of [cout hi | cout lo] by the next layer                   */
} pt_gemm_desc;

/* dtype PT_BF16X2 (forward GEMMs of f32-class inference; Encodec decode at the reference's fp32 precision): K, N, cin, c_split
 * count LOGICAL elements; every operand row holds its hi plane followed by its lo plane -- A: PT_V_PLAIN [M][>= 2K] (lo at + K),
 * PT_V_CONCAT (p: lo at + c_split, p2: lo at + K - c_split), PT_V_CONV (rows of [cin hi | cin lo]); B: PT_V_PLAIN [N][>= 2K];
 * K, cin and c_split multiples of 32, N of 8 (PT_ERR_ARG otherwise): a k-tile of the kernel carries the hi AND the lo plane of 32
 * logical columns of both operands and issues a_hi w_lo + a_lo w_hi + a_hi w_hi per fragment pair.  Outputs: PT_OUT_T = plane
 * rows (ldc >= 2N, see x2_block), PT_OUT_F32 = plain f32.  bias / act 0-1 / C2 + act2; no residuals, transposes or split-K. */
int pt_gemm(const pt_gemm_desc* d, int dtype, pt_stream stream);

/* Grouped weight gradients (bf16): up to 8 GEMMs dW_i (+)= alpha_i dY_i^T X_i in ONE launch + one fold launch.
 * Replaces the autograd weight-gradient ops of one transformer block / resnet of the reference (the backward of the nn.Linear
 * and Conv1d modules at tts/ldm/resnet.py:171,193,226-228, transformer_1d.py:134 and the diffusers Attention / FeedForward
 * linears) -- small outputs under an 8 192..32 768-token reduction.  Each descriptor is a pt_gemm weight-gradient descriptor:
 * out_kind = PT_OUT_F32_ATOMIC, A = dY (PT_V_PLAIN, trans = 1), B = X (trans = 1; all PT_V_PLAIN / PT_V_CONCAT or all
 * PT_V_CONV within one group), C = f32 gradient [M][ldc] accumulated in place (16-byte aligned rows preferred), optional
 * arow_sum bias gradient; split_k is chosen by the library so that the group's 256 x 256 tiles x K-slices fill `target_wgs`
 * workgroups (<= 0: 256, one per CU).  `ws` = f32 scratch for the split-K partial tiles, ws_floats >=
 * pt_wgrad_group_ws_floats(target_wgs); the caller owns it and must not touch it until the stream has passed the call. */
int64_t pt_wgrad_group_ws_floats(int target_wgs);
int pt_wgrad_group(const pt_gemm_desc* descs, int n, float* ws, int64_t ws_floats, int target_wgs, pt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Attention (non-causal or causal, no padding mask unless kv_len given), flash-style, f32 softmax.
 * q: [B*Nq][ldq] with head h at columns [h*D, (h+1)*D); k, v likewise over B*Nk rows.
 * Replaces F.scaled_dot_product_attention in diffusers AttnProcessor2_0 (K5 in SURVEY 2.4)
 * and its backward.  D in {32, 64, 128}.  lse: [B][H][Nq] f32.
 * kv_len (int32 [B] or NULL): keys >= kv_len[b] are masked out (build-defined; reference = NULL).
 * ---------------------------------------------------------------------------------------------- */
typedef struct pt_attn_desc {
  int64_t B, H, Nq, Nk, D;
  const void* q; int64_t ldq;
  const void* k; int64_t ldk;
  const void* v; int64_t ldv;
  void* o; int64_t ldo;
  float* lse;
  float scale;
  int32_t causal;
  const int32_t* kv_len;
  /* backward only */
  const void* d_o; int64_t lddo;
  float* delta;                  /* [B][H][Nq] f32 scratch: rowsum(dO * O) */
  void* dq; int64_t lddq;
  void* dk; int64_t lddk;
  void* dv; int64_t lddv;
} pt_attn_desc;

int pt_attn_fwd(const pt_attn_desc* d, int dtype, pt_stream stream);
int pt_attn_bwd(const pt_attn_desc* d, int dtype, pt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Normalisation / elementwise (HBM-bound).
 * ---------------------------------------------------------------------------------------------- */
/* nn.LayerNorm(C, eps) over rows of x[M][C]  (diffusers BasicTransformerBlock.norm1/2/3). */
int pt_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                     int64_t M, int64_t C, float eps, int dtype, pt_stream stream);
/* Replicated destinations (pt_layernorm_bwd, pt_groupnorm_bwd, pt_colsum): float atomics execute at the memory side and
 * hundreds of workgroups adding into ONE short row run ~14x below the chip-wide atomic rate.  With n_rep > 1 workgroup b adds
 * into ptr + (b % n_rep) * rep_stride instead of ptr (ptr = replica 0 inside a zeroed scratch arena) and
 * pt_fold_replicas sums the replicas into the real gradients once per step.  n_rep = 1: add straight into ptr. */
typedef struct pt_fold_seg {
  int64_t rep_off;       /* float offset of replica 0 in the arena                               */
  int64_t rep_stride;    /* floats between replicas                                              */
  int64_t dst_off;       /* float offset in the destination buffer                               */
  int32_t n;             /* elements                                                             */
  int32_t pad;
} pt_fold_seg;
/* dst[dst_off + i] += sum_r arena[rep_off + r * rep_stride + i] for every segment (max_n = largest n). */
int pt_fold_replicas(const float* arena, float* dst, const pt_fold_seg* segs_dev, int64_t n_segs, int n_rep,
                     int64_t max_n, pt_stream stream);

/* dst[c][r] = src[r][c] for every segment, one launch (bf16; rows, cols multiples of 64; 16-byte aligned pointers and row
 * pitches).  tile_begin = number of 64 x 64 tiles of the segments before this one; n_tiles = the total.  Keeps transposed
 * copies of the weight shadows fresh once per optimizer step: the autograd data gradient of nn.Linear (dx = dy W) then reads
 * W^T with the reduction index contiguous. */
typedef struct pt_transpose_seg {
  const void* src; void* dst;
  int64_t rows, cols, src_ld, dst_ld, tile_begin;
} pt_transpose_seg;
int pt_transpose_batch(const pt_transpose_seg* segs_dev, int64_t n_seg, int64_t n_tiles, int dtype, pt_stream stream);

/* dx = LN'(dy) [+ dres];  dgamma/dbeta += (f32 atomics). */
int pt_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                     const void* dres, void* dx, float* dgamma, float* dbeta,
                     int64_t M, int64_t C, int n_rep, int64_t rep_stride, int dtype, pt_stream stream);

/* nn.GroupNorm(G, C, eps) [+ SiLU] on token-major x = concat(x1[B][N][C1], x2[B][N][C2]) (x2 may be NULL)
 * (tts/ldm/resnet.py:238-240,267-273; transformer_1d.py:251; unet_1d_condition.py:731-733).
 * Statistics come in two forms.  Finalized: pt_groupnorm_stats(eps >= 0) zeroes mean/rstd, accumulates and finalizes them
 * (3 extra launches).  Raw: pt_groupnorm_stats(eps < 0) only ADDS the per-(batch, group) sum / sum of squares into
 * mean[] / rstd[], which the caller zeroed (e.g. one arena memset per step); apply / bwd are then given raw_eps = eps >= 0
 * and finalize on the fly (raw_eps < 0: the arrays hold finalized statistics). */
int pt_groupnorm_stats(const void* x1, const void* x2, float* mean, float* rstd,
                       int64_t B, int64_t N, int64_t C1, int64_t C2, int64_t G, float eps, int dtype, pt_stream stream);
/* statistics + normalisation in one call (writes finalized mean / rstd for the backward); bf16 inputs of up to 1024
 * tokens per item run as ONE kernel that keeps a 64-channel slab of the item in registers. */
int pt_groupnorm_fwd(const void* x1, const void* x2, const float* gamma, const float* beta, void* y, float* mean,
                     float* rstd, int64_t B, int64_t N, int64_t C1, int64_t C2, int64_t G, float eps, int silu,
                     int dtype, pt_stream stream);
int pt_groupnorm_apply(const void* x1, const void* x2, const float* mean, const float* rstd,
                       const float* gamma, const float* beta, void* y, void* xcat /* raw concat copy or NULL */,
                       int64_t B, int64_t N, int64_t C1, int64_t C2, int64_t G, int silu, float raw_eps, int dtype,
                       pt_stream stream);
/* backward: ws = f32 [B][G][2] scratch (zeroed by the call unless ws_zeroed).  dx1/dx2 = GN'(dy) [+ dres (concat layout)].
 * dx_item_sum (may be NULL; C2 must be 0): dx_item_sum[b][c] += sum_n dx1[(b, n)][c], row pitch item_ld floats -- the gradient of
 * the per-item time-embedding projection a ResnetBlock1D adds before norm2 (resnet.py:255-261), formed inside the slab kernel. */
int pt_groupnorm_bwd(const void* dy, const void* x1, const void* x2, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, const void* dres, void* dx1, void* dx2,
                     float* dgamma, float* dbeta, float* ws,
                     int64_t B, int64_t N, int64_t C1, int64_t C2, int64_t G, int silu, int accumulate_dx2,
                     float raw_eps, int ws_zeroed, int n_rep, int64_t rep_stride, float* dx_item_sum, int64_t item_ld,
                     int dtype, pt_stream stream);

/* GEGLU (diffusers FeedForward): out[m][j] = proj[m][j] * gelu_erf(proj[m][F + j]),  proj: [M][2F].
 * `bias` (f32 [2F] in the ORIGINAL column order, or NULL) is first added to proj in place.  interleaved != 0: proj / dproj
 * columns are in the interleaved order of pt_gemm act 2 / 3 (value of j at 64(j/32) + j%32, its gate 32 further): the
 * stand-alone form of those fused epilogues, for row counts they do not take. */
int pt_geglu_fwd(void* proj, const float* bias, void* out, int64_t M, int64_t F, int interleaved, int dtype, pt_stream stream);
int pt_geglu_bwd(const void* dout, const void* proj, void* dproj, int64_t M, int64_t F, int interleaved, int dtype, pt_stream stream);

/* SiLU on a flat f32/bf16 vector (time-embedding MLP, resnet.py:255-261). */
int pt_silu_fwd(const void* x, void* y, int64_t n, int dtype, pt_stream stream);
int pt_silu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype, pt_stream stream);

/* torch.nn.Dropout (diffusers Attention.to_out[1], FeedForward.net[1]; tts/models.py:95-100 forwards text_encoder_dropout):
 * y = x * keep[i] * scale [+ residual], keep = 1 byte per element (drawn by the caller), scale = 1 / (1 - p); the backward is the
 * same call on dy.  n a multiple of 16 bytes of elements; y may alias x. */
int pt_dropout(const void* x, const uint8_t* keep, const void* residual, void* y, int64_t n, float scale, int dtype, pt_stream stream);
/* y = a + b (flat), y may alias a. */
int pt_add(const void* a, const void* b, void* y, int64_t n, int dtype, pt_stream stream);
/* y[r] = x[2r] + x[2r+1] over rows of C (dgrad of nearest x2 upsample). rows = output rows. */
int pt_pairsum_rows(const void* x, void* y, int64_t rows, int64_t C, int dtype, pt_stream stream);
/* out[(m / seg_rows)][n] += sum over the rows m of each segment of dy[m][n]  (f32 atomics); seg_rows divides M.
 * seg_rows = M gives the bias gradient; seg_rows = rows per batch item gives the time-embedding gradient.
 * N need not be a multiple of the 16-byte chunk as long as ld is (pad columns are read, never written). */
int pt_colsum(const void* dy, int64_t ld, float* out, int64_t ld_out /* stride between segments in out; 0: N */,
              int64_t M, int64_t N, int64_t seg_rows, int n_rep, int64_t rep_stride, int dtype, pt_stream stream);

/* word_embedding(ids) + positional table (tts/models.py:112-115): out[b][s][:] = W[ids[b][s]][:] + pos[s][:]. */
int pt_embedding_fwd(const int32_t* ids, const void* W, const float* pos, void* out,
                     int64_t BS, int64_t S, int64_t d, int64_t vocab, int dtype, pt_stream stream);
int pt_embedding_bwd(const int32_t* ids, const void* dout, float* dW, int64_t BS, int64_t d, int64_t vocab,
                     int dtype, pt_stream stream);

/* diffusers Timesteps(C, flip_sin_to_cos=True, shift): out[b] = [cos(t w_i) | sin(t w_i)] (unet_1d_condition.py:209,622). */
int pt_timestep_embedding(const int64_t* t, void* out, int64_t B, int64_t C, int flip_sin_to_cos, float shift,
                          int dtype, pt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Training-step ends (train.py:86-107,116-120).
 * ---------------------------------------------------------------------------------------------- */
/* x_t = sqrt(abar[t]) x0 + sqrt(1-abar[t]) eps;  in: (B, n_q, T) f32 channel-first;
 * out: token-major [B*T][cpad] in `dtype` (channels >= n_q are zero). */
int pt_add_noise(const float* x0, const float* noise, const int64_t* t, const float* alphas_cumprod,
                 void* xt, int64_t B, int64_t n_q, int64_t T, int64_t cpad, int dtype, pt_stream stream);
/* token-major [B*T][cpad] -> (B, n_q, T) f32 (the `.sample` the reference returns) and back. */
int pt_tokens_to_bct(const void* x, float* out, int64_t B, int64_t n_q, int64_t T, int64_t cpad, int dtype, pt_stream stream);
int pt_bct_to_tokens(const float* x, void* out, int64_t B, int64_t n_q, int64_t T, int64_t cpad, int dtype, pt_stream stream);
/* loss += mean((pred - noise)^2) (f32 atomic into *loss, caller zeroes) and dpred = 2 (pred-noise) / numel * gscale.
 * pred token-major [B*T][cpad]; noise (B, n_q, T) f32; dpred token-major (pad channels zero). */
int pt_mse_loss(const void* pred, const float* noise, float* loss, void* dpred, float gscale,
                int64_t B, int64_t n_q, int64_t T, int64_t cpad, int dtype, pt_stream stream);

/* sum of squares of a flat f32 buffer into *out (f32 atomic; caller zeroes). */
int pt_sumsq(const float* g, float* out, int64_t n, pt_stream stream);

/* One fused AdamW step over a flat f32 master buffer (torch.optim.AdamW semantics, train.py:41-47,116-120):
 * clip = min(1, max_norm / (sqrt(*gnorm_sq) + 1e-6)) read on device; p -= lr*wd*p; m,v update; p -= step.
 * Also refreshes the activation-dtype weight shadow.  seg = table of n_seg tensors (pt_param_seg).
 * Gradients of layout-1 (Conv1d k=3) tensors are stored [Cout][3][Cin] (what the wgrad GEMM writes with contiguous
 * atomics) and the Adam moments m, v are indexed like the gradient; only p keeps the reference (Cout, Cin, 3) order. */
typedef struct pt_param_seg {
  int64_t offset;        /* element offset in the flat master / grad / m / v buffers            */
  int64_t numel;
  int64_t shadow_offset; /* element offset in the shadow buffer                                 */
  int32_t layout;        /* 0: copy; 1: Conv1d (Cout,Cin,3) -> shadow [Cout][3][cin_pad]; 2: GEGLU projection weight [2F][cin]
                            -> shadow rows interleaved (value row 32q+t at 64q+t, gate row F+32q+t at 64q+32+t), see pt_gemm act 2 */
  int32_t cin;           /* layout 1: Cin; layout 2: columns of the weight                        */
  int32_t cin_pad;       /* layout 1: padded Cin of the shadow (>= cin, multiple of 8)           */
  int32_t frozen;        /* 1: never updated (unused parameters: Transformer1DModel.proj_out)    */
} pt_param_seg;

int pt_adamw_step(float* p, const float* g, float* m, float* v, void* shadow, const pt_param_seg* seg_dev,
                  int64_t n_seg, int64_t n_total, const float* gnorm_sq, float max_norm,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                  int dtype, pt_stream stream);
/* The same update over positions [lo, hi) of the flat buffers only, walked in GRADIENT order (a Conv1d k=3 weight's gradient is
 * tap-major: position i then updates the master element at the transposed place of the same output row), so that a contiguous
 * range of the flat gradient buffer -- what a reduce-scatter hands one rank -- is a self-contained unit of optimizer work
 * (data-parallel training with a sharded optimizer pass; replaces DDP + a replicated torch.optim.AdamW, train.py:41-47,67-69,
 * 115-120).  lo % 4 == 0.  publish != 0: the updated values also overwrite g[lo, hi): an all-gather of the gradient buffer then
 * carries the new parameters to the other ranks, which adopt them with pt_import_params_range. */
int pt_adamw_step_range(float* p, float* g, float* m, float* v, void* shadow, const pt_param_seg* seg_dev, int64_t n_seg,
                        int64_t n_total, int64_t lo, int64_t hi, const float* gnorm_sq, float max_norm, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int64_t step, int publish, int dtype, pt_stream stream);
/* master[...] = values[lo, hi) (gradient order, as published by pt_adamw_step_range) + shadow refresh; frozen tensors keep theirs. */
int pt_import_params_range(float* p, const float* values, void* shadow, const pt_param_seg* seg_dev, int64_t n_seg, int64_t n_total,
                           int64_t lo, int64_t hi, int dtype, pt_stream stream);
/* shadow refresh only (after load_state_dict). */
int pt_pack_shadow(const float* p, void* shadow, const pt_param_seg* seg_dev, int64_t n_seg, int64_t n_total,
                   int dtype, pt_stream stream);


/* ------------------------------------------------------------------------------------------------
 * Encodec 24 kHz decoder (decode_codec.py:12-16 -> encodec.EncodecModel.decode).
 * Token-major activations (B*N, C); weights are EFFECTIVE (weight-norm folded on the host at load time).
 * Big layers (conv k7 128->512, the transposed convs, the C>=128 residual blocks) run on pt_gemm with the
 * PT_MAP_CAUSAL_REFLECT / PT_MAP_BACK row maps; the kernels below cover what a 128x128 GEMM tile cannot.
 * ---------------------------------------------------------------------------------------------- */
/* RVQ decode: out[(b,t)][:] = sum_q codebooks[q][codes[b][q][t]][:]   (codes int64 (B,n_q,T); dim multiple of 8).
 * PT_BF16X2: codebooks f32, out rows [dim hi | dim lo] bf16. */
int pt_rvq_decode(const int64_t* codes, const void* codebooks, void* out, int64_t B, int64_t n_q, int64_t T,
                  int64_t bins, int64_t dim, int dtype, pt_stream stream);

/* Row-streaming conv for few output channels (N <= 64): y[m][n] = act(bias[n] + sum_k A(m,k) w[n][k]) with
 * A(m, tap*cin+ci) = f(x[(b, src(n_row,tap))][ci]) for k < taps*cin and A(m, taps*cin + c) = f2(x2[m][c]) after that
 * (f, f2 = identity or ELU).  Weights stay in registers, rows stream HBM -> MFMA fragments directly, no LDS.
 * HBM-bound: bytes = M * (cin + cin2 + N) * sizeof(T) (tap re-reads are L1/L2 hits). */
typedef struct pt_rowconv_desc {
  int64_t B, n_rows;               /* M = B * n_rows output rows; taps never cross a batch item            */
  const void* x; int64_t ldx; int32_t cin; int32_t taps; int32_t rowmap; int32_t elu_x;
  const void* x2; int64_t ldx2; int32_t cin2; int32_t elu_x2;
  const void* w; int64_t ldw;      /* [N][ldw], ldw = K rounded up to 32, zero padded                       */
  const float* bias; int32_t N; int32_t act;   /* act: 0 none, 1 ELU                                        */
  void* y; int64_t ldy; int32_t y_f32;
  int32_t stride;                  /* PT_MAP_STRIDED_REFLECT: x has stride * n_rows rows per batch item               */
  int32_t f32_x3; int32_t _pad;    /* PT_F32: != 0 = bf16 x 3 products, as pt_gemm_desc.f32_x3                              */
} pt_rowconv_desc;
int pt_rowconv(const pt_rowconv_desc* d, int dtype, pt_stream stream);

/* One stage of the residual vector quantiser's ENCODE (data_preparation/generate_code.py:48 -> encodec quantizer.encode):
 * scores [M][bins] f32 = 2 x.e_j - |e_j|^2 (computed by the caller with an f32 pt_gemm: alpha = 2, bias = -|e|^2);
 * codes[b][q][t] = first argmax_j scores[(b,t)][j];  residual[(b,t)][:] -= codebook[argmax][:]   (all f32). */
int pt_rvq_search(const float* scores, const float* codebook, float* residual, int64_t* codes,
                  int64_t B, int64_t n_q, int64_t T, int64_t q, int64_t bins, int64_t dim, pt_stream stream);

/* Two-layer LSTM(H) over T steps + skip + ELU (encodec SLSTM; gate order i,f,g,o as torch.nn.LSTM):
 *   xg0   [B*T][4H]  = x W_ih0^T + b_ih0 + b_hh0  (computed by the caller with pt_gemm)
 *   layer 1 uses wcat1 [4H][2H] = [W_ih1 | W_hh1] and bias1 [4H] = b_ih1 + b_hh1
 *   out_elu[(b,t)][:] = ELU(h1_t + x[(b,t)][:])
 * h0_seq, h1_seq [B*T][H] and c0, c1 [B][H] (f32) are caller-provided scratch.  Two forms, chosen by the library:
 *   - persistent (PT_BF16, H = 512, h0_seq of B*T*H*2 >= 512 KiB + 512 bytes -- PT_F32: B*T*H*4 >= 1 MiB + 512 -- a device with >= 32 CUs): ONE launch per 8 * min(8, CUs / 32)
 *     batch rows runs all T steps with the recurrent weights resident in registers; clusters of 8 rows x 32 workgroups (one per
 *     CU) exchange the new hidden vectors every step through data-tagged 8-byte granules in h0_seq (its first 512 bytes hold the
 *     status word and the launch's XCD census).  Where the census finds every cluster on one XCD (a 256-CU device: read from the
 *     hardware per launch, never assumed) the exchange stays in that XCD's L2; otherwise it goes through memory (PT_LSTM_FORCE_
 *     REMOTE=1 forces that form).  Every workgroup of a launch must be resident at once; every wait is bounded;
 *   - persistent, f32-class (PT_F32, exact_f32 == 0): bf16 x 3 products; 8-byte granules of two hidden units, one 32-bit word
 *     each {hi bf16 | lo bf16 carrying a 2-bit hand-off tag in its lowest mantissa bits} -- a tag in every word, so no access of
 *     any width can pair a fresh tag with a stale value (PT_LSTM_F32_ROWS8=0's 16-row form: 16-byte granules {hi pair, tag, lo
 *     pair, tag}, one tag per 8-byte half); the same
 *     8-row x 32-workgroup clusters (eight waves per workgroup, hi / lo weight fragments in registers and LDS), XCD-local where the
 *     census allows; PT_LSTM_F32_ROWS8=0: clusters of 16 rows x 64 workgroups exchanging through memory;
 *   - persistent, exact f32 (PT_F32, exact_f32 != 0): that kernel on v_mfma_f32_16x16x4_f32 with the f32 hidden values themselves
 *     in the granules (the Encodec ENCODER: 15.8 -> 9.3 ms per 32 x 900 frames against the per-step kernels);
 *   - per step (other H, small devices or inputs, per_step != 0, PT_LSTM_PERSIST=0 / PT_LSTM_PERSIST_EXACT=0 -- both read per
 *     call): T + 1 dependent launches (layer 0 step s beside layer 1 step s - 1).
 * Persistent launches of ONE device are serialised by the library (a per-device event, recorded behind a call's last launch and
 * waited for -- on the device -- by the next call's first, under a host mutex that covers the enqueue only): two of them
 * overlapping would each wait for CUs the other holds.  Not applied to a stream under graph capture.
 * `status`: device pointer to ONE 32-bit word owned by the caller, or NULL.  The call clears it on the stream; after the call
 * has completed on the stream, 0 = ok and non-zero = a hand-off of the persistent form timed out (out_elu is then INVALID:
 * copy the word back -- e.g. to pinned memory on the same stream -- and check it before using the result).  With NULL the word is
 * the first 32 bits of h0_seq.  Latency-bound: reports steps/s, not a roofline fraction. */
/* PT_BF16X2: x and out_elu are plane rows [H hi | H lo] bf16; xg0, the weights and the scratch buffers are as for PT_F32
 * (h0_seq: B*T*H*4 bytes); only the persistent f32-class form exists -- where it does not apply (small inputs / devices,
 * per_step, exact_f32) the call returns PT_ERR_SHAPE and the caller converts and takes PT_F32. */
typedef struct pt_lstm2_desc {
  int64_t B, T, H;
  const void* x; const void* xg0; const void* whh0; const void* wcat1; const float* bias1;
  void* h0_seq; void* h1_seq; float* c0; float* c1; void* out_elu;
  void* status;
  int64_t exact_f32;     /* PT_F32: != 0 = the exact f32 MFMA, persistent or per step (the Encodec ENCODER, whose output
                            feeds integer code decisions); 0 = the persistent f32-class form where it applies (bf16 x 3 products:
                            ~1e-5 relative, for the decoder's 1e-3 waveform bound) */
  int64_t per_step;      /* != 0: the per-step kernels for THIS call (what a caller retries with after a timed-out hand-off) */
} pt_lstm2_desc;
int pt_lstm2_forward(const pt_lstm2_desc* d, int dtype, pt_stream stream);

/* Fused 24 kHz tail of the decoder (bf16): last transposed conv (cin 64 -> r = 2 x cout 32), its residual block (conv k3 32 -> 16,
 * 1x1 16 -> 32 + 1x1 shortcut, ELUs) and the final conv k7 32 -> 1, in one launch -- the part of EncodecModel.decode
 * (decode_codec.py:16) that runs at the output sample rate.  x: [B*n][ldx] ELU'd input rows; weights in the matrix forms of the
 * separate launches: wt [64][128] (row rho*32+co, col tap*64+ci), w3 [16][96], wf [32][64] (cols: 16 of the k3 branch | 32 of x1 |
 * zero pad), wfin [1][224]; biases f32; wav: [B][2n] f32.  Intermediates are rounded to bf16 where the separate launches round. */
typedef struct pt_encodec_tail_desc {
  int64_t B, n; int32_t cin, cout, r, _pad;
  const void* x; int64_t ldx;
  const void* wt; const float* bt; const void* w3; const float* b3; const void* wf; const float* bf; const void* wfin; const float* bfin;
  float* wav;
} pt_encodec_tail_desc;
/* PT_BF16X2 (all three fused kernels; csrc/encodec_x2.hip): x / y are plane rows ([cin hi | cin lo], ldx >= 2 cin; y likewise
 * over cout), the weight pointers are F32 matrices of the same [rows][K] shapes (wf of the tail: [32][64] with 48 used), split
 * into hi / lo fragments once per workgroup; intermediates live in LDS as planes, products are bf16 x 3. */
int pt_encodec_tail(const pt_encodec_tail_desc* d, int dtype, pt_stream stream);

/* Fused decoder stage (bf16) for (cin, cout, r) = (128, 64, 4): transposed conv + residual block (conv k3 cout -> cout/2, 1x1 +
 * 1x1 shortcut, ELUs) in one launch -- the 3 kHz -> 12 kHz stage of EncodecModel.decode.  x: [B*n][ldx] ELU'd input rows;
 * wt [r*cout][2*cin], w3 [cout/2][3*cout], wf [cout][cout/2 + cout] in the matrix forms of the separate launches; y: [B*r*n][ldy]
 * ELU'd output rows (the next stage's input). */
typedef struct pt_encodec_stage_desc {
  int64_t B, n; int32_t cin, cout, r, _pad;
  const void* x; int64_t ldx;
  const void* wt; const float* bt; const void* w3; const float* b3; const void* wf; const float* bf;
  void* y; int64_t ldy;
} pt_encodec_stage_desc;
int pt_encodec_stage(const pt_encodec_stage_desc* d, int dtype, pt_stream stream);
/* Fused residual block alone (bf16, cin = cout = 128, r = 1; wt / bt unused): x = RAW output rows [B*n][ldx] of the stage's
 * transposed conv, y = ELU(1x1([ELU(conv k3(ELU(x))) | x])) -- the residual block of the 600 Hz -> 3 kHz stage, whose transposed
 * conv (640 x 512 weights) stays a pt_gemm. */
int pt_encodec_res(const pt_encodec_stage_desc* d, int dtype, pt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * fp8 GEMMs (BASELINE configs[4], "fp8 MFMA GEMMs"): the forward and data-gradient GEMMs of the transformer's nn.Linear layers
 * with fp8 operands (v_mfma_f32_16x16x128_f8f6f4), f32 accumulation, bf16 output; per-tensor scales stay on the device.
 * ---------------------------------------------------------------------------------------------- */
#define PT_FP8_AMAX_BLOCKS 256
#define PT_FP8_STATE_FLOATS 258           /* {amax, scale, PT_FP8_AMAX_BLOCKS partial maxima (scratch)} */
/* Quantise a bf16 matrix x [rows][ldx] (cols % 16 == 0): state[0] = amax |x|, state[1] = scale = amax / FMAX (1 if amax == 0),
 * state (PT_FP8_STATE_FLOATS floats, no initialisation needed) also holds the reduction's scratch; out [rows][ld_out] = fp8(clamp(x / scale)) (round to nearest even).  out_t (may be NULL; needs rows, cols % 64 == 0) receives the
 * transpose [cols][ld_t] as well (weights: the dgrad GEMM reads W^T with the reduction index contiguous). */
int pt_fp8_quantize(const void* x, int64_t rows, int64_t cols, int64_t ldx, void* out, int64_t ld_out, void* out_t, int64_t ld_t,
                    float* state, int format, pt_stream stream);
/* C = (scale_a[0] * scale_b[0]) * A B^T (+ every store epilogue of pt_gemm: bias, residuals, act 1/2/3, C2): A [M][K] fp8 in
 * `a_format`, B [N][K] fp8 e4m3, both PT_V_PLAIN and not transposed (ld in elements = bytes), K % 16 == 0; C bf16.  alpha of the
 * descriptor still applies.  256 x 256 tiles on the eight-phase pipeline of pt_gemm; meant for M N >= ~128 tiles. */
int pt_gemm_fp8(const pt_gemm_desc* d, int a_format, const float* scale_a, const float* scale_b, pt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * north_star ops with NO reference implementation (SURVEY 8a'): build-defined, pinned to torch / numpy in tests.
 * ---------------------------------------------------------------------------------------------- */
/* codes[i] = clamp(rint((x[i] + 1) / 2 * (bins-1)), 0, bins-1): inverse of the collate normalisation (dataloader.py:64,143). */
/* One ancestral DDPM step (the sampler the reference implies with its DDPMScheduler(1000), train.py:32-36, but never wrote):
 * out = c_x0 * clamp((x - c_eps*eps) * c_inv, -clip, clip) + c_xt * x + sigma * z   (z may be NULL; clip <= 0: no clamp). */
int pt_ddpm_step(const float* x, const float* eps, const float* z, float* out, int64_t n, float c_eps, float c_inv,
                 float clip, float c_x0, float c_xt, float sigma, pt_stream stream);
int pt_codes_from_continuous(const float* x, int64_t* codes, int64_t n, int64_t bins, pt_stream stream);
/* Per row of logits[R][V] (RVQ-codebook logits head output): k == 1 greedy argmax (lowest index on ties); k > 1: softmax
 * over the k largest at `temperature`, inverse-CDF draw with the INJECTED uniform[row] in [0,1).  V <= 2048, k <= 64. */
int pt_sample_topk(const void* logits, int64_t ld, const float* uniforms, int64_t* out, int64_t R, int64_t V, int64_t k,
                   float temperature, int dtype, pt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Autoregressive decode step (north_star; no reference counterpart): the launches of ONE frame for <= 64 prompts, folded.
 * pt_decode_linear: y = epilogue([LayerNorm(x)] W^T) for M <= 64 rows, bf16, in one launch --
 *   ln_gamma / ln_beta != NULL: nn.LayerNorm(K, ln_eps) applied to the rows of x first (K <= 512), with pt_layernorm_fwd's arithmetic;
 *   bias [N] f32 or NULL, residual [M][ldr] bf16 or NULL (added after the bias), as pt_gemm;
 *   geglu != 0: W is a GEGLU projection in the interleaved shadow order (pt_param_seg layout 2), N = 2F, bias in the ORIGINAL order;
 *     y[m][F] = value * gelu_erf(gate) with the rounding points of pt_gemm followed by pt_geglu_fwd;
 *   seg_cols > 0: output columns [seg_cols, 2 seg_cols) go to y2 + t_dev[0] * t_stride (row pitch ld2) and [2 seg_cols, 3 seg_cols)
 *     to y3 + t_dev[0] * t_stride instead of y -- the fused q | k | v projection appending this frame's key and value to the
 *     (B, T, C) caches at the DEVICE-resident frame index (a captured decode step replays with a new index each time).
 * pt_ar_embed: out[b][:] = bf16(sum_q emb[q][prev[b][q]][:]) + pos[t_dev[0]][:]  (bf16; pt_rvq_decode + the position add).
 * pt_ar_advance (after pt_sample_topk): prev[i] = idx[i]; codes[i][t] = idx[i] (codes (B, n_q, T) int64, i = b n_q + q);
 *   t_dev[0] += 1; kv_len[b] += 1. */
typedef struct pt_decode_linear_desc {
  int64_t M, N, K;
  const void* x; int64_t ldx;
  const float* ln_gamma; const float* ln_beta; float ln_eps; int32_t geglu;
  const void* w; int64_t ldw;
  const float* bias;
  const void* residual; int64_t ldr;
  void* y; int64_t ldy;
  int64_t seg_cols; void* y2; int64_t ld2; void* y3; int64_t ld3; const int64_t* t_dev; int64_t t_stride;
} pt_decode_linear_desc;
int pt_decode_linear(const pt_decode_linear_desc* d, pt_stream stream);
int pt_ar_embed(const int64_t* prev, const void* emb, const void* pos, const int64_t* t_dev, void* out, int64_t B, int64_t n_q,
                int64_t bins, int64_t dim, pt_stream stream);
int pt_ar_advance(const int64_t* idx, int64_t* prev, int64_t* codes, int64_t* t_dev, int32_t* kv_len, int64_t B, int64_t n_q,
                  int64_t T, pt_stream stream);
/* dst[0 .. n) = src[t_dev[0] * ld .. + n)  (f32; the row of injected uniforms of the device-resident frame index). */
int pt_row_select(const float* src, int64_t ld, const int64_t* t_dev, float* dst, int64_t n, pt_stream stream);

/* A fixed sequence of launches issued by ONE call: the decode step's ~36 launches have constant arguments from frame to frame (every
 * per-frame quantity lives on the device), so the host builds the list once and replays it with one call per frame -- 2-3 us of
 * host time per launch instead of a Python round trip each, and no graph capture needed.  Stops at the first non-zero status. */
enum pt_op_kind { PT_OP_DECODE_LINEAR = 0, PT_OP_ATTN_FWD = 1, PT_OP_AR_EMBED = 2, PT_OP_SAMPLE_TOPK = 3, PT_OP_AR_ADVANCE = 4, PT_OP_ROW_SELECT = 5 };
typedef struct pt_ar_embed_desc { const int64_t* prev; const void* emb; const void* pos; const int64_t* t_dev; void* out; int64_t B, n_q, bins, dim; } pt_ar_embed_desc;
typedef struct pt_sample_desc { const void* logits; int64_t ld; const float* uniforms; int64_t* out; int64_t R, V, k; float temperature; int32_t dtype; } pt_sample_desc;
typedef struct pt_ar_advance_desc { const int64_t* idx; int64_t* prev; int64_t* codes; int64_t* t_dev; int32_t* kv_len; int64_t B, n_q, T; } pt_ar_advance_desc;
typedef struct pt_row_select_desc { const float* src; int64_t ld; const int64_t* t_dev; float* dst; int64_t n; } pt_row_select_desc;
typedef struct pt_op { int32_t kind; int32_t dtype; const void* desc; } pt_op;   /* dtype: PT_OP_ATTN_FWD only */
int pt_run_ops(const pt_op* ops, int64_t n, pt_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* PROMPT_TTS_HIP_H */
