/*
 * rcdm.h — C-ABI of librcdm_hip.so: the MI355X (gfx950) kernels underneath the RCDMs stage-2
 * denoiser (UNet3DConditionModel.forward + the CFG/DDIM loop).
 *
 * The reference (muzishen/RCDMs) has NO FFI of its own: every op on this path is a torch/ATen call
 * made from Python.  Each entry point below therefore names the reference Python call site(s) it
 * replaces (paths relative to the reference repo).  All functions:
 *   - take plain device pointers + sizes (no torch types), caller-owned buffers, an explicit
 *     hipStream_t (passed as void*), never allocate, never synchronise, are graph-capturable;
 *   - return 0 on success, a negative RCDM_E* code otherwise (never throw);
 *   - compute in fp16 storage / fp32 accumulate ("f16" below = IEEE binary16).
 *
 * Activation layout everywhere: channels-last rows.  A tensor the reference holds as
 * (b, C, f, H, W) is one row-major matrix X[M][ld] with M = b*f*H*W rows ordered (b, f, y, x) and C
 * contiguous channels per row; `ld` (>= C, multiple of 8) lets a tensor live inside a wider
 * "concat" buffer so torch.cat([h, skip], dim=1) (unet_blocks.py:644,754) costs nothing.
 */
#ifndef RCDM_H
#define RCDM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCDM_VERSION 0x000300 /* 0.3.0: + rcdm_conv3x3_wino*, rcdm_groupnorm_stats_prestat / _finalize (no descriptor changed) */
/* ABI rule: every descriptor struct below MUST be zero-initialised by the caller (memset / `= {0}` / calloc) before its
 * fields are set.  Descriptors grow at the END between versions and a zero in a field this header does not describe yet
 * always means "the behaviour of the previous version": 0.2.0 added rcdm_conv3x3_desc.c_in2 / lda2 (rcdm_conv3x3_add1x1)
 * and rcdm_attn_desc.flags (RCDM_ATTN_WIDE_RANGE) — a caller compiled against 0.1.0 that did not zero its structs passes
 * garbage there; rcdm_conv3x3 returns RCDM_EINVAL for c_in2 != 0 in every form, the phase form (upsample = 2) included. */

/* error codes */
#define RCDM_OK 0
#define RCDM_EINVAL (-1)   /* bad argument (null pointer, negative size, misaligned ld) */
#define RCDM_ESHAPE (-2)   /* shape not supported by this kernel */
#define RCDM_ELAUNCH (-3)  /* HIP launch / runtime error (see rcdm_last_hip_error) */
#define RCDM_EWORKSPACE (-4) /* workspace too small */
#define RCDM_ECOMM (-5)    /* RCCL missing or an RCCL call failed (see rcdm_comm_last_error) */

/* epilogue flags for rcdm_gemm / rcdm_conv3x3 (applied in fp32, one final rounding to f16) */
#define RCDM_EPI_BIAS 1      /* + bias[n]                              (fp32 [N])                  */
#define RCDM_EPI_ROWVEC 2    /* + rowvec[m / rows_per_sample][n]       (fp32, resnet.py:191-194)   */
#define RCDM_EPI_RESIDUAL 4  /* + residual[m][n]                       (f16, ldr)                  */
#define RCDM_EPI_GELU 16     /* out = gelu(acc + bias + rowvec) (+ residual): exact erf GELU, the "gelu" FeedForward of
                             * the stage-1 prior (diffusers GELU(approximate="none")); not with GEGLU  */
#define RCDM_EPI_GEGLU 8     /* out[m][j] = (h+bh) * gelu(g+bg); W/bias rows packed in groups of   */
                             /* 32 = 16 hidden rows then their 16 gate rows (rcdm_pack_geglu_rows) */

int rcdm_version(void);
/* last HIP error code seen by this library on this thread (0 = none) and its string */
int rcdm_last_hip_error(void);
const char* rcdm_last_hip_error_string(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM  out[M][N] = epi( A[M][K] * W[N][K]^T )        replaces: nn.Linear / 1x1 InflatedConv3d calls
 *   attention.py:121,140-141,164 (to_q/k/v/out), :330,:352 (proj_in/out 1x1 conv),
 *   motion_module.py:166,170 (proj_in/out), resnet.py:208 (conv_shortcut),
 *   diffusers FeedForward GEGLU proj + out (attention.py:514, motion_module.py:243).
 *   A f16 row stride lda; W f16 [N][K] (nn.Linear's own [out,in] layout); K % 8 == 0, N % 8 == 0.
 *   split_k: 0 = library heuristic, 1 = off, >1 = forced; needs workspace (rcdm_gemm_workspace_bytes).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t M, N, K;
  int32_t lda, ldc, ldr;      /* row strides in elements (ldr only with RCDM_EPI_RESIDUAL) */
  int32_t epilogue;           /* RCDM_EPI_* bits */
  int32_t rows_per_sample;    /* RCDM_EPI_ROWVEC: rowvec row = m / rows_per_sample */
  int32_t ldt;                /* RCDM_EPI_ROWVEC: rowvec row stride (floats) */
  float out_scale;            /* out = epi(...) * out_scale   (1/output_scale_factor, resnet.py:210) */
  int32_t split_k;
  int32_t dup_rows;           /* != 0: every output row m is ALSO stored at row m + dup_rows of `out` (the two CFG halves
                               * of a step share everything before the first cross-attention: computed once, stored twice) */
} rcdm_gemm_desc;

size_t rcdm_gemm_workspace_bytes(const rcdm_gemm_desc* d);
/* rcdm_gemm with the LayerNorm that follows it fused into the epilogue: out = epi(A W^T) as rcdm_gemm (bias / residual /
 * out_scale / dup_rows), and ln->out[m][:] = LayerNorm(out[m][:]) * gamma + beta (+ pe[(m / rows_per_frame) % frames][:])
 * computed from the f16-rounded `out` row, exactly what rcdm_layernorm would read back.  For N <= 320 (one 160x320 tile
 * spans the row): the token-matrix GEMMs of the 64x64 level (to_out / proj_in -> norm1/2/3, attention.py:479-526,
 * motion_module.py:234-246), where the separate LayerNorm is a 26-MB read + 26-MB write per call.  No GEGLU / GELU /
 * row vector / split-K; no workspace. */
typedef struct {
  const float* gamma;
  const float* beta;
  const float* pe;            /* NULL, or the positional-encoding table [frames][N] (fp32) */
  void* out;                  /* f16 [M][ld] */
  int32_t ld;
  int32_t rows_per_frame, frames;
  float eps;
} rcdm_ln_fuse;
int rcdm_gemm_ln(const rcdm_gemm_desc* d, const rcdm_ln_fuse* ln, const void* A, const void* W, const float* bias,
                 const void* residual, void* out, void* stream);

/* Deferred LayerNorm: the nn.LayerNorm between two token GEMMs (attention.py:482,502,514: norm1/2/3 in front of to_q|k|v,
 * attn2.to_q and the GEGLU projection; motion_module.py:236,243: norms[i] + pos_encoder, ff_norm) without a launch of its
 * own and without the normalised tensor ever existing in HBM.  With W' = W diag(gamma) and b' = b + W beta (folded once, at
 * pack time; for pos_encoder a per-frame row table b'_f = b' + W pe_f passed as `rowvec`):
 *     LayerNorm(x) W^T + b  =  rstd (x W'^T) - (rstd mean) S + b',      S[n] = sum_c f16(W'[n][c]).
 *   PRODUCER (the GEMM that writes the rows x, e.g. to_out + residual or proj_in): stat_out[slot][m] (float2 each, slot-major:
 *     a consumer's lanes read consecutive rows of one slot) = (sum, sum of squares) of the f16 values it stores to row m,
 *     one slot per column tile of its launch.  stat_parts
 *     must equal rcdm_gemm_stat_parts(d) (the column-tile count of the tile shape a statistics-producing launch of this
 *     shape uses; RCDM_ESHAPE otherwise).  Plain / bias / row-vector / residual epilogues, no GEGLU, no split-K.
 *   CONSUMER (A = the RAW rows x, W = f16(W'), bias = b', colsum = S in the packed column order of W): stat_in / parts_in
 *     = the producer's partials of the A rows, C = the LayerNorm width (= K), eps.  Its epilogue forms (rstd, mean rstd)
 *     per row from the partials (var = E[x^2] - mean^2 in fp32) and applies the identity before bias / row vector /
 *     GELU / GEGLU / residual.  Every epilogue form, no split-K; parts_in <= 20.
 *   One call may be both.  Both sides NULL = rcdm_gemm. */
typedef struct {
  float* stat_out;            /* producer: slot-major [stat_parts][stat_out_rows][2] fp32, or NULL */
  int32_t stat_parts;
  int32_t stat_out_rows;      /* rows per slot plane of stat_out: >= M + dup_rows */
  const float* stat_in;       /* consumer: [parts_in][stat_in_rows][2] fp32 (the producer's buffer), or NULL */
  int32_t parts_in;
  int32_t stat_in_rows;       /* rows per slot plane of stat_in (= the producer's stat_out_rows) */
  const float* colsum;        /* consumer: S [N] fp32 (16-byte aligned) */
  float eps;                  /* LayerNorm eps (1e-5) */
  int32_t C;                  /* LayerNorm width = K of the consumer */
} rcdm_lnx;
int rcdm_gemm_stat_parts(const rcdm_gemm_desc* d);   /* 0: a statistics-producing launch of this shape is not available */
/* The same for a call that is PRODUCER and, with consumer != 0, also CONSUMER: the consumer flag steers the tile choice too,
 * so a call that carries both sides must size stat_out from this query (consumer = 0 is rcdm_gemm_stat_parts). */
int rcdm_gemm_lnx_stat_parts(const rcdm_gemm_desc* d, int32_t consumer);
/* Two producers that fill ONE statistics buffer (the same projection on all rows and, later, on a row subset whose rows it
 * rewrites: the rank-1-context plan of the cross-attention) must use the same slot count.  rcdm_gemm_lnx therefore also
 * accepts a stat_parts other than rcdm_gemm_lnx_stat_parts(d, ..) where an LDS-DMA tile with exactly that column-tile
 * count exists (ceil(N / 64) or ceil(N / 128)) and takes that tile, unsplit; 1 / 0: */
int rcdm_gemm_lnx_parts_ok(const rcdm_gemm_desc* d, int32_t parts, int32_t consumer);
/* Workspace an rcdm_gemm_lnx call of this shape needs, with the SAME tile / split decision the call itself takes (statistics
 * producers and consumers are steered to other tile shapes than a plain rcdm_gemm of the shape: rcdm_gemm_workspace_bytes can
 * disagree).  Non-zero means the shape would run split-K, which rcdm_gemm_lnx refuses on either side (RCDM_ESHAPE). */
size_t rcdm_gemm_lnx_workspace_bytes(const rcdm_gemm_desc* d, int32_t producer, int32_t consumer);
int rcdm_gemm_lnx(const rcdm_gemm_desc* d, const rcdm_lnx* x, const void* A, const void* W, const float* bias,
                  const float* rowvec, const void* residual, void* out, void* workspace, size_t workspace_bytes,
                  void* stream);

/* tuning/test knob for rcdm_gemm and rcdm_conv3x3: -1 = automatic (default: chosen per shape; env
 * RCDM_IGEMM=dma128|dma256|dma64 overrides), tile (pixels x channels): 1 = 128x128, 2 = 256x256, 3 = 64x64,
 * 4 = 64x64 with a four-slot LDS ring, 5 = 128x64.
 * Changes the workspace size a shape needs: query rcdm_*_workspace_bytes after setting it. */
/* per-shape overrides of the tile heuristics: "taps,M,N,Cin,variant,split;..." (taps 1 | 9, variant 1 .. 10, split 0 = that
 * variant's heuristic), looked at before the library's own measured table; "off" ignores the table, "" adds nothing, NULL goes
 * back to the environment variable RCDM_SHAPE_RULES (same syntax).  For tuning another chip / model without a rebuild
 * (tools/tune_rules.py).  Changes workspace sizes like rcdm_set_igemm_variant. */
int rcdm_set_shape_rules(const char* rules);
int rcdm_set_igemm_variant(int32_t variant);  /* 6 / 7 / 8: ping-pong kernel at 160x320 / 160x256 / 256x256; 9: igemm16 (160x160); 10: 128x64 with a three-slot LDS ring (GEMMs; a conv runs as 5) */
/* Tuning switch: 0 = the shape heuristic never picks the 8-wave ping-pong kernel (igemm8.hip), 1 = it may
 * (default; environment RCDM_PP=0 sets the initial state).  Used for same-process A/B timing. */
int rcdm_set_igemm_pingpong(int32_t on);
/* Tuning / test switch: 1 = split-K launches of the 160x160 and the LDS-DMA tile kernels leave f16 slabs (half the bytes to and
 * from the reduce pass; one extra f16 rounding per partial sum; not the GEGLU projections, not the ping-pong kernel), 0 = fp32
 * slabs, -1 = default (environment RCDM_SLAB16, else 1: measured -0.11 ms per step, whole-UNet error unchanged). */
int rcdm_set_splitk_slab_f16(int32_t on);
/* debug: when non-NULL, every igemm block writes 4 int64 {start, end (s_memtime ticks), ticks spent in epilogues,
 * k-steps done} at trace[(blockIdx.y*gridDim.x + blockIdx.x)*4]; NULL (default) disables it. */
/* The launch geometry the library chooses for a shape (what rcdm_gemm / rcdm_gemm_lnx / rcdm_conv3x3* will do; diagnostics:
 * tools/ceiling.py prices tile quantisation with it): out8 = {tile variant, BM, BN, row tiles, column tiles, split-K factor,
 * resident blocks per CU of that tile, k-steps of 64}.  No launch, no device access. */
int rcdm_gemm_plan_query(const rcdm_gemm_desc* d, int32_t producer, int32_t consumer, int32_t* out8);
int rcdm_debug_set_igemm_trace(void* device_buffer);
/* debug, builds with -DRCDM_ATTN_TRACE only (tools/trace_attn.py): when non-NULL every wave of rcdm_flash_attn writes 8 int64 at
 * trace[(block * waves + wave) * 8]: {loop ticks, ticks at the barrier + K/V staging, in the QK^T issue, in the softmax, in the PV
 * issue, key tiles, 0, 0}.  The product build ignores the pointer. */
int rcdm_debug_set_attn_trace(void* device_buffer);
/* roofline calibration: `blocks` x 4 waves each run iters x 4 independent v_mfma_f32_32x32x16_f16 (32768 flop each) and
 * nothing else; ticks[block] (optional) = s_memtime ticks the first wave spent in its loop.  tools/mfma_peak.py */
int rcdm_debug_mfma_peak(int32_t blocks, int32_t iters, float* sink, long long* ticks, void* stream);
int rcdm_gemm(const rcdm_gemm_desc* d, const void* A, const void* W, const float* bias,
              const float* rowvec, const void* residual, void* out, void* workspace,
              size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * conv3x3 (pad 1) as implicit GEMM over channels-last rows.   replaces InflatedConv3d.forward
 *   resnet.py:10-18 with k=3: ResnetBlock3D conv1/conv2 (:188,:205), Downsample3D (:104, stride 2),
 *   Upsample3D (:65+:78: F.interpolate(nearest, x2) folded into the input indexing, never
 *   materialised), unet.py:403 conv_in, :457 conv_out.
 *   in : [n_img*h_in*w_in][lda] f16 (c_in channels used);  W f16 [c_out][9*c_in], k = tap*c_in + c,
 *   tap = ky*3+kx (i.e. torch weight.permute(0,2,3,1)); out rows = n_img*h_out*w_out where
 *   h_out = (h_in*(1+upsample) - 1)/stride + 1.  c_in % 8 == 0, c_out % 8 == 0.
 *   upsample = 2: the SAME operation as upsample = 1 (Upsample3D: nearest x2, then the 3x3 conv) computed as four 2x2
 *   "phase" convolutions over the SOURCE grid — output pixel (2y+a, 2x+b) of the upsampled image only ever sees the 2x2
 *   source pixels (y+a-1+r, x+b-1+c), each with the sum of the 3x3 taps that land on it — 4/9 of the multiply-adds.  W is
 *   then the phase image written by rcdm_pack_conv3x3_up2 (f16 [4][c_out][4*c_in]: the tap sums are formed in fp32 from the
 *   fp32 weights and rounded once), the epilogue may only carry RCDM_EPI_BIAS, stride 1, c_in % 64 == 0 (workspace:
 *   rcdm_conv3x3_workspace_bytes, as for the other forms).  Only shapes whose tiles fill the chip are taken: ask rcdm_conv3x3_up2_supported first (0: use
 *   upsample = 1 with the [c_out][9*c_in] weights; rcdm_conv3x3 itself returns RCDM_ESHAPE for such a descriptor).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t n_img, h_in, w_in, c_in, c_out;
  int32_t stride;     /* 1 | 2 */
  int32_t upsample;   /* 0 | 1 | 2 (2: phase form of 1, see above) */
  int32_t lda, ldc, ldr;
  int32_t epilogue;
  int32_t rows_per_sample, ldt;
  float out_scale;
  int32_t split_k;
  int32_t pad_after_only; /* 0: padding 1 on every side (default).  1: zero rows/columns only AFTER the image —
                           * diffusers Downsample2D(padding=0): F.pad(x, (0,1,0,1)) then a stride-2 conv (VAE encoder) */
  int32_t dup_rows;       /* as rcdm_gemm_desc.dup_rows */
  int32_t c_in2, lda2;    /* rcdm_conv3x3_add1x1 only (0 otherwise): channels and row stride of the second input */
} rcdm_conv3x3_desc;

size_t rcdm_conv3x3_workspace_bytes(const rcdm_conv3x3_desc* d);
int rcdm_conv3x3_plan_query(const rcdm_conv3x3_desc* d, int32_t* out8);   /* as rcdm_gemm_plan_query */
int rcdm_conv3x3_up2_supported(const rcdm_conv3x3_desc* d);   /* 1 | 0, d->upsample == 2 */
int rcdm_conv3x3(const rcdm_conv3x3_desc* d, const void* in, const void* W, const float* bias,
                 const float* rowvec, const void* residual, void* out, void* workspace,
                 size_t workspace_bytes, void* stream);
/* conv3x3(in) + conv1x1(in2) accumulated in ONE implicit GEMM over K = 9 c_in + c_in2.   replaces the tail of
 *   ResnetBlock3D.forward for c_in != c_out, resnet.py:205-212: conv2(hidden_states) + conv_shortcut(input_tensor)
 *   (the sum is formed in the fp32 accumulators; the stand-alone shortcut launch, its f16 output and the residual read
 *   of it are gone).  in2: [rows of `out`][lda2] f16, c_in2 channels used — row m of in2 is output pixel m, so stride 1,
 *   upsample 0, no pad_after_only.  W f16 [c_out][9*c_in + c_in2]: the rcdm_conv3x3 layout followed by the 1x1 weight's
 *   [c_out][c_in2] columns; bias = the sum of the two biases.  c_in % 64 == 0 and c_in2 % 64 == 0 (whole k-steps).
 *   Everything else (epilogue flags, residual, split_k, workspace query with the same descriptor) as rcdm_conv3x3;
 *   rcdm_conv3x3 / rcdm_conv3x3_gnstat themselves return RCDM_EINVAL for a descriptor with c_in2 != 0. */
int rcdm_conv3x3_add1x1(const rcdm_conv3x3_desc* d, const void* in, const void* in2, const void* W, const float* bias,
                        const float* rowvec, const void* residual, void* out, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm (+SiLU).  replaces torch.nn.GroupNorm applied to the 5-D tensor — statistics over
 *   (C/groups, f, H, W) ACROSS frames — at resnet.py:185-186,196,202 and unet.py:455-456
 *   (samples = b, rows_per_sample = f*H*W), and the per-frame 4-D form at attention.py:328 and
 *   motion_module.py:162 (samples = b*f, rows_per_sample = H*W, eps 1e-6, no SiLU).
 *   Three launches: stats (deterministic fixed-order partials, no float atomics), finalize, apply; ONE launch when a
 *   sample has <= 512 rows (the 8x8 / per-frame 16x16 levels): a block owns a whole (sample, group bundle) slab.
 *   C % (2*groups) == 0, C % 8 == 0.  stats workspace: rcdm_groupnorm_workspace_bytes.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t samples, rows_per_sample, C, groups;
  int32_t ldx, ldy;
  float eps;
  int32_t silu;  /* 0 | 1 */
} rcdm_groupnorm_desc;

size_t rcdm_groupnorm_workspace_bytes(const rcdm_groupnorm_desc* d);
int rcdm_groupnorm_silu(const rcdm_groupnorm_desc* d, const void* x, const float* gamma,
                        const float* beta, void* y, void* workspace, size_t workspace_bytes,
                        void* stream);
/* A split-K GEMM / conv whose reduce pass ALSO leaves the partial statistics of the GroupNorm that reads its output next —
 * resnet.py:185-202 (conv1 -> norm2), attention.py:328 / motion_module.py:162 (the norm in front of a transformer / motion
 * module, fed by conv2 or by the feed-forward GEMM) — so that the norm needs no statistics launch and no extra read of the
 * tensor:  rcdm_gemm_gnstat / rcdm_conv3x3_gnstat(d, gn, ..., gn_workspace)  then  rcdm_groupnorm_silu_prestat(gn, out, ...,
 * gn_workspace).  `gn` describes the norm over exactly the rows the launch writes (samples * rows_per_sample == M, C == N,
 * ldx == ldc).  The statistics are taken from the STORED halfs in the order of the stand-alone pass: results are bit-identical
 * to rcdm_gemm / rcdm_conv3x3 followed by rcdm_groupnorm_silu.  *_gnstat_ok: 1 when the pair qualifies (the launch is split-K
 * — only then does a reduce pass exist —, no GEGLU / dup_rows / upsample == 2, the norm takes its three-launch form); the
 * calls return RCDM_ESHAPE otherwise.  gn_workspace: rcdm_groupnorm_workspace_bytes(gn), the same bytes passed on. */
int rcdm_gemm_gnstat_ok(const rcdm_gemm_desc* d, const rcdm_groupnorm_desc* gn);
int rcdm_gemm_gnstat(const rcdm_gemm_desc* d, const rcdm_groupnorm_desc* gn, const void* A, const void* W, const float* bias,
                     const float* rowvec, const void* residual, void* out, void* workspace, size_t workspace_bytes,
                     void* gn_workspace, size_t gn_workspace_bytes, void* stream);
int rcdm_conv3x3_gnstat_ok(const rcdm_conv3x3_desc* d, const rcdm_groupnorm_desc* gn);
int rcdm_conv3x3_gnstat(const rcdm_conv3x3_desc* d, const rcdm_groupnorm_desc* gn, const void* in, const void* W,
                        const float* bias, const float* rowvec, const void* residual, void* out, void* workspace,
                        size_t workspace_bytes, void* gn_workspace, size_t gn_workspace_bytes, void* stream);
int rcdm_conv3x3_add1x1_gnstat(const rcdm_conv3x3_desc* d, const rcdm_groupnorm_desc* gn, const void* in, const void* in2,
                               const void* W, const float* bias, const float* rowvec, const void* residual, void* out,
                               void* workspace, size_t workspace_bytes, void* gn_workspace, size_t gn_workspace_bytes,
                               void* stream);   /* rcdm_conv3x3_add1x1 with the statistics-carrying reduce (rcdm_conv3x3_gnstat_ok) */
/* finalize + apply only: the partial statistics are already in `workspace` (left there by a *_gnstat call with this
 * descriptor).  rcdm_groupnorm_prestat_ok: 1 when this descriptor's norm has a statistics pass to replace (three-launch
 * form; the smallest tensors run as one launch). */
int rcdm_groupnorm_prestat_ok(const rcdm_groupnorm_desc* d);
int rcdm_groupnorm_silu_prestat(const rcdm_groupnorm_desc* d, const void* x, const float* gamma, const float* beta, void* y,
                                void* workspace, size_t workspace_bytes, void* stream);
/* tuning / test switch: 1 = norms with >= 4 samples and <= 128 partial blocks per sample (the per-frame norms of the
 * transformers / motion modules) run as TWO launches — statistics, then an apply kernel whose blocks finalise their sample's
 * groups themselves (bit-identical to the three-launch form) — 0 = always statistics / finalize / apply, -1 = default
 * (0: the two-launch form measured 0.05 ms per step slower; environment RCDM_GN_FOLD=1 sets 1). */
int rcdm_set_groupnorm_fold(int32_t on);
/* statistics only: stat[sample][group][2] = (mean, 1 / sqrt(var + eps)), fp32 — the first two launches of the three-launch
 * form, for a consumer that applies the normalisation itself (rcdm_rowchain's gn_stat: the norm in front of proj_in,
 * attention.py:328-330, motion_module.py:162-166).  ldy / silu of the descriptor are ignored; workspace as above. */
int rcdm_groupnorm_stats(const rcdm_groupnorm_desc* d, const void* x, float* stat, void* workspace,
                         size_t workspace_bytes, void* stream);
/* the same when the partial statistics are already in `workspace` (left there by a *_gnstat call with this descriptor, as for
 * rcdm_groupnorm_silu_prestat): the finalize launch only.  RCDM_ESHAPE when the descriptor's norm has no three-launch form. */
int rcdm_groupnorm_stats_prestat(const rcdm_groupnorm_desc* d, float* stat, void* workspace, size_t workspace_bytes,
                                 void* stream);
/* finalize only, on partials a producer wrote in its OWN geometry: partial[samples][groups][splits][3] = (count, mean, M2)
 * fp32 -> stat[samples][groups][2] = (mean, 1 / sqrt(var + eps)); the fixed-order combination of the three-launch form.
 * Used behind rcdm_conv3x3_wino(gn_out, ...), whose output transform leaves one partial per (sample, group, 2x2 tile). */
int rcdm_groupnorm_finalize(int32_t samples, int32_t groups, int32_t splits, float eps, const float* partial, float* stat,
                            void* stream);
/* the apply launch alone: y = (x - mean) rstd gamma + beta (+ SiLU) with (mean, rstd) = stat[sample][group] already final
 * (rcdm_groupnorm_finalize / rcdm_groupnorm_stats) — the third launch of the three-launch form. */
int rcdm_groupnorm_apply(const rcdm_groupnorm_desc* d, const void* x, const float* stat, const float* gamma, const float* beta,
                         void* y, void* stream);

/* conv3x3, stride 1, padding 1, as Winograd F(2x2, 3x3) (wino.hip; round 6): the same operation as rcdm_conv3x3 /
 *   rcdm_conv3x3_add1x1 — ResnetBlock3D's conv1 / conv2 (+ conv_shortcut), resnet.py:188,205-212 — with 4/9 of the
 *   multiply-adds, for the levels where the implicit GEMM is a chain of latencies (the 16x16 / 8x8 latents: few rows, wide
 *   channels).  Three launches: input transform B^T d B (fp32 arithmetic on the f16 pixels, one rounding), 16 (+ 4) batched
 *   160x160-tile GEMMs with fp32 results in the workspace, output transform A^T M A + epilogue in fp32 (one rounding).
 *   `d` is the rcdm_conv3x3_desc of the equivalent call: stride 1, upsample 0, no pad_after_only / dup_rows, h_in and w_in
 *   even, c_in % 64 == 0 (c_in2 % 64 == 0), epilogue bits BIAS | ROWVEC | RESIDUAL; split_k 0 = heuristic (one resident
 *   round of the chip).  U = rcdm_pack_conv3x3_wino(w): f16 [16][c_out][c_in], position p = 4 i + j holds (G g G^T)[i][j]
 *   formed in fp32 from the fp32 weights and rounded once.  in2 / W2 (both or neither; d->c_in2, d->lda2): the second input
 *   of rcdm_conv3x3_add1x1 and its PLAIN f16 [c_out][c_in2] 1x1 weight (rcdm_pack_f16) — it needs no transform, its four
 *   output-parity GEMMs ride as four more batch entries; bias = the sum of the two biases.
 *   gn != NULL: `in` is the RAW input of a GroupNorm (+ SiLU when gn->silu) whose (mean, rstd) pairs are in gn_stat
 *   ([samples][groups][2] fp32, rcdm_groupnorm_stats) and the input transform applies x * rstd * gamma + (beta - mean * rstd *
 *   gamma) (+ SiLU) on the way — resnet.py:185-186 / :202 in front of conv1 / conv2 — before the zero padding; gn->C == c_in,
 *   gn->samples * gn->rows_per_sample == n_img * h_in * w_in; ldx / ldy / eps of gn are not used here.
 *   Numerics: the MFMA operands are f16(B^T d B) (|.| <= 4 max|d|) and f16(G g G^T): about twice the operand-rounding
 *   noise of rcdm_conv3x3 (which multiplies the f16 pixels and f16 weights themselves); accumulation, both transforms and the
 *   epilogue are fp32.  Results are NOT bit-identical to rcdm_conv3x3.  rcdm_conv3x3_wino_supported: 1 | 0.
 *   gn_out != NULL: the GroupNorm that reads `out` NEXT (over exactly the rows written: gn_out->C == c_out, samples *
 *   rows_per_sample == n_img * h_in * w_in); the output transform then also leaves its partial statistics, taken from the stored
 *   halfs, in gn_out_partial — fp32 [samples][groups][rows_per_sample / 4][3] = (count, mean, M2), one per 2x2 tile — for
 *   rcdm_groupnorm_finalize(samples, groups, rows_per_sample / 4, eps, ...): that norm needs no statistics pass (c_out <= 8192). */
int rcdm_conv3x3_wino_supported(const rcdm_conv3x3_desc* d);
size_t rcdm_conv3x3_wino_workspace_bytes(const rcdm_conv3x3_desc* d);
int rcdm_conv3x3_wino_plan_query(const rcdm_conv3x3_desc* d, int32_t* out8);   /* as rcdm_gemm_plan_query; variant 11, column tiles x entries */
int rcdm_pack_conv3x3_wino(const float* w, int32_t c_out, int32_t c_in, void* dst, void* stream);
/* Upsample3D.forward (resnet.py:60-79: F.interpolate(nearest, x2) + conv3x3) as ONE rcdm_gemm + this gather (round 6): on the
 * upsampled grid output pixel (Y, X) = sum over the nine taps of w[ky][kx] . s((Y + ky - 1) >> 1, (X + kx - 1) >> 1), so every
 * product is a tap's 1x1 image of a SOURCE pixel — nine products per source pixel and channel pair (the four-phase form of
 * rcdm_conv3x3 upsample = 2 needs sixteen).  P = rcdm_gemm(source rows [n_img*h*w][c_in], W9) with W9 = f16 [9*c_out][c_in],
 * row tap*c_out + c = weight[c][:][ky][kx] (tap = 3 ky + kx; the ordinary rounded weights, no sums), no bias: f16
 * [n_img*h*w][ldp >= 9 c_out].  The gather adds, per output pixel of the (2h x 2w) image, the nine planes' values of the source
 * pixels its taps land on (taps outside the upsampled image skipped = its zero padding) + bias[c_out] (may be NULL), fp32 sum,
 * one rounding: out f16 [n_img*4*h*w][ldc].  Numerics: the nine products are rounded to f16 before the sum (one more rounding
 * per term than the implicit GEMM's fp32 accumulation). */
int rcdm_upsample_taps_gather(const void* P, int32_t ldp, int32_t n_img, int32_t h, int32_t w, int32_t c_out, const float* bias,
                              void* out, int32_t ldc, void* stream);
/* the same gather with upsample = 0: a plain stride-1, padding-1 conv3x3 as rcdm_gemm (N = 9 c_out tap planes over its own
 * pixels) + gather — no multiply saved, but a conv with very few output channels (unet.py:457 conv_out: 320 -> 4) becomes a
 * plain GEMM whose N is 9x wider than the conv's, instead of an implicit GEMM that pads 4 channels to a 64-wide tile.
 * upsample = 1 is rcdm_upsample_taps_gather. */
int rcdm_conv_taps_gather(const void* P, int32_t ldp, int32_t n_img, int32_t h, int32_t w, int32_t c_out, int32_t upsample,
                          const float* bias, void* out, int32_t ldc, void* stream);
/* tuning / test switch: 1 = the batched GEMM of rcdm_conv3x3_wino leaves f16 slabs (half the bytes between it and the output
 * transform, whole-row stores; every transform-domain sum is rounded to f16 before A^T M A), 0 = fp32 slabs, -1 = default
 * (environment RCDM_WINO_SLAB16, else 1). */
int rcdm_set_wino_slab_f16(int32_t on);
int rcdm_conv3x3_wino(const rcdm_conv3x3_desc* d, const rcdm_groupnorm_desc* gn, const float* gn_stat,
                      const float* gn_gamma, const float* gn_beta, const void* in, const void* in2, const void* U,
                      const void* W2, const float* bias, const float* rowvec, const void* residual, void* out,
                      void* workspace, size_t workspace_bytes, const rcdm_groupnorm_desc* gn_out, float* gn_out_partial,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (eps 1e-5, torch default) with optional fused positional-encoding add
 *   replaces nn.LayerNorm at attention.py:482,502,514 and motion_module.py:236,243; with pe != NULL
 *   also PositionalEncoding.forward motion_module.py:265-267 applied after the "(b f) d c -> (b d) f c"
 *   regroup (:299-302): y[m] += pe[(m / rows_per_frame) % frames].   C % 8 == 0, C <= 2048.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t M, C, ldx, ldy;
  float eps;
  int32_t rows_per_frame, frames; /* only with pe */
} rcdm_layernorm_desc;

int rcdm_layernorm(const rcdm_layernorm_desc* d, const void* x, const float* gamma,
                   const float* beta, const float* pe, void* y, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Flash-style multi-head attention, softmax(scale * Q K^T) V, no mask, online softmax in fp32.
 *   replaces CrossAttention._attention attention.py:170-199 (+ reshape_heads_to_batch_dim :93-105):
 *   self-attention over the hw latent patches of one frame and cross-attention over the L_text
 *   context rows of that frame.  Q rows [batch*Lq][ldq] with head h at columns [h*d, (h+1)*d);
 *   K,V rows [batch*Lk][ldk|ldv]; out [batch*Lq][ldo].  d % 8 == 0, d <= 160.
 *   Range: scaled scores |scale * log2(e) * q.k| < 2^15 (the d = 40, Lk >= 256 kernel keeps its running max as an f16 inside
 *   the Q fragment and re-rounds Q * scale * log2(e) to f16: relative error ln2 * 2^-12 * |scaled score| on a probability).
 *   A caller that cannot bound its scores below that sets RCDM_ATTN_WIDE_RANGE in `flags` (or the environment sets
 *   RCDM_ATTN_MSUB=0): the fp32-fma softmax kernel, which has neither limit.  rcdms_amd/engine.py sets it from a
 *   data-independent bound on |q| |k| (LayerNorm output norm x Frobenius norms of the folded per-head weights).
 * ---------------------------------------------------------------------------------------------- */
#define RCDM_ATTN_WIDE_RANGE 1
typedef struct {
  int32_t batch, heads, Lq, Lk, d;
  int32_t ldq, ldk, ldv, ldo;
  float scale;
  int32_t flags;              /* 0, or RCDM_ATTN_WIDE_RANGE */
} rcdm_attn_desc;

int rcdm_flash_attn(const rcdm_attn_desc* d, const void* Q, const void* K, const void* V, void* out,
                    void* stream);
/* the same with a mask: key k of batch b is visible to query q iff key_valid[b*Lk + k] != 0 (key_valid may be NULL =
 * all valid) and, with causal != 0, k <= q.  Replaces the additive attention_mask of the stage-1 prior transformer
 * (myprior_transformer.py:389-393: (1 - text_mask) * -10000 padded, + causal_attention_mask :250-256), whose masked
 * probabilities are exactly 0 in fp32.  A query with no visible key yields a zero row. */
int rcdm_flash_attn_masked(const rcdm_attn_desc* d, const void* Q, const void* K, const void* V,
                           const unsigned char* key_valid, int32_t causal, void* out, void* stream);

/* Cross-attention with a short key sequence (Lk <= 96): CrossAttention.forward with encoder_hidden_states,
 * attention.py:139-168, at the 85 / 91 context rows of the stage-2 UNet.  A wave holds all scores of 32 queries of one
 * head in registers (no key loop, no online rescale); a block of 8 waves shares one head's operands through LDS.  K and V are passed as the per-context FRAGMENT-MAJOR image rcdm_xattn_pack_kv
 * writes once per context (they do not change over the denoising steps): every MFMA operand fragment is one contiguous
 * 1-KiB block in lane order; rcdm_xattn_image_bytes sizes it.  `d` is the rcdm_attn_desc of the equivalent
 * rcdm_flash_attn call (ldk / ldv unused by rcdm_xattn); results agree with rcdm_flash_attn to f16 rounding. */
size_t rcdm_xattn_image_bytes(int32_t batch, int32_t heads, int32_t d);
int rcdm_xattn_pack_kv(const void* K, const void* V, int32_t batch, int32_t Lk, int32_t heads, int32_t d, int32_t ldk,
                       int32_t ldv, void* image, void* stream);
int rcdm_xattn(const rcdm_attn_desc* d, const void* Q, const void* image, void* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Temporal self-attention over the f frames of every (sample, pixel, head).
 *   replaces VersatileAttention.forward motion_module.py:294-354 between to_q/k/v and to_out: the
 *   "(b f) d c -> (b d) f c" regroup, 5x5 softmax(QK^T*scale)V and the inverse regroup become index
 *   arithmetic.  qkv rows ordered (b, f, pixel), [q | k | v] at columns [0,C) [C,2C) [2C,3C).
 *   frames <= 8, d % 8 == 0.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t samples, frames, pixels, heads, d;
  int32_t ldqkv, ldo;
  float scale;
} rcdm_temporal_attn_desc;

int rcdm_temporal_attn(const rcdm_temporal_attn_desc* d, const void* qkv, void* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused feed-forward, row-stationary (rowff.hip):  out[m][:] = x[m][:] + FF_geglu(LayerNorm(x[m][:]))  in one launch.
 *   replaces the chain nn.LayerNorm -> diffusers FeedForward(dim, activation_fn="geglu") -> + hidden_states at
 *   attention.py:514 (norm3 / ff of BasicTransformerBlock.forward) and motion_module.py:243 (ff_norm / ff of
 *   TemporalTransformerBlock.forward): Linear(C -> 8C) -> split (hidden, gate) -> hidden * gelu(gate) (exact erf
 *   GELU) -> Linear(4C -> C) -> + bias -> + x.  The 4C-wide hidden tensor is never written: a wave keeps its 16 token
 *   rows and their fp32 output accumulator in registers and streams the weights through LDS.
 *   Weights come as the fragment-major stream rcdm_pack_ff_stream writes once (12 C^2 halfs, rcdm_ff_stream_bytes):
 *   w1 = ff.net.0.proj.weight fp32 [8C][C], b1 = ff.net.0.proj.bias [8C], w2 = ff.net.2.weight fp32 [C][4C];
 *   b1_packed [8C] floats.  b2 = ff.net.2.bias [C].  out may alias x.  Supported C: rcdm_ff_fused_supported.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t M, C;
  int32_t ldx, ldo;   /* row strides in elements, multiples of 8 */
  float eps;          /* LayerNorm eps (1e-5) */
} rcdm_ff_desc;
size_t rcdm_ff_stream_bytes(int32_t C);
int rcdm_ff_fused_supported(int32_t C);
int rcdm_pack_ff_stream(const float* w1, const float* b1, const float* w2, int32_t C, void* wstream, float* b1_packed,
                        void* stream);
int rcdm_ff_fused(const rcdm_ff_desc* d, const void* x, const float* ln_gamma, const float* ln_beta, const void* wstream,
                  const float* b1_packed, const float* b2, void* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row-stationary chain (rowff.hip), one launch:
 *     tok[m][:] = a_in[m][:] W_a^T + a_bias (+ res[m][:])                  -> stored (f16 [M][ldt])
 *     y[m][:]   = LayerNorm(tok[m][:]) * gamma + beta (+ pe[(m / rows_per_frame) % frames][:])
 *     tail 1 / 3:  out[m][0 : tail*C] = y[m][:] W_t^T          (no bias; W_t = fp32 [tail*C][C], e.g. [to_q; to_k; to_v])
 *     tail 0:      out[m][:] = tok[m][:] + FF_geglu(y[m][:])   (as rcdm_ff_fused; may be written in place of tok)
 *     tail 2:      out[m][:] = z_res[m][:] + ((tok[m][:] + FF_geglu(y[m][:])) W_z^T + z_bias)   — the block's proj_out and the
 *                  residual add of Transformer3DModel / TemporalTransformer3DModel.forward (attention.py:352-363,
 *                  motion_module.py:170-176) ride too; the feed-forward's output rows are NOT stored (tok keeps the rows
 *                  stage A wrote).  Needs res, no pe.  W_z = fp32 [C][C], passed to rcdm_pack_rowchain as wt.
 *   replaces, in BasicTransformerBlock.forward / Transformer3DModel.forward (attention.py:330,479-514) and
 *   TemporalTransformer3DModel / TemporalTransformerBlock.forward (motion_module.py:166,234-243,299-302):
 *     proj_in -> norm1 -> [to_q | to_k | to_v]                         (res = NULL, tail 3)
 *     attn1.to_out[0] + hidden_states -> norm2 -> attn2.to_q           (res = tok, tail 1)
 *     attention_blocks[i].to_out[0] + hidden_states -> norms[i + 1] + pos_encoder -> [to_q | to_k | to_v]   (tail 3, pe)
 *     attn2.to_out[0] / attention_blocks[-1].to_out[0] + hidden_states -> norm3 / ff_norm -> ff -> + hidden_states  (tail 0)
 *   W_a = fp32 [C][C] (nn.Linear layout).  The weights come as one fragment-major stream (rcdm_pack_rowchain,
 *   rcdm_rowchain_stream_bytes); b1_packed / b2 only with tail 0.  tok may alias res, out may alias tok (tail 0) — a block
 *   reads and writes only its own rows.  Supported C: rcdm_rowchain_supported; supported (C, tail, pe) combinations:
 *   rcdm_rowchain_config_supported — pe (with rows_per_frame, frames <= 8) rides with tail 1 / 3 only: with the feed-forward
 *   tails the per-block parameter region (up to 19 C floats) would not fit the 160 KB of LDS (RCDM_ESHAPE).
 *   Preconditions (RCDM_EINVAL / RCDM_ESHAPE when violated): every pointer 16-byte aligned (the fp32 vectors a_bias, ln_*,
 *   pe, b1_packed, b2, z_bias, gn_* are fetched as float4, the rows as 16-byte pieces), and M * ld * 2 < 2^31 for every row
 *   operand (32-bit byte offsets into 2-GB buffer resources).  The same two rules hold for rcdm_ff_fused.
 *   gn_stat != NULL (only with res == NULL): a_in is the RAW input of the GroupNorm in front of proj_in and the kernel
 *   applies  a_in[m][c] * rstd * gn_gamma[c] + (gn_beta[c] - mean * rstd * gn_gamma[c])  (rounded to f16 like the separate
 *   launch) with (mean, rstd) = gn_stat[m / gn_rows][c / (C / gn_groups)] from rcdm_groupnorm_stats; gn_rows % 16 == 0,
 *   gn_rows >= 160.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t M, C;
  int32_t lda, ldr, ldt, ldo;      /* row strides in elements, multiples of 8 (ldr only with res) */
  int32_t tail;                    /* 0 = feed-forward, 1 = GEMM to C columns, 2 = feed-forward + projection, 3 = GEMM to 3C */
  int32_t rows_per_frame, frames;  /* only with pe */
  float eps;                       /* LayerNorm eps (1e-5) */
  int32_t gn_groups, gn_rows;      /* only with gn_stat: groups and rows per sample of the GroupNorm */
  int32_t ldz;                     /* row stride of z_res (tail 2) */
} rcdm_rowchain_desc;
int rcdm_rowchain_supported(int32_t C);
int rcdm_rowchain_config_supported(int32_t C, int32_t tail, int32_t pe_frames);   /* pe_frames = 0: no pe */
size_t rcdm_rowchain_stream_bytes(int32_t C, int32_t tail);
/* wa [C][C]; tail 1 / 3: wt [tail*C][C]; tail 0 / 2: w1 [8C][C], b1 [8C], w2 [C][4C] and b1_packed [8C] (out); tail 2: wt [C][C] */
int rcdm_pack_rowchain(const float* wa, int32_t C, int32_t tail, const float* wt, const float* w1, const float* b1,
                       const float* w2, void* wstream, float* b1_packed, void* stream);
int rcdm_rowchain(const rcdm_rowchain_desc* d, const void* a_in, const void* res, void* tok, const float* a_bias,
                  const float* ln_gamma, const float* ln_beta, const float* pe, const void* wstream, const float* b1_packed,
                  const float* b2, void* out, const float* gn_stat, const float* gn_gamma, const float* gn_beta,
                  const void* z_res, const float* z_bias, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row softmax: y[m][n] = softmax over n of (scale * x[m][n]); f16 rows, fp32 math, N % 8 == 0, N <= 4096, scale > 0.
 *   With two rcdm_gemm calls (scores = Q K^T, out = P V^T^T) it is the attention of heads too wide for
 *   rcdm_flash_attn: the single 512-channel head of the SD-1.5 VAE mid block (diffusers 0.24.0 AutoencoderKL,
 *   called at RCDMs_pipeline.py:281,429 — SURVEY §8f N3).
 * ---------------------------------------------------------------------------------------------- */
int rcdm_softmax_rows(int32_t M, int32_t N, int32_t ldx, int32_t ldy, float scale, const void* x, void* y, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Timestep embedding: diffusers 0.24.0 Timesteps(dim, flip_sin_to_cos=True, freq_shift=0) as used
 *   at unet.py:100,383: out[r][0:half] = cos(t_r*w_i), out[r][half:] = sin(t_r*w_i),
 *   w_i = exp(-ln(10000)*i/half).  t fp32 device array [rows]; out fp32 [rows][dim].
 * ---------------------------------------------------------------------------------------------- */
int rcdm_timestep_embed(const float* t, int32_t rows, int32_t dim, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Tiny-M fp32 linear: out[r][n] = act_out( sum_k act_in(x[r][k]) * W[n][k] + bias[n] ), rows <= 8.
 *   replaces TimestepEmbedding (linear_1 -> SiLU -> linear_2, unet.py:103,389) and every
 *   ResnetBlock3D.time_emb_proj(silu(temb)) (resnet.py:191), all 22 batched as one N = sum(Cout).
 *   W f16 [N][K]; silu_in / silu_out are 0|1.  K % 8 == 0.
 * ---------------------------------------------------------------------------------------------- */
int rcdm_small_linear(const float* x, int32_t rows, int32_t K, const void* W, const float* bias,
                      int32_t N, int32_t silu_in, int32_t silu_out, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Assemble the UNet input rows from the sampling-loop state: replaces
 *   torch.cat([latents]*2) + cat([x, mask, masked_latents], dim=1) RCDMs_pipeline.py:482-486 and the
 *   NCFHW->channels-last f16 conversion.  latents fp32 (S,4,f,H,W); mask fp32 (reps*S,1,f,H,W);
 *   masked fp32 (reps*S,4,f,H,W); reps = 2 with CFG (uncond copies first), 1 without.
 *   out rows [(reps*S)*f*H*W][ld] f16, channels 0..8 written, 9..c_pad-1 zeroed.
 * ---------------------------------------------------------------------------------------------- */
int rcdm_assemble_input(const float* latents, const float* mask, const float* masked, int32_t S,
                        int32_t reps, int32_t frames, int32_t H, int32_t W, void* out, int32_t ld,
                        int32_t c_pad, void* stream);

/* generic layout converters for UNet3DConditionModel.forward called directly with a 5-D tensor
 * (unet.py:322-330): fp32 (b,C,f,H,W) -> f16 rows [b*f*H*W][ld] (channels >= C zeroed up to c_pad)
 * and f16 rows -> fp32 (b,C,f,H,W). */
int rcdm_ncfhw_to_rows(const float* x, int32_t b, int32_t C, int32_t frames, int32_t H, int32_t W,
                       void* out, int32_t ld, int32_t c_pad, void* stream);
int rcdm_rows_to_ncfhw(const void* rows, int32_t ld, int32_t b, int32_t C, int32_t frames, int32_t H,
                       int32_t W, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused classifier-free guidance + DDIM step (eta = 0, epsilon prediction), in place on `latents`:
 *   replaces noise_pred.chunk(2); eps = eps_u + s*(eps_c - eps_u) RCDMs_pipeline.py:492-494 and
 *   diffusers 0.24.0 DDIMScheduler.step (:497):  x0 = (x - sqrt(1-a_t) eps)/sqrt(a_t);
 *   x' = sqrt(a_prev) x0 + sqrt(1-a_prev) eps.
 *   eps rows f16 [(reps*S)*f*H*W][ld] (uncond sample block first); latents fp32 (S,4,f,H,W).
 *   coef: device fp32 table [n_steps][4] = {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)};
 *   step_counter: device int32; the kernel uses row coef[*step_counter] (the counter is advanced by
 *   rcdm_advance_step so a captured graph replays with no host-side parameter change).
 * ---------------------------------------------------------------------------------------------- */
int rcdm_cfg_ddim_step(const void* eps, int32_t ld, float* latents, int32_t S, int32_t reps,
                       int32_t frames, int32_t H, int32_t W, float guidance_scale, const float* coef,
                       const int32_t* step_counter, void* stream);
/* Fused classifier-free guidance + one PLMS step of diffusers 0.24.0 PNDMScheduler(skip_prk_steps=True) — the other scheduler
 *   type RCDMsPipeline's constructor accepts (RCDMs_pipeline.py:72-79), stepped at :497 — in place on `latents`:
 *   e = eps_u + s (eps_c - eps_u); e' = the linear-multistep combination of e with up to three stored predictions;
 *   x' = sqrt(a'/a) x - (a' - a) e' / (a sqrt(1 - a') + sqrt(a (1 - a) a')).
 *   table: device fp32 [n_calls][12], one row per model evaluation (n_calls = num_inference_steps + 1: the second
 *   timestep is evaluated twice), as rcdms_amd.scheduler.PNDMScheduler.plms_table() lays it out:
 *   (a, b, w_now, w1, w2, w3, slot_now, s1, s2, s3, mode, 0); history: device fp32 [5][S*4*f*H*W], caller-owned, carries
 *   the stored predictions (slots 0..3) and the first sample (slot 4) between calls; no initialisation needed.
 *   step_counter as in rcdm_cfg_ddim_step. */
int rcdm_cfg_pndm_step(const void* eps, int32_t ld, float* latents, float* history, int32_t S, int32_t reps,
                       int32_t frames, int32_t H, int32_t W, float guidance_scale, const float* table,
                       const int32_t* step_counter, void* stream);
/* Stage-1 prior, per step of prior_pipeline.py:311-344.
 * rcdm_prior_assemble: tok[(b, l)] (f16, B*L rows of C) <- base rows, except l == time_row <- temb (one fp32 row of C);
 *   x16[b] (f16, E) <- latents[b % n_lat] (fp32): the `torch.cat([latents] * 2)` of :314 and the sequence concat of
 *   myprior_transformer.py:363-387 without re-projecting what does not change between steps.
 * rcdm_cfg_unclip_step: x0 = clamp(u + s (c - u), +-clip_range) with u / c = rows r / n + r of pred (f16, ld);
 *   latents[r] = coef[*step][0] x0 + coef[*step][1] latents[r] + coef[*step][2] noise[*step][r]  — the CFG combine
 *   (:328-333) fused with diffusers 0.24.0 UnCLIPScheduler.step for prediction_type "sample" / "fixed_small_log";
 *   coef fp32 [n_steps][3]; noise fp32 [n_steps][n][E] or NULL; clip_range <= 0 disables the clamp. */
int rcdm_prior_assemble(const void* base, const float* temb, const float* latents, int32_t n_lat, void* tok, void* x16,
                        int32_t B, int32_t L, int32_t C, int32_t E, int32_t time_row, void* stream);
int rcdm_cfg_unclip_step(const void* pred, int32_t ld, float* latents, int32_t n, int32_t reps, int32_t E,
                         float guidance_scale, float clip_range, const float* coef, const float* noise,
                         const int32_t* step_counter, void* stream);
/* t_out[0..rows) = timesteps[*step_counter] (fp32) — feeds rcdm_timestep_embed inside a graph */
int rcdm_load_timestep(const float* timesteps, const int32_t* step_counter, float* t_out,
                       int32_t rows, void* stream);
int rcdm_advance_step(int32_t* step_counter, void* stream);
/* dst[0..row_floats) = table[*step_counter][0..row_floats) (fp32): the per-step row of a table computed once per
 * schedule — the whole timestep-embedding chain of unet.py:381-389 and the 22 time_emb_proj outputs (resnet.py:191)
 * depend on the timestep only, so the sampling loop evaluates them for all T steps when the schedule is set and a
 * captured step starts with this one copy instead of five serial launches.  row_floats % 4 == 0, 16-byte aligned. */
int rcdm_load_table_row(const float* table, const int32_t* step_counter, float* dst, size_t row_floats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Weight repacking (fp32 reference layout -> f16 kernel layout), device to device:
 *   rcdm_pack_f16: elementwise fp32 -> f16.
 *   rcdm_pack_conv3x3: torch (Cout,Cin,3,3) fp32 -> f16 [Cout][9*cin_pad], k = tap*cin_pad + c.
 *   rcdm_pack_conv3x3_up2: torch (Cout,Cin,3,3) fp32 -> f16 [4 phases][Cout][4*Cin] for rcdm_conv3x3 with upsample = 2:
 *   phase = 2a + b (output parity), k = (2r + c)*Cin + ci, value = sum of w[co][ci][ky][kx] over the taps with
 *   (a+ky-1)>>1 == a-1+r and (b+kx-1)>>1 == b-1+c.
 *   rcdm_pack_geglu_rows: FeedForward.net.0.proj weight (8C,K)/bias(8C) -> rows reordered so every
 *   32-row group holds 16 "hidden" rows then the matching 16 "gate" rows (RCDM_EPI_GEGLU): the two
 *   land in the same lane of the 16x16x32 and of the 32x32x16 MFMA accumulator layouts.
 * ---------------------------------------------------------------------------------------------- */
int rcdm_pack_f16(const float* src, void* dst, size_t n, void* stream);
int rcdm_pack_conv3x3(const float* w, int32_t c_out, int32_t c_in, int32_t cin_pad, void* dst,
                      void* stream);
int rcdm_pack_conv3x3_up2(const float* w, int32_t c_out, int32_t c_in, void* dst, void* stream);
/* C[n][m] = A[n][k] B[k][m], fp32 row-major, device to device: composition of two linear maps at pack time (proj_out behind
 * ff.net.2, the context stacks' projections) before the product is rounded to f16.  Fixed summation order; not a hot-path call. */
int rcdm_matmul_f32(const float* A, const float* B, float* C, int32_t n, int32_t k, int32_t m, void* stream);
int rcdm_pack_geglu_rows(const float* w, const float* bias, int32_t n_out /*8C*/, int32_t K,
                         void* w_dst, float* bias_dst, void* stream);

/* Mish activation, fp32 elementwise: y = x * tanh(softplus(x)).  Replaces `Mish.forward`
 * (src/models/resnet.py:215-217); never instantiated by configs/testing.yaml, kept for drop-in completeness. */
int rcdm_mish(const float* x, float* y, size_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * hipGraph plumbing: capture the ~10^3 launches of one denoising step once, replay per step.
 * ---------------------------------------------------------------------------------------------- */
int rcdm_graph_begin_capture(void* stream);
int rcdm_graph_end_capture(void* stream, void** graph_exec_out);
int rcdm_graph_launch(void* graph_exec, void* stream);
int rcdm_graph_destroy(void* graph_exec);

/* HIP-event timing on an explicit stream (bench.py roofline leg) */
int rcdm_event_create(void** ev_out);
int rcdm_event_record(void* ev, void* stream);
int rcdm_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out); /* synchronises on stop */
int rcdm_event_destroy(void* ev);
int rcdm_stream_synchronize(void* stream);

/* ------------------------------------------------------------------------------------------------
 * Thin RCCL layer (librccl is dlopen'ed on first use; a single-GPU process never maps it).  The reference has no
 * inter-GPU communication at all: stage2_batchtest_rcdms_model.py:457-468 spawns one process per device and every
 * process loads every checkpoint.  These serve the packed-weight broadcast and the CFG-split latency mode (two GPUs
 * per story, one classifier-free-guidance half each, RCDMs_pipeline.py:482-497: the halves' noise predictions are
 * all-gathered inside the step graph).  Byte-typed, in place on the caller's HIP stream, graph-capturable.
 *   rcdm_comm_unique_id: rank 0 of a group fills 128 bytes, the caller ships them to the other ranks out of band
 *                        (file, socket, torch.distributed object broadcast);
 *   rcdm_comm_create:    collective over the group's ranks (blocks until all have called it); the communicator is bound
 *                        to the HIP device that is current at this call, and later calls from a thread whose current
 *                        device differs are refused with RCDM_EINVAL;
 *   rcdm_bcast:          `bytes` at `buf` from rank `root` (0 <= root < nranks, else RCDM_EINVAL) to every rank's `buf`;
 *   rcdm_allgather:      recv[r * bytes_per_rank ...] on every rank = rank r's send (send may alias its own slot).
 * ---------------------------------------------------------------------------------------------- */
int rcdm_comm_unique_id(void* id128);
int rcdm_comm_create(const void* id128, int32_t nranks, int32_t rank, void** comm_out);
int rcdm_comm_destroy(void* comm);
int rcdm_bcast(void* comm, void* buf, size_t bytes, int32_t root, void* stream);
int rcdm_allgather(void* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream);
int rcdm_comm_last_error(void); /* last non-zero ncclResult_t seen on this thread */

#ifdef __cplusplus
}
#endif
#endif /* RCDM_H */
