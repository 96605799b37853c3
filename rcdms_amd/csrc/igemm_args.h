// igemm_args.h — launch arguments shared by the two implicit-GEMM kernel families of librcdm_hip.so:
// igemm.hip (igemm_dma_kernel: 4-wave 128x128 / 64x64 / 128x64 tiles and the 8-wave 256x256 one-barrier loop) and
// igemm8.hip (igemm_pp_kernel: 8-wave ping-pong loop, 160x320 / 160x256 / 256x256 tiles).
#pragma once
#include "common.h"

constexpr int BK = 64;
#ifndef RCDM_PRIO_LOADS
#define RCDM_PRIO_LOADS 1   // 0: round-2 priorities (MFMA runs of the ping-pong kernel at s_setprio 1, nothing else raised), for A/B builds
#endif

struct IgemmArgs {
  const f16* A;
  const f16* W;
  const float* bias;
  const float* rowvec;
  const f16* res;
  f16* out;
  float* partial;
  int M, N, Cin, Ktot;
  int Hi, Wi, Ho, Wo, stride, up;
  int pad;  // rows / columns of zero padding before the image: 1, or 0 for the pad-after-only form
  int lda, ldc, ldr, ldt, rows_per_sample;
  int epi;
  float out_scale;
  long long dup;     // element offset of the second copy of every output row (0 = none): dup_rows * ldc
  int tilesM, tilesN, kc, nk, splits, nk_per_split;
  long long* trace;  // debug: per-block s_memtime stamps (rcdm_debug_set_igemm_trace), normally null
  int dbg;           // ping-pong loop switches: 8 = rotate the k order per block (RCDM_PP_ROTATE, default on)
  // fused LayerNorm of the output rows (rcdm_gemm_ln; kEpiLN in epi, the 160x320 ping-pong tile only)
  const float* ln_g;
  const float* ln_b;
  const float* ln_pe;
  f16* ln_out;
  int ln_ld, ln_rpf, ln_frames;
  float ln_eps;
  // deferred LayerNorm (rcdm_gemm_lnx).  Producer side: stat_out[slot][m] = (sum, sum of squares) of the f16 values
  // this launch stores to row m, one slot per column tile (stat_parts == tilesN).  Consumer side: the A rows are the RAW
  // input x of a LayerNorm whose gamma is folded into W and whose beta into the bias, so LayerNorm(x) W^T =
  // rstd (x W'^T) - rstd mean S, S[n] = sum_c W'[n][c]: lnx_stat holds the producer's partials of the A rows, the epilogue
  // forms (rstd, mean rstd) per row and applies them before bias / row vector / GEGLU / residual.
  float* stat_out;
  int stat_parts, stat_ld;      // slot-major: stat_out[slot][stat_ld rows][2]
  const float* lnx_stat;
  const float* lnx_S;
  int lnx_parts, lnx_ld;
  float lnx_invC, lnx_eps;
  // nearest-2x upsample + conv3x3 as four 2x2 PHASE convolutions over the SOURCE grid (rcdm_conv3x3, upsample = 2; the
  // ping-pong kernel with TAPS = 4): output pixel (2y + a, 2x + b) sees only the 2x2 source pixels (y + a - 1 + r,
  // x + b - 1 + c), r, c in {0, 1}, each with the SUM of the 3x3 taps that land on it.  Virtual row m = phase * ph_rows +
  // source pixel (phase = 2a + b; a tile lies inside one phase: ph_rows % BM == 0); W = [phase][N][4 taps][Cin]; the
  // epilogue stores virtual row m to output row img * 4HW + (2y + a) * 2W + 2x + b.  0 = not a phase launch.
  int ph_rows;
  // split-K launches whose reduce pass also leaves the GroupNorm partial statistics of the rows it writes (rcdm_gemm_gnstat /
  // rcdm_conv3x3_gnstat; splitk_reduce_gn_kernel): the statistics geometry of gn_plan.h for the norm that reads `out` next
  float* gn_partial;            // null: plain reduce
  int gn_samples, gn_P, gn_G, gn_cg, gn_CH, gn_RPB, gn_splits, gn_rps;
  // conv3x3 + a 1x1 convolution of a SECOND input accumulated into the same output (rcdm_conv3x3_add1x1: ResnetBlock3D's
  // conv2(...) + conv_shortcut(input_tensor), resnet.py:205-212, as one implicit GEMM over K = 9 Cin + Cin2): the k-steps
  // [nk1, nk) read A2 — row m of A2 is output pixel m (stride 1, no upsample) — against W columns 9 Cin + c, i.e. a tenth
  // "tap" with its own tensor, row stride and channel count.  nk1 = INT_MAX: no second input.
  const f16* A2;
  int lda2, Cin2;
  int nk1 = 0x7fffffff;         // (default member initialiser: `IgemmArgs a{}` must not switch the second input on)
  // split-K slabs as f16 (round 6; the split launches of the 160x160 kernel and of the LDS-DMA tile kernels): the partial sums travel to the reduce pass through the
  // staged, coalesced f16 epilogue — half the slab bytes each way — at one extra f16 rounding per partial.  0 = fp32 slabs.
  int slab16 = 0;
};
constexpr int kNoSeg2 = 0x7fffffff;
// output row of virtual row m of a phase launch (IgemmArgs::ph_rows): the (2y + a, 2x + b) pixel of the upsampled image
__device__ __forceinline__ int phase_out_row(const IgemmArgs& p, int m) {
  const int ph = m / p.ph_rows, pm = m - ph * p.ph_rows;
  const int hw = p.Hi * p.Wi, img = pm / hw, rem = pm - img * hw;
  const int y = rem / p.Wi, x = rem - y * p.Wi;
  return img * 4 * hw + (2 * y + (ph >> 1)) * 2 * p.Wi + 2 * x + (ph & 1);
}
constexpr int kLnxMaxParts = 20;   // partial slots per row a consumer can sum (N = 1280 behind 64-wide producer tiles)
constexpr int kEpiLN = 1 << 20;  // internal epilogue bit (not part of the C-ABI flags)

// GEGLU packing (rcdm_pack_geglu_rows): packed rows/columns come in groups of 32 = 16 "hidden" + their 16 "gate"
// (16 = one 16x16x32 fragment, half a 32x32x16 one: value and gate sit in the same lane for both MFMA shapes, and every
// tile width that is a multiple of 32 — 128, 160, 256, 320 — holds whole groups);
// packed column n of a hidden value <-> output column (n>>5)*16 + (n&15); its gate sits at n + 16.
constexpr int kGegluGroup = 16;
__host__ __device__ __forceinline__ int geglu_out_col(int n) { return (n >> 5) * 16 + (n & 15); }

// igemm8.hip: ping-pong tile shapes (index into kPPShapes), launched by igemm.hip's dispatcher
struct PPShape { int bm, bn; };
constexpr int kNumPPShapes = 3;
extern const PPShape kPPShapes[kNumPPShapes];
// igemm16.hip: 160x160 tiles, 4 waves, two blocks per CU
int rcdm_igemm16_launch(const IgemmArgs& a, int taps, hipStream_t stream);
// taps = 1 | 9 | 4 (4: the phase form of an upsampling conv, a.ph_rows set); a.tilesM/tilesN/splits/nk_per_split/partial
// already planned for the shape.  Returns an RCDM_* code.
int rcdm_igemm_pp_launch(const IgemmArgs& a, int taps, int shape, hipStream_t stream);
