// wino.hip — conv3x3 (stride 1, padding 1) over channels-last f16 rows as Winograd F(2x2, 3x3): sixteen GEMMs over the
// transformed 4x4 input patches instead of nine taps over the pixels — 4/9 of the multiply-adds.  Round 6 (VERDICT r5 #1a).
//
// Replaces (reference): InflatedConv3d.forward src/models/resnet.py:10-18 as called by ResnetBlock3D.forward
// resnet.py:188,205 (conv1 / conv2, with the 1x1 conv_shortcut of :208 riding as four more GEMM entries) at the levels where
// the implicit GEMM is a chain of latencies and not a stream: the 16x16 and 8x8 latents of the UNet (unet_blocks.py).
//
//   Y = A^T [ (G g G^T) . (B^T d B) ] A        per 2x2 output tile, input patch d = in[2ty-1 .. 2ty+2][2tx-1 .. 2tx+2]
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1]
//
// Three launches: (1) wino_in_kernel — V[pos][tile][c] = (B^T d B)[pos], fp32 arithmetic on the f16 pixels, ONE rounding to
// f16; optionally the GroupNorm apply + SiLU of the norm in front of the conv (resnet.py:185-186,202) on the way in, so the
// normalised tensor is never written; (2) wino_gemm_kernel — the 160x160-tile LDS-DMA loop of igemm16.hip over
// E = 16 (+ 4) batch entries x split-K slices x tiles, dealt to the XCDs as contiguous runs (whole positions per L2), results
// to slabs [entry][split][tile][N] — f16 through the staged epilogue by default (rcdm_set_wino_slab_f16), fp32 otherwise:
// entry e < 16 is V[e] (T x K) against U[e] (N x K); entries 16 + 2a + b (only with a second input) are the 1x1 convolution
// of the pixels (2ty + a, 2tx + b) of in2 against W2 — plain GEMM rows picked by parity, no transform; (3) wino_out_kernel —
// the fixed-order sum of a tile's slabs, A^T M A in fp32, the 1x1 terms, bias / per-sample row vector / residual / scale, one
// rounding to f16, and optionally the per-tile partial statistics of the GroupNorm that reads the output next.
// Numerics: the products see f16(B^T d B) and f16(G g G^T) instead of f16 pixels and f16 weights: |B^T d B| <= 4 max|d|
// (typically 2x), so the operand rounding noise is about twice the direct form's; with f16 slabs every position sum is
// rounded once more before the output transform; both transforms and the accumulation are fp32.  DESIGN.md section 4g.
//
// Also here, because they are the same kind of algebra on the 3x3 convolution: upsample_gather_kernel
// (rcdm_conv_taps_gather / rcdm_upsample_taps_gather) — Upsample3D (resnet.py:60-79) and conv_out (unet.py:457) as ONE plain
// GEMM over nine stacked tap planes + a gather.
#include "common.h"
#include "igemm_args.h"
#include "pp_sync.h"
#include "igemm_epilogue.h"
#include <stdlib.h>

namespace {

struct WinoArgs {
  // conv geometry
  const f16* in;
  const f16* in2;
  const f16* U;      // f16 [16][N][K]   (rcdm_pack_conv3x3_wino)
  const f16* W2;     // f16 [N][K2]      (1x1 weight of the second input) or null
  const float* bias;
  const float* rowvec;
  const f16* res;
  f16* out;
  f16* V;            // workspace: f16 [16][T][K]
  float* slab;       // workspace: fp32 (or, slab16 != 0, f16) [E][splits][T][N]
  int slab16;
  int n_img, H, W, th, tw, T, N, K, K2;
  int lda, lda2, ldc, ldr, ldt, rows_per_sample;
  int epi;
  float out_scale;
  int E, splits, nk, nk2, nkps, nkps2, tilesM, tilesN;
  // GroupNorm (+ SiLU) applied to `in` by the input transform (null: `in` is used as it is)
  const float* gn_stat;   // [samples][G][2] = (mean, rstd)
  const float* gn_gamma;
  const float* gn_beta;
  int gn_cg, gn_G, gn_rps, gn_silu;
  // partial statistics of the GroupNorm that reads `out` NEXT, left by the output transform (null: none): one partial
  // (count, mean, M2) per (sample, group, tile of the sample), the layout gn_finalize_kernel reads with splits = tiles per sample
  float* go_partial;
  int go_G, go_cg, go_rps, go_splits;
};

// ---------------------------------------------------------------------------------------------------------------------
// (1) input transform.  One thread per (tile, 4-channel group): 16 pixel loads of 8 bytes (zero outside the image — the
// conv's padding, applied AFTER the norm / activation), B^T d B in fp32, 16 stores of 8 bytes; consecutive threads hold
// consecutive channel groups, so every load / store instruction of a wave covers 512 contiguous bytes.
__global__ __launch_bounds__(256) void wino_in_kernel(const WinoArgs p) {
  const int nch = p.K >> 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.T * nch) return;
  const int t = idx / nch, c4 = idx - t * nch;
  const int tpi = p.th * p.tw;
  const int img = t / tpi, rem = t - img * tpi;
  const int ty = rem / p.tw, tx = rem - ty * p.tw;
  // all 16 loads are issued unconditionally through a buffer descriptor: a pixel outside the image gets an offset past
  // num_records and reads as zero in hardware (a load under a per-element condition is a branch + wait per element: the
  // first version's 16 round trips took 11-17 us per launch)
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  union H4 { u32x2_t v; f16 e[4]; f16x4 h; };
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0x7FFFFFFF, 0x00020000);
  const unsigned base = ((unsigned)(img * p.H * p.W) * (unsigned)p.lda + (unsigned)c4 * 4u) * 2u;
  H4 d[4][4];
  bool ok[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y = 2 * ty - 1 + i, x = 2 * tx - 1 + j;
      ok[i][j] = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      const unsigned off = ok[i][j] ? base + (unsigned)(y * p.W + x) * (unsigned)p.lda * 2u : 0x80000000u;
      d[i][j].v = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
    }
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  const bool gn = p.gn_stat != nullptr;
  if (gn) {
    const int s = (img * p.H * p.W) / p.gn_rps;
    const f32x4 ga = *(const f32x4*)(p.gn_gamma + c4 * 4), be = *(const f32x4*)(p.gn_beta + c4 * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int g = (c4 * 4 + e) / p.gn_cg;
      const f32x2 mr = *(const f32x2*)(p.gn_stat + (size_t)(s * p.gn_G + g) * 2);
      sc[e] = mr.y * ga[e];
      sh[e] = __builtin_fmaf(-(mr.x * mr.y), ga[e], be[e]);
    }
  }
  float f[4][4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = (float)d[i][j].e[e];
        if (gn) {
          v = __builtin_fmaf(v, sc[e], sh[e]);
          if (p.gn_silu) v = silu_f(v);
          if (!ok[i][j]) v = 0.f;
        }
        f[i][j][e] = v;
      }
  // rows: B^T d
  float r[4][4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      r[0][j][e] = f[0][j][e] - f[2][j][e];
      r[1][j][e] = f[1][j][e] + f[2][j][e];
      r[2][j][e] = f[2][j][e] - f[1][j][e];
      r[3][j][e] = f[1][j][e] - f[3][j][e];
    }
  // columns: (B^T d) B, rounded once
  f16* vb = p.V + (size_t)t * p.K + c4 * 4;
  const size_t pstride = (size_t)p.T * p.K;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    H4 o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[0].e[e] = (f16)(r[i][0][e] - r[i][2][e]);
      o[1].e[e] = (f16)(r[i][1][e] + r[i][2][e]);
      o[2].e[e] = (f16)(r[i][2][e] - r[i][1][e]);
      o[3].e[e] = (f16)(r[i][1][e] - r[i][3][e]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) *(u32x2_t*)(vb + (size_t)(i * 4 + j) * pstride) = o[j].v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// (2) the batched GEMM: igemm16.hip's loop (160x160 tile, 4 waves, two blocks per CU, one barrier per 64-deep k-step, 2-stage
// LDS-DMA ring, XOR swizzle on the source side), entry and split-K slice from blockIdx.y, fp32 tile to the entry's slab.
template <bool S16>   // S16: the tile goes to an f16 slab through the staged, coalesced epilogue of igemm_epilogue.h
__global__ __launch_bounds__(256, 2) void wino_gemm_kernel(const WinoArgs p) {
  constexpr int BM = 160, BN = 160, FM = 5, FN = 5;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int NP = BM / 8 / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // Work item = (entry, split-K slice, tile), ordered entry-major with the second input's four entries first (their K is the
  // other tensor's width, usually the longer loop).  The hardware deals workgroups to the 8 XCDs round-robin by linear id, so
  // block b -> XCD b % 8 gets a CONTIGUOUS run of items: whole entries per XCD (two positions each at E = 16), i.e. every
  // byte of U and V is fetched into exactly one L2.  (Tiles of one entry spread over all XCDs — the first version — pulled
  // each V[pos] into all eight L2s: 260 MB through the fabric instead of 78 MB at the 16x16 level, 45 us instead of 36.)
  const int ntile = p.tilesM * p.tilesN, W = p.E * p.splits * ntile;
  const int n_extra = p.E - 16;
  int item;
  {
    const int lin = blockIdx.x, xcd = lin & 7, j = lin >> 3;
    const int WX = n_extra * p.splits * ntile, WM = W - WX;   // items of the second input's entries (first in item order) / of the 16 positions
    if (((WX | WM) & 7) == 0) {
      // every XCD takes its share of the long entries first, then its run of positions
      const int xq = WX >> 3, mq = WM >> 3;
      item = j < xq ? xcd * xq + j : WX + xcd * mq + (j - xq);
    } else {
      const int q = W >> 3, r = W & 7;
      item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
  }
  const int yes = item / ntile, tl = item - yes * ntile;
  const int ye = yes / p.splits, split = yes - ye * p.splits;
  const int entry = ye < n_extra ? 16 + ye : ye - n_extra;
  const bool extra = entry >= 16;
  const int nk = extra ? p.nk2 : p.nk, nkps = extra ? p.nkps2 : p.nkps;
  const int ks_begin = split * nkps;
  const int nkl = min(nk, ks_begin + nkps) - ks_begin;
  const int Kd = extra ? p.K2 : p.K;             // contraction width = W row length
  const size_t slab_off = ((size_t)entry * p.splits + split) * p.T * p.N;

  const int tn = tl / p.tilesM, tm = tl - tn * p.tilesM;
  const int cm0 = tm * BM, cn0 = tn * BN;
  const f16* Abase = extra ? p.in2 : p.V + (size_t)entry * p.T * p.K;
  const f16* Wbase = extra ? p.W2 : p.U + (size_t)entry * p.N * p.K;
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)Abase, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)Wbase, 0, 0x7FFFFFFF, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const int lrow = lane >> 3, lch = lane & 7;

  unsigned a_off[NP], w_off[NP];
  int a_c[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = (wave * NP + i) * 8 + lrow;
    const int m = cm0 + row;
    a_c[i] = (lch ^ ((row >> 1) & 7)) * 8;
    a_off[i] = OOB;
    if (m < p.T) {
      if (extra) {   // tile m -> pixel (2ty + a, 2tx + b) of in2
        const int tpi = p.th * p.tw;
        const int img = m / tpi, rem = m - img * tpi;
        const int ty = rem / p.tw, tx = rem - ty * p.tw;
        const int ab = entry - 16;
        const int prow = img * p.H * p.W + (2 * ty + (ab >> 1)) * p.W + 2 * tx + (ab & 1);
        a_off[i] = (unsigned)prow * (unsigned)p.lda2 * 2u;
      } else {
        a_off[i] = (unsigned)m * (unsigned)p.K * 2u;
      }
    }
    const int n = cn0 + row;
    w_off[i] = n < p.N ? (unsigned)n * (unsigned)Kd * 2u : OOB;
  }
  auto issue = [&](int g, int stage) __attribute__((always_inline)) {
    const int c0 = (ks_begin + g) * BK;
    char* sbase = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int c = c0 + a_c[i];
      const unsigned vo = (c < Kd && a_off[i] != OOB) ? a_off[i] + (unsigned)c * 2u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsrcA, (__attribute__((address_space(3))) void*)(sbase + (wave * NP + i) * 1024), 16, vo, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int c = c0 + a_c[i];
      const unsigned vo = (c < Kd && w_off[i] != OOB) ? w_off[i] + (unsigned)c * 2u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsrcW, (__attribute__((address_space(3))) void*)(sbase + A_BYTES + (wave * NP + i) * 1024), 16, vo, 0, 0, 0);
    }
  };

  // weights are the MFMA A operand, tiles the B operand -> D[channel][tile row]
  const int l15 = lane & 15, kg = lane >> 4;
  const int sw = (lane >> 1) & 7;
  const int koff0 = ((kg ^ sw) << 4), koff1 = (((4 + kg) ^ sw) << 4);
  const int rowA = (wm * (BM / 2) + l15) * 128, rowB = A_BYTES + (wn * (BN / 2) + l15) * 128;
  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nkl > 0) issue(0, 0);
  for (int g = 0; g < nkl; ++g) {
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (g + 1 < nkl) issue(g + 1, (g + 1) & 1);
    const char* sb = smem + (g & 1) * STAGE;
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int koff = kk ? koff1 : koff0;
      f16x8 wf[FN], xf[FM];
#pragma unroll
      for (int i = 0; i < FN; ++i) wf[i] = *(const f16x8*)(sb + rowB + i * 2048 + koff);
#pragma unroll
      for (int j = 0; j < FM; ++j) xf[j] = *(const f16x8*)(sb + rowA + j * 2048 + koff);
      __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(3);
    }
  }
  const int row0 = wm * (BM / 2), col0 = wn * (BN / 2);
  if constexpr (S16) {
    // f16 slab: accumulators -> f16 tile in LDS -> whole-row 16-byte stores (the plain-projection path of tile_epilogue)
    wait_lgkm0();
    tick_barrier();   // every wave is done reading the ring
    IgemmArgs q{};
    q.out = (f16*)p.slab + slab_off;
    q.M = p.T; q.N = p.N; q.ldc = p.N; q.epi = 0; q.out_scale = 1.0f; q.dup = 0;
    tile_epilogue<FM, FN, false, 256, BM, BN, false, false>(q, smem, acc, cm0, cn0, row0, col0, l15, kg, t, nullptr);
  } else {
    // fp32 tile to the slab, whole rows per store through LDS (an empty slice — more splits than k-steps — writes zeros)
    wait_lgkm0();
    tick_barrier();   // every wave is done reading the ring
    slab_store_staged<FM, FN, 256, BM, BN, 80>(p.slab + slab_off, p.N, p.T, p.N, smem, acc, cm0, cn0, row0, col0, l15, kg, t);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// (3) output transform + epilogue.  One BLOCK per tile, one thread per 4-channel group: the 16 transform-domain values of the
// group (each the fixed-order sum of its split-K slabs), A^T M A, + the four 1x1 terms, + bias + row vector, + residual, x scale.
// SP: split count compiled in (1, 2: all slab loads requested before the first addition), 0 = any (one round trip per slice).
// go_partial: the block also leaves the tile's partial statistics for the GroupNorm that reads `out` next — taken from the
// STORED halfs (what a statistics pass would read back), per-channel sums through LDS, then one thread per group in channel
// order: deterministic.
template <bool S16>
__device__ __forceinline__ f32x4 slab_load4(const WinoArgs& p, size_t off) {
  if constexpr (S16) {
    union { uint2 u; f16 e[4]; } h;
    h.u = *(const uint2*)((const f16*)p.slab + off);
    return f32x4{(float)h.e[0], (float)h.e[1], (float)h.e[2], (float)h.e[3]};
  } else {
    return *(const f32x4*)(p.slab + off);
  }
}

template <int SP, bool S16>
__global__ __launch_bounds__(512) void wino_out_kernel(const WinoArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nch = p.N >> 2;
  const int t = blockIdx.x;
  const int tpi = p.th * p.tw;
  const int img = t / tpi, rem = t - img * tpi;
  const int ty = rem / p.tw, tx = rem - ty * p.tw;
  const size_t splane = (size_t)p.T * p.N, eplane = (size_t)p.splits * splane;
  int prow[4];
#pragma unroll
  for (int ab = 0; ab < 4; ++ab) prow[ab] = img * p.H * p.W + (2 * ty + (ab >> 1)) * p.W + 2 * tx + (ab & 1);
  union H4 { uint2 u; f16 e[4]; };
  for (int cg4 = threadIdx.x; cg4 < nch; cg4 += blockDim.x) {
    const int n = cg4 * 4;
    const size_t src = (size_t)t * p.N + n;
    // the residual rows are requested first (the coldest operand of the thread)
    H4 rr[4];
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
      rr[ab].u = make_uint2(0, 0);
      if (p.epi & RCDM_EPI_RESIDUAL) rr[ab].u = *(const uint2*)(p.res + (size_t)prow[ab] * p.ldr + n);
    }
    f32x4 m[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) m[e] = slab_load4<S16>(p, src + e * eplane);
    if constexpr (SP == 2) {
      f32x4 m2[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) m2[e] = slab_load4<S16>(p, src + e * eplane + splane);
#pragma unroll
      for (int e = 0; e < 16; ++e) m[e] += m2[e];
    } else if constexpr (SP == 0) {
      for (int s = 1; s < p.splits; ++s) {
        f32x4 m2[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) m2[e] = slab_load4<S16>(p, src + e * eplane + s * splane);
#pragma unroll
        for (int e = 0; e < 16; ++e) m[e] += m2[e];
      }
    }
    f32x4 o[4];
    {
      f32x4 r0[4], r1[4];   // rows: A^T M
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r0[j] = m[0 * 4 + j] + m[1 * 4 + j] + m[2 * 4 + j];
        r1[j] = m[1 * 4 + j] - m[2 * 4 + j] - m[3 * 4 + j];
      }
      o[0] = r0[0] + r0[1] + r0[2];
      o[1] = r0[1] - r0[2] - r0[3];
      o[2] = r1[0] + r1[1] + r1[2];
      o[3] = r1[1] - r1[2] - r1[3];
    }
    if (p.E > 16) {
      f32x4 x[4];
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) x[ab] = slab_load4<S16>(p, src + (16 + ab) * eplane);
      for (int s = 1; s < p.splits; ++s) {
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) x[ab] += slab_load4<S16>(p, src + (16 + ab) * eplane + s * splane);
      }
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) o[ab] += x[ab];
    }
    f32x4 add = {0.f, 0.f, 0.f, 0.f};
    if (p.epi & RCDM_EPI_BIAS) add += *(const f32x4*)(p.bias + n);
    if (p.epi & RCDM_EPI_ROWVEC) add += *(const f32x4*)(p.rowvec + (size_t)(prow[0] / p.rows_per_sample) * p.ldt + n);   // (a tile lies inside one image)
    f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cq = cs;
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
      H4 q;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        q.e[e] = (f16)((o[ab][e] + add[e] + (float)rr[ab].e[e]) * p.out_scale);
        const float f = (float)q.e[e];
        cs[e] += f;
        cq[e] += f * f;
      }
      *(uint2*)(p.out + (size_t)prow[ab] * p.ldc + n) = q.u;
    }
    if (p.go_partial) {   // per-channel (sum, sum of squares) over the tile's four pixels
      *(f32x4*)(smem + (size_t)n * 4) = cs;
      *(f32x4*)(smem + (size_t)(p.N + n) * 4) = cq;
    }
  }
  if (p.go_partial) {
    __syncthreads();
    const int g = threadIdx.x;
    if (g < p.go_G) {
      const float* fs = (const float*)smem;
      float gs = 0.f, gq = 0.f;
      for (int c = g * p.go_cg; c < (g + 1) * p.go_cg; ++c) {
        gs += fs[c];
        gq += fs[p.N + c];
      }
      const float nn = 4.f * (float)p.go_cg;
      const float mean = gs / nn;
      float m2 = gq - gs * mean;
      if (m2 < 0.f) m2 = 0.f;
      const int s = prow[0] / p.go_rps;
      const int sp = t - s * p.go_splits;
      float* o = p.go_partial + (((size_t)s * p.go_G + g) * p.go_splits + sp) * 3;
      o[0] = nn;
      o[1] = mean;
      o[2] = m2;
    }
  }
}

// weight transform: U[pos][co][ci] = f16((G g G^T)[pos]) from torch (Cout, Cin, 3, 3) fp32
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, f16* __restrict__ U, int c_out, int c_in) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)c_out * c_in;
  if (idx >= total) return;
  const float* g = w + idx * 9;
  float gg[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) gg[i][j] = g[i * 3 + j];
  float r[4][3];   // G g
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    r[0][j] = gg[0][j];
    r[1][j] = 0.5f * (gg[0][j] + gg[1][j] + gg[2][j]);
    r[2][j] = 0.5f * (gg[0][j] - gg[1][j] + gg[2][j]);
    r[3][j] = gg[2][j];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float u0 = r[i][0], u1 = 0.5f * (r[i][0] + r[i][1] + r[i][2]), u2 = 0.5f * (r[i][0] - r[i][1] + r[i][2]), u3 = r[i][2];
    U[(size_t)(i * 4 + 0) * total + idx] = (f16)u0;
    U[(size_t)(i * 4 + 1) * total + idx] = (f16)u1;
    U[(size_t)(i * 4 + 2) * total + idx] = (f16)u2;
    U[(size_t)(i * 4 + 3) * total + idx] = (f16)u3;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Upsample3D (nearest 2x, then conv3x3; resnet.py:60-79) as ONE plain GEMM + a gather.  On the upsampled grid output pixel
// (Y, X) = sum over the nine taps of w[ky][kx] . s((Y + ky - 1) >> 1, (X + kx - 1) >> 1): every product is a tap's 1x1 image of
// a SOURCE pixel, P[src pixel][tap * C_out + c] = x[src pixel][:] . w[c][:][tap] — nine products per source pixel and channel
// pair where the four 2x2 phase convolutions (rcdm_conv3x3, upsample = 2) need sixteen and the literal form thirty-six.  The
// products are one rcdm_gemm with N = 9 C_out over the source rows (no transform, the ordinary f16 weights, re-ordered);
// this kernel sums, per output pixel, the nine planes' values of the source pixels its taps land on (taps outside the
// upsampled image are skipped: its zero padding), adds the bias and rounds once.  One thread per (output pixel, 8 channels);
// each P value is read by four neighbouring outputs (L2).
struct UpGatherArgs {
  const f16* P;
  const float* bias;
  f16* out;
  int ldp, ldc, n_img, H, W, C;
  int up;   // 1: the taps index the nearest-2x upsampled image (source pixel = tap position >> 1); 0: a plain stride-1 conv3x3
};
__global__ __launch_bounds__(256) void upsample_gather_kernel(const UpGatherArgs p) {
  const int nch = p.C >> 3;
  const size_t total = ((size_t)p.n_img * p.H * p.W << (2 * p.up)) * nch;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int o = (int)(idx / nch), c8 = (int)(idx - (size_t)o * nch);
  const int W2 = p.W << p.up, H2 = p.H << p.up;
  const int img = o / (H2 * W2), rem = o - img * (H2 * W2);
  const int Y = rem / W2, X = rem - Y * W2;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.P, 0, 0x7FFFFFFF, 0x00020000);
  Pack16 v[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = Y + ky - 1, xx = X + kx - 1;
      const bool ok = (unsigned)yy < (unsigned)H2 && (unsigned)xx < (unsigned)W2;
      const unsigned row = (unsigned)(img * p.H * p.W + (yy >> p.up) * p.W + (xx >> p.up));
      const unsigned off = ok ? (row * (unsigned)p.ldp + (unsigned)((ky * 3 + kx) * p.C + c8 * 8)) * 2u : 0x80000000u;
      v[ky * 3 + kx].v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    }
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (p.bias) {
    const f32x4 b0 = *(const f32x4*)(p.bias + c8 * 8), b1 = *(const f32x4*)(p.bias + c8 * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[e] = b0[e];
      acc[4 + e] = b1[e];
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += (float)v[k].e[e];
  Pack16 q;
#pragma unroll
  for (int e = 0; e < 8; ++e) q.e[e] = (f16)acc[e];
  *(uint4*)(p.out + (size_t)o * p.ldc + c8 * 8) = q.u;
}

int g_wino_cus = 0;
int wino_cus() {
  if (g_wino_cus <= 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      g_wino_cus = n;
    else
      g_wino_cus = 256;
  }
  return g_wino_cus;
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// geometry + workspace carve-up; returns an RCDM_* code
int wino_plan(const rcdm_conv3x3_desc* d, WinoArgs& a) {
  if (!d) return RCDM_EINVAL;
  if (d->n_img <= 0 || d->h_in <= 0 || d->w_in <= 0 || d->c_in <= 0 || d->c_out <= 0) return RCDM_EINVAL;
  if (d->stride != 1 || d->upsample != 0 || d->pad_after_only || d->dup_rows) return RCDM_ESHAPE;
  if ((d->h_in & 1) || (d->w_in & 1)) return RCDM_ESHAPE;
  if ((d->c_in & 63) || (d->c_out & 7) || (d->c_in2 & 63)) return RCDM_ESHAPE;
  if ((d->lda & 7) || (d->ldc & 7) || (d->c_in2 && (d->lda2 & 7))) return RCDM_EINVAL;
  if (d->epilogue & ~(RCDM_EPI_BIAS | RCDM_EPI_ROWVEC | RCDM_EPI_RESIDUAL)) return RCDM_ESHAPE;
  if ((d->epilogue & RCDM_EPI_RESIDUAL) && (d->ldr & 7)) return RCDM_EINVAL;
  a = WinoArgs{};
  a.n_img = d->n_img; a.H = d->h_in; a.W = d->w_in;
  a.th = d->h_in / 2; a.tw = d->w_in / 2;
  a.T = d->n_img * a.th * a.tw;
  a.N = d->c_out; a.K = d->c_in; a.K2 = d->c_in2;
  a.lda = d->lda; a.lda2 = d->lda2; a.ldc = d->ldc; a.ldr = d->ldr; a.ldt = d->ldt;
  a.rows_per_sample = d->rows_per_sample > 0 ? d->rows_per_sample : 1;
  a.epi = d->epilogue;
  a.out_scale = d->out_scale == 0.f ? 1.f : d->out_scale;
  a.E = d->c_in2 ? 20 : 16;
  a.nk = d->c_in / BK; a.nk2 = d->c_in2 / BK;
  a.tilesM = (a.T + 159) / 160; a.tilesN = (a.N + 159) / 160;
  // 32-bit byte offsets into 2-GB buffer resources
  if ((size_t)a.T * a.K * 2 >= (1ull << 31)) return RCDM_ESHAPE;
  if ((size_t)d->n_img * d->h_in * d->w_in * d->lda * 2 >= (1ull << 31)) return RCDM_ESHAPE;
  if (d->c_in2 && (size_t)d->n_img * d->h_in * d->w_in * d->lda2 * 2 >= (1ull << 31)) return RCDM_ESHAPE;
  if ((size_t)a.N * (a.K > a.K2 ? a.K : a.K2) * 2 >= (1ull << 31)) return RCDM_ESHAPE;
  // split-K: up to one block per CU (measured at the 8x8 level, 128 tile-entries: split 2 beats 1 and 4 — the slabs of every
  // extra slice are read back by the output transform), a slice keeps >= 5 k-steps
  int splits = d->split_k;
  if (splits <= 0) {
    const int blocks = a.E * a.tilesM * a.tilesN;
    splits = wino_cus() / (blocks > 0 ? blocks : 1);
    if (splits > a.nk / 5) splits = a.nk / 5;
    if (splits > 8) splits = 8;
  }
  if (splits < 1) splits = 1;
  if (splits > a.nk) splits = a.nk;
  a.splits = splits;
  a.nkps = (a.nk + splits - 1) / splits;
  a.nkps2 = (a.nk2 + splits - 1) / splits;
  return RCDM_OK;
}
size_t wino_v_bytes(const WinoArgs& a) { return align256((size_t)16 * a.T * a.K * 2); }
size_t wino_slab_bytes(const WinoArgs& a) { return align256((size_t)a.E * a.splits * a.T * a.N * 4); }   // (sized for fp32 in either mode)

// 1: the batched GEMM writes f16 slabs (half the slab traffic, coalesced stores; every transform-domain value rounded to f16
// before the output transform), 0: fp32 slabs.  -1 = not set: environment RCDM_WINO_SLAB16, default 1 (measured: -10 us per 16x16-level conv, whole-UNet rel-RMS +0.2 ... +2.7 %).
int g_wino_slab16 = -1;
int wino_slab16() {
  if (g_wino_slab16 < 0) {
    const char* e = getenv("RCDM_WINO_SLAB16");
    g_wino_slab16 = e ? (atoi(e) != 0) : 1;
  }
  return g_wino_slab16;
}

}  // namespace

extern "C" {

int rcdm_conv3x3_wino_supported(const rcdm_conv3x3_desc* d) {
  WinoArgs a;
  return wino_plan(d, a) == RCDM_OK ? 1 : 0;
}

size_t rcdm_conv3x3_wino_workspace_bytes(const rcdm_conv3x3_desc* d) {
  WinoArgs a;
  if (wino_plan(d, a) != RCDM_OK) return 0;
  return wino_v_bytes(a) + wino_slab_bytes(a);
}

int rcdm_conv3x3_wino_plan_query(const rcdm_conv3x3_desc* d, int32_t* out8) {
  WinoArgs a;
  const int rc = wino_plan(d, a);
  if (rc != RCDM_OK || !out8) return rc != RCDM_OK ? rc : RCDM_EINVAL;
  out8[0] = 11; out8[1] = 160; out8[2] = 160; out8[3] = a.tilesM; out8[4] = a.tilesN * a.E; out8[5] = a.splits; out8[6] = 2;
  out8[7] = a.nk;
  return RCDM_OK;
}

int rcdm_conv_taps_gather(const void* P, int32_t ldp, int32_t n_img, int32_t h, int32_t w, int32_t c_out, int32_t upsample,
                          const float* bias, void* out, int32_t ldc, void* stream) {
  if (!P || !out || n_img <= 0 || h <= 0 || w <= 0 || c_out <= 0 || (upsample != 0 && upsample != 1)) return RCDM_EINVAL;
  if ((c_out & 7) || (ldp & 7) || (ldc & 7) || ldp < 9 * c_out || ldc < c_out) return RCDM_EINVAL;
  if ((((uintptr_t)P | (uintptr_t)out | (uintptr_t)bias) & 15)) return RCDM_EINVAL;
  if ((size_t)n_img * h * w * ldp * 2 >= (1ull << 31)) return RCDM_ESHAPE;   // 32-bit byte offsets into a 2-GB buffer resource
  UpGatherArgs a{(const f16*)P, bias, (f16*)out, ldp, ldc, n_img, h, w, c_out, upsample};
  const size_t total = ((size_t)n_img * h * w << (2 * upsample)) * (c_out >> 3);
  hipLaunchKernelGGL(upsample_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return rcdm_check_launch();
}

int rcdm_upsample_taps_gather(const void* P, int32_t ldp, int32_t n_img, int32_t h, int32_t w, int32_t c_out, const float* bias,
                              void* out, int32_t ldc, void* stream) {
  return rcdm_conv_taps_gather(P, ldp, n_img, h, w, c_out, 1, bias, out, ldc, stream);
}

int rcdm_set_wino_slab_f16(int32_t on) {
  g_wino_slab16 = on < 0 ? -1 : (on ? 1 : 0);
  return RCDM_OK;
}

int rcdm_pack_conv3x3_wino(const float* w, int32_t c_out, int32_t c_in, void* dst, void* stream) {
  if (!w || !dst || c_out <= 0 || c_in <= 0) return RCDM_EINVAL;
  const size_t total = (size_t)c_out * c_in;
  hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (f16*)dst, c_out, c_in);
  return rcdm_check_launch();
}

int rcdm_conv3x3_wino(const rcdm_conv3x3_desc* d, const rcdm_groupnorm_desc* gn, const float* gn_stat, const float* gn_gamma,
                      const float* gn_beta, const void* in, const void* in2, const void* U, const void* W2, const float* bias,
                      const float* rowvec, const void* residual, void* out, void* workspace, size_t workspace_bytes,
                      const rcdm_groupnorm_desc* gn_out, float* gn_out_partial, void* stream_) {
  WinoArgs a;
  const int rc = wino_plan(d, a);
  if (rc != RCDM_OK) return rc;
  if (!in || !U || !out || !workspace) return RCDM_EINVAL;
  if ((d->c_in2 != 0) != (in2 != nullptr) || (d->c_in2 != 0) != (W2 != nullptr)) return RCDM_EINVAL;
  if ((a.epi & RCDM_EPI_BIAS) && !bias) return RCDM_EINVAL;
  if ((a.epi & RCDM_EPI_ROWVEC) && (!rowvec || (d->ldt & 3))) return RCDM_EINVAL;
  if ((a.epi & RCDM_EPI_RESIDUAL) && !residual) return RCDM_EINVAL;
  if (workspace_bytes < wino_v_bytes(a) + wino_slab_bytes(a)) return RCDM_EWORKSPACE;
  if (((uintptr_t)in | (uintptr_t)out | (uintptr_t)U | (uintptr_t)workspace | (uintptr_t)in2 | (uintptr_t)W2 | (uintptr_t)residual |
       (uintptr_t)bias | (uintptr_t)rowvec) & 15)
    return RCDM_EINVAL;
  if (gn) {
    if (!gn_stat || !gn_gamma || !gn_beta) return RCDM_EINVAL;
    if (gn->C != d->c_in || gn->groups <= 0 || gn->C % gn->groups || gn->samples * gn->rows_per_sample != d->n_img * d->h_in * d->w_in ||
        gn->rows_per_sample % (d->h_in * d->w_in))
      return RCDM_ESHAPE;
    a.gn_stat = gn_stat; a.gn_gamma = gn_gamma; a.gn_beta = gn_beta;
    a.gn_G = gn->groups; a.gn_cg = gn->C / gn->groups; a.gn_rps = gn->rows_per_sample; a.gn_silu = gn->silu;
  }
  if (gn_out) {
    // the norm over exactly the rows this launch writes; a tile (2x2 pixels of one image) lies inside one sample
    if (!gn_out_partial) return RCDM_EINVAL;
    if (gn_out->C != d->c_out || gn_out->groups <= 0 || gn_out->groups > 64 || gn_out->C % gn_out->groups ||
        gn_out->samples * gn_out->rows_per_sample != d->n_img * d->h_in * d->w_in || gn_out->rows_per_sample % (d->h_in * d->w_in) ||
        a.N * 8 > 64 * 1024)
      return RCDM_ESHAPE;
    a.go_partial = gn_out_partial;
    a.go_G = gn_out->groups; a.go_cg = gn_out->C / gn_out->groups; a.go_rps = gn_out->rows_per_sample;
    a.go_splits = gn_out->rows_per_sample / 4;
  }
  a.in = (const f16*)in; a.in2 = (const f16*)in2; a.U = (const f16*)U; a.W2 = (const f16*)W2;
  a.bias = bias; a.rowvec = rowvec; a.res = (const f16*)residual; a.out = (f16*)out;
  a.V = (f16*)workspace;
  a.slab16 = wino_slab16();
  a.slab = (float*)((char*)workspace + wino_v_bytes(a));
  hipStream_t stream = (hipStream_t)stream_;
  {
    const size_t items = (size_t)a.T * (a.K >> 2);
    hipLaunchKernelGGL(wino_in_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, a);
  }
  {
    constexpr int kLds = 2 * (160 + 160) * 128;
    static bool attr_set[64] = {};
    if (rcdm_first_on_device(attr_set)) {
      (void)hipFuncSetAttribute((const void*)wino_gemm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
      (void)hipFuncSetAttribute((const void*)wino_gemm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    }
    if (a.slab16) hipLaunchKernelGGL(wino_gemm_kernel<true>, dim3(a.tilesM * a.tilesN * a.E * a.splits), dim3(256), kLds, stream, a);
    else hipLaunchKernelGGL(wino_gemm_kernel<false>, dim3(a.tilesM * a.tilesN * a.E * a.splits), dim3(256), kLds, stream, a);
  }
  {
    int threads = ((a.N >> 2) + 63) / 64 * 64;
    if (threads > 512) threads = 512;
    const size_t lds = a.go_partial ? (size_t)a.N * 8 : 0;
    const dim3 grid((unsigned)a.T);
    if (a.slab16) {
      if (a.splits == 1) hipLaunchKernelGGL((wino_out_kernel<1, true>), grid, dim3(threads), lds, stream, a);
      else if (a.splits == 2) hipLaunchKernelGGL((wino_out_kernel<2, true>), grid, dim3(threads), lds, stream, a);
      else hipLaunchKernelGGL((wino_out_kernel<0, true>), grid, dim3(threads), lds, stream, a);
    } else {
      if (a.splits == 1) hipLaunchKernelGGL((wino_out_kernel<1, false>), grid, dim3(threads), lds, stream, a);
      else if (a.splits == 2) hipLaunchKernelGGL((wino_out_kernel<2, false>), grid, dim3(threads), lds, stream, a);
      else hipLaunchKernelGGL((wino_out_kernel<0, false>), grid, dim3(threads), lds, stream, a);
    }
  }
  return rcdm_check_launch();
}

}  // extern "C"
