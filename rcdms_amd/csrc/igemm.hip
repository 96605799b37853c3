// igemm.hip — MFMA implicit-GEMM for gfx950: Linear / 1x1 conv (taps = 1) and conv3x3 (taps = 9)
// over channels-last f16 rows, fp32 accumulate, fused epilogues.
//
// Replaces (reference, muzishen/RCDMs): InflatedConv3d.forward src/models/resnet.py:10-18, the
// nn.Linear calls of CrossAttention src/models/attention.py:121,140-141,164, Transformer3DModel
// proj_in/proj_out :330,:352, TemporalTransformer3DModel proj_in/out motion_module.py:166,170,
// diffusers FeedForward (GEGLU) and ResnetBlock3D conv_shortcut resnet.py:208.
//
// Structure (one kernel template, three tile shapes):
//   * PERSISTENT blocks: grid = min(#tiles, CUs x blocks/CU); a block walks tiles t, t+G, ... and keeps ONE
//     continuous stream of k-steps across tile boundaries, so the next tile's first operand tiles are already in
//     flight while the current tile's epilogue runs (most GEMMs of this UNet have only 5-20 k-steps per tile:
//     per-tile fill/drain, not the MFMA loop, is what costs).
//   * global -> LDS by buffer_load ... lds (LDS-DMA: no VGPR staging, no ds_write), 2-stage ring, counted
//     s_waitcnt vmcnt + one raw s_barrier per k-step.  Out-of-range rows / taps / channel chunks get the voffset
//     0x80000000 (>= num_records): the hardware writes zeros, so there are no branches in the loader.
//   * The DMA writes LDS lane-linearly (wave-uniform base + lane*16): rows are unpadded 128 B and the bank-conflict
//     fix is an XOR swizzle applied on the SOURCE side — LDS slot (row r, 16-B chunk c') holds global chunk
//     c' ^ ((r>>1)&7); a fragment read of chunk kc of row r reads slot kc ^ ((r>>1)&7).  For the 32x32x16 operand
//     pattern (lanes 0-31 = rows, lane>>5 = chunk parity) every ds_read_b128 lane group hits 16 distinct slots.
//   * The WEIGHT tile is the MFMA A operand and the ACTIVATION tile the B operand, so D[channel][pixel]: a lane
//     owns one pixel and 4 consecutive channels per register quad -> the tile is staged through the ring stage
//     that was just consumed (as f16, 8 bytes per quad; as fp32 for split-K slabs), then a coalesced
//     16-byte-per-lane epilogue (bias / per-sample row vector / GELU / GEGLU / residual / scale in fp32) while the
//     other stage already receives the next tile.
//   tiles <BM, BN, WM x WN waves>: 128x128 (2x2, two blocks per CU) and 256x256 (2x4, one block per CU).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "igemm_args.h"
#include "gn_plan.h"
#ifndef RCDM_LNX_ABLATE
#define RCDM_LNX_ABLATE 0   // debug builds (tools/lnx_bench.py): 1 = no partial-statistics loads, 64 = no accumulator transform, 128 = no table write, 256 = no producer statistics
#endif

namespace {

// v: 8 accumulated values of row m at packed columns n..n+7 (GEGLU: g = the matching gate columns n+16..).
// n is a multiple of 8, so every per-column vector (bias, row vector) is fetched as two aligned float4.
__device__ __forceinline__ void load8(const float* src, float (&d)[8]) {
  const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
  d[0] = a[0]; d[1] = a[1]; d[2] = a[2]; d[3] = a[3];
  d[4] = b[0]; d[5] = b[1]; d[6] = b[2]; d[7] = b[3];
}

// v[0..8) += eight consecutive slab values at element offset `off` (fp32 slabs, or f16 ones: IgemmArgs::slab16)
__device__ __forceinline__ void slab_add8(const IgemmArgs& p, size_t off, float (&v)[8]) {
  if (p.slab16) {
    Pack16 h;
    h.u = *(const uint4*)((const f16*)p.partial + off);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += (float)h.e[e];
  } else {
    const f32x4 a = *(const f32x4*)(p.partial + off), b = *(const f32x4*)(p.partial + off + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] += a[e];
      v[4 + e] += b[e];
    }
  }
}

__device__ __forceinline__ uint4 epilogue_store(const IgemmArgs& p, int m, int n, float (&v)[8], float (&g)[8]) {   // returns the 8 halfs it stored
  int oc = n;
  if (p.epi & RCDM_EPI_GEGLU) {
    if (p.epi & RCDM_EPI_BIAS) {
      float bh[8], bg[8];
      load8(p.bias + n, bh);
      load8(p.bias + n + kGegluGroup, bg);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] += bh[e];
        g[e] += bg[e];
      }
    }
    gelu8(g);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= g[e];
    oc = geglu_out_col(n);
  } else if (p.epi & RCDM_EPI_BIAS) {
    float bb[8];
    load8(p.bias + n, bb);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += bb[e];
  }
  if (p.epi & RCDM_EPI_ROWVEC) {
    float rv[8];
    load8(p.rowvec + (size_t)(m / p.rows_per_sample) * p.ldt + oc, rv);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += rv[e];
  }
  if (p.epi & RCDM_EPI_GELU) gelu8(v);
  if (p.epi & RCDM_EPI_RESIDUAL) {
    Pack16 r;
    r.u = *(const uint4*)(p.res + (size_t)m * p.ldr + oc);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += (float)r.e[e];
  }
  Pack16 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o.e[e] = (f16)(v[e] * p.out_scale);
  *(uint4*)(p.out + (size_t)m * p.ldc + oc) = o.u;
  if (p.dup) *(uint4*)(p.out + (size_t)m * p.ldc + oc + p.dup) = o.u;
  return o.u;
}

// Every load the block has in flight — the next tile's first DMA pieces (issued a k-step and a staging phase ago) and
// this tile's residual / bias prefetches — is waited for right before the tile's first output store, with the
// compiler-visible form of s_waitcnt: (a) once stores are in flight vmcnt cannot separate them from DMA pieces, so this
// is the last point where "operands landed" can be established cheaply; (b) the builtin lets the compiler's hazard
// tracking see that no load is pending into a register the main loop reuses (prefetches that a branch never consumes
// would otherwise make it guard those registers with its own vmcnt(0) after every DMA issue in the k-loop).
#define RCDM_PRE_STORE_WAIT() __builtin_amdgcn_s_waitcnt(0x0F70) /* vmcnt(0); expcnt / lgkmcnt untouched */

// -DRCDM_TRACE_PHASES (debug builds for tools/trace_phases.py): the trace record grows from 4 to 8 int64 per block and
// the fused epilogue is stamped after its first barrier, after staging and after the pre-store wait.
#ifdef RCDM_TRACE_PHASES
#define RCDM_PHASE_BEGIN() if (p.trace) tq = __builtin_amdgcn_s_memtime()
#define RCDM_PHASE_STAMP(acc_)                                  \
  if (p.trace) {                                                \
    const long long n_ = __builtin_amdgcn_s_memtime();          \
    acc_ += n_ - tq;                                            \
    tq = n_;                                                    \
  }
constexpr int TRACE_SLOTS = 8;
#else
#define RCDM_PHASE_BEGIN()
#define RCDM_PHASE_STAMP(acc_)
#ifdef RCDM_TRACE_LX   // kernel-start stamps (statistics fetched / first / second barrier) in slots 4-6
constexpr int TRACE_SLOTS = 8;
#else
constexpr int TRACE_SLOTS = 4;
#endif
#endif

template <int TAPS, int BM_, int BN_, int WM, int WN, int NSTAGE, bool E16, int LX = 0>  // LX: deferred-LayerNorm epilogues (rcdm_gemm_lnx) compiled in: 1 = row statistics only, 2 = consumer (+ statistics)
// (second launch bound = waves per SIMD the tile's LDS footprint allows: 4 blocks of 64x64, 3 of 128x64, 2 of 128x128 per CU)
#ifdef RCDM_DMA_MINW2   // A/B builds: the round-3 bound (2 waves per SIMD for every tile)
#define RCDM_DMA_MINW(wm, wn, bm, bn, ns) 2
#else
#define RCDM_DMA_MINW(wm, wn, bm, bn, ns) (((wm) * (wn) == 4 && (bm) * (bn) <= 64 * 64 && (ns) == 2) ? 4 : ((wm) * (wn) == 4 && (bm) * (bn) <= 128 * 64) ? 3 : 2)
#endif
__global__ __launch_bounds__(WM * WN * 64, RCDM_DMA_MINW(WM, WN, BM_, BN_, NSTAGE))
void igemm_dma_kernel(const IgemmArgs p) {
  constexpr int NW = WM * WN;             // waves
  constexpr int FM = BM_ / WM / 32;       // pixel fragments per wave
  constexpr int FN = BN_ / WN / 32;       // channel fragments per wave
  constexpr int A_BYTES = BM_ * 128, B_BYTES = BN_ * 128, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int AI = BM_ / 8 / NW;        // 1-KiB DMA pieces per wave per stage, activations
  constexpr int BI = BN_ / 8 / NW;        // weights
  static_assert(BM_ % (8 * NW) == 0 && BN_ % (8 * NW) == 0, "DMA pieces must divide evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // deferred LayerNorm, consumer side (LX & 2): LTPR threads per tile row sum the row's partial statistics at the tile's
  // FIRST k-step — the loads go out in front of the wait for that step's operands and cost no round trip of their own —
  // and (rstd, mean rstd) sits in a [BM][2] table behind the ring until the epilogue reads it
  constexpr bool LXC = (LX & 2) != 0;
  constexpr int LTPR = (WM * WN * 64) / BM_;
  static_assert(!LXC || LTPR == 1 || LTPR == 2 || LTPR == 4, "threads per row of the LayerNorm table");
  float* ltab = (float*)(smem + NSTAGE * (BM_ + BN_) * 128);
  float* lvS = ltab + 2 * BM_;   // [BN] S of the tile's channels
  constexpr bool lnx_k = LXC;   // the (LX & 2) instantiation is only launched for consumers (p.lnx_stat != nullptr)

  const int ntiles = p.tilesM * p.tilesN;
  const int G = gridDim.x;
  const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - 1 - (int)blockIdx.x) / G + 1 : 0;
  const int ks_begin = blockIdx.y * p.nk_per_split;
  const int ks_end = min(p.nk, ks_begin + p.nk_per_split);
  const int nkl = ks_end - ks_begin;
  if (my_tiles == 0 || nkl <= 0) return;

  // XCD-aware, L2-blocked tile order.  Block b runs on XCD b%8, so each XCD is given a contiguous run of the tile
  // sequence; the sequence itself walks super-rows of GM row panels column by column (n outer, m inner inside the
  // super-row), so the ~64 tiles an XCD has in flight form a GM x (64/GM) patch: every activation panel is shared
  // by 64/GM tiles and every weight panel by GM tiles *at the same time*.  With plain n-fastest order a wide N
  // (GEGLU: 40-80 column tiles) streams the whole weight matrix through the 4 MB L2 once per row panel: rocprofv3
  // FETCH_SIZE showed 16-19x the operand bytes (0.4-0.5 GB per launch, HBM-bound) on the 32x32 / 16x16 levels.
  constexpr int GM = 8;
  auto tile_of = [&](int i, int& m0, int& n0) __attribute__((always_inline)) {
    const int lin = (int)blockIdx.x + i * G;
    const int xcd = lin & 7, q = ntiles >> 3, r = ntiles & 7;
    const int tl = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    const int per_group = GM * p.tilesN;
    const int grp = tl / per_group, rem = tl - grp * per_group;
    const int gm = min(GM, p.tilesM - grp * GM);  // rows in this (possibly last, short) super-row
    const int tn = rem / gm, tm = grp * GM + (rem - tn * gm);
    m0 = tm * BM_;
    n0 = tn * BN_;
  };

  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(TAPS == 9 ? p.A2 : p.A), 0, 0x7FFFFFFF, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const int Hv = p.Hi << p.up, Wv = p.Wi << p.up;

  // ---- loader state of the tile currently being ISSUED (static-indexed arrays: fully unrolled) ----------------
  const int lrow = lane >> 3, lch = lane & 7;
  unsigned a_off[AI], w_off[BI];
  int a_img[AI], a_iy[AI], a_ix[AI], a_c[AI], w_c[BI];
  auto setup_loader = [&](int m0, int n0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int row = (wave * AI + i) * 8 + lrow;
      const int m = m0 + row;
      a_c[i] = (lch ^ ((row >> 1) & 7)) * 8;
      a_off[i] = OOB;
      a_img[i] = a_iy[i] = a_ix[i] = 0;
      if (TAPS == 1) {
        if (m < p.M) a_off[i] = (unsigned)m * (unsigned)p.lda * 2u;
      } else {
        const int hw = p.Ho * p.Wo;
        const int img = m / hw, rem = m - img * hw;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        a_img[i] = img;
        a_iy[i] = m < p.M ? oy * p.stride - p.pad : -(1 << 20);
        a_ix[i] = ox * p.stride - p.pad;
      }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int row = (wave * BI + i) * 8 + lrow;
      const int n = n0 + row;
      w_c[i] = (lch ^ ((row >> 1) & 7)) * 8;
      w_off[i] = n < p.N ? (unsigned)n * (unsigned)p.Ktot * 2u : OOB;
    }
  };

  auto issue = [&](int ks, int stage) __attribute__((always_inline)) {
    int tap = 0, kci = ks;
    if (TAPS != 1) {
      // channel chunk outer, tap inner: the nine shifted windows of one 64-channel slab are read back to back, so
      // the re-reads hit L2 (one slab of all tiles in flight on an XCD is ~2 MB).  Tap-outer order re-read the whole
      // Cin-deep panel (16 MB in flight at Cin = 960) per tap: FETCH_SIZE was 10x the input bytes.
      kci = ks / 9;
      tap = ks - kci * 9;
    }
    // second input (IgemmArgs::A2, conv launches only): the k-steps from nk1 on are a tenth "tap" — the output pixel itself
    // in another tensor with its own row stride and channel count, against W columns 9 Cin + c
    const bool s2 = TAPS == 9 && ks >= p.nk1;   // (wave-uniform)
    if (s2) {
      kci = ks - p.nk1;
      tap = 9;
    }
    const int c0 = kci * BK;
    const int dy = s2 ? 1 : tap / 3, dx = s2 ? 1 : tap - dy * 3;
    const int cin = s2 ? p.Cin2 : p.Cin;
    const unsigned lda = (unsigned)(s2 ? p.lda2 : p.lda);
    char* sbase = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int c = c0 + a_c[i];
      unsigned vo;
      if (TAPS == 1) {
        vo = (c < p.Cin && a_off[i] != OOB) ? a_off[i] + (unsigned)c * 2u : OOB;
      } else {
        const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
        const bool ok = (c < cin) && ((unsigned)iy < (unsigned)Hv) && ((unsigned)ix < (unsigned)Wv);
        const int sy = iy >> p.up, sx = ix >> p.up;
        vo = ok ? ((unsigned)((a_img[i] * p.Hi + sy) * p.Wi + sx) * lda + (unsigned)c) * 2u : OOB;
      }
      if (TAPS == 9 && s2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrcA2, (__attribute__((address_space(3))) void*)(sbase + (wave * AI + i) * 1024), 16, vo, 0, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrcA, (__attribute__((address_space(3))) void*)(sbase + (wave * AI + i) * 1024), 16, vo, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int c = c0 + w_c[i];
      const unsigned vo = (c < cin && w_off[i] != OOB)
                              ? w_off[i] + ((unsigned)tap * (unsigned)p.Cin + (unsigned)c) * 2u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsrcW, (__attribute__((address_space(3))) void*)(sbase + A_BYTES + (wave * BI + i) * 1024), 16, vo, 0, 0, 0);
    }
  };

  // ---- compute mapping -------------------------------------------------------------------------------------
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, hi = lane >> 5;
  const int sw = (lr >> 1) & 7;
  int koff[4];  // byte offset of k-chunk (kk*2 + hi) inside a swizzled 128-B row
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) koff[kk] = (((kk * 2 + hi) ^ sw) << 4);
  const int rowA = (wm * FM * 32 + lr) * 128, rowB = A_BYTES + (wn * FN * 32 + lr) * 128;
  const bool geglu = (p.epi & RCDM_EPI_GEGLU) != 0;

  f32x16 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  long long ts0 = 0, ts_loop = 0, ts_epi = 0;
  long long ts_a = 0, ts_b = 0, ts_c = 0, tq = 0;  // RCDM_TRACE_PHASES only
  if (p.trace) ts0 = __builtin_amdgcn_s_memtime();
  int cm0, cn0;               // tile being computed
  tile_of(0, cm0, cn0);
  // deferred LayerNorm: this thread's share of the first tile's row statistics and its 4 entries of S are requested
  // FIRST — the loader set-up and the prologue DMA below cover their round trip (asked for behind the DMA they cost the
  // 16x16-level attn2.to_q +2.3 us: tools/lnx_bench.py, -DRCDM_LNX_ABLATE=1)
  f32x2 lx_rs = {1.f, 0.f};
  f32x4 lx_s4 = {0.f, 0.f, 0.f, 0.f};
  LnxRow<LTPR, LXC ? kLnxMaxParts : 1> lx_row0;
  if constexpr (LXC) {
    const int lm = cm0 + t / LTPR;
#if !(RCDM_LNX_ABLATE & 1)
    lx_row0.load(p.lnx_stat, p.lnx_ld, lm, lm < p.M, p.lnx_parts, t % LTPR);
#endif
    if (t < BN_ / 4 && cn0 + 4 * t < p.N) lx_s4 = *(const f32x4*)(p.lnx_S + cn0 + 4 * t);
  }
#ifdef RCDM_TRACE_LX
  if (p.trace) ts_a = __builtin_amdgcn_s_memtime() - ts0;
#endif
  setup_loader(cm0, cn0);
  const int total = my_tiles * nkl;
  int i_tile = 0, i_ks = 0;   // next (tile, k-step) to issue
  int i_stage = 0;            // ring slot the next issue goes to
  auto issue_next = [&]() __attribute__((always_inline)) {
    if (i_ks == nkl) {
      i_ks = 0;
      ++i_tile;
      int im0, in0;
      tile_of(i_tile, im0, in0);
      setup_loader(im0, in0);
    }
    issue(ks_begin + i_ks, i_stage);
    ++i_ks;
    i_stage = i_stage + 1 == NSTAGE ? 0 : i_stage + 1;
  };
  // prologue: NSTAGE-1 steps in flight
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < total) issue_next();
  // deferred LayerNorm: this thread's share of the NEXT tile's row statistics -> (rstd, mean rstd), and its 4 entries of S.
  // Fetched where registers are free and a memory wait follows anyway (here: behind the prologue DMA; for later tiles of a
  // persistent block: at the end of the previous epilogue); written to the LDS table after the tile's first barrier.
  auto lx_fetch = [&](int m0, int n0) __attribute__((always_inline)) {
    const int lm = m0 + t / LTPR;
    float r_ = 1.f, m_ = 0.f;
#if !(RCDM_LNX_ABLATE & 1)
    lnx_row<LTPR, kLnxMaxParts>(p.lnx_stat, p.lnx_ld, lm, lm < p.M, p.lnx_parts, t % LTPR, p.lnx_invC, p.lnx_eps, r_, m_);
#endif
    lx_rs = f32x2{r_, m_};
    lx_s4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t < BN_ / 4 && n0 + 4 * t < p.N) lx_s4 = *(const f32x4*)(p.lnx_S + n0 + 4 * t);
  };
#ifdef RCDM_TRACE_LX
  if (p.trace) ts_b = __builtin_amdgcn_s_memtime() - ts0;
#endif
#if !(RCDM_LNX_ABLATE & 1)
  if constexpr (LXC) {
    float r_, m_;
    lx_row0.finish(p.lnx_invC, p.lnx_eps, r_, m_);
    lx_rs = f32x2{r_, m_};
  }
#endif
#ifdef RCDM_TRACE_LX
  if (p.trace) ts_c = __builtin_amdgcn_s_memtime() - ts0 + (long long)(lx_rs.x == 12345.f);   // ticks from kernel start to "statistics finished"
#endif
  int c_ks = 0, c_tile = 0;   // k-step / tile being computed
  int c_stage = 0;            // ring slot being computed
  int post_epi = 0;           // 2: an epilogue just ran (operands of the next step already landed), 1: its stores may be in flight

  constexpr int PIECES = AI + BI;  // DMA instructions per wave per step
  for (int g = 0; g < total; ++g) {
    // this wave's DMA pieces of step g have landed (loads retire in order: at most the pieces of the NSTAGE-2
    // younger steps may still be in flight); the barrier makes everybody's visible and also guarantees every
    // wave is done reading the slot of step g-1, which the issue below refills
    const int younger = min(NSTAGE - 2, total - 1 - g);
    const bool lx_now = lnx_k && c_ks == 0;
    if (post_epi == 2) {
      // first step of a new tile: every DMA piece issued so far was waited for inside the epilogue, BEFORE its output
      // stores were issued — nothing to wait for here, and the stores' round trip to L2 (vmcnt counts them, and they
      // retire out of order with loads, so a counted wait cannot skip them) is hidden under this step's MFMAs
      post_epi = 1;
    } else {
      if (post_epi == 1 || younger <= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (post_epi == 1: the epilogue's stores may still be counted)
      } else if (NSTAGE >= 4 && younger >= 2) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTAGE >= 4 ? 2 * PIECES : 0) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTAGE >= 3 ? PIECES : 0) : "memory");
      }
      post_epi = 0;
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (LXC && !(RCDM_LNX_ABLATE & 128)) if (lx_now) {   // (after the barrier: the previous tile's epilogue is done reading the table)
      if (t % LTPR == 0) *(f32x2*)(ltab + 2 * (t / LTPR)) = lx_rs;
      if (t < BN_ / 4) *(f32x4*)(lvS + 4 * t) = lx_s4;
    }
    // 8-wave blocks put two waves on every SIMD, released by the same barrier: if both issued their DMA pieces first
    // (each piece stalls the issuing wave for ~100 cycles) the SIMD's matrix pipe would idle through both bursts.
    // The second half of the block therefore issues after its first two k-chunks, under the first half's MFMAs.
    const bool more = g + NSTAGE - 1 < total;
    const int issue_kk = (NW == 8 && wave >= 4) ? 2 : 0;
    const char* sb = smem + c_stage * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if ((kk == 0 || (NW == 8 && kk == 2)) && more && issue_kk == kk) issue_next();
      f16x8 wf[FN], xf[FM];
#pragma unroll
      for (int i = 0; i < FN; ++i) wf[i] = *(const f16x8*)(sb + rowB + i * 32 * 128 + koff[kk]);
#pragma unroll
      for (int j = 0; j < FM; ++j) xf[j] = *(const f16x8*)(sb + rowA + j * 32 * 128 + koff[kk]);
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    const int e_stage = c_stage;  // slot just consumed
    c_stage = c_stage + 1 == NSTAGE ? 0 : c_stage + 1;
    if (++c_ks < nkl) continue;
    long long te0 = 0;
    if (p.trace) te0 = __builtin_amdgcn_s_memtime();

    // ---- tile done: epilogue.  The slot just consumed is idle until step g+NSTAGE is issued after the next
    // barrier, so it serves as the fp32 staging tile ([RP rows][BN] floats, 16-B chunks XOR-swizzled by row) for a
    // coalesced 16-byte-per-lane store phase; the DMA of the next tile's first step keeps flowing into the other
    // stage.  Raw s_barrier + lgkmcnt only: a __syncthreads() here would drain that DMA (vmcnt(0)).
    c_ks = 0;
    if constexpr (E16 && (LX & 2) != 0 && !(RCDM_LNX_ABLATE & 64)) {
      // deferred LayerNorm: x W'^T of the raw rows -> rstd acc - (mean rstd) S = LayerNorm(x) W'^T, on the accumulators,
      // before any of the epilogue's prefetches is live; everything below (bias b', row table, GEGLU, residual) runs unchanged
      // on top.  The table was written behind the first k-step's barrier: a one-step tile needs one more barrier here.
      if (nkl == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      f32x2 rs[FM];
#pragma unroll
      for (int j = 0; j < FM; ++j) rs[j] = *(const f32x2*)(ltab + 2 * ((wm * FM + j) * 32 + lr));
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 s4 = *(const f32x4*)(lvS + (wn * FN + i) * 32 + 8 * q + 4 * hi);   // S of this quad's 4 channels
#pragma unroll
          for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[i][j][4 * q + e] = __builtin_fmaf(acc[i][j][4 * q + e], rs[j].x, -rs[j].y * s4[e]);
        }
    }
    if constexpr (E16) {
      // ---- f16 staging (every launch without split-K): the accumulators are rounded to f16 (the reference's fp16
      // Linear / Conv outputs are rounded at the same point, give or take the bias) and staged as [rows][BN] halfs:
      // half the LDS bytes of an fp32 tile through ds_write_b64, and a 128x128 tile fits the consumed ring slot in ONE
      // pass (two barriers instead of four).  16-byte chunks are XOR-swizzled by (row >> 1) & 7 (two-way conflicts on
      // the 16 writes of a wave, none on the reads).  Bias, row vector, GELU / GEGLU, residual and scale are applied
      // in fp32 on the way out.  The post phase is VALU- and latency-bound (measured: ~3.4 k of a 128x128 tile's ~4.8 k
      // epilogue ticks), so (a) every global read it needs — bias, the row vector of the (at most two) samples the
      // tile touches, residual rows — is issued before the staging barriers and covered by ONE wait, so that no
      // per-item vmcnt wait ends up behind the previous item's store; (b) staged reads of all items are issued
      // together; (c) out = (h + bias + rowvec + res) * scale runs as v_fma_mix_f32 / v_fma_mixlo|hi_f16 on the f16
      // inputs directly: 2 instructions per output instead of cvt + add + add + mul + cvt.
      constexpr int NT = NW * 64;
      constexpr int RPH = (STAGE_BYTES / (BN_ * 2)) >= BM_ ? BM_ : (STAGE_BYTES / (BN_ * 2)) / 32 * 32;
      constexpr int NPASS = BM_ / RPH;
      static_assert(BM_ % RPH == 0 && RPH % 32 == 0, "staging passes must tile the block");
      f16* sH = (f16*)(smem + e_stage * STAGE_BYTES);
      constexpr int P_TPR = BN_ / 8, P_RPI = NT / P_TPR, P_ITEMS = RPH / P_RPI;
      constexpr int G_TPR = BN_ / 16, G_RPI = NT / G_TPR, G_ITEMS = RPH / G_RPI;
      static_assert(NT % P_TPR == 0 && RPH % P_RPI == 0 && NT % G_TPR == 0 && RPH % G_RPI == 0, "epilogue sweep");
      const int pc8 = t % P_TPR, pr0 = t / P_TPR;
      const int oc8 = t % G_TPR, gr0 = t / G_TPR;
      const int hc = (oc8 >> 1) * 4 + (oc8 & 1);  // GEGLU: 16-B chunk (8 halfs) of the hidden columns; the gate is 2 chunks on
      const int pn = geglu ? cn0 + hc * 8 : cn0 + pc8 * 8;   // first packed column this thread handles
      const bool pn_ok = pn < p.N;
      constexpr bool WHOLE = NPASS * P_ITEMS <= 8;
      constexpr int RCH = WHOLE ? P_ITEMS : (P_ITEMS < 4 ? P_ITEMS : 4);  // staged reads in flight per thread
      Pack16 resv[WHOLE ? NPASS : 1][P_ITEMS];
      const bool has_res = !geglu && (p.epi & RCDM_EPI_RESIDUAL);
      const bool has_rv = !geglu && (p.epi & RCDM_EPI_ROWVEC);
      // per-sample row vector: a tile of BM rows touches at most two samples when rows_per_sample >= BM
      const bool rv_pair = has_rv && p.rows_per_sample >= BM_;
      const int smp0 = rv_pair ? cm0 / p.rows_per_sample : 0;
      const int m_switch = rv_pair ? (smp0 + 1) * p.rows_per_sample : 0x7fffffff;
      float bA[8], bB[8], rvA[8], rvB[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bA[e] = bB[e] = rvA[e] = rvB[e] = 0.f;
      if ((p.epi & RCDM_EPI_BIAS) && pn_ok) {
        load8(p.bias + pn, bA);
        if (geglu) load8(p.bias + pn + kGegluGroup, bB);
      }
      if (rv_pair && pn_ok) {
        load8(p.rowvec + (size_t)smp0 * p.ldt + pn, rvA);
        if (m_switch < p.M && m_switch < cm0 + BM_) load8(p.rowvec + (size_t)(smp0 + 1) * p.ldt + pn, rvB);
      }
      // deferred LayerNorm of the A rows (rcdm_gemm_lnx): LTPR threads per tile row sum that row's partial statistics;
      // (rstd, mean rstd) go to a [BM][2] table behind the ring, published by the staging barrier of the first pass
      constexpr bool lnx = LXC;
      const bool stat_on = LX != 0 && p.stat_out != nullptr && !(RCDM_LNX_ABLATE & 256);
      const int stat_tn = cn0 / BN_;
      const int dup_rows = p.dup ? (int)(p.dup / p.ldc) : 0;
      if (WHOLE && !geglu) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
          for (int it = 0; it < P_ITEMS; ++it) {
            const int m = cm0 + ps * RPH + pr0 + it * P_RPI;
            resv[WHOLE ? ps : 0][it].u = make_uint4(0, 0, 0, 0);
            if (has_res && m < p.M && pn_ok) resv[WHOLE ? ps : 0][it].u = *(const uint4*)(p.res + (size_t)m * p.ldr + pn);
          }
      }
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        if (!WHOLE && !geglu) {
#pragma unroll
          for (int it = 0; it < P_ITEMS; ++it) {
            const int m = cm0 + ps * RPH + pr0 + it * P_RPI;
            resv[0][it].u = make_uint4(0, 0, 0, 0);
            if (has_res && m < p.M && pn_ok) resv[0][it].u = *(const uint4*)(p.res + (size_t)m * p.ldr + pn);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RCDM_PHASE_BEGIN();
        __builtin_amdgcn_s_barrier();  // operand reads (ps = 0) / previous pass's staged reads are complete
        RCDM_PHASE_STAMP(ts_a);
#pragma unroll
        for (int j = 0; j < FM; ++j) {
          const int blk = wm * FM + j;
          if (blk / (RPH / 32) == ps) {
            const int prow = (blk % (RPH / 32)) * 32 + lr;
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                union { f16 h[4]; uint2 u; } pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk.h[e] = (f16)acc[i][j][4 * q + e];
                const int piece = ((wn * FN + i) * 32 + 8 * q + 4 * hi) >> 2;
                *(uint2*)(sH + prow * BN_ + ((piece ^ (((prow >> 1) & 7) << 1)) << 2)) = pk.u;
              }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        RCDM_PHASE_STAMP(ts_b);
        RCDM_PRE_STORE_WAIT();
        RCDM_PHASE_STAMP(ts_c);
        const int mbase = cm0 + ps * RPH;
        const float sc = p.out_scale;
        if (geglu) {
          const int oc = geglu_out_col(pn);
#pragma unroll
          for (int it0 = 0; it0 < G_ITEMS; it0 += RCH) {
            Pack16 hh[RCH], gg[RCH];
#pragma unroll
            for (int k = 0; k < RCH; ++k)
              if (it0 + k < G_ITEMS) {
                const int row = gr0 + (it0 + k) * G_RPI;
                const int sx = (row >> 1) & 7;
                hh[k].u = *(const uint4*)(sH + row * BN_ + ((hc ^ sx) << 3));
                gg[k].u = *(const uint4*)(sH + row * BN_ + (((hc + 2) ^ sx) << 3));
              }
#pragma unroll
            for (int k = 0; k < RCH; ++k)
              if (it0 + k < G_ITEMS) {
                const int m = mbase + gr0 + (it0 + k) * G_RPI;
                if (m < p.M && pn_ok) {
                  Pack16 o;
                  o.u = geglu8(hh[k].u, gg[k].u, bA, bB, sc);
                  *(uint4*)(p.out + (size_t)m * p.ldc + oc) = o.u;
                  if (p.dup) *(uint4*)(p.out + (size_t)m * p.ldc + oc + p.dup) = o.u;
                }
              }
          }
        } else {
          const bool gelu_on = (p.epi & RCDM_EPI_GELU) != 0;
          float cA[8], cB[8];  // (bias + row vector) * scale of the tile's first / second sample
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            cA[e] = (bA[e] + rvA[e]) * sc;
            cB[e] = (bA[e] + rvB[e]) * sc;
          }
#pragma unroll
          for (int it0 = 0; it0 < P_ITEMS; it0 += RCH) {
            Pack16 hh[RCH];
            float stv1[RCH], stv2[RCH];
#pragma unroll
            for (int k = 0; k < RCH; ++k) stv1[k] = stv2[k] = 0.f;
#pragma unroll
            for (int k = 0; k < RCH; ++k)
              if (it0 + k < P_ITEMS) {
                const int row = pr0 + (it0 + k) * P_RPI;
                hh[k].u = *(const uint4*)(sH + row * BN_ + ((pc8 ^ ((row >> 1) & 7)) << 3));
              }
#pragma unroll
            for (int k = 0; k < RCH; ++k)
              if (it0 + k < P_ITEMS) {
                const int it = it0 + k;
                const int m = mbase + pr0 + it * P_RPI;
                float& st1 = stv1[k];          // producer side of a deferred LayerNorm: sums over this thread's 8 outputs
                float& st2 = stv2[k];
                if (m < p.M && pn_ok) {
                  const Pack16& rr = resv[WHOLE ? ps : 0][it];
                  Pack16 o;
                  if (gelu_on || (has_rv && !rv_pair)) {
                    // rare forms (stage-1 GELU feed-forward; row vector with fewer rows per sample than the tile): plain
                    // fp32 arithmetic
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (float)hh[k].e[e] + bA[e];
                    if (has_rv && rv_pair) {
                      const bool second = m >= m_switch;
#pragma unroll
                      for (int e = 0; e < 8; ++e) v[e] += second ? rvB[e] : rvA[e];
                    } else if (has_rv) {
                      float rv[8];
                      load8(p.rowvec + (size_t)(m / p.rows_per_sample) * p.ldt + pn, rv);
#pragma unroll
                      for (int e = 0; e < 8; ++e) v[e] += rv[e];
                    }
                    if (gelu_on) {
                      gelu8(v);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) o.e[e] = (f16)((v[e] + (float)rr.e[e]) * sc);
                  } else {
                    const bool second = m >= m_switch;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                      const float c0 = second ? cB[2 * d] : cA[2 * d], c1 = second ? cB[2 * d + 1] : cA[2 * d + 1];
                      const float t0 = mix_f16_f32(hh[k].v[d], 0, sc, c0);
                      const float t1 = mix_f16_f32(hh[k].v[d], 1, sc, c1);
                      o.v[d] = mix_f16_pack(rr.v[d], sc, t0, t1);
                    }
                  }
                  *(uint4*)(p.out + (size_t)m * p.ldc + pn) = o.u;
                  if (p.dup) *(uint4*)(p.out + (size_t)m * p.ldc + pn + p.dup) = o.u;
                  if (stat_on) {   // (v_dot2c_f32_f16 was tried for two elements per instruction: wrong sums on gfx950)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                      const float f = (float)o.e[e];
                      st1 += f;
                      st2 = __builtin_fmaf(f, f, st2);
                    }
                  }
                }
              }
            if (stat_on) {
              // wave-uniform; every lane of the P_TPR-lane row group takes part in the DPP sums.  The batch's 2 RCH reductions
              // are independent chains the scheduler interleaves; one store per row and column tile
#pragma unroll
              for (int k = 0; k < RCH; ++k) {
                stv1[k] = group_sum<P_TPR>(stv1[k]);
                stv2[k] = group_sum<P_TPR>(stv2[k]);
              }
              if (pc8 == 0) {
#pragma unroll
                for (int k = 0; k < RCH; ++k) {
                  const int m = mbase + pr0 + (it0 + k) * P_RPI;
                  if (it0 + k < P_ITEMS && m < p.M) {
                    *(f32x2*)(p.stat_out + ((size_t)stat_tn * p.stat_ld + m) * 2) = f32x2{stv1[k], stv2[k]};
                    if (dup_rows) *(f32x2*)(p.stat_out + ((size_t)stat_tn * p.stat_ld + m + dup_rows) * 2) = f32x2{stv1[k], stv2[k]};
                  }
                }
              }
            }
          }
        }
      }
    } else {
      // ---- split-K launches only: the fp32 tile goes to this split's slab (rcdm_*_workspace_bytes), staged through
      // the consumed ring slot ([RP rows][BN] floats, 16-B chunks XOR-swizzled by row) so that the slab is written
      // in coalesced 2 x 16-byte pieces per lane; bias / row vector / residual / GEGLU belong to splitk_reduce_kernel
      constexpr int RP = (STAGE_BYTES / (BN_ * 4)) >= 64 ? 64 : 32;   // rows per pass
      constexpr int NPASS = BM_ / RP;
      constexpr int NT = NW * 64;
      constexpr int TPR = BN_ / 8, RPI = NT / TPR, ITEMS = RP / RPI;  // threads per row, rows per sweep
      static_assert(NT % TPR == 0 && RP % RPI == 0, "epilogue sweep must tile the pass exactly");
      float* sC = (float*)(smem + e_stage * STAGE_BYTES);
      float* dst = p.partial + (size_t)blockIdx.y * p.M * p.N;
      const int c8 = t % TPR, r0 = t / TPR;
      const int n = cn0 + c8 * 8;
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // operand reads (ps = 0) / previous pass's staged reads are complete
#pragma unroll
        for (int j = 0; j < FM; ++j) {
          const int blk = wm * FM + j;
          if (blk / (RP / 32) == ps) {
            const int prow = (blk % (RP / 32)) * 32 + lr;
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int chunk = ((wn * FN + i) * 32 + 8 * q + 4 * hi) >> 2;
                f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                *(f32x4*)(sC + prow * BN_ + ((chunk ^ (prow & 7)) << 2)) = v;
              }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        RCDM_PRE_STORE_WAIT();
        const int mbase = cm0 + ps * RP;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
          const int row = r0 + it * RPI;
          const int m = mbase + row;
          if (m < p.M && n < p.N) {
            const f32x4 v0 = *(const f32x4*)(sC + row * BN_ + (((2 * c8) ^ (row & 7)) << 2));
            const f32x4 v1 = *(const f32x4*)(sC + row * BN_ + (((2 * c8 + 1) ^ (row & 7)) << 2));
            if (p.slab16) {   // f16 slab (IgemmArgs::slab16): the same [split][M][N] planes, halfs
              Pack16 h;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                h.e[e] = (f16)v0[e];
                h.e[4 + e] = (f16)v1[e];
              }
              *(uint4*)((f16*)p.partial + ((size_t)blockIdx.y * p.M + m) * p.N + n) = h.u;
            } else {
              *(f32x4*)(dst + (size_t)m * p.N + n) = v0;
              *(f32x4*)(dst + (size_t)m * p.N + n + 4) = v1;
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    ++c_tile;
    post_epi = 2;
    if (c_tile < my_tiles) {
      tile_of(c_tile, cm0, cn0);
      if constexpr (LXC) lx_fetch(cm0, cn0);
    }
    if (p.trace) ts_epi += __builtin_amdgcn_s_memtime() - te0;
  }
  if (p.trace && t == 0) {
    long long* o = p.trace + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * TRACE_SLOTS;
    o[0] = ts0;
    o[1] = __builtin_amdgcn_s_memtime();
    o[2] = ts_epi;
    o[3] = my_tiles * nkl;
    if (TRACE_SLOTS == 8) {
      o[4] = ts_a;
      o[5] = ts_b;
      o[6] = ts_c;
      o[7] = my_tiles;
    }
  }
  (void)ts_loop; (void)ts_a; (void)ts_b; (void)ts_c; (void)tq;
}

// split-K second pass: fixed-order sum of the fp32 slabs + the epilogue (8 output columns per thread).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const IgemmArgs p) {
  const bool geglu = (p.epi & RCDM_EPI_GEGLU) != 0;
  const int nout8 = (geglu ? p.N / 2 : p.N) / 8;
  const size_t total = (size_t)p.M * nout8;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(idx / nout8);
    const int oc = (int)(idx - (size_t)m * nout8) * 8;
    const int n = geglu ? (oc >> 4) * 32 + (oc & 15) : oc;
    float v[8], g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = g[e] = 0.f;
    for (int s = 0; s < p.splits; ++s) {
      const size_t off = ((size_t)s * p.M + m) * p.N + n;
      slab_add8(p, off, v);
      if (geglu) slab_add8(p, off + kGegluGroup, g);
    }
    epilogue_store(p, p.ph_rows ? phase_out_row(p, m) : m, n, v, g);   // (phase launches carry a bias at most: no per-row operand)
  }
}

// split-K second pass + the statistics pass of the GroupNorm that reads the result next (rcdm_*_gnstat): the grid, the
// thread -> (row, 16-byte chunk) mapping and the order of every addition are gn_stats_kernel's (norm.hip), so the partials —
// and with them the norm's output — are bit-identical to reduce + rcdm_groupnorm_silu.  Per element: the fixed-order sum of
// the fp32 slabs and the epilogue (what splitk_reduce_kernel does), the row stored as f16, and the statistics taken from the
// STORED halfs (what gn_stats_kernel would read back).  One launch and one read of the tensor less per norm.
// grid (gn_splits, gn_samples); block gn_CH * gn_RPB threads; no GEGLU, no phase rows (checked by the launcher).
__global__ __launch_bounds__(512) void splitk_reduce_gn_kernel(const IgemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* part = (float*)smem;
  const int t = threadIdx.x;
  const int ch = t % p.gn_CH, rl = t / p.gn_CH;
  const int s = blockIdx.y, sp = blockIdx.x;
  const int r_begin = sp * p.gn_rps;
  const int r_end = min(p.gn_P, r_begin + p.gn_rps);
  float sum[8], sq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sum[e] = sq[e] = 0.f;
  const int n = ch * 8;
  // GN_U rows per thread and pass, as in gn_stats_kernel (the split is sized so that this is normally the block's only pass):
  // per slab, the loads of all GN_U rows are requested before the first addition — `splits` round trips per pass
  for (int r = r_begin + rl; r < r_end; r += GN_U * p.gn_RPB) {
    float v[GN_U][8];
#pragma unroll
    for (int u = 0; u < GN_U; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[u][e] = 0.f;
    // (both slabs of a split-2 launch requested together — 210 VGPRs — measured +0.06 ms per step against this form)
    for (int k = 0; k < p.splits; ++k) {
      if (p.slab16) {
        Pack16 h[GN_U];
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
          const int ru = min(r + u * p.gn_RPB, r_end - 1);
          h[u].u = *(const uint4*)((const f16*)p.partial + ((size_t)k * p.M + (size_t)s * p.gn_P + ru) * p.N + n);
        }
#pragma unroll
        for (int u = 0; u < GN_U; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[u][e] += (float)h[u].e[e];
        continue;
      }
      f32x4 a[GN_U], b[GN_U];
#pragma unroll
      for (int u = 0; u < GN_U; ++u) {
        const int ru = min(r + u * p.gn_RPB, r_end - 1);
        const float* src = p.partial + ((size_t)k * p.M + (size_t)s * p.gn_P + ru) * p.N + n;
        a[u] = *(const f32x4*)src;
        b[u] = *(const f32x4*)(src + 4);
      }
#pragma unroll
      for (int u = 0; u < GN_U; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[u][e] += a[u][e];
          v[u][4 + e] += b[u][e];
        }
    }
#pragma unroll
    for (int u = 0; u < GN_U; ++u) {
      const int ru = r + u * p.gn_RPB;
      if (ru < r_end) {
        float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // (GEGLU gates: never with statistics)
        Pack16 o;
        o.u = epilogue_store(p, s * p.gn_P + ru, n, v[u], g);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = (float)o.e[e];
          sum[e] += f;
          sq[e] += f * f;
        }
      }
    }
  }
  GnArgs ga{};
  ga.partial = p.gn_partial; ga.G = p.gn_G; ga.cg = p.gn_cg; ga.CH = p.gn_CH; ga.RPB = p.gn_RPB; ga.splits = p.gn_splits;
  gn_block_partials(ga, part, t, sum, sq, s, sp, r_end - r_begin);
}

// the reduce launch of a split-K GEMM / conv (every kernel family writes the same [split][M][N] fp32 slabs)
int launch_splitk_reduce(const IgemmArgs& a, hipStream_t stream) {
  if (a.gn_partial) {
    const int threads = a.gn_CH * a.gn_RPB;
    const size_t lds = (size_t)(threads + a.gn_CH) * 16 * sizeof(float);
    hipLaunchKernelGGL(splitk_reduce_gn_kernel, dim3(a.gn_splits, a.gn_samples), dim3(threads), lds, stream, a);
    return rcdm_check_launch();
  }
  const bool geglu = (a.epi & RCDM_EPI_GEGLU) != 0;
  const size_t total = (size_t)a.M * ((geglu ? a.N / 2 : a.N) / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, a);
  return rcdm_check_launch();
}

// variant: 1 = 128x128 (2 blocks/CU), 2 = 256x256 (1), 3 = 64x64 two-slot ring (4), 4 = 64x64 four-slot ring (2),
// 5 = 128x64 (3); 6 / 7 / 8 = the ping-pong kernel of igemm8.hip at 160x320 / 160x256 / 256x256; 9 = the 160x160 kernel of
// igemm16.hip (2); 10 = 128x64 with a three-slot ring (2; GEMMs)   (-1 = heuristic)
int g_force_variant = -1;
// f16 split-K slabs on the 160x160 and the LDS-DMA kernels (IgemmArgs::slab16): -1 = not set (environment RCDM_SLAB16, default below)
int g_slab16 = -1;
int slab16_mode() {
  if (g_slab16 < 0) {
    const char* e = getenv("RCDM_SLAB16");
    g_slab16 = e ? (atoi(e) != 0) : 1;   // (five same-box pairs: -0.11 ms per step, whole-UNet error unchanged)
  }
  return g_slab16;
}
struct TileCfg { int bm, bn, blocks_per_cu; };
constexpr int kFirstPP = 6;
constexpr int kVar16 = 9;  // igemm16.hip: 160x160, two blocks per CU
// 10: the igemm_dma loop at 128x64 with a THREE-slot LDS ring, two blocks per CU (GEMMs only; a conv runs as 5).  Measured
// in the replayed graph (profiles/r4_shape_rules_ab.txt): -0.09 ms per step on the thirty-five N = C = K = 1280 projections of
// the 16x16 level, worse everywhere else — as were a four-slot 128x64 ring and a three-slot 128x128 ring (one block per CU
// each; built, tested, removed): one more stage in flight pays only where it does not cost the third co-resident block
// more than the latency it hides.
constexpr int kFirstDeep = 10, kNumVariants = 11;
const TileCfg kTiles[kNumVariants] = {{128, 128, 2}, {128, 128, 2}, {256, 256, 1}, {64, 64, 4}, {64, 64, 2}, {128, 64, 3},
                                      {160, 320, 1}, {160, 256, 1}, {256, 256, 1}, {160, 160, 2},
                                      {128, 64, 2}};
inline bool is_pp(int v) { return v >= kFirstPP && v < kVar16; }
inline bool is_dma(int v) { return v < kFirstPP || v >= kFirstDeep; }
int g_pp_mode = -1;  // RCDM_PP=0: never pick the ping-pong kernel (A/B switch)
int g_num_cus = 0;
long long* g_trace = nullptr;

int num_cus() {
  if (g_num_cus <= 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      g_num_cus = n;
    else
      g_num_cus = 256;
  }
  return g_num_cus;
}

// Split-K for the ping-pong kernel: its tiles are big, so shapes with fewer tiles than CUs (M = 10240 / 2560 rows with
// N = 640 / 1280) are cut along K until one round of the chip is full; a slice keeps >= 16 k-steps.
// taps of a launch: 1 (GEMM), 9 (conv3x3), 4 (phase form); the W row holds taps * Cin columns (+ Cin2 of a second input)
inline int taps_of(const IgemmArgs& a) { return (a.Ktot - a.Cin2) / a.Cin; }

int pp_splits(int tiles, int nk) {
  int s = num_cus() / (tiles > 0 ? tiles : 1);
  if (s > nk / 16) s = nk / 16;
  if (s > 8) s = 8;
  return s < 1 ? 1 : s;
}

// The ping-pong kernel (igemm8.hip).  Measured against the 128x128 / 256x256 one-barrier kernels (tools/kbench.py,
// profiles/r2_pp_kbench.txt): it wins where its big tile comes out as whole rounds of the chip AND the k-loop is long
// enough to amortise a prologue / epilogue that nothing overlaps (one block per CU): the conv3x3 of the 64x64 level
// (160x320: exactly 256 tiles, 1.37-1.42x), the other convs with >= 2560 rows (1.04-1.07x), and the M = 40960 GEMMs
// with N <= 960 (qkv 1.13x, feed-forward out 1.12x).  The K = 640 / 1280 GEMMs of the 32x32 / 16x16 levels stay on the
// two-blocks-per-CU kernel, whose second block hides the epilogue.  Returns the shape index or -1.
int pick_pp(const IgemmArgs& a) {
  const int taps = taps_of(a);
  const int nk = (a.Cin + BK - 1) / BK * taps;
  if (a.M < 2048 || a.N < 256) return -1;
  if (taps == 1) {
    if (a.M < 20480 || a.N > 1024) return -1;
    // (nk 10 / 15: the 1x1 shortcut convs of the 64x64 level, tools/autotune.py + same-box A/B in the graph)
    if (!(nk >= 10 || (a.N >= 640 && nk >= 5))) return -1;
  } else if (nk < 40) {
    return -1;
  }
  const int cus = num_cus();
  int best = -1;
  float best_score = 0.80f;
  for (int sh = 0; sh < kNumPPShapes; ++sh) {
    const int bm = kPPShapes[sh].bm, bn = kPPShapes[sh].bn;
    const int tm = (a.M + bm - 1) / bm, tn = (a.N + bn - 1) / bn, tiles = tm * tn;
    const int sp = tiles < cus ? pp_splits(tiles, nk) : 1;
    const int work = tiles * sp, rounds = (work + cus - 1) / cus;
    const float useful = (float)a.M * (float)a.N / ((float)tiles * bm * bn);
    const float fill = (float)work / (float)(rounds * cus);
    float score = useful * fill * (sp > 1 ? 0.90f : 1.0f);
    if (sh == 2) score *= 1.03f;  // 256x256 moves fewer operand bytes per flop
    if (score > best_score) {
      best_score = score;
      best = sh;
    }
  }
  return best;
}

// The 160x160 two-blocks-per-CU kernel (igemm16.hip), measured against every other variant (tools/kbench.py,
// profiles/r2_kbench.txt): it wins on the wide-N GEMMs with K <= 1280 and >= 2560 rows — fused [q;k;v] and GEGLU
// projections of the 64x64 / 32x32 / 16x16 levels: 7-15 % (no padded columns at N = 960 / 1920, 20 % fewer operand bytes
// per flop than 128x128, and unlike the ping-pong kernel its epilogue hides under the CU's other block) — and on the convs
// of the 32x32 level (3 % over the ping-pong kernel, which needs split-K there).  N = C GEMMs (HBM-bound or too few
// tiles), K >= 2560 (split-K shapes) and the 8x8 level stay where they were.
bool pick_16(const IgemmArgs& a) {
  static int mode = -1;  // RCDM_I16=0: never (A/B switch)
  if (mode < 0) {
    const char* e = getenv("RCDM_I16");
    mode = e ? atoi(e) : 1;
  }
  if (!mode) return false;
  const int taps = taps_of(a);
  if (taps == 1) {
    // plain projections (no epilogue work: the staged halfs are copied out) that split into whole rounds of 160x160 tiles:
    // the cross-attention queries of the 32x32 level and of the shared-prefix half batch (tools/autotune.py: 17.3 -> 13.8 us)
    const bool plain = a.epi == 0 || (a.lnx_stat && !(a.epi & (RCDM_EPI_RESIDUAL | RCDM_EPI_GELU | RCDM_EPI_GEGLU)));
    if (plain && a.out_scale == 1.0f && a.M % 160 == 0 && a.N % 160 == 0 && (a.M / 160) * (a.N / 160) >= 256 &&
        a.Cin <= 640 && a.N <= 640)
      return true;
    return a.N >= 960 && a.N >= 3 * a.Cin && a.M >= 2048 && a.Cin <= 1280;  // wide N only: qkv (3C), GEGLU (8C)
  }
  return a.M >= 5120 && a.M < 20480 && a.N >= 640 && a.N <= 1280;
}

// Per-shape overrides of the heuristics below: {taps, M, N, C_in} -> tile variant (1 .. 10) and split-K factor (0 = that
// variant's own heuristic).  kShapeRules holds what tools/autotune.py found AND a same-box A/B of the whole step
// confirmed; RCDM_SHAPE_RULES="taps,M,N,Cin,variant,split;..." adds rules at run time (first match wins: the environment's
// rules are looked at first), RCDM_SHAPE_RULES=off ignores the table — for tuning another chip or another model without
// a rebuild.  A rule is skipped where its variant cannot run the launch (row statistics: variants 1 .. 5; deferred-
// LayerNorm consumers: not the ping-pong kernel).
struct ShapeRule { int taps, M, N, Cin, variant, split; };
const ShapeRule kShapeRules[] = {
    // the headline workload (b = 2 x 5 frames, 64x64 latents), round 4: profiles/r4_autotune.txt, profiles/r4_shape_rules_ab.txt
    {9, 40960, 320, 320, kVar16, 1},   // the convs of the 64x64 level on 160x160 tiles, two blocks per CU, instead of the
    {9, 40960, 320, 640, kVar16, 1},   //   160x320 ping-pong tile (-3 % back to back, -0.15 ms per step together in the graph)
    {9, 40960, 320, 960, kVar16, 1},
    {9, 20480, 320, 320, kVar16, 0},   // ... and of the shared-prefix half batch
    {1, 2560, 1280, 1280, 10, 0},      // the N = C projections of the 16x16 level (to_out, proj_in, to_q) on the three-slot 128x64 ring: -0.09 ms
    {1, 640, 1280, 2560, 3, 3},        // 1x1 shortcuts of the 8x8 up blocks: 64x64 tiles split 3 ways
    // tools/tune_rules.py (every shape timed inside the step's launch sequence), then tools/ab_rules.sh: -0.07 ms together
    {1, 2560, 1280, 6400, kVar16, 0},  // proj_out-composed feed-forward GEMMs (K = 5C): 16x16 level on 160x160 tiles,
    {1, 10240, 640, 3200, 1, 0},       //   32x32 level on 128x128, 8x8 level on the three-slot 128x64 ring
    {1, 640, 1280, 6400, 10, 0},
    {1, 2560, 1280, 2560, 10, 0},      // 1x1 shortcuts / downsample-level projections of the 16x16 level: three-slot 128x64 ring
    {1, 2560, 1280, 1920, 10, 0},
    {1, 2560, 1280, 640, 10, 0},
    {1, 40960, 320, 640, kVar16, 0},   // 1x1 shortcuts of the 64x64 up blocks on 160x160 tiles
    {1, 40960, 320, 960, kVar16, 0},
    // the shapes only BASELINE config 1's plan has (b = 2 x 5 frames, 32x32 latents): tools/tune_rules.py --latent 32, confirmed
    // together in the graph (183.1 -> 173.9 ms per 20-step story; profiles/r4_shape_rules_ab.txt)
    {1, 2560, 640, 3200, 4, 0}, {1, 10240, 320, 1600, 10, 0}, {1, 2560, 640, 640, 4, 0}, {9, 160, 1280, 1280, 5, 0},
    {1, 160, 1280, 6400, 10, 0}, {1, 160, 10240, 1280, 4, 0}, {9, 10240, 320, 640, kVar16, 0},
    {1, 160, 1280, 2560, 10, 4}, {9, 10240, 320, 960, kVar16, 4}, {9, 2560, 640, 1920, kVar16, 0},
    {1, 10240, 320, 320, 10, 0}, {9, 2560, 640, 1280, kVar16, 0}, {9, 2560, 640, 960, kVar16, 8},
    {9, 5120, 320, 320, 1, 4}, {9, 640, 1280, 640, 1, 8}, {9, 2560, 640, 320, 1, 4}, {9, 640, 640, 640, 1, 8},
    {9, 2560, 320, 320, kVar16, 8}, {1, 2560, 640, 1920, 4, 0}, {1, 640, 1280, 1920, 10, 4},
    {9, 10240, 8, 320, kVar16, 8}, {1, 10240, 320, 960, 10, 0}, {1, 2560, 640, 1280, 4, 0}, {9, 5120, 320, 64, 5, 0},
    {1, 2560, 640, 960, 4, 0}, {1, 5120, 320, 320, 4, 0}, {1, 640, 1280, 640, 4, 0}, {1, 2560, 640, 320, 4, 0},
    // the shapes only BASELINE config 3's plan has (4 stories = b 8, 64x64 latents, L = 91): --stories 4 --ctx-len 91, confirmed
    // together (2741 -> 2683 ms per story batch): at this batch the 160x160 two-blocks-per-CU kernel beats the ping-pong tiles on
    // every conv of the 64x64 / 32x32 levels in the step's own sequence, not back to back
    {1, 40960, 5120, 640, 2, 0}, {1, 10240, 1280, 6400, kVar16, 1}, {9, 40960, 640, 640, kVar16, 0},
    {9, 163840, 320, 640, kVar16, 1}, {9, 163840, 320, 960, kVar16, 0}, {9, 40960, 640, 1920, kVar16, 1},
    {9, 40960, 640, 1280, kVar16, 1}, {9, 40960, 640, 960, kVar16, 0}, {1, 10240, 1280, 1280, kVar16, 1},
    {9, 40960, 640, 320, kVar16, 1}, {1, 10240, 1280, 2560, kVar16, 1}, {1, 40960, 640, 1920, kVar16, 0},
    {1, 40960, 640, 1280, kVar16, 1}, {9, 81920, 320, 64, kVar16, 1}, {1, 10240, 1280, 1920, kVar16, 0},
    {1, 40960, 640, 320, 5, 1}, {1, 81920, 320, 320, 5, 1}, {1, 10240, 1280, 640, kVar16, 0},
    // BASELINE config 5 (stage-1 prior, 970 token rows): tools/tune_rules.py --prior, confirmed with tools/bench_prior.py (1.83 -> 1.92 stories/s)
    {1, 970, 2048, 2048, 4, 0}, {1, 970, 6144, 2048, 5, 1}, {1, 10, 1280, 2048, 10, 4},
    {0, 0, 0, 0, 0, 0},   // (terminator)
};
constexpr int kMaxEnvRules = 128;
ShapeRule g_env_rules[kMaxEnvRules];
int g_n_env = -1;            // -1: RCDM_SHAPE_RULES not parsed yet
bool g_rule_table_on = true;
char g_rules_text[8192] = "";
bool g_rules_from_api = false;
const ShapeRule* find_shape_rule(const IgemmArgs& a) {
  ShapeRule* env_rules = g_env_rules;
  int& n_env = g_n_env;
  bool& table_on = g_rule_table_on;
  if (n_env < 0) {
    int n = 0;
    const char* e = g_rules_from_api ? g_rules_text : getenv("RCDM_SHAPE_RULES");
    table_on = true;
    if (e && !strcmp(e, "off")) {
      table_on = false;
    } else if (e) {
      while (*e && n < kMaxEnvRules) {
        ShapeRule r{};
        int used = 0;
        if (sscanf(e, "%d,%d,%d,%d,%d,%d%n", &r.taps, &r.M, &r.N, &r.Cin, &r.variant, &r.split, &used) == 6 && r.variant >= 1 &&
            r.variant < kNumVariants && r.split >= 0)
          env_rules[n++] = r;
        e += used;
        while (*e && *e != ';') ++e;
        if (*e == ';') ++e;
        if (!used) break;
      }
    }
    n_env = n;
  }
  if (a.ph_rows) return nullptr;
  const int taps = taps_of(a);
  auto fits = [&](const ShapeRule& r) {
    if (r.taps != taps || r.M != a.M || r.N != a.N || r.Cin != a.Cin) return false;
    if (a.stat_out && !is_dma(r.variant)) return false;
    if (a.lnx_stat && is_pp(r.variant)) return false;
    if ((a.stat_out || a.lnx_stat) && r.split > 1) return false;
    return true;
  };
  for (int i = 0; i < n_env; ++i)
    if (fits(env_rules[i])) return &env_rules[i];
  if (table_on)
    for (const ShapeRule* r = kShapeRules; r->variant; ++r)
      if (fits(*r)) return r;
  return nullptr;
}

int pick_variant(const IgemmArgs& a) {
  if (g_force_variant < 0) {
    const char* e = getenv("RCDM_IGEMM");
    g_force_variant = 99;
    if (e && !strcmp(e, "dma128")) g_force_variant = 1;
    if (e && !strcmp(e, "dma256")) g_force_variant = 2;
    if (e && !strcmp(e, "dma64")) g_force_variant = 3;
  }
  if (g_force_variant != 99) return g_force_variant == 0 ? 1 : g_force_variant;
  if (const ShapeRule* r = find_shape_rule(a)) return r->variant;
  if (g_pp_mode < 0) {
    const char* e = getenv("RCDM_PP");
    g_pp_mode = e ? atoi(e) : 1;
  }
  if (g_pp_mode && !a.stat_out) {   // row statistics come out of the igemm_dma epilogue only
    if (pick_16(a)) return kVar16;
    const int pp = pick_pp(a);
    if (pp >= 0) return kFirstPP + pp;
  }
  // measured (tools/kbench.py, MI355X).  128x128 with two blocks per CU is the default.  Shapes that leave most CUs
  // without a 128x128 tile (8x8 / 16x16 levels, context K/V projections) run as 64x64 or 128x64 tiles so that several
  // blocks per CU keep more DMA in flight; N = 320 / 960 (half a 128-wide tile wasted) with a short K take 128x64;
  // the deep-K convs of the 32x32 / 16x16 levels take 256x256.
  const int nk = (a.Cin + BK - 1) / BK * taps_of(a) + (a.Cin2 + BK - 1) / BK;
  if (a.Ktot != a.Cin) {
    if (a.N <= 64) return 5;  // conv_out (4 -> 8 channels): half the weight tile of 128x128 is padding (51 -> 28 us)
    return 1;   // (the 256x256 LDS-DMA tile's conv instantiation spills 48 B: only when forced; the ping-pong 256x256 tile covers its shapes)
  }
  const int t128 = ((a.M + 127) / 128) * ((a.N + 127) / 128);
  if (t128 <= 64 && nk <= 24) return 4;  // (the two-slot 64x64 ring is 8-10 % faster back to back, tools/autotune.py, but +0.1 ms per step in the graph)
  if (nk >= 20 && nk < 40 && a.N >= 512 && a.M >= 512) {
    // 256x256 (staggered 8-wave loop, ~8 % faster per flop) when its last round of tiles is not emptier than 128x128's
    const int cus = num_cus();
    const int t256 = ((a.M + 255) / 256) * ((a.N + 255) / 256);
    const float e1 = (float)t128 / (float)(((t128 + 2 * cus - 1) / (2 * cus)) * 2 * cus);
    const float e2 = (float)t256 / (float)(((t256 + cus - 1) / cus) * cus);
    if (1.08f * e2 > e1 + 0.01f) return 2;
  }
  if (nk >= 40) return 1;  // deep K: split-K over 128x128 tiles fills the chip
  if (t128 <= 160) return 3;
  if (t128 <= 256) return 5;
  if ((a.N % 128) == 64 && nk <= 10) return 5;
  return 1;
}

// Split-K only pays when (a) all tiles x splits still run as ONE round of resident blocks (a second, partly filled
// round costs more than the idle CUs it fills) and (b) every slice keeps >= ~20 k-steps, because the fp32 slabs and
// the reduce pass are not free (measured with tools/splitk_test.py: e.g. M=2560 N=1280 K=1280 is 21 us unsplit and
// 33 us split 3 ways; the 8x8-level convs (50 tiles, 180 k-steps) drop from 150 us to 40 us split 8 ways).
int plan_splits(int tiles, int slots, int nk, int requested) {
  if (requested == 1) return 1;
  if (requested > 1) return requested < nk ? requested : nk;
  int s = slots / (tiles > 0 ? tiles : 1);
  if (s > nk / 20) s = nk / 20;
  if (s > 16) s = 16;
  if (s < 1) s = 1;
  return s;
}

int fill_common(IgemmArgs& a, int requested_split, int* variant_out = nullptr, int forced_variant = -1) {
  int variant = forced_variant >= 0 ? forced_variant : pick_variant(a);
  // the 256x256 LDS-DMA tile is at the 256-register cap: its statistics / deferred-LayerNorm instantiations spilled (12 /
  // 200 B of scratch) and are not built — such launches take the 128x128 tile, also when variant 2 is forced
  if (variant == 2 && (a.stat_out || a.lnx_stat)) variant = 1;
  if (variant_out) *variant_out = variant;
  const TileCfg& tc = kTiles[variant];
  a.tilesM = (a.M + tc.bm - 1) / tc.bm;
  a.tilesN = (a.N + tc.bn - 1) / tc.bn;
  a.kc = (a.Cin + BK - 1) / BK;
  const int taps = taps_of(a);
  a.nk = taps * a.kc;
  a.nk1 = kNoSeg2;
  if (a.Cin2 > 0) {          // second input: its k-steps follow the nine taps' (from_conv: Cin and Cin2 are multiples of BK)
    a.nk1 = a.nk;
    a.nk += a.Cin2 / BK;
  }
  int s;
  const ShapeRule* rule = (g_force_variant == 99 && requested_split <= 0 && forced_variant < 0) ? find_shape_rule(a) : nullptr;
  if (rule && rule->split > 0) {
    s = rule->split;
  } else if (is_pp(variant) && requested_split <= 0) {
    const int tiles = a.tilesM * a.tilesN;
    s = tiles < num_cus() ? pp_splits(tiles, a.nk) : 1;
  } else {
    s = plan_splits(a.tilesM * a.tilesN, num_cus() * tc.blocks_per_cu, a.nk, requested_split);
  }
  if (s > a.nk) s = a.nk;
  a.nk_per_split = (a.nk + s - 1) / s;
  a.splits = (a.nk + a.nk_per_split - 1) / a.nk_per_split;
  return RCDM_OK;
}

// A statistics producer whose caller asks for another slot count than this shape's own tile choice gives (two producers
// that fill ONE statistics buffer — e.g. the same projection run on all rows and on a row subset — must agree on it): the
// LDS-DMA tile whose column-tile count is `parts` (64- or 128-wide tiles), -1 when there is none.
int variant_for_parts(const IgemmArgs& a, int parts) {
  const int n64 = (a.N + 63) / 64, n128 = (a.N + 127) / 128;
  if (parts == n128) return 1;
  if (parts == n64) return ((a.M + 127) / 128) * n64 < num_cus() ? 4 : 5;
  return -1;
}

int check_common(const IgemmArgs& a) {
  if (!a.A || !a.W || !a.out) return RCDM_EINVAL;
  if (a.M <= 0 || a.N <= 0 || a.Cin <= 0) return RCDM_EINVAL;
  if ((a.Cin & 7) || (a.N & 7) || (a.lda & 7) || (a.ldc & 7)) return RCDM_ESHAPE;
  if ((a.epi & RCDM_EPI_BIAS) && !a.bias) return RCDM_EINVAL;
  if ((a.epi & RCDM_EPI_ROWVEC) && (!a.rowvec || a.rows_per_sample <= 0 || (a.ldt & 3))) return RCDM_EINVAL;
  if ((a.epi & RCDM_EPI_RESIDUAL) && (!a.res || (a.ldr & 7))) return RCDM_EINVAL;
  if ((a.epi & RCDM_EPI_GEGLU) && (a.N % 32)) return RCDM_ESHAPE;
  if ((a.epi & RCDM_EPI_GEGLU) && (a.epi & RCDM_EPI_GELU)) return RCDM_EINVAL;
  if (a.dup < 0) return RCDM_EINVAL;
  // buffer-load offsets are 32-bit with 0x80000000 reserved as "out of range"
  const size_t in_rows = (a.Ktot == a.Cin) ? (size_t)a.M : (size_t)(a.M / (a.Ho * a.Wo)) * a.Hi * a.Wi;
  if (in_rows * (size_t)a.lda * 2 >= 0x7FFFFFFFull || (size_t)a.N * a.Ktot * 2 >= 0x7FFFFFFFull) return RCDM_ESHAPE;
  return RCDM_OK;
}

template <typename K>
void set_lds(K kernel, int bytes) {
  (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <int TAPS>
int launch(IgemmArgs& a, int variant, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  constexpr int LTAB = 3072;                       // behind the ring, deferred LayerNorm: [BM][2] floats (rstd, mean rstd) + [BN] floats S
  constexpr int LDS_128 = 2 * (128 + 128) * 128 + LTAB;   // 64 KB  (two blocks per CU)
  constexpr int LDS_256 = 2 * (256 + 256) * 128 + LTAB;   // 128 KB
  constexpr int LDS_64 = 2 * (64 + 64) * 128 + LTAB;      // 32 KB  (four blocks per CU)
  constexpr int LDS_64D = 4 * (64 + 64) * 128 + LTAB;     // 64 KB  (three steps in flight)
  constexpr int LDS_128x64 = 2 * (128 + 64) * 128 + LTAB; // 48 KB  (three blocks per CU)
  constexpr int LDS_128x64_3 = 3 * (128 + 64) * 128 + LTAB;   // 72 KB (two)
  static bool attr_set[64] = {};
  if (rcdm_first_on_device(attr_set)) {
    set_lds(igemm_dma_kernel<TAPS, 128, 128, 2, 2, 2, false>, LDS_128);
    set_lds(igemm_dma_kernel<TAPS, 256, 256, 2, 4, 2, false>, LDS_256);
    set_lds(igemm_dma_kernel<TAPS, 64, 64, 2, 2, 2, false>, LDS_64);
    set_lds(igemm_dma_kernel<TAPS, 64, 64, 2, 2, 4, false>, LDS_64D);
    set_lds(igemm_dma_kernel<TAPS, 128, 64, 2, 2, 2, false>, LDS_128x64);
    set_lds(igemm_dma_kernel<TAPS, 128, 128, 2, 2, 2, true>, LDS_128);
    set_lds(igemm_dma_kernel<TAPS, 256, 256, 2, 4, 2, true>, LDS_256);
    set_lds(igemm_dma_kernel<TAPS, 64, 64, 2, 2, 2, true>, LDS_64);
    set_lds(igemm_dma_kernel<TAPS, 64, 64, 2, 2, 4, true>, LDS_64D);
    set_lds(igemm_dma_kernel<TAPS, 128, 64, 2, 2, 2, true>, LDS_128x64);
    if constexpr (TAPS == 1) {
#define RCDM_DEEP_LDS(BM, BN, NS, LDS)                                                      \
      set_lds(igemm_dma_kernel<TAPS, BM, BN, 2, 2, NS, false>, LDS);                        \
      set_lds(igemm_dma_kernel<TAPS, BM, BN, 2, 2, NS, true>, LDS);                         \
      set_lds(igemm_dma_kernel<TAPS, BM, BN, 2, 2, NS, true, 1>, LDS);                      \
      set_lds(igemm_dma_kernel<TAPS, BM, BN, 2, 2, NS, true, 2>, LDS)
      RCDM_DEEP_LDS(128, 64, 3, LDS_128x64_3);
#undef RCDM_DEEP_LDS
      set_lds(igemm_dma_kernel<TAPS, 128, 128, 2, 2, 2, true, 1>, LDS_128);
      set_lds(igemm_dma_kernel<TAPS, 64, 64, 2, 2, 2, true, 1>, LDS_64);
      set_lds(igemm_dma_kernel<TAPS, 64, 64, 2, 2, 4, true, 1>, LDS_64D);
      set_lds(igemm_dma_kernel<TAPS, 128, 64, 2, 2, 2, true, 1>, LDS_128x64);
      set_lds(igemm_dma_kernel<TAPS, 128, 128, 2, 2, 2, true, 2>, LDS_128);
      set_lds(igemm_dma_kernel<TAPS, 64, 64, 2, 2, 2, true, 2>, LDS_64);
      set_lds(igemm_dma_kernel<TAPS, 64, 64, 2, 2, 4, true, 2>, LDS_64D);
      set_lds(igemm_dma_kernel<TAPS, 128, 64, 2, 2, 2, true, 2>, LDS_128x64);
    }
  }
  if (a.splits > 1) {
    const size_t need = (size_t)a.splits * a.M * a.N * sizeof(float);
    if (!workspace || workspace_bytes < need) return RCDM_EWORKSPACE;
    a.partial = (float*)workspace;
  } else {
    a.partial = nullptr;
  }
  a.trace = g_trace;
  a.dbg = 0;
  if (a.stat_out && (!is_dma(variant) || a.splits > 1 || (a.epi & RCDM_EPI_GEGLU) || a.stat_parts != a.tilesN))
    return RCDM_ESHAPE;
  if ((a.stat_out || a.lnx_stat) && TAPS != 1) return RCDM_ESHAPE;
  if (a.lnx_stat && (a.splits > 1 || !a.lnx_S || a.lnx_parts < 1 || a.lnx_parts > kLnxMaxParts)) return RCDM_ESHAPE;
  if (a.lnx_stat && variant == kFirstPP + 2) return RCDM_ESHAPE;   // the 256x256 ping-pong tile has no consumer epilogue (register cap); only reachable when that variant is forced
  if (variant == kVar16) {
    a.slab16 = a.splits > 1 && slab16_mode() && !(a.epi & RCDM_EPI_GEGLU);
    int rc = rcdm_igemm16_launch(a, TAPS, stream);
    if (rc) return rc;
    if (a.splits > 1) {
      rc = launch_splitk_reduce(a, stream);
    }
    return rc;
  }
  if (is_pp(variant)) {
    static int rotate = -1;
    if (rotate < 0) {
      const char* e = getenv("RCDM_PP_ROTATE");
      rotate = e ? atoi(e) : 0;   // measured: no gain (the weight stream is not hot-spotting L2 channels)
    }
    a.dbg = rotate ? 8 : 0;
    int rc = rcdm_igemm_pp_launch(a, TAPS, variant - kFirstPP, stream);
    if (rc) return rc;
    if (a.splits > 1) {
      rc = launch_splitk_reduce(a, stream);
    }
    return rc;
  }
  a.slab16 = a.splits > 1 && slab16_mode() && !(a.epi & RCDM_EPI_GEGLU);
  const int ntiles = a.tilesM * a.tilesN;
  int gx = num_cus() * kTiles[variant].blocks_per_cu;
  if (a.splits > 1) gx = (gx + a.splits - 1) / a.splits;
  gx = (gx + 7) / 8 * 8;  // keep the b%8 -> XCD pattern aligned across the persistent stride
  static int persist_mode = -1;  // RCDM_PERSIST=0: one tile per block (A/B switch); default: persistent blocks
  if (persist_mode < 0) {
    const char* e = getenv("RCDM_PERSIST");
    persist_mode = e ? atoi(e) : 1;
  }
  // (GEGLU launches used to run one tile per block: a persistent block had to drain its epilogue stores before it
  // could trust the next tile's DMA.  With the loads waited for BEFORE the stores that stall is gone.)
  if (persist_mode == 0) gx = ntiles;
  if (gx > ntiles) gx = ntiles;
  dim3 grid(gx, a.splits);
  // E16 = true: direct launch with the fused epilogue (f16 staging); false: split-K slab writer (fp32 staging)
  const bool e16 = a.splits == 1;
  const bool lx = a.stat_out || a.lnx_stat;   // (TAPS == 1 and e16 only: checked above)
#define RCDM_IGEMM_LAUNCH(BM, BN, WM, WN, NS, THREADS, LDS)                                                           \
  do {                                                                                                                \
    if constexpr (TAPS == 1) {                                                                                        \
      if (a.lnx_stat) {                                                                                               \
        hipLaunchKernelGGL((igemm_dma_kernel<TAPS, BM, BN, WM, WN, NS, true, 2>), grid, dim3(THREADS), LDS, stream, a); \
        break;                                                                                                        \
      }                                                                                                               \
      if (lx) {                                                                                                       \
        hipLaunchKernelGGL((igemm_dma_kernel<TAPS, BM, BN, WM, WN, NS, true, 1>), grid, dim3(THREADS), LDS, stream, a); \
        break;                                                                                                        \
      }                                                                                                               \
    }                                                                                                                 \
    if (e16) hipLaunchKernelGGL((igemm_dma_kernel<TAPS, BM, BN, WM, WN, NS, true>), grid, dim3(THREADS), LDS, stream, a); \
    else hipLaunchKernelGGL((igemm_dma_kernel<TAPS, BM, BN, WM, WN, NS, false>), grid, dim3(THREADS), LDS, stream, a);    \
  } while (0)
  if (TAPS != 1 && variant >= kFirstDeep) variant = 5;
  switch (variant) {
    case 10: if constexpr (TAPS == 1) { RCDM_IGEMM_LAUNCH(128, 64, 2, 2, 3, 256, LDS_128x64_3); } break;
    case 2:   // (no statistics / consumer instantiations of this tile: fill_common)
      if (e16) hipLaunchKernelGGL((igemm_dma_kernel<TAPS, 256, 256, 2, 4, 2, true>), grid, dim3(512), LDS_256, stream, a);
      else hipLaunchKernelGGL((igemm_dma_kernel<TAPS, 256, 256, 2, 4, 2, false>), grid, dim3(512), LDS_256, stream, a);
      break;
    case 3: RCDM_IGEMM_LAUNCH(64, 64, 2, 2, 2, 256, LDS_64); break;
    case 4: RCDM_IGEMM_LAUNCH(64, 64, 2, 2, 4, 256, LDS_64D); break;
    case 5: RCDM_IGEMM_LAUNCH(128, 64, 2, 2, 2, 256, LDS_128x64); break;
    default: RCDM_IGEMM_LAUNCH(128, 128, 2, 2, 2, 256, LDS_128);
  }
#undef RCDM_IGEMM_LAUNCH
  int rc = rcdm_check_launch();
  if (rc) return rc;
  if (a.splits > 1) {
    rc = launch_splitk_reduce(a, stream);
  }
  return rc;
}

// statistics geometry of the norm behind a split-K launch (rcdm_*_gnstat): 0 when the pair qualifies — `a` planned (splits
// known), the norm reads exactly the rows this launch writes (same row count, width, row stride), takes the three-launch
// form, and the epilogue has no GEGLU / second row copy / phase rows.  With gn_ws: also points a.gn_partial into it.
int attach_gnstat(IgemmArgs& a, const rcdm_groupnorm_desc* gn, void* gn_ws, size_t gn_ws_bytes, bool need_ws) {
  if (!gn) return RCDM_EINVAL;
  if (a.splits <= 1 || (a.epi & RCDM_EPI_GEGLU) || a.dup || a.ph_rows) return RCDM_ESHAPE;
  GnArgs g{};
  int rc = rcdm_gn_plan(gn, g);
  if (rc) return rc;
  if (!rcdm_gn_three_launch(g)) return RCDM_ESHAPE;
  if ((long long)g.samples * g.P != a.M || g.C != a.N || gn->ldx != a.ldc) return RCDM_ESHAPE;
  if (g.CH * g.RPB > 512 || (size_t)(g.CH * g.RPB + g.CH) * 16 * sizeof(float) > 64 * 1024) return RCDM_ESHAPE;   // (the kernel's launch bound; C <= 4096)
  const size_t need = ((size_t)g.samples * g.splits * g.G * 3 + (size_t)g.samples * g.G * 2) * sizeof(float);
  if (need_ws) {
    if (!gn_ws || gn_ws_bytes < need) return RCDM_EWORKSPACE;
    a.gn_partial = (float*)gn_ws;
  }
  a.gn_samples = g.samples; a.gn_P = g.P; a.gn_G = g.G; a.gn_cg = g.cg; a.gn_CH = g.CH; a.gn_RPB = g.RPB;
  a.gn_splits = g.splits; a.gn_rps = g.rows_per_split;
  return RCDM_OK;
}

void from_gemm(const rcdm_gemm_desc* d, IgemmArgs& a) {
  a.M = d->M; a.N = d->N; a.Cin = d->K; a.Ktot = d->K;
  a.Hi = a.Wi = a.Ho = a.Wo = 1; a.stride = 1; a.up = 0; a.pad = 1;
  a.lda = d->lda; a.ldc = d->ldc; a.ldr = d->ldr; a.ldt = d->ldt;
  a.rows_per_sample = d->rows_per_sample; a.epi = d->epilogue; a.out_scale = d->out_scale;
  a.dup = (long long)d->dup_rows * d->ldc;
}

int from_conv(const rcdm_conv3x3_desc* d, IgemmArgs& a) {
  if (d->stride != 1 && d->stride != 2) return RCDM_ESHAPE;
  if (d->upsample != 0 && d->upsample != 1) return RCDM_ESHAPE;
  if (d->n_img <= 0 || d->h_in <= 0 || d->w_in <= 0) return RCDM_EINVAL;
  const int hv = d->h_in << d->upsample, wv = d->w_in << d->upsample;
  a.Ho = (hv - 1) / d->stride + 1;
  a.Wo = (wv - 1) / d->stride + 1;
  a.Hi = d->h_in; a.Wi = d->w_in; a.stride = d->stride; a.up = d->upsample;
  if (d->pad_after_only != 0 && d->pad_after_only != 1) return RCDM_ESHAPE;
  // F.pad(x, (0,1,0,1)) + a stride-2 conv without padding: (h + 1 - 3) / 2 + 1 = h / 2 rows for even h — the same
  // count as the symmetric form; odd sizes would differ, and the form only exists for stride 2
  if (d->pad_after_only && (d->stride != 2 || d->upsample || (d->h_in & 1) || (d->w_in & 1))) return RCDM_ESHAPE;
  a.pad = d->pad_after_only ? 0 : 1;
  a.M = d->n_img * a.Ho * a.Wo; a.N = d->c_out; a.Cin = d->c_in; a.Ktot = 9 * d->c_in;
  a.lda = d->lda; a.ldc = d->ldc; a.ldr = d->ldr; a.ldt = d->ldt;
  a.rows_per_sample = d->rows_per_sample; a.epi = d->epilogue; a.out_scale = d->out_scale;
  a.dup = (long long)d->dup_rows * d->ldc;
  if (d->c_in2 < 0) return RCDM_EINVAL;
  if (d->c_in2 > 0) {   // rcdm_conv3x3_add1x1: a 1x1 convolution of a second input in the same accumulators
    if (d->stride != 1 || d->upsample || d->pad_after_only) return RCDM_ESHAPE;
    if ((d->c_in % BK) || (d->c_in2 % BK) || (d->lda2 & 7) || d->lda2 < d->c_in2) return RCDM_ESHAPE;
    if ((size_t)a.M * (size_t)d->lda2 * 2 >= 0x7FFFFFFFull) return RCDM_ESHAPE;
    a.Cin2 = d->c_in2; a.lda2 = d->lda2; a.Ktot += d->c_in2;
  }
  return RCDM_OK;
}

// upsample = 2: the nearest-2x upsample + conv3x3 as four 2x2 phase convolutions over the source grid (igemm_args.h,
// IgemmArgs::ph_rows; weights in the rcdm.h phase layout).  4/9 of the flops of the upsample = 1 form.  Runs on the
// ping-pong kernel only, unsplit: returns the tile shape, or -1 when the shape does not fill the chip that way (the caller
// keeps the upsample = 1 form — at the 8x8 -> 16x16 level the plain form with split-K is faster).
int plan_up2(const rcdm_conv3x3_desc* d, IgemmArgs& a) {
  if (d->upsample != 2 || d->stride != 1 || d->pad_after_only || d->dup_rows) return -1;
  if (d->epilogue & ~RCDM_EPI_BIAS) return -1;
  if (d->n_img <= 0 || d->h_in <= 0 || d->w_in <= 0 || d->c_in <= 0 || d->c_out <= 0 || (d->c_in % BK)) return -1;
  const long long src = (long long)d->n_img * d->h_in * d->w_in;
  if (4 * src >= 0x7FFFFFFFll || 4ll * d->c_out * 4 * d->c_in * 2 >= 0x7FFFFFFFll) return -1;   // virtual rows are ints; weight offsets 32-bit
  a.Hi = a.Ho = d->h_in; a.Wi = a.Wo = d->w_in; a.stride = 1; a.up = 0; a.pad = 1;
  a.ph_rows = (int)src; a.M = 4 * a.ph_rows; a.N = d->c_out; a.Cin = d->c_in; a.Ktot = 4 * d->c_in;
  a.lda = d->lda; a.ldc = d->ldc; a.ldr = 0; a.ldt = 0; a.rows_per_sample = 1; a.epi = d->epilogue; a.out_scale = d->out_scale;
  a.dup = 0;
  const int cus = num_cus();
  int best = -1;
  float best_score = 0.70f;
  const int forced = (g_force_variant >= kFirstPP && g_force_variant < kFirstPP + kNumPPShapes) ? g_force_variant - kFirstPP : -1;
  for (int sh = 0; sh < kNumPPShapes; ++sh) {
    const int bm = kPPShapes[sh].bm, bn = kPPShapes[sh].bn;
    if (a.ph_rows % bm) continue;
    const int tm = a.M / bm, tn = (a.N + bn - 1) / bn, tiles = tm * tn;
    if (forced >= 0) {   // rcdm_set_igemm_variant(6 | 7 | 8): that tile shape whatever the fill (tests)
      if (sh == forced) best = sh;
      continue;
    }
    const int sp = tiles < cus ? pp_splits(tiles, 4 * (a.Cin / BK)) : 1;   // few tiles (the 8x8 -> 16x16 upsampler): cut K like the plain form does
    const int work = tiles * sp, rounds = (work + cus - 1) / cus;
    const float score = ((float)a.N / (float)(tn * bn)) * ((float)work / (float)(rounds * cus)) * (sp > 1 ? 0.90f : 1.0f);
    if (score > best_score) {
      best_score = score;
      best = sh;
    }
  }
  if (best < 0) return -1;
  a.tilesM = a.M / kPPShapes[best].bm;
  a.tilesN = (a.N + kPPShapes[best].bn - 1) / kPPShapes[best].bn;
  a.kc = a.Cin / BK;
  a.nk = 4 * a.kc;
  int s = d->split_k > 0 ? d->split_k : (a.tilesM * a.tilesN < cus ? pp_splits(a.tilesM * a.tilesN, a.nk) : 1);
  if (s > a.nk) s = a.nk;
  a.nk_per_split = (a.nk + s - 1) / s;
  a.splits = (a.nk + a.nk_per_split - 1) / a.nk_per_split;
  return best;
}

// the plain (upsample 0 | 1) conv launch behind rcdm_conv3x3 / _add1x1 / _gnstat / _add1x1_gnstat; gn: the norm whose
// partial statistics the split-K reduce leaves (nullptr: none)
int conv_launch(const rcdm_conv3x3_desc* d, const rcdm_groupnorm_desc* gn, const void* in, const void* in2, const void* W,
                const float* bias, const float* rowvec, const void* residual, void* out, void* workspace,
                size_t workspace_bytes, void* gn_workspace, size_t gn_workspace_bytes, void* stream) {
  IgemmArgs a{};
  int rc = from_conv(d, a);
  if (rc) return rc;
  a.A = (const f16*)in; a.W = (const f16*)W; a.bias = bias; a.rowvec = rowvec;
  a.res = (const f16*)residual; a.out = (f16*)out;
  a.A2 = (const f16*)in2;
  rc = check_common(a);
  if (rc) return rc;
  if (a.epi & RCDM_EPI_GEGLU) return RCDM_ESHAPE;
  int variant = 0;
  fill_common(a, d->split_k, &variant);
  if (gn) {
    rc = attach_gnstat(a, gn, gn_workspace, gn_workspace_bytes, true);
    if (rc) return rc;
  }
  return launch<9>(a, variant, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // namespace

extern "C" {

int rcdm_conv3x3_up2_supported(const rcdm_conv3x3_desc* d) {
  if (!d) return 0;
  IgemmArgs a{};
  return plan_up2(d, a) >= 0 ? 1 : 0;
}

int rcdm_set_igemm_variant(int32_t v) {
  if (v < -1 || v >= kNumVariants) return RCDM_EINVAL;
  g_force_variant = v < 0 ? 99 : v;
  return RCDM_OK;
}

int rcdm_set_shape_rules(const char* rules) {
  if (rules && strlen(rules) >= sizeof(g_rules_text)) return RCDM_EINVAL;
  g_rules_from_api = rules != nullptr;
  if (rules) strcpy(g_rules_text, rules);
  g_n_env = -1;   // parsed again at the next launch
  return RCDM_OK;
}

int rcdm_set_splitk_slab_f16(int32_t on) {
  g_slab16 = on < 0 ? -1 : (on ? 1 : 0);
  return RCDM_OK;
}

int rcdm_set_igemm_pingpong(int32_t on) {
  g_pp_mode = on ? 1 : 0;
  return RCDM_OK;
}

int rcdm_debug_set_igemm_trace(void* device_buffer) {
  g_trace = (long long*)device_buffer;
  return RCDM_OK;
}

size_t rcdm_gemm_workspace_bytes(const rcdm_gemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
  IgemmArgs a{};
  from_gemm(d, a);
  fill_common(a, d->split_k);
  return a.splits > 1 ? (size_t)a.splits * a.M * a.N * sizeof(float) : 0;
}

size_t rcdm_gemm_lnx_workspace_bytes(const rcdm_gemm_desc* d, int32_t producer, int32_t consumer) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
  IgemmArgs a{};
  from_gemm(d, a);
  if (producer) a.stat_out = (float*)16;      // any non-null value: the flags steer the tile choice, nothing is dereferenced
  if (consumer) a.lnx_stat = (const float*)16;
  fill_common(a, d->split_k);
  return a.splits > 1 ? (size_t)a.splits * a.M * a.N * sizeof(float) : 0;
}

int rcdm_gemm(const rcdm_gemm_desc* d, const void* A, const void* W, const float* bias, const float* rowvec,
              const void* residual, void* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d) return RCDM_EINVAL;
  IgemmArgs a{};
  from_gemm(d, a);
  a.A = (const f16*)A; a.W = (const f16*)W; a.bias = bias; a.rowvec = rowvec;
  a.res = (const f16*)residual; a.out = (f16*)out;
  int rc = check_common(a);
  if (rc) return rc;
  int variant = 0;
  fill_common(a, d->split_k, &variant);
  return launch<1>(a, variant, workspace, workspace_bytes, (hipStream_t)stream);
}

int rcdm_gemm_gnstat_ok(const rcdm_gemm_desc* d, const rcdm_groupnorm_desc* gn) {
  if (!d || !gn || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
  IgemmArgs a{};
  from_gemm(d, a);
  fill_common(a, d->split_k);
  return attach_gnstat(a, gn, nullptr, 0, false) == RCDM_OK;
}

int rcdm_gemm_gnstat(const rcdm_gemm_desc* d, const rcdm_groupnorm_desc* gn, const void* A, const void* W, const float* bias,
                     const float* rowvec, const void* residual, void* out, void* workspace, size_t workspace_bytes,
                     void* gn_workspace, size_t gn_workspace_bytes, void* stream) {
  if (!d) return RCDM_EINVAL;
  IgemmArgs a{};
  from_gemm(d, a);
  a.A = (const f16*)A; a.W = (const f16*)W; a.bias = bias; a.rowvec = rowvec;
  a.res = (const f16*)residual; a.out = (f16*)out;
  int rc = check_common(a);
  if (rc) return rc;
  int variant = 0;
  fill_common(a, d->split_k, &variant);
  rc = attach_gnstat(a, gn, gn_workspace, gn_workspace_bytes, true);
  if (rc) return rc;
  return launch<1>(a, variant, workspace, workspace_bytes, (hipStream_t)stream);
}

int rcdm_gemm_stat_parts(const rcdm_gemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
  IgemmArgs a{};
  from_gemm(d, a);
  a.stat_out = (float*)16;   // any non-null value: the tile choice of a statistics-producing launch
  fill_common(a, d->split_k);
  return a.splits > 1 ? 0 : a.tilesN;
}

int rcdm_gemm_lnx_stat_parts(const rcdm_gemm_desc* d, int32_t consumer) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
  IgemmArgs a{};
  from_gemm(d, a);
  a.stat_out = (float*)16;                       // producer flag, as in rcdm_gemm_stat_parts ...
  if (consumer) a.lnx_stat = (const float*)16;   // ... and the consumer flag the launch will carry: both steer the tile choice
  fill_common(a, d->split_k);
  return a.splits > 1 ? 0 : a.tilesN;
}

int rcdm_gemm_lnx(const rcdm_gemm_desc* d, const rcdm_lnx* x, const void* A, const void* W, const float* bias,
                  const float* rowvec, const void* residual, void* out, void* workspace, size_t workspace_bytes,
                  void* stream) {
  if (!d || !x) return RCDM_EINVAL;
  IgemmArgs a{};
  from_gemm(d, a);
  a.A = (const f16*)A; a.W = (const f16*)W; a.bias = bias; a.rowvec = rowvec;
  a.res = (const f16*)residual; a.out = (f16*)out;
  int rc = check_common(a);
  if (rc) return rc;
  if (x->stat_out) {
    if (x->stat_parts <= 0 || ((uintptr_t)x->stat_out & 7)) return RCDM_EINVAL;
    a.stat_out = x->stat_out;
    a.stat_parts = x->stat_parts;
    a.stat_ld = x->stat_out_rows;
    if (a.stat_ld < a.M + d->dup_rows) return RCDM_EINVAL;
  }
  if (x->stat_in) {
    if (!x->colsum || x->parts_in <= 0 || x->C <= 0 || ((uintptr_t)x->stat_in & 7) || ((uintptr_t)x->colsum & 15)) return RCDM_EINVAL;
    if (x->parts_in > kLnxMaxParts) return RCDM_ESHAPE;
    a.lnx_stat = x->stat_in; a.lnx_S = x->colsum; a.lnx_parts = x->parts_in; a.lnx_ld = x->stat_in_rows;
    if (a.lnx_ld < a.M) return RCDM_EINVAL;
    a.lnx_invC = 1.0f / (float)x->C; a.lnx_eps = x->eps;
  }
  int variant = 0;
  fill_common(a, d->split_k, &variant);
  if (a.stat_out && a.stat_parts != a.tilesN) {   // the caller's slot count, where an LDS-DMA tile has it (rcdm_gemm_lnx_parts_ok)
    const int alt = variant_for_parts(a, a.stat_parts);
    if (alt < 0) return RCDM_ESHAPE;
    fill_common(a, 1, &variant, alt);             // (a statistics launch is never split)
  }
  return launch<1>(a, variant, workspace, workspace_bytes, (hipStream_t)stream);
}

int rcdm_gemm_lnx_parts_ok(const rcdm_gemm_desc* d, int32_t parts, int32_t consumer) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0 || parts <= 0 || (d->epilogue & RCDM_EPI_GEGLU)) return 0;
  IgemmArgs a{};
  from_gemm(d, a);
  a.stat_out = (float*)16;
  if (consumer) a.lnx_stat = (const float*)16;
  fill_common(a, d->split_k);
  if (a.splits == 1 && a.tilesN == parts) return 1;
  return variant_for_parts(a, parts) >= 0;
}

int rcdm_gemm_ln(const rcdm_gemm_desc* d, const rcdm_ln_fuse* ln, const void* A, const void* W, const float* bias,
                 const void* residual, void* out, void* stream) {
  if (!d || !ln || !ln->gamma || !ln->beta || !ln->out) return RCDM_EINVAL;
  if (ln->pe && (ln->rows_per_frame <= 0 || ln->frames <= 0)) return RCDM_EINVAL;
  IgemmArgs a{};
  from_gemm(d, a);
  a.A = (const f16*)A; a.W = (const f16*)W; a.bias = bias; a.rowvec = nullptr;
  a.res = (const f16*)residual; a.out = (f16*)out;
  int rc = check_common(a);
  if (rc) return rc;
  // one 160x320 ping-pong tile must span the output row; only the plain epilogues
  if (a.N > kPPShapes[0].bn || (ln->ld & 7) || d->split_k > 1) return RCDM_ESHAPE;
  if (a.epi & (RCDM_EPI_GEGLU | RCDM_EPI_GELU | RCDM_EPI_ROWVEC)) return RCDM_ESHAPE;
  a.tilesM = (a.M + kPPShapes[0].bm - 1) / kPPShapes[0].bm;
  a.tilesN = 1;
  a.kc = (a.Cin + BK - 1) / BK;
  a.nk = a.kc;
  a.splits = 1;
  a.nk_per_split = a.nk;
  a.partial = nullptr;
  a.trace = nullptr;
  a.dbg = 0;
  a.epi |= kEpiLN;
  a.ln_g = ln->gamma; a.ln_b = ln->beta; a.ln_pe = ln->pe; a.ln_out = (f16*)ln->out; a.ln_ld = ln->ld;
  a.ln_rpf = ln->pe ? ln->rows_per_frame : 1; a.ln_frames = ln->pe ? ln->frames : 1; a.ln_eps = ln->eps;
  return rcdm_igemm_pp_launch(a, 1, 0, (hipStream_t)stream);
}

size_t rcdm_conv3x3_workspace_bytes(const rcdm_conv3x3_desc* d) {
  if (!d) return 0;
  IgemmArgs a{};
  if (d->upsample == 2) {
    if (d->c_in2) return 0;   // (no phase form with a second input: every launch entry refuses the pair)
    return plan_up2(d, a) >= 0 && a.splits > 1 ? (size_t)a.splits * a.M * a.N * sizeof(float) : 0;
  }
  if (from_conv(d, a) || a.Cin <= 0 || a.N <= 0) return 0;
  fill_common(a, d->split_k);
  return a.splits > 1 ? (size_t)a.splits * a.M * a.N * sizeof(float) : 0;
}

int rcdm_conv3x3(const rcdm_conv3x3_desc* d, const void* in, const void* W, const float* bias, const float* rowvec,
                 const void* residual, void* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d) return RCDM_EINVAL;
  if (d->c_in2) return RCDM_EINVAL;   // (a descriptor of rcdm_conv3x3_add1x1 — refused in every form, the phase form included)
  IgemmArgs a{};
  if (d->upsample == 2) {
    const int shape = plan_up2(d, a);
    if (shape < 0) return RCDM_ESHAPE;
    if (!in || !W || !out || ((a.epi & RCDM_EPI_BIAS) && !bias)) return RCDM_EINVAL;
    if ((a.N & 7) || (a.lda & 7) || (a.ldc & 7) || (size_t)a.ph_rows * (size_t)a.lda * 2 >= 0x7FFFFFFFull) return RCDM_ESHAPE;
    a.A = (const f16*)in; a.W = (const f16*)W; a.bias = bias; a.out = (f16*)out;
    a.trace = nullptr;
    a.dbg = 0;
    if (a.splits > 1) {
      if (!workspace || workspace_bytes < (size_t)a.splits * a.M * a.N * sizeof(float)) return RCDM_EWORKSPACE;
      a.partial = (float*)workspace;
    }
    int rc = rcdm_igemm_pp_launch(a, 4, shape, (hipStream_t)stream);
    if (rc || a.splits == 1) return rc;
    const size_t total = (size_t)a.M * (a.N / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return rcdm_check_launch();
  }
  return conv_launch(d, nullptr, in, nullptr, W, bias, rowvec, residual, out, workspace, workspace_bytes, nullptr, 0, stream);
}

// Launch geometry the library chooses for a shape (diagnostics: tools/ceiling.py prices tile quantisation with it): out[8] =
// {tile variant, BM, BN, row tiles, column tiles, split-K factor, resident blocks per CU of that tile, k-steps of 64}.
static void plan_out(const IgemmArgs& a, int variant, int32_t* out) {
  out[0] = variant; out[1] = kTiles[variant].bm; out[2] = kTiles[variant].bn; out[3] = a.tilesM; out[4] = a.tilesN;
  out[5] = a.splits; out[6] = kTiles[variant].blocks_per_cu; out[7] = a.nk;
}

int rcdm_gemm_plan_query(const rcdm_gemm_desc* d, int32_t producer, int32_t consumer, int32_t* out8) {
  if (!d || !out8 || d->M <= 0 || d->N <= 0 || d->K <= 0) return RCDM_EINVAL;
  IgemmArgs a{};
  from_gemm(d, a);
  if (producer) a.stat_out = (float*)16;
  if (consumer) a.lnx_stat = (const float*)16;
  int variant = 0;
  fill_common(a, d->split_k, &variant);
  plan_out(a, variant, out8);
  return RCDM_OK;
}

int rcdm_conv3x3_plan_query(const rcdm_conv3x3_desc* d, int32_t* out8) {
  if (!d || !out8) return RCDM_EINVAL;
  IgemmArgs a{};
  if (d->upsample == 2) {
    const int shape = plan_up2(d, a);
    if (shape < 0) return RCDM_ESHAPE;
    plan_out(a, kFirstPP + shape, out8);
    return RCDM_OK;
  }
  if (from_conv(d, a) || a.Cin <= 0 || a.N <= 0) return RCDM_EINVAL;
  int variant = 0;
  fill_common(a, d->split_k, &variant);
  plan_out(a, variant, out8);
  return RCDM_OK;
}

int rcdm_conv3x3_add1x1(const rcdm_conv3x3_desc* d, const void* in, const void* in2, const void* W, const float* bias,
                        const float* rowvec, const void* residual, void* out, void* workspace, size_t workspace_bytes,
                        void* stream) {
  if (!d || d->upsample == 2 || d->c_in2 <= 0 || !in2) return RCDM_EINVAL;
  return conv_launch(d, nullptr, in, in2, W, bias, rowvec, residual, out, workspace, workspace_bytes, nullptr, 0, stream);
}

int rcdm_conv3x3_gnstat_ok(const rcdm_conv3x3_desc* d, const rcdm_groupnorm_desc* gn) {
  if (!d || !gn || d->upsample == 2) return 0;
  IgemmArgs a{};
  if (from_conv(d, a) || a.Cin <= 0 || a.N <= 0) return 0;
  fill_common(a, d->split_k);
  return attach_gnstat(a, gn, nullptr, 0, false) == RCDM_OK;
}

int rcdm_conv3x3_gnstat(const rcdm_conv3x3_desc* d, const rcdm_groupnorm_desc* gn, const void* in, const void* W,
                        const float* bias, const float* rowvec, const void* residual, void* out, void* workspace,
                        size_t workspace_bytes, void* gn_workspace, size_t gn_workspace_bytes, void* stream) {
  if (!d || !gn) return RCDM_EINVAL;
  if (d->upsample == 2) return RCDM_ESHAPE;
  if (d->c_in2) return RCDM_EINVAL;
  return conv_launch(d, gn, in, nullptr, W, bias, rowvec, residual, out, workspace, workspace_bytes, gn_workspace,
                     gn_workspace_bytes, stream);
}

int rcdm_conv3x3_add1x1_gnstat(const rcdm_conv3x3_desc* d, const rcdm_groupnorm_desc* gn, const void* in, const void* in2,
                               const void* W, const float* bias, const float* rowvec, const void* residual, void* out,
                               void* workspace, size_t workspace_bytes, void* gn_workspace, size_t gn_workspace_bytes,
                               void* stream) {
  if (!d || !gn || d->upsample == 2 || d->c_in2 <= 0 || !in2) return RCDM_EINVAL;
  return conv_launch(d, gn, in, in2, W, bias, rowvec, residual, out, workspace, workspace_bytes, gn_workspace,
                     gn_workspace_bytes, stream);
}

}  // extern "C"
