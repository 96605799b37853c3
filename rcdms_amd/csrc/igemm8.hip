// igemm8.hip — 8-wave PING-PONG implicit GEMM for gfx950 (Linear / 1x1 conv and conv3x3 over channels-last f16
// rows, fp32 accumulate): the deep-K, many-row shapes of the UNet (GEGLU / qkv / feed-forward GEMMs and the convs of the
// 64x64 .. 16x16 levels).  Same operands, epilogues and C-ABI entry points as igemm.hip (rcdm_gemm / rcdm_conv3x3 pick
// the kernel by shape); replaces the same reference calls (InflatedConv3d.forward src/models/resnet.py:10-18, the
// nn.Linear calls of attention.py:121,140-141,164, diffusers FeedForward).
//
// Why a second loop: igemm_dma_kernel keeps ONE barrier per 64-deep k-step and lets every wave do
// [DMA issue -> ds_read -> MFMA] in the same order, so both waves of a SIMD stall on the vector-memory queue at the
// same time (measured: MFMA pipe 33-35 % busy at the 64x64 level, 62 % inside the k-loop).  Here the block's 8 waves
// form two groups of 4 (one wave of each group per SIMD) that run HALF A K-STEP OUT OF PHASE:
//     tick            4g        4g+1      4g+2      4g+3
//     group 0 (rows 0..BM/2)    L(g,0)    C(g,0)    L(g,1)    C(g,1)
//     group 1 (rows BM/2..BM)   C(g-1,1)  L(g,0)    C(g,0)    L(g,1)
// L = load phase: ds_read the fragments of one 32-deep half k-step into registers + issue this wave's LDS-DMA pieces;
// C = compute phase: FMW x FNW v_mfma_f32_16x16x32_f16 back to back at raised priority.  One s_barrier per tick keeps
// the two groups in lock-step, so the matrix pipe of every SIMD always has one wave in C while its partner is in L.
// Tiles are multiples of 16 (16x16x32 fragments), chosen so that M = 40960 / 10240 / 2560 rows split into an EXACT
// number of rounds over the 256 CUs: 160x320 (256 row tiles at the 64x64 level), 160x256, 256x256.
//
// LDS (all 160 KB for the 160x320 and 256x256 tiles): a ring of 2 pixel tiles [BM rows][128 B] and a ring of 3 weight
// tiles [BN rows][128 B], written by buffer_load ... lds (lane-linear 1-KiB pieces, XOR swizzle on the source side exactly
// as igemm.hip).  Measured with the compute ablated (RCDM_PP_ABLATE): the k-loop without DMA runs 0.93 us per 160x320x64
// k-step, the DMA stream alone 1.15-1.2 us — latency, not bandwidth: a part is ~1.1 us from issue to landed under load,
// and with two weight tiles a weight part could only be issued 2-3 ticks (0.5-0.7 us) before its first read (a deeper
// PIXEL ring changed nothing).  Hence three weight tiles: every part now has >= 4 ticks in flight.
// A tile is released in parts as the out-of-phase groups finish with it and refilled one part per L phase:
//     group 0, L(g,0) tick 4g  : pixel rows of group 1 (free since tick 4g-1)               -> k-step g+1
//     group 1, L(g,0) tick 4g+1: upper half of the weight rows (tile of k-step g-1)         -> k-step g+2
//     group 0, L(g,1) tick 4g+2: lower half of the weight rows                              -> k-step g+2
//     group 1, L(g,1) tick 4g+3: pixel rows of group 0 (last read one tick earlier)         -> k-step g+2
// Waits are counted vmcnt ("all but the newest pixel part + weight part of this wave"), placed before the barriers that
// close ticks 4g (group 0), 4g+3 (both).
// Pieces that do not divide over 4 waves (BM/2 = 80 rows = 10 pieces: waves 0,1 issue 3, waves 2,3 issue 2) make the
// vmcnt immediates wave-dependent: picked by a wave-uniform branch.
#include "common.h"
#include "igemm_args.h"
#include "pp_sync.h"
#include "igemm_epilogue.h"

const PPShape kPPShapes[kNumPPShapes] = {{160, 320}, {160, 256}, {256, 256}};

// cache policy of the two LDS-DMA streams (buffer_load ... lds aux bits: 1 = sc0, 2 = nt, 16 = sc1); experiments only
#ifndef RCDM_PP_AAUX
#define RCDM_PP_AAUX 0
#endif
#ifndef RCDM_PP_WAUX
#define RCDM_PP_WAUX 0
#endif

namespace {

template <int TAPS, int FMW, int FNW, bool SLAB>
__global__ __launch_bounds__(512, 2) void igemm_pp_kernel(const IgemmArgs p) {
  constexpr int BM = 2 * FMW * 16, BN = 4 * FNW * 16;
  constexpr int HM = BM / 2, HN = BN / 2;         // rows of one pixel half / one weight half
  constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
  constexpr int W_RING = 2 * A_BYTES;             // 3 weight tiles behind the 2 pixel tiles
  constexpr int PP = HM / 8;                      // 1-KiB pieces per pixel half
  constexpr int NPP = (PP + 3) / 4;               // ... per wave (the last one may be missing: see npp)
  constexpr int NPW = HN / 32;                    // weight-half pieces per wave
  static_assert(2 * A_BYTES + 3 * W_BYTES <= 160 * 1024, "ring exceeds the LDS");
  static_assert(HN % 32 == 0 && HM % 8 == 0, "tile halves must be whole pieces");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = wave >> 2, wn = wave & 3;

  const int ks_begin = blockIdx.y * p.nk_per_split;
  const int nkl = min(p.nk, ks_begin + p.nk_per_split) - ks_begin;
  if (nkl <= 0) return;

  // XCD-aware, L2-blocked tile order (see igemm.hip): block b runs on XCD b % 8 and each XCD gets a contiguous run of
  // the sequence "super-rows of 8 row panels, column by column".
  int cm0, cn0;
  {
    const int ntiles = p.tilesM * p.tilesN, lin = blockIdx.x;
    const int xcd = lin & 7, q = ntiles >> 3, r = ntiles & 7;
    const int tl = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    constexpr int GM = 8;
    const int per_group = GM * p.tilesN;
    const int sg = tl / per_group, rem = tl - sg * per_group;
    const int gm = min(GM, p.tilesM - sg * GM);
    const int tn = rem / gm, tm = sg * GM + (rem - tn * gm);
    cm0 = tm * BM;
    cn0 = tn * BN;
  }

  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(TAPS == 9 ? p.A2 : p.A), 0, 0x7FFFFFFF, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const int Hv = p.Hi << p.up, Wv = p.Wi << p.up;

  const int phase = TAPS == 4 ? cm0 / p.ph_rows : 0;   // phase launch (IgemmArgs::ph_rows): the tile's output parity (2a + b)
  // ---- loader state: every group fills the OTHER group's pixel rows; group 0 the lower weight rows, group 1 the upper
  const int myhalf = grp ^ 1, mywhalf = grp;
  const int lrow = lane >> 3, lch = lane & 7;
  unsigned a_off[NPP], w_off[NPW];
  int a_img[NPP], a_iy[NPP], a_ix[NPP], a_c[NPP], a_lds[NPP], w_c[NPW], w_lds[NPW];
#pragma unroll
  for (int i = 0; i < NPP; ++i) {
    const int q = wn + 4 * i;
    const bool live = q < PP;
    const int row = myhalf * HM + q * 8 + lrow;  // row of the pixel tile
    const int m = cm0 + row;
    a_lds[i] = live ? (myhalf * HM + q * 8) * 128 : -1;
    a_c[i] = (lch ^ ((row >> 1) & 7)) * 8;
    a_off[i] = OOB;
    a_img[i] = a_ix[i] = 0;
    a_iy[i] = -(1 << 20);
    if (TAPS == 1) {
      if (live && m < p.M) a_off[i] = (unsigned)m * (unsigned)p.lda * 2u;
    } else if (live && m < p.M) {
      const int hw = p.Ho * p.Wo;
      const int pm = TAPS == 4 ? m - phase * p.ph_rows : m;   // (phase launch: the source pixel of this virtual row)
      const int img = pm / hw, rem = pm - img * hw;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_img[i] = img * p.Hi * p.Wi;  // first pixel row of the image
      a_iy[i] = oy * p.stride - p.pad + (TAPS == 4 ? phase >> 1 : 0);
      a_ix[i] = ox * p.stride - p.pad + (TAPS == 4 ? phase & 1 : 0);
    }
  }
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int q = wn + 4 * i;
    const int row = mywhalf * HN + q * 8 + lrow;  // row of the weight tile
    const int n = cn0 + row;
    w_lds[i] = W_RING + (mywhalf * HN + q * 8) * 128;
    w_c[i] = (lch ^ ((row >> 1) & 7)) * 8;
    w_off[i] = n < p.N ? (unsigned)(phase * p.N + n) * (unsigned)p.Ktot * 2u : OOB;
  }

  // Every tile of an XCD that shares a weight panel would otherwise stream the SAME weight lines at the same moment
  // (32 CUs on one L2 channel at a time).  Each block therefore starts its k-loop at a different k-step (conv: at a
  // different 64-channel slab, so the nine taps of a slab stay adjacent) and wraps around.
  int rot = 0;
  if (p.dbg & 8) {
    const int idx = blockIdx.x >> 3;  // position inside the XCD's run of tiles
    rot = TAPS == 1 ? idx % nkl : (TAPS * idx) % nkl;
  }
  auto kpos = [&](int ks, int& c0, int& tap) __attribute__((always_inline)) {
    ks += rot;
    if (ks >= nkl) ks -= nkl;
    ks += ks_begin;
    tap = 0;
    int kci = ks;
    if (TAPS != 1) {  // channel chunk outer, tap inner (the nine windows of one 64-channel slab re-hit L2)
      kci = ks / TAPS;
      tap = ks - kci * TAPS;
      if (TAPS == 9 && ks >= p.nk1) {   // second input (IgemmArgs::A2): a tenth "tap" = the output pixel itself in another tensor
        kci = ks - p.nk1;
        tap = 9;
      }
    }
    c0 = kci * BK;
  };
  auto issue_px = [&](int ks) __attribute__((always_inline)) {
    const int slot = ks & 1;
    int c0, tap;
    kpos(ks, c0, tap);
    const bool s2 = TAPS == 9 && tap == 9;   // (wave-uniform)
    const int dy = s2 ? 1 : TAPS == 4 ? tap >> 1 : tap / 3, dx = s2 ? 1 : TAPS == 4 ? tap & 1 : tap - dy * 3;
    const int cin = s2 ? p.Cin2 : p.Cin;
    const unsigned lda = (unsigned)(s2 ? p.lda2 : p.lda);
#if defined(RCDM_PP_ABLATE) && (RCDM_PP_ABLATE & 8)   // bound of a halo-staged pixel tile: the pixel pieces of 2 taps in 9 only (garbage results)
    if (TAPS != 1 && tap >= 2) return;
#endif
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const int c = c0 + a_c[i];
      unsigned vo;
      if (TAPS == 1) {
        vo = (c < p.Cin && a_off[i] != OOB) ? a_off[i] + (unsigned)c * 2u : OOB;
      } else {
        const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
        const bool ok = (c < cin) && ((unsigned)iy < (unsigned)Hv) && ((unsigned)ix < (unsigned)Wv);
        const int sy = iy >> p.up, sx = ix >> p.up;
        // computed unconditionally (a select, not a branch: no address is dereferenced here)
        const unsigned off = ((unsigned)(a_img[i] + sy * p.Wi + sx) * lda + (unsigned)c) * 2u;
        vo = ok ? off : OOB;
      }
      if (a_lds[i] >= 0) {  // wave-uniform: the waves whose last piece does not exist issue one DMA fewer
        if (TAPS == 9 && s2)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              rsrcA2, (__attribute__((address_space(3))) void*)(smem + slot * A_BYTES + a_lds[i]), 16, vo, 0, 0, RCDM_PP_AAUX);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              rsrcA, (__attribute__((address_space(3))) void*)(smem + slot * A_BYTES + a_lds[i]), 16, vo, 0, 0, RCDM_PP_AAUX);
      }
    }
  };
  auto issue_w = [&](int ks) __attribute__((always_inline)) {
    const int slot = ks % 3;
    int c0, tap;
    kpos(ks, c0, tap);
    const int cin = (TAPS == 9 && tap == 9) ? p.Cin2 : p.Cin;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int c = c0 + w_c[i];
      const unsigned vo = (c < cin && w_off[i] != OOB)
                              ? w_off[i] + ((unsigned)tap * (unsigned)p.Cin + (unsigned)c) * 2u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsrcW, (__attribute__((address_space(3))) void*)(smem + slot * W_BYTES + w_lds[i]), 16, vo, 0, 0, RCDM_PP_WAUX);
    }
  };

  // ---- compute mapping: 16x16x32 fragments; weights are the A operand, pixels the B operand -> D[channel][pixel]
  const int l15 = lane & 15, kg = lane >> 4;
  const int sw = (lane >> 1) & 7;  // (row >> 1) & 7 of this lane's fragment rows (fragment bases are multiples of 16)
  const int koff0 = ((kg ^ sw) << 4), koff1 = (((4 + kg) ^ sw) << 4);
  const int rowA = (grp * HM + l15) * 128, rowB = W_RING + (wn * (BN / 4) + l15) * 128;

  f32x4 acc[FNW][FMW];
#pragma unroll
  for (int i = 0; i < FNW; ++i)
#pragma unroll
    for (int j = 0; j < FMW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 wf[FNW], xf[FMW];

#ifdef RCDM_PP_ABLATE  // debug builds only (rcdms_amd.build.build_variant): 1 no DMA after the prologue, 2 no MFMA, 4 no ds_read
  constexpr bool no_dma = (RCDM_PP_ABLATE & 1) != 0, no_mfma = (RCDM_PP_ABLATE & 2) != 0, no_read = (RCDM_PP_ABLATE & 4) != 0;
#else
  constexpr bool no_dma = false, no_mfma = false, no_read = false;
#endif
  // RCDM_PP_ISSUE_FIRST: the DMA pieces of an L phase are issued BEFORE its ds_reads (the issue stalls on the
  // vector-memory queue; reads queued behind it complete under that stall instead of in front of it)
#ifdef RCDM_PP_ISSUE_FIRST
  constexpr bool issue_first = true;
#else
  constexpr bool issue_first = false;
#endif
  auto load_frags = [&](int aslot, int wslot, int koff) __attribute__((always_inline)) {
    if (no_read) return;
    const char* sa = smem + aslot * A_BYTES + rowA + koff;
    const char* sb = smem + wslot * W_BYTES + rowB + koff;
#pragma unroll
    for (int i = 0; i < FNW; ++i) wf[i] = *(const f16x8*)(sb + i * 2048);
#pragma unroll
    for (int j = 0; j < FMW; ++j) xf[j] = *(const f16x8*)(sa + j * 2048);
  };
  auto compute = [&]() __attribute__((always_inline)) {
    if (no_mfma) return;
    // The MFMA run at LOW priority, everything else of this wave (the load phases: address arithmetic, ds_read, DMA issue)
    // at high priority.  The SIMD issues oldest-first among equal priorities, and a wave whose next MFMA waits for the
    // matrix pipe keeps the other wave's VALU instructions from issuing (tools/ubench/pipe_overlap.hip: a VALU wave beside
    // two MFMA-streaming waves gets one instruction per 23 clocks at equal priority, one per 6.4 at s_setprio 3, with the
    // MFMA rate unchanged).  Round 2 had it the other way round (MFMA run at priority 1); -0.05 ms per step.
#if RCDM_PRIO_LOADS
    __builtin_amdgcn_s_setprio(0);
#else
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int i = 0; i < FNW; ++i)
#pragma unroll
      for (int j = 0; j < FMW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
#if RCDM_PRIO_LOADS
    __builtin_amdgcn_s_setprio(3);
#else
    __builtin_amdgcn_s_setprio(0);
#endif
  };

  // DMAs this wave has in flight when "its newest pixel part + its newest weight part" are outstanding
  const int npp = (wn + 4 * (NPP - 1) < PP) ? NPP : NPP - 1;
  const int keep = npp + NPW;

  // ---- prologue: all of k-step 0 and what the steady state would have issued before tick 0: both weight halves of
  // k-step 1 and (group 1) the group-0 pixel rows of k-step 1.  Issue order = the order the loop's counted waits assume.
  if (grp == 1) {
    issue_px(0);
    issue_w(0);
    if (nkl > 1) {
      issue_w(1);
      issue_px(1);
      wait_vm_n(keep);
    } else {
      wait_vm<0>();
    }
  } else {
    issue_px(0);
    issue_w(0);
    if (nkl > 1) {
      issue_w(1);
      wait_vm<NPW>();
    } else {
      wait_vm<0>();
    }
  }
  tick_barrier();

  if (grp == 0) {
    int wslot = 0;  // g % 3
    for (int g = 0; g < nkl; ++g) {
      const int aslot = g & 1;
      // tick 4g: L(g,0).  Group 1 reads its pixel rows of THIS k-step one tick from now (issued in tick 4g-4): all but
      // the newest pixel part and the weight part before it must have landed.
      const bool more_p0 = g + 1 < nkl && !no_dma;
      if (issue_first) {
        if (more_p0) issue_px(g + 1);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(aslot, wslot, koff0);
      } else {
        load_frags(aslot, wslot, koff0);
        if (more_p0) issue_px(g + 1);
      }
      wait_lgkm0();
      if (more_p0) wait_vm_n(keep); else wait_vm<0>();
      tick_barrier();
      // tick 4g+1: C(g,0)
      compute();
      tick_barrier();
      // tick 4g+2: L(g,1): the weight tile of k-step g-1 is free since tick 4g-1 -> lower half of k-step g+2
      const bool more_w = g + 2 < nkl && !no_dma;
      if (issue_first) {
        if (more_w) issue_w(g + 2);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(aslot, wslot, koff1);
      } else {
        load_frags(aslot, wslot, koff1);
        if (more_w) issue_w(g + 2);
      }
      wait_lgkm0();
      tick_barrier();
      // tick 4g+3: C(g,1); the lower weight rows of k-step g+1 (issued in tick 4g-2) must have landed
      compute();
      if (more_w) wait_vm_n(keep); else wait_vm<0>();
      tick_barrier();
      wslot = wslot == 2 ? 0 : wslot + 1;
    }
    tick_barrier();  // tick 4 nkl: group 1's last compute phase
  } else {
    tick_barrier();  // tick 0: group 0's first load phase
    int wslot = 0;
    for (int g = 0; g < nkl; ++g) {
      const int aslot = g & 1;
      const bool more = g + 2 < nkl && !no_dma;
      // tick 4g+1: L(g,0): upper half of the weight rows of k-step g+2
      if (issue_first) {
        if (more) issue_w(g + 2);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(aslot, wslot, koff0);
      } else {
        load_frags(aslot, wslot, koff0);
        if (more) issue_w(g + 2);
      }
      wait_lgkm0();
      tick_barrier();
      // tick 4g+2: C(g,0)
      compute();
      tick_barrier();
      // tick 4g+3: L(g,1); group 0 finished with its pixel rows of k-step g one tick ago -> refill them for g+2;
      // its pixel rows and the upper weight rows of k-step g+1 must have landed
      if (issue_first) {
        if (more) issue_px(g + 2);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(aslot, wslot, koff1);
      } else {
        load_frags(aslot, wslot, koff1);
        if (more) issue_px(g + 2);
      }
      wait_lgkm0();
      if (more) wait_vm_n(keep); else wait_vm<0>();
      tick_barrier();
      // tick 4g+4: C(g,1)
      compute();
      tick_barrier();
      wslot = wslot == 2 ? 0 : wslot + 1;
    }
  }
  // every wave is past the last tick: no LDS read and no DMA is outstanding anywhere in the block

  // deferred-LayerNorm consumer (only when this tile shape is forced: pick_variant keeps such launches off this kernel): the
  // row / column tables behind the staging tile, one barrier, then the accumulators are transformed (igemm_epilogue.h)
  constexpr bool kLX = TAPS == 1 && !SLAB && FMW < 8;   // (not on the 256x256 tile: at the 256-register cap already; rcdm_gemm_lnx refuses it there)
  const float* ltab = nullptr;
  if constexpr (kLX) {
    if (p.lnx_stat != nullptr) {
      f32x2 pre = {1.f, 0.f};
      if (t < BM) {
        float r_ = 1.f, m_ = 0.f;
        lnx_row<1, kLnxMaxParts>(p.lnx_stat, p.lnx_ld, cm0 + t, cm0 + t < p.M, p.lnx_parts, 0, p.lnx_invC, p.lnx_eps, r_, m_);
        pre = f32x2{r_, m_};
      }
      float* tab = (float*)(smem + BM * (2 * BN + 16));
      lnx_write_tables<BM, BN>(p, tab, cm0, cn0, t, pre);
      wait_lgkm0();
      tick_barrier();
      ltab = tab;
    }
  }
  tile_epilogue<FMW, FNW, SLAB, 512, BM, BN, TAPS == 1, kLX, TAPS == 4>(p, smem, acc, cm0, cn0, grp * HM, wn * (BN / 4), l15, kg, t, ltab);
}

template <int FMW, int FNW>
constexpr int pp_lds_bytes() {
  constexpr int BM = 2 * FMW * 16, BN = 4 * FNW * 16;
  constexpr int ring = (2 * BM + 3 * BN) * 128, stage = BM * (2 * BN + 16);
  return ring > stage ? ring : stage;
}

template <int TAPS, int FMW, int FNW>
int launch_pp(const IgemmArgs& a, hipStream_t stream) {
  constexpr int LDS = pp_lds_bytes<FMW, FNW>();
  static bool attr_set[64] = {};
  if (rcdm_first_on_device(attr_set)) {
    (void)hipFuncSetAttribute((const void*)igemm_pp_kernel<TAPS, FMW, FNW, false>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute((const void*)igemm_pp_kernel<TAPS, FMW, FNW, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  }
  dim3 grid(a.tilesM * a.tilesN, a.splits);
  if (a.splits > 1)
    hipLaunchKernelGGL((igemm_pp_kernel<TAPS, FMW, FNW, true>), grid, dim3(512), LDS, stream, a);
  else
    hipLaunchKernelGGL((igemm_pp_kernel<TAPS, FMW, FNW, false>), grid, dim3(512), LDS, stream, a);
  return rcdm_check_launch();
}

template <int TAPS>
int launch_shape(const IgemmArgs& a, int shape, hipStream_t stream) {
  switch (shape) {
    case 0: return launch_pp<TAPS, 5, 5>(a, stream);
    case 1: return launch_pp<TAPS, 5, 4>(a, stream);
    case 2: return launch_pp<TAPS, 8, 4>(a, stream);
    default: return RCDM_EINVAL;
  }
}

}  // namespace

int rcdm_igemm_pp_launch(const IgemmArgs& a, int taps, int shape, hipStream_t stream) {
  if (taps == 1) return launch_shape<1>(a, shape, stream);
  if (taps == 9) return launch_shape<9>(a, shape, stream);
  if (taps == 4 && shape >= 0 && shape < kNumPPShapes && a.ph_rows > 0 && a.ph_rows % kPPShapes[shape].bm == 0)
    return launch_shape<4>(a, shape, stream);
  return RCDM_EINVAL;
}
