// igemm16.hip — 160x160-tile implicit GEMM (Linear / 1x1 conv and conv3x3 over channels-last f16 rows, fp32 accumulate),
// 4 waves per block, two blocks per CU, v_mfma_f32_16x16x32_f16.  Same operands, epilogues and C-ABI entry points as
// igemm.hip (rcdm_gemm / rcdm_conv3x3 pick the kernel by shape); replaces the same reference calls (nn.Linear of
// attention.py:121,140-141,164,330,352 / motion_module.py:166,170, diffusers FeedForward, InflatedConv3d resnet.py:10-18).
//
// Why this tile: every implicit-GEMM kernel of the library is bound by operand delivery L2 -> LDS (~30 B/clk per CU,
// DESIGN.md §3), so operand bytes per flop decide a shape's ceiling: 15.6 B per kflop at 128x128, 12.5 at 160x160.  And
// every dimension of this UNet is a multiple of 160 (channels are multiples of 320, rows of 2560): 160-wide tiles have
// no padded columns where 128-wide ones waste 6 % (N = 960, 1920) to 17 % (N = 320).  16x16 fragments make the 80x80
// wave tile possible (5 x 5 fragments, 100 accumulator registers).  Two blocks of 80 KB LDS share a CU, so one block's
// epilogue and tile fill run under the other's k-loop — what the one-block-per-CU ping-pong kernel (igemm8.hip) cannot
// do, which is why that kernel only wins where the k-loop is long.
//
// Loop: one s_barrier per 64-deep k-step, 2-stage LDS ring filled by buffer_load ... lds (5 pixel + 5 weight 1-KiB
// pieces per wave and step, XOR swizzle on the source side as in igemm.hip); the pieces of step g+1 are issued right
// after the barrier that opens step g.  One tile per block, XCD-aware tile order; epilogue through igemm_epilogue.h.
#include "common.h"
#include "igemm_args.h"
#include "pp_sync.h"
#include "igemm_epilogue.h"
#ifndef RCDM_I16_ABLATE
#define RCDM_I16_ABLATE 0  // debug builds: 1 = no epilogue, 2 = epilogue without global stores, 8 = no DMA after the first stage
#endif

namespace {

template <int TAPS, bool SLAB>
__global__ __launch_bounds__(256, 2) void igemm16_kernel(const IgemmArgs p) {
  constexpr int BM = 160, BN = 160, FM = 5, FN = 5;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int NP = BM / 8 / 4;  // 1-KiB pieces per wave, per operand and step
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ks_begin = blockIdx.y * p.nk_per_split;
  const int nkl = min(p.nk, ks_begin + p.nk_per_split) - ks_begin;
  if (nkl <= 0) return;

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

  // deferred LayerNorm of the A rows (rcdm_gemm_lnx): thread t < BM requests row t's partial statistics FIRST — the loader
  // set-up and the first operand stage cover the round trip — and carries (rstd, mean rstd) in two registers to the epilogue
  f32x2 lx_pre = {1.f, 0.f};
  const bool lx_on = TAPS == 1 && !SLAB && p.lnx_stat != nullptr;
  LnxRow<1, kLnxMaxParts> lx_row0;
  if (lx_on && t < BM) lx_row0.load(p.lnx_stat, p.lnx_ld, cm0 + t, cm0 + t < p.M, p.lnx_parts, 0);

  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(TAPS == 9 ? p.A2 : p.A), 0, 0x7FFFFFFF, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const int Hv = p.Hi << p.up, Wv = p.Wi << p.up;
  const int lrow = lane >> 3, lch = lane & 7;

  // ---- loader state (static-indexed arrays: fully unrolled)
  unsigned a_off[NP], w_off[NP];
  int a_img[NP], a_iy[NP], a_ix[NP], a_c[NP], w_c[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = (wave * NP + i) * 8 + lrow;
    const int m = cm0 + row;
    a_c[i] = (lch ^ ((row >> 1) & 7)) * 8;
    a_off[i] = OOB;
    a_img[i] = a_ix[i] = 0;
    a_iy[i] = -(1 << 20);
    if (TAPS == 1) {
      if (m < p.M) a_off[i] = (unsigned)m * (unsigned)p.lda * 2u;
    } else if (m < p.M) {
      const int hw = p.Ho * p.Wo;
      const int img = m / hw, rem = m - img * hw;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_img[i] = img * p.Hi * p.Wi;
      a_iy[i] = oy * p.stride - p.pad;
      a_ix[i] = ox * p.stride - p.pad;
    }
    const int n = cn0 + row;
    w_c[i] = a_c[i];
    w_off[i] = n < p.N ? (unsigned)n * (unsigned)p.Ktot * 2u : OOB;
  }
  auto issue = [&](int g, int stage) __attribute__((always_inline)) {
    const int ks = ks_begin + g;
    int tap = 0, kci = ks;
    // second input (IgemmArgs::A2, conv launches only): the k-steps from nk1 on are a tenth "tap" — the output pixel itself
    // (dy = dx = 1 from the padded origin) in ANOTHER tensor with its own row stride and channel count; W column 9 Cin + c
    const bool s2 = TAPS == 9 && ks >= p.nk1;
    if (TAPS != 1) {  // channel chunk outer, tap inner
      kci = ks / 9;
      tap = ks - kci * 9;
      if (s2) {
        kci = ks - p.nk1;
        tap = 9;
      }
    }
    const int c0 = kci * BK;
    const int dy = s2 ? 1 : tap / 3, dx = s2 ? 1 : tap - dy * 3;
    const int cin = s2 ? p.Cin2 : p.Cin;
    const unsigned lda = (unsigned)(s2 ? p.lda2 : p.lda);
    char* sbase = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int c = c0 + a_c[i];
      unsigned vo;
      if (TAPS == 1) {
        vo = (c < p.Cin && a_off[i] != OOB) ? a_off[i] + (unsigned)c * 2u : OOB;
      } else {
        const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
        const bool ok = (c < cin) && ((unsigned)iy < (unsigned)Hv) && ((unsigned)ix < (unsigned)Wv);
        const int sy = iy >> p.up, sx = ix >> p.up;
        const unsigned off = ((unsigned)(a_img[i] + sy * p.Wi + sx) * lda + (unsigned)c) * 2u;
        vo = ok ? off : OOB;
      }
      if (TAPS == 9 && s2)   // (wave-uniform: ks is)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrcA2, (__attribute__((address_space(3))) void*)(sbase + (wave * NP + i) * 1024), 16, vo, 0, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrcA, (__attribute__((address_space(3))) void*)(sbase + (wave * NP + i) * 1024), 16, vo, 0, 0, 0);
    }
#if RCDM_I16_ABLATE & 16   // upper bound of a direct-to-VGPR weight operand: no weight DMA after the first stage
    if (g > 0) return;
#endif
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int c = c0 + w_c[i];
      const unsigned vo = (c < cin && w_off[i] != OOB)
                              ? w_off[i] + ((unsigned)tap * (unsigned)p.Cin + (unsigned)c) * 2u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsrcW, (__attribute__((address_space(3))) void*)(sbase + A_BYTES + (wave * NP + i) * 1024), 16, vo, 0, 0, 0);
    }
  };

  // ---- compute mapping: weights are the A operand, pixels the B operand -> D[channel][pixel]
  const int l15 = lane & 15, kg = lane >> 4;
  const int sw = (lane >> 1) & 7;
  const int koff0 = ((kg ^ sw) << 4), koff1 = (((4 + kg) ^ sw) << 4);
  const int rowA = (wm * (BM / 2) + l15) * 128, rowB = A_BYTES + (wn * (BN / 2) + l15) * 128;
  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // epilogue placement: the tables sit in the ring stage the LAST k-step does not read (stage nkl & 1), the f16 staging tile
  // ([BM] rows of 2 BN + 16 bytes = 53760 B) clear of them: behind the tables when that is stage 0, in front when stage 1
  constexpr int kStg = BM * (2 * BN + 16);
  static_assert(lnx_table_bytes(BM, BN) <= 4096 && 4096 + kStg <= 2 * STAGE && kStg >= STAGE, "epilogue layout");
  const bool lx_low = lx_on && !(nkl & 1);
  float* const lx_tab = (float*)(smem + (lx_low ? 0 : kStg));
  char* const stg = smem + (lx_low ? 4096 : 0);
  issue(0, 0);
  if (lx_on && t < BM) {
    float r_ = 1.f, m_ = 0.f;
    lx_row0.finish(p.lnx_invC, p.lnx_eps, r_, m_);
    lx_pre = f32x2{r_, m_};
  }
  for (int g = 0; g < nkl; ++g) {
    // this wave's pieces of step g have landed (nothing newer is in flight); the barrier makes everybody's visible and
    // says everybody is done reading the stage of step g-1, which the issue below refills
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
#if !(RCDM_I16_ABLATE & 8)
    if (g + 1 < nkl) issue(g + 1, (g + 1) & 1);
#endif
    const char* sb = smem + (g & 1) * STAGE;
#if RCDM_PRIO_LOADS   // everything but the MFMA runs of a k-step at raised priority (see igemm8.hip compute())
    __builtin_amdgcn_s_setprio(3);
#endif
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int koff = kk ? koff1 : koff0;
      f16x8 wf[FN], xf[FM];
#pragma unroll
      for (int i = 0; i < FN; ++i) wf[i] = *(const f16x8*)(sb + rowB + i * 2048 + koff);
#pragma unroll
      for (int j = 0; j < FM; ++j) xf[j] = *(const f16x8*)(sb + rowA + j * 2048 + koff);
#if RCDM_PRIO_LOADS
      __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
#if RCDM_PRIO_LOADS
      __builtin_amdgcn_s_setprio(3);
#endif
    }
  }
  // the ring stage the last k-step did not read is dead (read by nobody, filled by nobody): the deferred-LayerNorm tables go
  // there NOW, so that the barrier below — which the epilogue needs anyway — publishes them (igemm_epilogue.h
  // lnx_write_tables).  Not inside the loop: an ordinary global load anywhere in its body makes hipcc drain the LDS-DMA
  // queue (vmcnt(0)) in front of every step's ds_reads (+0.3 ms per step, measured).
  if constexpr (TAPS == 1 && !SLAB)
    if (lx_on) lnx_write_tables<BM, BN>(p, lx_tab, cm0, cn0, t, lx_pre);
  wait_lgkm0();
  tick_barrier();  // every wave is done reading the ring: the LDS is free for the epilogue's staging tile
#if RCDM_I16_ABLATE & 1
  {  // debug build: k-loop only (the accumulators stay live through a store that never happens)
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 1.2345678e33f) p.out[0] = (f16)s;
  }
#else
  if constexpr (SLAB) {
    if (p.slab16) {   // f16 slab: the plain-projection path of the staged epilogue, pointed at this split's slab
      IgemmArgs q{};
      q.out = (f16*)p.partial + (size_t)blockIdx.y * p.M * p.N;
      q.M = p.M; q.N = p.N; q.ldc = p.N; q.epi = 0; q.out_scale = 1.0f; q.dup = 0;
      tile_epilogue<FM, FN, false, 256, BM, BN, false, false>(q, smem, acc, cm0, cn0, wm * (BM / 2), wn * (BN / 2), l15, kg, t, nullptr);
      return;
    }
  }
  tile_epilogue<FM, FN, SLAB, 256, BM, BN, false, TAPS == 1 && !SLAB>(p, stg, acc, cm0, cn0, wm * (BM / 2), wn * (BN / 2), l15, kg, t,
                                                                       lx_on ? lx_tab : nullptr);
#endif
}

constexpr int kLds16 = 2 * (160 + 160) * 128;  // 81920 B >= the 160 x (320 + 16) B staging tile

template <int TAPS>
int launch16(const IgemmArgs& a, hipStream_t stream) {
  static bool attr_set[64] = {};
  if (rcdm_first_on_device(attr_set)) {
    (void)hipFuncSetAttribute((const void*)igemm16_kernel<TAPS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds16);
    (void)hipFuncSetAttribute((const void*)igemm16_kernel<TAPS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds16);
  }
  dim3 grid(a.tilesM * a.tilesN, a.splits);
  if (a.splits > 1)
    hipLaunchKernelGGL((igemm16_kernel<TAPS, true>), grid, dim3(256), kLds16, stream, a);
  else
    hipLaunchKernelGGL((igemm16_kernel<TAPS, false>), grid, dim3(256), kLds16, stream, a);
  return rcdm_check_launch();
}

}  // namespace

int rcdm_igemm16_launch(const IgemmArgs& a, int taps, hipStream_t stream) {
  if (taps == 1) return launch16<1>(a, stream);
  if (taps == 9) return launch16<9>(a, stream);
  return RCDM_EINVAL;
}
