// rowff.hip — ROW-STATIONARY fused feed-forward for gfx950:  out = x + FeedForward_geglu(LayerNorm(x))  in ONE launch.
// Replaces the chain  nn.LayerNorm -> diffusers FeedForward(dim, activation_fn="geglu") -> + hidden_states  of
// BasicTransformerBlock.forward (src/models/attention.py:514: norm3 -> ff -> +) and TemporalTransformerBlock.forward
// (src/models/motion_module.py:243: ff_norm -> ff -> +), which the library otherwise runs as three launches
// (rcdm_layernorm, rcdm_gemm with the GEGLU epilogue, rcdm_gemm with bias + residual) with the 4C-wide hidden tensor
// written to and read back from HBM (105 MB per call at the 64x64 level of the 512x512 UNet).
//
// Decomposition (round 3): the implicit-GEMM kernels tile BOTH dimensions, so their operands
// stream L2 -> LDS for every tile and every intermediate tensor round-trips HBM.  Here a WAVE owns 16 PF token rows for
// the whole chain and only the weights move:
//   * the wave's LayerNorm'ed rows live in registers as the B operands of v_mfma_f32_16x16x32_f16 (PF x C / 32
//     fragments), loaded once;
//   * the fp32 output accumulator of those rows lives in registers (PF x C / 16 fragments);
//   * hidden units are processed in groups of 32: GEMM1 (K = C) for 16 hidden + 16 gate columns, twice, gives the GEGLU
//     inputs in the MFMA accumulator layout, lane (pixel, kg) holding hidden units 4 kg .. 4 kg + 3 of each 16 — which
//     IS the B-operand layout of the next MFMA if the k-slots of W2 are permuted to match (done once, in the weight
//     pack): h = (a_h + b_h) * gelu(a_g + b_g) goes from accumulator registers straight into GEMM2 (K = 32,
//     C / 16 MFMAs per 16 rows into the resident output accumulator).  The hidden tensor never exists, not even in LDS;
//   * the only LDS traffic of the loop is the weight stream: W1 / W2 are packed FRAGMENT-MAJOR in consumption order
//     (rcdm_pack_ff_stream: every MFMA A operand is one contiguous 1-KiB block in lane order), so a buffer_load ... lds
//     piece is one fragment, LDS is written and read linearly (no swizzle, no bank conflicts), and the whole LDS is one
//     ring of R chunks of 2 C / 32 fragments that all NW waves of the block consume in lock step (one s_barrier per
//     chunk, counted vmcnt).  Weight bytes L2 -> LDS per block: 24 C^2 (2.4 MB at C = 320) for 16 PF NW rows.
// Every wave reads every weight fragment from LDS once per PF MFMAs; with PF = 1 and ten waves per CU (the first version)
// the LDS pipe was as loaded as the matrix pipe and the two did not overlap (measured: reads alone 1000 clocks per chunk,
// MFMAs alone 1000, together 1500).  Hence PF = 3 on FOUR waves, one per SIMD, each with the whole 512-register file:
// 192-row blocks, 0.4 x the LDS traffic per flop.
#include "common.h"
#include "pp_sync.h"
#include <type_traits>

#ifndef RCDM_FF_PD
#define RCDM_FF_PD 4   // fragment reads in flight ahead of the MFMAs that consume them
#endif
#ifndef RCDM_FF_ABLATE
#define RCDM_FF_ABLATE 0  // debug builds (wrong results, timing only): 1 no DMA after the prefill, 2 no MFMA, 4 no fragment
#endif                    // reads, 8 no barrier in the loop, 16 no epilogue, 32 no GEGLU arithmetic

namespace {

struct FFArgs {
  const f16* x;        // [M][ldx]
  f16* out;            // [M][ldo]  (may alias x: every block reads and writes only its own rows)
  const float* ln_g;   // [C]
  const float* ln_b;   // [C]
  const f16* wstream;  // rcdm_pack_ff_stream
  const float* b1p;    // [8C] packed: per (group, pair) 16 hidden biases then their 16 gate biases
  const float* b2;     // [C]
  int M, ldx, ldo;
  float eps;
};

template <int C, int NW, int PF, int R, int PD>
__global__ __launch_bounds__(NW * 64) void ff_rows_kernel(const FFArgs p) {
  constexpr int NK = C / 32;        // k-steps of GEMM1 = register fragments of 16 LayerNorm'ed rows
  constexpr int NOF = C / 16;       // output fragments (16 channels each)
  constexpr int CHF = 2 * NK;       // fragments per chunk (== NOF)
  constexpr int CHB = CHF * 1024;   // chunk bytes
  constexpr int NG = C / 8;         // groups of 32 hidden units (4C / 32)
  constexpr int NCH = 3 * NG;       // chunks: A_g / B_g = GEMM1 of the two 16 + 16 column sets of group g, C_g = GEMM2
  constexpr int NT = NW * 64;
  constexpr int PPW = CHF / NW;     // DMA pieces per wave and chunk
  constexpr int WR = 16 * PF;       // rows per wave
  static_assert(CHF % NW == 0, "chunk fragments must divide over the waves");
  static_assert(NOF == CHF, "chunk size");
  static_assert((3 * CHF) % PD == 0, "the three chunk positions of the loop body must repeat their rotation phases");
  static_assert(CHF >= 20, "five GEGLU slices per value");
  constexpr int BIAS0 = R * CHB;            // b1p (8C floats), then gamma, beta (C floats each)
  constexpr int GAM0 = BIAS0 + 8 * C * 4, BET0 = GAM0 + C * 4;
  constexpr int RS = 2 * C + 16;            // staged output row (bytes)
  static_assert(NW * WR * RS <= R * CHB, "epilogue staging exceeds the ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l15 = lane & 15, kg = lane >> 4;
  const int row0 = blockIdx.x * (NW * WR) + wave * WR;

  const __amdgpu_buffer_rsrc_t rsrcW =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.wstream, 0, (unsigned)NCH * CHB, 0x00020000);
  int dma_voff = lane * 16;
  auto issue_chunk = [&](int c, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wave * PPW + i;
      // per-lane part of the address in the (loop-invariant) voffset, the chunk / piece part in the scalar offset.  The
      // descriptor's range check sees the voffset only: past the end of the stream it is set out of range explicitly
      // (such a piece moves nothing and writes zeros)
      const unsigned voff = c < NCH ? (unsigned)dma_voff : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void*)(smem + slot * CHB + piece * 1024),
                                               16, voff, c < NCH ? c * CHB + piece * 1024 : 0, 0, 0);
    }
  };

  // ---- prologue: plain loads first (they are waited for with the DMA prefill still in flight behind them)
  Pack16 xr[PF][NK];
#pragma unroll
  for (int f = 0; f < PF; ++f) {
    const int row = row0 + 16 * f + l15;
    const f16* xp = p.x + (size_t)(row < p.M ? row : 0) * p.ldx + 8 * kg;
#pragma unroll
    for (int s = 0; s < NK; ++s) xr[f][s].u = *(const uint4*)(xp + 32 * s);
  }
  constexpr int NB4 = (8 * C + 2 * C) / 4;  // float4 items of [b1p | gamma | beta]
  constexpr int NB4_PER = (NB4 + NT - 1) / NT;
  f32x4 bq[NB4_PER];
#pragma unroll
  for (int i = 0; i < NB4_PER; ++i) {
    const int idx = min(i * NT + t, NB4 - 1);
    const float* src = idx < 2 * C ? p.b1p + 4 * idx : (idx < 2 * C + C / 4 ? p.ln_g + 4 * (idx - 2 * C) : p.ln_b + 4 * (idx - 2 * C - C / 4));
    bq[i] = *(const f32x4*)src;
  }
  int cslot = 0;  // slot of the chunk consumed next
#pragma unroll
  for (int c = 0; c < R - 1; ++c) issue_chunk(c, c);
#pragma unroll
  for (int i = 0; i < NB4_PER; ++i) {
    const int idx = i * NT + t;
    if (idx < NB4) *(f32x4*)(smem + BIAS0 + 16 * idx) = bq[i];
  }

  // LayerNorm statistics of this lane's rows (two passes over registers: mean, then squared deviations)
  float mean[PF], rstd[PF];
#pragma unroll
  for (int f = 0; f < PF; ++f) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)xr[f][k].e[j];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    mean[f] = s * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = (float)xr[f][k].e[j] - mean[f];
        q += d * d;
      }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    rstd[f] = rsqrtf(q * (1.0f / C) + p.eps);
  }
  wait_lgkm0();
  wait_vm<(R - 2) * PPW>();  // this wave's pieces of chunk 0 (the compiler's own wait for the plain loads came earlier)
  tick_barrier();            // gamma / beta / b1p and chunk 0 are in LDS
  f16x8 xf[PF][NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const f32x4 g0 = *(const f32x4*)(smem + GAM0 + (32 * k + 8 * kg) * 4), g1 = *(const f32x4*)(smem + GAM0 + (32 * k + 8 * kg + 4) * 4);
    const f32x4 b0 = *(const f32x4*)(smem + BET0 + (32 * k + 8 * kg) * 4), b1 = *(const f32x4*)(smem + BET0 + (32 * k + 8 * kg + 4) * 4);
#pragma unroll
    for (int f = 0; f < PF; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xf[f][k][j] = (f16)(((float)xr[f][k].e[j] - mean[f]) * rstd[f] * g0[j] + b0[j]);
        xf[f][k][4 + j] = (f16)(((float)xr[f][k].e[4 + j] - mean[f]) * rstd[f] * g1[j] + b1[j]);
      }
  }

  f32x4 oacc[PF][NOF];
#pragma unroll
  for (int f = 0; f < PF; ++f)
#pragma unroll
    for (int i = 0; i < NOF; ++i) oacc[f][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- the weight stream is ONE software pipeline over all NCH * CHF fragments: the read of fragment f + PD is issued
  // with the MFMAs of fragment f, across chunk boundaries (no drain at the barriers).  Chunk hand-over at the top of
  // chunk c: this wave's pieces of chunk c + 1 have landed (counted vmcnt); the barrier makes everybody's visible — so
  // reads may run ahead into chunk c + 1 during chunk c — and says everybody is done with chunk c - 1, whose slot the
  // issue refills with chunk c + R - 1.  Past the end of the stream the refill pieces are out of range: they move
  // nothing, write zeros into slots nobody reads again, and keep the counted wait the same for every chunk.
  //
  // Chunk order in the stream (rcdm_pack_ff_stream): A_0, then per group g: B_g, A_{g+1}, C_g  (A / B = GEMM1 of the
  // first / second 16 hidden + 16 gate columns of the group, C = GEMM2).  The GEGLU arithmetic of a GEMM1 chunk's
  // accumulators runs UNDER the next chunk's MFMAs (every wave of the block is in the same chunk, so VALU work placed
  // between chunks would leave all four matrix pipes idle at once).
  int c = 0;
  // the one lane-dependent value the loop keeps: DMA voffset and LDS read base.  Opaque, and the epilogue re-derives its
  // lane indices from it, so that no second copy of the thread id has to survive the loop
  int lane16 = lane * 16;
  asm volatile("" : "+v"(lane16));
  dma_voff = lane16;
  auto rd = [&](const char* q) __attribute__((always_inline)) -> f16x8 {
    if (RCDM_FF_ABLATE & 4) { f16x8 z; for (int e = 0; e < 8; ++e) z[e] = (f16)(float)lane; return z; }
    return *(const f16x8*)q;
  };
  auto mm = [&](f16x8 a, f16x8 b, f32x4 acc) __attribute__((always_inline)) -> f32x4 {
    if (RCDM_FF_ABLATE & 2) { asm volatile("" ::"v"(a), "v"(b)); return acc; }
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  };
  f16x8 fr[PD];
#pragma unroll
  for (int i = 0; i < PD; ++i) fr[i] = rd(smem + lane16 + i * 1024);

  // GEGLU of one GEMM1 chunk's accumulators, h = (a_h + b_h) * gelu(a_g + b_g) for the lane's four hidden units, cut into
  // CHF slices of 3-4 plain fp32 VALU instructions (packed fp32 issues slower next to MFMAs), one slice per MFMA step of
  // the chunk it runs under.  gelu(x) = max(x, 0) - |x| q(|x|), q = poly(t) exp(-x^2 / 2) / 2,
  // t = 1 / (1 + p |x| / sqrt 2)  (Abramowitz & Stegun 7.1.26, common.h gelu2).  Value e uses slices 5 e .. 5 e + 4.
  // The empty asm statements pin every slice where the source puts it: without them the optimizer sinks the whole (pure)
  // computation to its first use, after the chunk's MFMAs.
  struct GegluState { float bh, x, t, pl, ex; };
  auto geglu_slice = [&](int i, GegluState& st, const f32x4& ah, const f32x4& ag, int gp, f16* dst) __attribute__((always_inline)) {
    const int e = i / 5, stage = i - 5 * e;
    if (e >= 4) return;
    const float* bp = (const float*)(smem + BIAS0 + (lane16 >> 8) * 16) + gp * 32 + e;  // this value's two biases (LDS broadcast reads)
    if (RCDM_FF_ABLATE & 32) {
      if (stage == 4) dst[e] = (f16)((ah[e] + bp[0]) * (ag[e] + bp[16]));
      return;
    }
    if (stage == 0) {
      st.x = ag[e] + bp[16];
      st.bh = bp[0];
      st.t = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(st.x), 0.2316418882f, 1.0f));
      st.ex = st.x * st.x;
      asm volatile("" : "+v"(st.t), "+v"(st.ex));
    } else if (stage == 1) {
      st.ex = __builtin_amdgcn_exp2f(st.ex * -0.72134752044f);
      st.pl = __builtin_fmaf(st.t, 0.5307027145f, -0.7265760135f);
      st.pl = __builtin_fmaf(st.pl, st.t, 0.7107068705f);
      asm volatile("" : "+v"(st.pl), "+v"(st.ex));
    } else if (stage == 2) {
      st.pl = __builtin_fmaf(st.pl, st.t, -0.142248368f);
      st.pl = __builtin_fmaf(st.pl, st.t, 0.127414796f);
      st.t = st.pl * st.t;
      asm volatile("" : "+v"(st.t));
    } else if (stage == 3) {
      st.t = st.t * st.ex;
      st.pl = __builtin_fmaxf(st.x, 0.0f);
      st.t = __builtin_fmaf(-__builtin_fabsf(st.x), st.t, st.pl);
      asm volatile("" : "+v"(st.t));
    } else {
      float hv = (ah[e] + st.bh) * st.t;
      asm volatile("" : "+v"(hv));
      dst[e] = (f16)hv;
    }
  };
  // one chunk: CHF steps of [PF MFMAs on fragment i | read of fragment i + PD | slice i of the GEGLUs], each step its own
  // scheduling region (sched_barrier): the interleave is what the source says.
  // PHT: the pipeline registers rotate continuously over the stream, fragment f lives in fr[f % PD]; a chunk that starts
  // at stream fragment k CHF begins at rotation (k CHF) % PD (compile-time: register indices must be static)
  auto run_chunk = [&](auto&& mfma_i, auto&& slice_i, auto pht) __attribute__((always_inline)) {
    constexpr int PH = decltype(pht)::value;
    wait_vm<(R - 3) * PPW>();
    if (!(RCDM_FF_ABLATE & 8)) tick_barrier();
    if (!(RCDM_FF_ABLATE & 1)) issue_chunk(c + R - 1, cslot == 0 ? R - 1 : cslot - 1);
    const char* sb = smem + cslot * CHB + lane16;
    const char* sn = smem + (cslot == R - 1 ? 0 : cslot + 1) * CHB + lane16;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < CHF; ++i) {
      const f16x8 cur = fr[(i + PH) % PD];
      mfma_i(i, cur);
      fr[(i + PH) % PD] = rd(i + PD < CHF ? sb + (i + PD) * 1024 : sn + (i + PD - CHF) * 1024);
      slice_i(i);
      __builtin_amdgcn_sched_barrier(0);
    }
    ++c;
    cslot = cslot == R - 1 ? 0 : cslot + 1;
  };
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  // two accumulator sets: A chunks accumulate into (ahA, agA), read by the GEGLU that runs under the following B chunk;
  // B chunks into (ahB, agB), read under the following A chunk.  The first MFMA of a chunk takes C = 0.
  f32x4 ahA[PF], agA[PF], ahB[PF], agB[PF];
  union HB { f16x8 v; f16 e[8]; } hb[PF];
  GegluState gs[PF];
#pragma unroll
  for (int f = 0; f < PF; ++f) ahA[f] = agA[f] = ahB[f] = agB[f] = z4;
  auto g1A = [&](int i, f16x8 w) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < PF; ++f) {
      if (i & 1) agA[f] = mm(w, xf[f][i >> 1], i < 2 ? z4 : agA[f]); else ahA[f] = mm(w, xf[f][i >> 1], i < 2 ? z4 : ahA[f]);
    }
  };
  auto g1B = [&](int i, f16x8 w) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < PF; ++f) {
      if (i & 1) agB[f] = mm(w, xf[f][i >> 1], i < 2 ? z4 : agB[f]); else ahB[f] = mm(w, xf[f][i >> 1], i < 2 ? z4 : ahB[f]);
    }
  };
  auto g2 = [&](int i, f16x8 w) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < PF; ++f) oacc[f][i] = mm(w, hb[f].v, oacc[f][i]);
  };
  auto nothing = [&](int) __attribute__((always_inline)) {};
  int g = 0;
  auto sliceA = [&](int i) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < PF; ++f) geglu_slice(i, gs[f], ahA[f], agA[f], 2 * g, hb[f].e);
  };
  auto sliceB = [&](int i) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < PF; ++f) geglu_slice(i, gs[f], ahB[f], agB[f], 2 * g + 1, hb[f].e + 4);
  };

  // rotation phases of the chunk positions: A_0 is chunk 0, the loop body is chunks 3g+1 (B), 3g+2 (A), 3g+3 (C); the
  // last group has no A chunk, so its C chunk is chunk 3 NG - 1
  using P0 = std::integral_constant<int, 0>;
  using PB = std::integral_constant<int, (1 * CHF) % PD>;
  using PA = std::integral_constant<int, (2 * CHF) % PD>;
  using PC = std::integral_constant<int, (3 * CHF) % PD>;
  using PL = std::integral_constant<int, ((3 * NG - 1) * CHF) % PD>;
  run_chunk(g1A, nothing, P0{});  // A_0
  for (; g < NG - 1; ++g) {
    run_chunk(g1B, sliceA, PB{});   // B_g under GEGLU(A_g)
    run_chunk(g1A, sliceB, PA{});   // A_{g+1} under GEGLU(B_g)
    run_chunk(g2, nothing, PC{});   // C_g
  }
  run_chunk(g1B, sliceA, PB{});
#pragma unroll
  for (int i = 0; i < CHF; ++i) sliceB(i);
  run_chunk(g2, nothing, PL{});  // C of the last group
  wait_lgkm0();
  wait_vm<0>();    // the zero-fill pieces issued past the end of the stream
  tick_barrier();  // every wave is done reading the ring and nothing is in flight into it

  // ---- epilogue: accumulators -> f16 rows in this wave's private staging region -> coalesced 16-byte pass with
  // bias + residual (the same two roundings as the unfused rcdm_gemm epilogue)
  if (RCDM_FF_ABLATE & 16) {
    float sacc = 0.f;
#pragma unroll
    for (int f = 0; f < PF; ++f)
#pragma unroll
      for (int i = 0; i < NOF; ++i) sacc += oacc[f][i][0] + oacc[f][i][1] + oacc[f][i][2] + oacc[f][i][3];
    if (sacc == 1.2345678e33f) p.out[0] = (f16)sacc;
    return;
  }
  char* st = smem + wave * WR * RS;
  const int lane_e = lane16 >> 4, l15_e = lane_e & 15, kg_e = lane_e >> 4;
#pragma unroll
  for (int f = 0; f < PF; ++f)
#pragma unroll
    for (int i = 0; i < NOF; ++i) {
      union { f16 h[4]; uint2 u; } pk;
#pragma unroll
      for (int e = 0; e < 4; ++e) pk.h[e] = (f16)oacc[f][i][e];
      *(uint2*)(st + (16 * f + l15_e) * RS + (16 * i + 4 * kg_e) * 2) = pk.u;
    }
  wait_lgkm0();
  constexpr int CPR = C / 8, ITEMS = WR * CPR, NIT = (ITEMS + 63) / 64;
  constexpr int U = 5;  // residual loads in flight per lane
  static_assert(NIT % U == 0, "epilogue batches");
  for (int it0 = 0; it0 < NIT; it0 += U) {
    Pack16 rr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = min((it0 + u) * 64 + lane_e, ITEMS - 1);
      const int r = idx / CPR, c8 = idx - r * CPR;
      const int m = min(row0 + r, p.M - 1);
      rr[u].u = *(const uint4*)(p.x + (size_t)m * p.ldx + c8 * 8);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = (it0 + u) * 64 + lane_e;
      const int r = idx / CPR, c8 = idx - r * CPR;
      const int m = row0 + r;
      if (idx < ITEMS && m < p.M) {
        Pack16 v, o;
        v.u = *(const uint4*)(st + r * RS + c8 * 16);
        const f32x4 a0 = *(const f32x4*)(p.b2 + c8 * 8), a1 = *(const f32x4*)(p.b2 + c8 * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o.e[e] = (f16)((float)v.e[e] + a0[e] + (float)rr[u].e[e]);
          o.e[4 + e] = (f16)((float)v.e[4 + e] + a1[e] + (float)rr[u].e[4 + e]);
        }
        *(uint4*)(p.out + (size_t)m * p.ldo + c8 * 8) = o.u;
      }
    }
  }
}

// fp32 [8C][C] / [8C] / [C][4C] -> the fragment-major stream + packed bias the kernel consumes
__global__ void pack_ff_stream_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                      int C, f16* __restrict__ ws, float* __restrict__ b1p) {
  const int NK = C / 32, CHF = 2 * NK;
  const size_t total = (size_t)12 * C * C;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(idx & 7), L = (int)((idx >> 3) & 63);
    const size_t fr = idx >> 9;
    const int f = (int)(fr % CHF), c = (int)(fr / CHF);
    // chunk order: A_0, then per group g: B_g, A_{g+1} (absent for the last group), C_g
    int g, part;  // part 0 = A (first 16 hidden + gate columns of the group), 1 = B (second 16), 2 = C (GEMM2)
    if (c == 0) { g = 0; part = 0; }
    else {
      const int NGr = C / 8, q = c - 1;
      if (q < 3 * (NGr - 1)) {
        const int gg = q / 3, r = q - 3 * gg;
        if (r == 0) { g = gg; part = 1; } else if (r == 1) { g = gg + 1; part = 0; } else { g = gg; part = 2; }
      } else {
        g = NGr - 1; part = (q - 3 * (NGr - 1)) == 0 ? 1 : 2;
      }
    }
    const int l15 = L & 15, kg = L >> 4;
    float v;
    if (part < 2) {
      const int s = f >> 1, gate = f & 1;
      const int hr = 32 * g + 16 * part + l15;
      v = w1[(size_t)(gate ? 4 * C + hr : hr) * C + 32 * s + 8 * kg + j];
    } else {
      const int hid = 32 * g + (j < 4 ? 4 * kg + j : 16 + 4 * kg + (j - 4));
      v = w2[(size_t)(16 * f + l15) * (4 * C) + hid];
    }
    ws[idx] = (f16)v;
    if (idx < (size_t)8 * C) {  // packed bias: [(g, pair)][16 hidden | 16 gate]
      const int q = (int)idx, gp = q >> 5, l = q & 31;
      const int hr = 16 * gp + (l & 15);
      b1p[q] = b1[l < 16 ? hr : 4 * C + hr];
    }
  }
}

template <int C, int NW, int PF, int R, int PD>
int launch_ff(const FFArgs& a, hipStream_t stream) {
  constexpr int CHB = 2 * (C / 32) * 1024;
  constexpr int LDS = R * CHB + 10 * C * 4;
  static_assert(LDS <= 160 * 1024, "LDS");
  int dev = 0;
  (void)hipGetDevice(&dev);
  static bool attr_set[64] = {};
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)ff_rows_kernel<C, NW, PF, R, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) !=
        hipSuccess)
      return RCDM_ELAUNCH;
    attr_set[dev] = true;
  }
  const int rows = NW * 16 * PF;
  const int nblocks = (a.M + rows - 1) / rows;
  hipLaunchKernelGGL((ff_rows_kernel<C, NW, PF, R, PD>), dim3(nblocks), dim3(NW * 64), LDS, stream, a);
  return rcdm_check_launch();
}

int g_ff_variant = -1;  // -1: automatic (= 0); 0: 10 waves x 16 rows; 1: 4 waves x 48 rows

}  // namespace

extern "C" {

size_t rcdm_ff_stream_bytes(int32_t C) { return C > 0 ? (size_t)24 * C * C : 0; }

int rcdm_ff_fused_supported(int32_t C) { return C == 320 ? 1 : 0; }

int rcdm_set_ff_variant(int32_t v) {
  if (v < -1 || v > 1) return RCDM_EINVAL;
  g_ff_variant = v;
  return RCDM_OK;
}

int rcdm_pack_ff_stream(const float* w1, const float* b1, const float* w2, int32_t C, void* wstream, float* b1_packed,
                        void* stream) {
  if (!w1 || !b1 || !w2 || !wstream || !b1_packed || C <= 0) return RCDM_EINVAL;
  if (C % 32) return RCDM_ESHAPE;
  const size_t n = (size_t)12 * C * C;
  size_t g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(pack_ff_stream_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w1, b1, w2, C, (f16*)wstream,
                     b1_packed);
  return rcdm_check_launch();
}

int rcdm_ff_fused(const rcdm_ff_desc* d, const void* x, const float* ln_gamma, const float* ln_beta, const void* wstream,
                  const float* b1_packed, const float* b2, void* out, void* stream) {
  if (!d || !x || !ln_gamma || !ln_beta || !wstream || !b1_packed || !b2 || !out) return RCDM_EINVAL;
  if (d->M <= 0 || d->ldx < d->C || d->ldo < d->C || (d->ldx & 7) || (d->ldo & 7)) return RCDM_EINVAL;
  FFArgs a{(const f16*)x, (f16*)out, ln_gamma, ln_beta, (const f16*)wstream, b1_packed, b2, d->M, d->ldx, d->ldo, d->eps};
  switch (d->C) {
    case 320:
      // measured (tools/kbench.py ff, 40960 rows): ten waves x 16 rows 118 us, four waves x 48 rows 147 us (one wave per
      // SIMD issues one instruction per ~4 clocks: MFMAs + GEGLU arithmetic + reads do not fit), the unfused chain 163 + 15
      if (g_ff_variant == 1) return launch_ff<320, 4, 3, 7, RCDM_FF_PD>(a, (hipStream_t)stream);
      return launch_ff<320, 10, 1, 7, 2>(a, (hipStream_t)stream);
    default: return RCDM_ESHAPE;
  }
}

}  // extern "C"
