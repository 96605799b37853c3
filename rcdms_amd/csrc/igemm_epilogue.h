// igemm_epilogue.h — tile epilogue shared by the phase-structured implicit-GEMM kernels (igemm8.hip: 8 waves,
// igemm16.hip: 4 waves).  Called when every wave of the block is past its k-loop: no LDS read and no DMA is outstanding,
// so the whole dynamic LDS is free for staging.
// Accumulator layout (v_mfma_f32_16x16x32_f16, weights = A operand): acc[i][j][e] = D[channel col0 + 16 i + 4 kg + e]
// [pixel row0 + 16 j + l15] of the BM x BN tile at (cm0, cn0).
#pragma once
#include "common.h"
#include "igemm_args.h"
#include "pp_sync.h"
#ifndef RCDM_LNX_ABLATE
#define RCDM_LNX_ABLATE 0   // debug builds (tools/lnx_bench.py): 1 = no partial-statistics loads
#endif

__device__ __forceinline__ void epi_store16(f16* dst, uint4 v) { *(uint4*)dst = v; }  // (non-temporal: measured, no change)

// Deferred LayerNorm of the A rows (rcdm_gemm_lnx), table side.  Called by the kernel INSIDE its last k-step, after that
// step's barrier, with `tab` in the ring stage the last step does not read (nobody reads or fills it any more): thread
// t < BM publishes row t's (rstd, mean rstd) — lx_pre, summed from the partial statistics it requested at kernel start —
// and threads t < BN / 4 the tile's per-column vectors: S, bias + row vector of the first sample the tile meets, bias + row
// vector of the second.  The barrier that ends the k-loop makes them visible: no barrier of their own.
// Layout behind tab: [BM][2] (rstd, mean rstd) | [BN] S | [BN] bias (+ row vector, first sample) | [BN] the same, second sample.
template <int BM, int BN>
__device__ __forceinline__ void lnx_write_tables(const IgemmArgs& p, float* tab, int cm0, int cn0, int t, f32x2 lx_pre) {
  const bool lx_rvp = (p.epi & RCDM_EPI_ROWVEC) != 0 && p.rows_per_sample >= BM;   // the tile meets at most two samples
  const int smp0 = lx_rvp ? cm0 / p.rows_per_sample : 0;
  const int sw = lx_rvp ? (smp0 + 1) * p.rows_per_sample : 0x7fffffff;
  const int vn = cn0 + 4 * t;
  f32x4 vS = {0.f, 0.f, 0.f, 0.f}, vB = vS, vR0 = vS, vR1 = vS;
  if (t < BN / 4 && vn < p.N) {
    vS = *(const f32x4*)(p.lnx_S + vn);
    if (p.epi & RCDM_EPI_BIAS) vB = *(const f32x4*)(p.bias + vn);
    if (lx_rvp) {
      vR0 = *(const f32x4*)(p.rowvec + (size_t)smp0 * p.ldt + vn);
      if (sw < p.M && sw < cm0 + BM) vR1 = *(const f32x4*)(p.rowvec + (size_t)(smp0 + 1) * p.ldt + vn);
    }
  }
  if (t < BM) *(f32x2*)(tab + 2 * t) = lx_pre;
  if (t < BN / 4) {
    *(f32x4*)(tab + 2 * BM + 4 * t) = vS;
    *(f32x4*)(tab + 2 * BM + BN + 4 * t) = vB + vR0;
    *(f32x4*)(tab + 2 * BM + 2 * BN + 4 * t) = vB + vR1;
  }
}
constexpr int lnx_table_bytes(int BM, int BN) { return (2 * BM + 3 * BN) * 4; }

// fp32 tile -> slab through LDS, whole rows per store instruction (round 6).  The accumulator layout gives a lane 4 consecutive
// channels of ONE row: stored directly, a wave's store instruction covers 16 rows x 64 bytes — half-line pieces the memory system
// merges badly (the fp32 slab epilogue of the 16x16-level Winograd GEMM cost 15 us where the f16 staged one cost 5).  Here the
// tile goes through LDS in passes of RP rows ([RP][BN] floats, rows padded by 16 B): every wave writes the fragments whose rows
// fall in the pass, then all NT threads store 16 bytes per lane along the rows (BN * 4 contiguous bytes per row).
// Caller: every wave past its k-loop and past a barrier (the ring is free); RP % 16 == 0, RP * (4 BN + 16) bytes of LDS at stg.
template <int FMW, int FNW, int NT, int BM, int BN, int RP>
__device__ __forceinline__ void slab_store_staged(float* dst, int ldn, int Mlim, int Nlim, char* stg, f32x4 (&acc)[FNW][FMW],
                                                  int cm0, int cn0, int row0, int col0, int l15, int kg, int t) {
  static_assert(RP % 16 == 0 && BM % RP == 0, "pass rows");
  constexpr int RS = 4 * BN + 16;
  constexpr int CPR = BN / 4, ITEMS = RP * CPR;
#pragma unroll
  for (int pass = 0; pass < BM / RP; ++pass) {
#pragma unroll
    for (int j = 0; j < FMW; ++j) {
      const int row = row0 + j * 16 + l15 - pass * RP;
      if ((row0 + j * 16) / RP == pass) {   // (wave-uniform)
#pragma unroll
        for (int i = 0; i < FNW; ++i) *(f32x4*)(stg + row * RS + (col0 + i * 16 + 4 * kg) * 4) = acc[i][j];
      }
    }
    wait_lgkm0();
    tick_barrier();
    for (int base = 0; base < ITEMS; base += NT * 4) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = min(base + u * NT + t, ITEMS - 1);
        const int row = idx / CPR, c = idx - row * CPR;
        v[u] = *(const f32x4*)(stg + row * RS + c * 16);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = base + u * NT + t;
        const int row = idx / CPR, c = idx - row * CPR;
        const int m = cm0 + pass * RP + row, n = cn0 + c * 4;
        if (idx < ITEMS && m < Mlim && n < Nlim) *(f32x4*)(dst + (size_t)m * ldn + n) = v[u];
      }
    }
    if (pass + 1 < BM / RP) {
      wait_lgkm0();
      tick_barrier();   // everybody has read this pass before the next one overwrites it
    }
  }
}

// stg: where the f16 staging tile goes ([BM] rows of 2 BN + 16 bytes); ltab: the tables lnx_write_tables left (nullptr: none)
template <int FMW, int FNW, bool SLAB, int NT, int BM, int BN, bool LN_OK = false, bool LX = false, bool PH = false>  // LX: rcdm_gemm_lnx consumer epilogues compiled in (GEMM launches only); PH: phase launch (bias-only epilogues, rows remapped)
__device__ __forceinline__ void tile_epilogue(const IgemmArgs& p, char* stg, f32x4 (&acc)[FNW][FMW], int cm0, int cn0,
                                              int row0, int col0, int l15, int kg, int t, const float* ltab = nullptr) {
  char* const smem = stg;
#if RCDM_PRIO_LOADS   // the epilogue's VALU work at raised priority against the co-resident block's MFMA runs
  __builtin_amdgcn_s_setprio(3);
#endif
  if constexpr (SLAB) {
    // ---- split-K: the fp32 tile goes to this split's slab (16 B per lane, 64-B runs per pixel row); bias / row
    // vector / residual / GEGLU belong to splitk_reduce_kernel
    float* dst = p.partial + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int j = 0; j < FMW; ++j) {
      const int m = cm0 + row0 + j * 16 + l15;
#pragma unroll
      for (int i = 0; i < FNW; ++i) {
        const int n = cn0 + col0 + i * 16 + 4 * kg;
        if (m < p.M && n < p.N) *(f32x4*)(dst + (size_t)m * p.N + n) = acc[i][j];
      }
    }
  } else {
    // ---- fused epilogue: accumulators -> f16 tile in LDS ([BM][BN] halfs, rows padded by 16 B: the 16 lanes of a
    // ds_write_b64 group fall in 16 different bank pairs) -> coalesced 16-byte-per-lane pass with bias / per-sample row
    // vector / GELU / GEGLU / residual / scale in fp32
    constexpr int RS = 2 * BN + 16;
    // deferred LayerNorm of the A rows (rcdm_gemm_lnx): LayerNorm(x) W^T + b = rstd (x W'^T) - (mean rstd) S + b'.  The identity
    // is applied to the fp32 ACCUMULATORS, before anything is rounded — whatever the epilogue form: the staged halfs are then
    // the normalised projection (+ bias + row vector), what the reference's fp16 Linear rounds, never the raw x W'^T whose
    // magnitude grows with |mean S| (a raw row above 65504 would be inf) — and GEGLU / residual / scale run unchanged on top,
    // with bias and row vector already in.  The tables come from lnx_write_tables (no barrier here: the k-loop's last one
    // published them).
    int epi = p.epi;
    if constexpr (LX) if (ltab != nullptr) {
      const float* lvS = ltab + 2 * BM;
      const float* lvB = lvS + BN;
      const float* lvB2 = lvB + BN;
      const bool lx_rv = (p.epi & RCDM_EPI_ROWVEC) != 0;
      const bool lx_rvp = lx_rv && p.rows_per_sample >= BM;     // both row vectors a tile can meet are in the tables
      const int lx_switch = lx_rvp ? (cm0 / p.rows_per_sample + 1) * p.rows_per_sample : 0x7fffffff;
      epi &= ~RCDM_EPI_BIAS;
      if (lx_rvp) epi &= ~RCDM_EPI_ROWVEC;                      // (else the items add the row vector per row, as without lnx)
      f32x2 rsj[FMW];
      bool secj[FMW];
#pragma unroll
      for (int j = 0; j < FMW; ++j) {
        const int row = row0 + j * 16 + l15;
        rsj[j] = *(const f32x2*)(ltab + 2 * row);
        secj[j] = cm0 + row >= lx_switch;
      }
#pragma unroll
      for (int i = 0; i < FNW; ++i) {
        const int col = col0 + i * 16 + 4 * kg;
        const f32x4 s4 = *(const f32x4*)(lvS + col), b4 = *(const f32x4*)(lvB + col), c4 = *(const f32x4*)(lvB2 + col);
#pragma unroll
        for (int j = 0; j < FMW; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[i][j][e] = __builtin_fmaf(acc[i][j][e], rsj[j].x, __builtin_fmaf(-rsj[j].y, s4[e], secj[j] ? c4[e] : b4[e]));
      }
    }
#pragma unroll
    for (int j = 0; j < FMW; ++j) {
      const int row = row0 + j * 16 + l15;
#pragma unroll
      for (int i = 0; i < FNW; ++i) {
        const int col = col0 + i * 16 + 4 * kg;
        union { f16 h[4]; uint2 u; } pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk.h[e] = (f16)acc[i][j][e];
        *(uint2*)(smem + row * RS + col * 2) = pk.u;
      }
    }
    wait_lgkm0();
    tick_barrier();
    const float sc = p.out_scale;
#ifdef RCDM_PP_EPI_U
    constexpr int U = RCDM_PP_EPI_U;
#else
    constexpr int U = 8;  // staged reads / residual loads in flight per thread (measured: 8 is 1-2 % faster than 4 on the 64x64 convs)
#endif
    if (epi & RCDM_EPI_GEGLU) {
      constexpr int CPR = BN / 16;  // output chunks (8 hidden columns) per row
      constexpr int ITEMS = BM * CPR;
      const int oc0 = geglu_out_col(cn0);
      for (int base = 0; base < ITEMS; base += NT * U) {
        Pack16 hh[U], gg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = min(base + u * NT + t, ITEMS - 1);
          const int row = idx / CPR, c = idx - row * CPR;
          const int hc = (c >> 1) * 4 + (c & 1);
          hh[u].u = *(const uint4*)(smem + row * RS + hc * 16);
          gg[u].u = *(const uint4*)(smem + row * RS + (hc + 2) * 16);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = base + u * NT + t;
          const int row = idx / CPR, c = idx - row * CPR;
          const int hc = (c >> 1) * 4 + (c & 1);
          const int m = cm0 + row, pn = cn0 + hc * 8;
          if (idx < ITEMS && m < p.M && pn < p.N) {
            float bh[8], bg[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bh[e] = bg[e] = 0.f;
            if (epi & RCDM_EPI_BIAS) {
              const f32x4 a0 = *(const f32x4*)(p.bias + pn), a1 = *(const f32x4*)(p.bias + pn + 4);
              const f32x4 b0 = *(const f32x4*)(p.bias + pn + kGegluGroup), b1 = *(const f32x4*)(p.bias + pn + kGegluGroup + 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                bh[e] = a0[e]; bh[4 + e] = a1[e];
                bg[e] = b0[e]; bg[4 + e] = b1[e];
              }
            }
            Pack16 o;
            o.u = geglu8(hh[u].u, gg[u].u, bh, bg, sc);
#if defined(RCDM_I16_ABLATE) && (RCDM_I16_ABLATE & 2)
            if (o.u.x != 0x7e7e7e7eu) continue;
#endif
            epi_store16(p.out + (size_t)m * p.ldc + oc0 + c * 8, o.u);
            if (p.dup) *(uint4*)(p.out + (size_t)m * p.ldc + oc0 + c * 8 + p.dup) = o.u;
          }
        }
      }
    } else {
      constexpr int CPR = BN / 8;
      constexpr int ITEMS = BM * CPR;
      const bool has_res = (epi & RCDM_EPI_RESIDUAL) != 0;
      if constexpr (LN_OK && BN >= 320) {
        if (p.epi & kEpiLN) {
          // ---- rcdm_gemm_ln: the tile spans the whole output row (N <= BN).  Eight lanes share a row (five 16-byte chunks
          // each, c = 8 k + l), NT / 8 rows per pass: out = f16((acc + bias + residual) * scale) is stored, and the LayerNorm
          // of exactly those rounded values — mean, then squared deviations, two DPP group sums — goes to ln_out.  Same
          // arithmetic as layernorm_grp_kernel<8, 5> reading `out` back.
          const int l = t & 7, rsub = t >> 3, nchunks = p.N >> 3;
          const float invN = 1.0f / (float)p.N;
          // the residual tile of ALL passes is requested first: one HBM round trip for the block instead of one per pass
          constexpr int RPP = NT / 8, NPASS = (BM + RPP - 1) / RPP;
          Pack16 rall[NPASS][5];
#pragma unroll
          for (int pass = 0; pass < NPASS; ++pass)
#pragma unroll
            for (int k = 0; k < 5; ++k) {
              const int row = pass * RPP + rsub, c = k * 8 + l;
              rall[pass][k].u = make_uint4(0, 0, 0, 0);
              if (has_res && row < BM && cm0 + row < p.M && c < nchunks)
                rall[pass][k].u = *(const uint4*)(p.res + (size_t)(cm0 + row) * p.ldr + c * 8);
            }
#pragma unroll
          for (int pass = 0; pass < NPASS; ++pass) {
            const int row = pass * RPP + rsub, m = cm0 + row;
            const bool live = row < BM && m < p.M;
            Pack16 hh[5];
            Pack16 (&rr)[5] = rall[pass];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
              const int c = k * 8 + l;
              hh[k].u = make_uint4(0, 0, 0, 0);
              if (live && c < nchunks) hh[k].u = *(const uint4*)(smem + row * RS + c * 16);
            }
            float v[5][8];
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
              const int c = k * 8 + l;
              Pack16 o;
              f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;   // the affine vectors are re-read per pass (L1 hits): held across
              if (p.epi & RCDM_EPI_BIAS) {                 // the passes they cost 120 registers and spilled
                const int cc = min(c, nchunks - 1);
                a0 = *(const f32x4*)(p.bias + cc * 8);
                a1 = *(const f32x4*)(p.bias + cc * 8 + 4);
              }
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                o.e[e] = (f16)(((float)hh[k].e[e] + (e < 4 ? a0[e & 3] : a1[e & 3]) + (float)rr[k].e[e]) * sc);
                v[k][e] = (live && c < nchunks) ? (float)o.e[e] : 0.f;
                sum += v[k][e];
              }
              if (live && c < nchunks) {
                epi_store16(p.out + (size_t)m * p.ldc + c * 8, o.u);
                if (p.dup) epi_store16(p.out + (size_t)m * p.ldc + c * 8 + p.dup, o.u);
              }
            }
            const float mean = group_sum<8>(sum) * invN;
            float sq = 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k)
              if (k * 8 + l < nchunks) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  v[k][e] -= mean;
                  sq += v[k][e] * v[k][e];
                }
              }
            const float rstd = rsqrtf(group_sum<8>(sq) * invN + p.ln_eps);
            if (live) {
              const float* pe_row = p.ln_pe ? p.ln_pe + (size_t)((m / p.ln_rpf) % p.ln_frames) * p.N : nullptr;
#pragma unroll
              for (int k = 0; k < 5; ++k) {
                const int c = k * 8 + l;
                if (c < nchunks) {
                  f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
                  if (pe_row) {
                    p0 = *(const f32x4*)(pe_row + c * 8);
                    p1 = *(const f32x4*)(pe_row + c * 8 + 4);
                  }
                  const f32x4 g0 = *(const f32x4*)(p.ln_g + c * 8), g1 = *(const f32x4*)(p.ln_g + c * 8 + 4);
                  const f32x4 b0 = *(const f32x4*)(p.ln_b + c * 8), b1 = *(const f32x4*)(p.ln_b + c * 8 + 4);
                  Pack16 y;
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    y.e[e] = (f16)(v[k][e] * rstd * g0[e] + (b0[e] + p0[e]));
                    y.e[4 + e] = (f16)(v[k][4 + e] * rstd * g1[e] + (b1[e] + p1[e]));
                  }
                  epi_store16(p.ln_out + (size_t)m * p.ln_ld + c * 8, y.u);
                }
              }
            }
          }
          return;
        }
      }
      if (epi == 0 && sc == 1.0f) {
        // plain projection (fused q/k/v): the staged halfs are the result; no conversion round trip
        for (int base = 0; base < ITEMS; base += NT * U) {
          uint4 hh[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int idx = min(base + u * NT + t, ITEMS - 1);
            const int row = idx / CPR, c8 = idx - row * CPR;
            hh[u] = *(const uint4*)(smem + row * RS + c8 * 16);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int idx = base + u * NT + t;
            const int row = idx / CPR, c8 = idx - row * CPR;
            const int m = cm0 + row, n = cn0 + c8 * 8;
            if (idx < ITEMS && m < p.M && n < p.N) {
#if defined(RCDM_I16_ABLATE) && (RCDM_I16_ABLATE & 2)
              if (hh[u].x != 0x7e7e7e7eu) continue;
#endif
              if constexpr (PH) {
                epi_store16(p.out + (size_t)phase_out_row(p, m) * p.ldc + n, hh[u]);
                continue;
              }
              epi_store16(p.out + (size_t)m * p.ldc + n, hh[u]);
              if (p.dup) *(uint4*)(p.out + (size_t)m * p.ldc + n + p.dup) = hh[u];
            }
          }
        }
        return;
      }
      for (int base = 0; base < ITEMS; base += NT * U) {
        Pack16 hh[U], rr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = min(base + u * NT + t, ITEMS - 1);
          const int row = idx / CPR, c8 = idx - row * CPR;
          const int m = cm0 + row, n = cn0 + c8 * 8;
          hh[u].u = *(const uint4*)(smem + row * RS + c8 * 16);
          rr[u].u = make_uint4(0, 0, 0, 0);
          if (has_res && base + u * NT + t < ITEMS && m < p.M && n < p.N)
            rr[u].u = *(const uint4*)(p.res + (size_t)m * p.ldr + n);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = base + u * NT + t;
          const int row = idx / CPR, c8 = idx - row * CPR;
          const int m = cm0 + row, n = cn0 + c8 * 8;
          if (idx < ITEMS && m < p.M && n < p.N) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)hh[u].e[e];
            if (epi & RCDM_EPI_BIAS) {
              const f32x4 a0 = *(const f32x4*)(p.bias + n), a1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[e] += a0[e];
                v[4 + e] += a1[e];
              }
            }
            if (epi & RCDM_EPI_ROWVEC) {
              const float* rv = p.rowvec + (size_t)(m / p.rows_per_sample) * p.ldt + n;
              const f32x4 a0 = *(const f32x4*)rv, a1 = *(const f32x4*)(rv + 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[e] += a0[e];
                v[4 + e] += a1[e];
              }
            }
            if (epi & RCDM_EPI_GELU) gelu8(v);
            Pack16 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.e[e] = (f16)((v[e] + (float)rr[u].e[e]) * sc);
#if defined(RCDM_I16_ABLATE) && (RCDM_I16_ABLATE & 2)
            if (o.u.x != 0x7e7e7e7eu) continue;
#endif
            if constexpr (PH) {
              epi_store16(p.out + (size_t)phase_out_row(p, m) * p.ldc + n, o.u);
              continue;
            }
            epi_store16(p.out + (size_t)m * p.ldc + n, o.u);
            if (p.dup) *(uint4*)(p.out + (size_t)m * p.ldc + n + p.dup) = o.u;
          }
        }
      }
    }
  }
}
