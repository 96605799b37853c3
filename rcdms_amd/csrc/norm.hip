// norm.hip — GroupNorm(+SiLU) and LayerNorm(+positional encoding) over channels-last f16 rows,
// fp32 statistics, wavefront reductions, deterministic (fixed-order partials, no float atomics).
//
// Replaces (reference): torch.nn.GroupNorm on the 5-D tensor, i.e. statistics ACROSS the f frames,
// src/models/resnet.py:185-186,196,202 and unet.py:455-456; the per-frame form
// attention.py:328 / motion_module.py:162; nn.LayerNorm attention.py:482,502,514,
// motion_module.py:236,243; PositionalEncoding.forward motion_module.py:265-267.
#include <stdlib.h>
#include "common.h"
#include "gn_plan.h"

namespace {

int g_gn_fold_mode = -1;

// grid (splits, samples); block CH*RPB threads; thread (rl, ch) owns 16-B chunk ch of rows rl+k*RPB.
__global__ void gn_stats_kernel(const GnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* part = (float*)smem;  // [threads][16]
  const int t = threadIdx.x;
  const int ch = t % p.CH, rl = t / p.CH;
  const int s = blockIdx.y, sp = blockIdx.x;
  const int r_begin = sp * p.rows_per_split;
  const int r_end = min(p.P, r_begin + p.rows_per_split);
  float sum[8], sq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sum[e] = sq[e] = 0.f;
  const f16* base = p.x + (size_t)s * p.P * p.ldx + ch * 8;
  // GN_U independent 16-byte loads in flight per thread (the split is sized so that this is normally the block's ONE
  // memory round trip: a launch this short is a chain of latencies, not a stream); rows past r_end add nothing
  for (int r = r_begin + rl; r < r_end; r += GN_U * p.RPB) {
    Pack16 v[GN_U];
#pragma unroll
    for (int u = 0; u < GN_U; ++u) {
      const int ru = r + u * p.RPB;
      v[u].u = make_uint4(0, 0, 0, 0);
      if (ru < r_end) v[u].u = *(const uint4*)(base + (size_t)ru * p.ldx);
    }
#pragma unroll
    for (int u = 0; u < GN_U; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = (float)v[u].e[e];
        sum[e] += f;
        sq[e] += f * f;
      }
  }
  gn_block_partials(p, part, t, sum, sq, s, sp, r_end - r_begin);
}

// one wave per (sample, group): lanes take splits lane, lane+64, ... in order, then a fixed xor-butterfly
// of Chan's pairwise combination — deterministic (the tree shape never changes), no atomics.
__device__ __forceinline__ void chan_combine(float& n, float& mean, float& m2, float nb, float mb, float qb) {
  const float nt = n + nb;
  if (nt > 0.f) {
    const float delta = mb - mean;
    const float f = nb / nt;
    mean += delta * f;
    m2 += qb + delta * delta * n * f;
    n = nt;
  }
}

// (mean, rstd) of one (sample, group) from its `splits` partials, by ONE full wave: lanes take the partials lane, lane + 64,
// ... in order, then a fixed xor-butterfly of Chan's combination.  The arithmetic gn_finalize_kernel has had since round 1
// (so a fold of this into a consumer is bit-identical to the separate launch); result valid in every lane.
__device__ __forceinline__ void gn_finalize_group(const float* base, int splits, int lane, float eps, float& mean_out, float& rstd_out) {
  float n = 0.f, mean = 0.f, m2 = 0.f;
  // the (sample, group)'s partials are contiguous: all of a lane's (<= 8) are requested before the first combine, one
  // round trip instead of splits / 64 dependent ones
  for (int sp0 = 0; sp0 < splits; sp0 += 512) {
    float pn[8], pm[8], pq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int sp = sp0 + lane + 64 * i;
      pn[i] = pm[i] = pq[i] = 0.f;
      if (sp < splits) {
        pn[i] = base[sp * 3];
        pm[i] = base[sp * 3 + 1];
        pq[i] = base[sp * 3 + 2];
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) chan_combine(n, mean, m2, pn[i], pm[i], pq[i]);
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float nb = __shfl_xor(n, off, 64), mb = __shfl_xor(mean, off, 64), qb = __shfl_xor(m2, off, 64);
    // combine in a lane-order-independent way: the lower lane of each pair is always the "left" operand
    float ln = n, lm = mean, lq = m2, rn = nb, rm = mb, rq = qb;
    if (lane & off) { ln = nb; lm = mb; lq = qb; rn = n; rm = mean; rq = m2; }
    chan_combine(ln, lm, lq, rn, rm, rq);
    n = ln; mean = lm; m2 = lq;
  }
  const float var = n > 0.f ? m2 / n : 0.f;  // biased, as torch.nn.GroupNorm
  mean_out = mean;
  rstd_out = rsqrtf(var + eps);
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const GnArgs p) {
  const int lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (idx >= p.samples * p.G) return;
  float mean, rstd;
  gn_finalize_group(p.partial + (size_t)idx * p.splits * 3, p.splits, lane, p.eps, mean, rstd);
  if (lane == 0) {
    p.stat[idx * 2] = mean;
    p.stat[idx * 2 + 1] = rstd;
  }
}

// grid (blocks_per_sample, samples); same thread->chunk mapping as the stats kernel.
// FOLD: no gn_finalize launch in front — every block finalises its sample's G groups itself (blockDim is a multiple of 64
// then: full waves for the butterfly; threads >= CH * RPB only take part in that).  Used where a sample has few partials
// and many samples share the chip (the per-frame norms of the transformers / motion modules: 10 samples x <= 128 splits),
// so that a block re-reads <= 50 KB of partials from L2; the cross-frame ResNet norms (2 samples x ~400 splits, ~850 apply
// blocks) keep the separate launch.
template <bool FOLD>
__global__ void gn_apply_kernel(const GnArgs p) {
  __shared__ float gstat[64 * 2];
  const int t = threadIdx.x;
  const int ch = t % p.CH, rl = t / p.CH;
  const int s = blockIdx.y;
  if constexpr (FOLD) {
    // a wave finalises its groups g = wave, wave + nw, ... EIGHT at a time: all their partials (<= 128 splits: two per lane
    // and group) are requested first, and the eight butterflies run interleaved — one memory round trip and one shuffle
    // chain per batch instead of one per group.  Per group the same operations in the same order as gn_finalize_group
    // (a zero partial is an exact no-op of chan_combine).
    const int lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6;
    for (int g0 = wave; g0 < p.G; g0 += 8 * nw) {
      float pn[8][2], pm[8][2], pq[8][2];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int g = g0 + k * nw;
        const float* base = p.partial + ((size_t)s * p.G + min(g, p.G - 1)) * p.splits * 3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int sp = lane + 64 * i;
          const bool ok = g < p.G && sp < p.splits;
          const float* q = base + (ok ? sp : 0) * 3;
          const float a = q[0], b = q[1], c = q[2];
          pn[k][i] = ok ? a : 0.f;
          pm[k][i] = ok ? b : 0.f;
          pq[k][i] = ok ? c : 0.f;
        }
      }
      float n[8], mean[8], m2[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        n[k] = mean[k] = m2[k] = 0.f;
        chan_combine(n[k], mean[k], m2[k], pn[k][0], pm[k][0], pq[k][0]);
        chan_combine(n[k], mean[k], m2[k], pn[k][1], pm[k][1], pq[k][1]);
      }
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float nb = __shfl_xor(n[k], off, 64), mb = __shfl_xor(mean[k], off, 64), qb = __shfl_xor(m2[k], off, 64);
          float ln = n[k], lm = mean[k], lq = m2[k], rn = nb, rm = mb, rq = qb;
          if (lane & off) { ln = nb; lm = mb; lq = qb; rn = n[k]; rm = mean[k]; rq = m2[k]; }
          chan_combine(ln, lm, lq, rn, rm, rq);
          n[k] = ln; mean[k] = lm; m2[k] = lq;
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int g = g0 + k * nw;
          if (g < p.G) {
            const float var = n[k] > 0.f ? m2[k] / n[k] : 0.f;
            gstat[g * 2] = mean[k];
            gstat[g * 2 + 1] = rsqrtf(var + p.eps);
          }
        }
      }
    }
    __syncthreads();
    if (rl >= p.RPB) return;
  }
  float sc[8], sh[8];
  {
    const f32x4 g0 = *(const f32x4*)(p.gamma + ch * 8), g1 = *(const f32x4*)(p.gamma + ch * 8 + 4);
    const f32x4 b0 = *(const f32x4*)(p.beta + ch * 8), b1 = *(const f32x4*)(p.beta + ch * 8 + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (ch * 8 + e) / p.cg;
      const float mean = FOLD ? gstat[g * 2] : p.stat[(s * p.G + g) * 2];
      const float rstd = FOLD ? gstat[g * 2 + 1] : p.stat[(s * p.G + g) * 2 + 1];
      const float ga = e < 4 ? g0[e & 3] : g1[e & 3], be = e < 4 ? b0[e & 3] : b1[e & 3];
      sc[e] = rstd * ga;
      sh[e] = __builtin_fmaf(-(mean * rstd), ga, be);   // explicit: rowff.hip's prologue form must round the same way
    }
  }
  const f16* xb = p.x + (size_t)s * p.P * p.ldx + ch * 8;
  f16* yb = p.y + (size_t)s * p.P * p.ldy + ch * 8;
  const int rstep = gridDim.x * p.RPB;
  for (int r = blockIdx.x * p.RPB + rl; r < p.P; r += 4 * rstep) {
    Pack16 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)  // four loads in flight (eight measured 5-10 % slower here)
      if (r + u * rstep < p.P) v[u].u = *(const uint4*)(xb + (size_t)(r + u * rstep) * p.ldx);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r + u * rstep >= p.P) break;
      Pack16 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = __builtin_fmaf((float)v[u].e[e], sc[e], sh[e]);
        if (p.silu) f = silu_f(f);
        o.e[e] = (f16)f;
      }
      *(uint4*)(yb + (size_t)(r + u * rstep) * p.ldy) = o.u;
    }
  }
}

// Single-launch GroupNorm for the smallest levels (16x16 per frame, 8x8: 30 of the 81 GroupNorms of a UNet call).  There the
// tensor is a few MB, the three-launch form is mostly launch latency (a dependent graph node costs >= 1.6 us even when
// empty), and one block can own a whole (sample, group-bundle) slab: block (bundle, sample) covers GB consecutive groups
// = NCHK whole 16-byte chunks of every row (GB chosen so that GB * cg is a multiple of 8), sums x and x^2 over the slab
// (thread-sequential, then a fixed-order LDS reduction: deterministic), and applies (+SiLU) reading the slab again
// from L2.  Same arithmetic as one split of gn_stats + gn_finalize + gn_apply.
struct GnFusedArgs {
  const f16* x;
  f16* y;
  const float* gamma;
  const float* beta;
  int P, cg, GB, NCHK, RPB, ldx, ldy, silu;
  float eps;
  float* stat_out;   // non-null: statistics only — (mean, rstd) of the block's groups to stat_out[sample][group][2], no apply pass
  int G;
};

__global__ __launch_bounds__(256) void gn_fused_kernel(const GnFusedArgs p) {
  __shared__ float part[256 * 16];
  __shared__ float gstat[16 * 2];
  const int t = threadIdx.x;
  const int ch = t % p.NCHK, rl = t / p.NCHK;
  const bool live = rl < p.RPB;
  const int s = blockIdx.y, c0 = blockIdx.x * p.GB * p.cg;  // first channel of this bundle
  const f16* xb = p.x + (size_t)s * p.P * p.ldx + c0 + ch * 8;
  f16* yb = p.y + (size_t)s * p.P * p.ldy + c0 + ch * 8;
  float sum[8], sq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sum[e] = sq[e] = 0.f;
  if (live) {
    for (int r = rl; r < p.P; r += 4 * p.RPB) {
      Pack16 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ru = r + u * p.RPB;
        v[u].u = make_uint4(0, 0, 0, 0);
        if (ru < p.P) v[u].u = *(const uint4*)(xb + (size_t)ru * p.ldx);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = (float)v[u].e[e];
          sum[e] += f;
          sq[e] += f * f;
        }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    part[t * 16 + e] = sum[e];
    part[t * 16 + 8 + e] = sq[e];
  }
  __syncthreads();
  // column sums first (NCHK * 16 values, each over the RPB row-threads, spread over the block), then the few groups
  __shared__ float colsum[32 * 16];
  for (int o = t; o < p.NCHK * 16; o += 256) {
    const int c = o >> 4, k = o & 15;
    float a = 0.f;
    for (int r = 0; r < p.RPB; ++r) a += part[(r * p.NCHK + c) * 16 + k];
    colsum[o] = a;
  }
  __syncthreads();
  if (t < p.GB) {
    float gs = 0.f, gq = 0.f;
    for (int c = t * p.cg; c < (t + 1) * p.cg; ++c) {
      gs += colsum[(c >> 3) * 16 + (c & 7)];
      gq += colsum[(c >> 3) * 16 + 8 + (c & 7)];
    }
    const float n = (float)p.P * (float)p.cg;
    const float mean = gs / n;
    float m2 = gq - gs * mean;
    if (m2 < 0.f) m2 = 0.f;
    gstat[t * 2] = mean;
    gstat[t * 2 + 1] = rsqrtf(m2 / n + p.eps);  // biased variance, as torch.nn.GroupNorm
    if (p.stat_out) {
      float* o = p.stat_out + ((size_t)s * p.G + blockIdx.x * p.GB + t) * 2;
      o[0] = gstat[t * 2];
      o[1] = gstat[t * 2 + 1];
    }
  }
  if (p.stat_out) return;   // (rcdm_groupnorm_stats on a small tensor: one launch instead of statistics + finalize)
  __syncthreads();
  if (!live) return;
  float sc[8], sh[8];
  {
    const f32x4 g0 = *(const f32x4*)(p.gamma + c0 + ch * 8), g1 = *(const f32x4*)(p.gamma + c0 + ch * 8 + 4);
    const f32x4 b0 = *(const f32x4*)(p.beta + c0 + ch * 8), b1 = *(const f32x4*)(p.beta + c0 + ch * 8 + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (ch * 8 + e) / p.cg;
      const float mean = gstat[g * 2], rstd = gstat[g * 2 + 1];
      const float ga = e < 4 ? g0[e & 3] : g1[e & 3], be = e < 4 ? b0[e & 3] : b1[e & 3];
      sc[e] = rstd * ga;
      sh[e] = __builtin_fmaf(-(mean * rstd), ga, be);   // explicit: rowff.hip's prologue form must round the same way
    }
  }
  for (int r = rl; r < p.P; r += 4 * p.RPB) {
    Pack16 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (r + u * p.RPB < p.P) v[u].u = *(const uint4*)(xb + (size_t)(r + u * p.RPB) * p.ldx);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r + u * p.RPB >= p.P) break;
      Pack16 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = __builtin_fmaf((float)v[u].e[e], sc[e], sh[e]);
        if (p.silu) f = silu_f(f);
        o.e[e] = (f16)f;
      }
      *(uint4*)(yb + (size_t)(r + u * p.RPB) * p.ldy) = o.u;
    }
  }
}

// bundle size for the single-launch form: the smallest GB with GB * cg % 8 == 0 that divides the group count; 0 = none
int gn_fused_bundle(int G, int cg) {
  for (int gb = 1; gb <= 8; gb <<= 1)
    if ((gb * cg) % 8 == 0 && G % gb == 0 && gb * cg / 8 <= 32) return gb;
  return 0;
}

int gn_plan(const rcdm_groupnorm_desc* d, GnArgs& a) {   // (rcdm_gn_plan for igemm.hip: below)
  if (d->samples <= 0 || d->rows_per_sample <= 0 || d->C <= 0 || d->groups <= 0) return RCDM_EINVAL;
  if ((d->C & 7) || (d->C % d->groups) || (d->ldx & 7) || (d->ldy & 7)) return RCDM_ESHAPE;
  if (d->groups > 64 || d->C > 8192) return RCDM_ESHAPE;
  a.samples = d->samples; a.P = d->rows_per_sample; a.C = d->C; a.G = d->groups;
  a.cg = d->C / d->groups; a.CH = d->C / 8;
  a.RPB = 256 / a.CH;
  if (a.RPB < 1) a.RPB = 1;
  if (a.CH * a.RPB < a.G) a.RPB = (a.G + a.CH - 1) / a.CH;  // need >= G threads for the group pass
  a.ldx = d->ldx; a.ldy = d->ldy; a.eps = d->eps; a.silu = d->silu;
  // one pass of GN_U rows per thread where that gives a chip-filling grid (<= 1024 blocks per launch), else several
  int splits = (a.P + GN_U * a.RPB - 1) / (GN_U * a.RPB);
  const int fill = (768 + a.samples - 1) / a.samples;              // small tensors: still >= 768 blocks per launch ...
  const int max_splits = (a.P + 2 * a.RPB - 1) / (2 * a.RPB);      // ... while every thread keeps >= 2 rows
  if (splits < fill) splits = fill < max_splits ? fill : max_splits;
  const int cap = (1024 + a.samples - 1) / a.samples;
  if (splits > cap) splits = cap;
  if (splits > 512) splits = 512;
  if (splits < 1) splits = 1;
  a.rows_per_split = (a.P + splits - 1) / splits;
  a.splits = (a.P + a.rows_per_split - 1) / a.rows_per_split;
  return RCDM_OK;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: a wave normalises R rows at a time (all R*NCH 16-byte loads issued before the first
// reduction, so one wave keeps several KB in flight); each row is held in registers, exact two-pass.
template <int NCH, int R>
__global__ __launch_bounds__(256) void layernorm_kernel(const f16* __restrict__ x, f16* __restrict__ y,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const float* __restrict__ pe, int M, int C, int ldx,
                                                        int ldy, float eps, int rows_per_frame, int frames) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= M) return;
  const int nchunks = C >> 3;
  Pack16 raw[R][NCH];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      raw[r][i].u = make_uint4(0, 0, 0, 0);
      if (c < nchunks && row0 + r < M) raw[r][i].u = *(const uint4*)(x + (size_t)(row0 + r) * ldx + c * 8);
    }
  float gm[NCH][8], bt[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    // two aligned float4 per 8-channel chunk (the scalar form was 16 dword loads per chunk per lane)
    f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, b0 = g0, b1 = g0;
    if (c < nchunks) {
      g0 = *(const f32x4*)(gamma + c * 8);
      g1 = *(const f32x4*)(gamma + c * 8 + 4);
      b0 = *(const f32x4*)(beta + c * 8);
      b1 = *(const f32x4*)(beta + c * 8 + 4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      gm[i][e] = g0[e];
      gm[i][4 + e] = g1[e];
      bt[i][e] = b0[e];
      bt[i][4 + e] = b1[e];
    }
  }
  const float invC = 1.0f / (float)C;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = row0 + r;
    float v[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = (float)raw[r][i].e[e];
        sum += v[i][e];
      }
    const float mean = wave_sum(sum) * invC;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dlt = v[i][e] - mean;
          sq += dlt * dlt;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(sq) * invC + eps);
    if (row < M) {
      const float* pe_row = pe ? pe + (size_t)((row / rows_per_frame) % frames) * C : nullptr;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunks) {
          Pack16 o;
          f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
          if (pe_row) {
            p0 = *(const f32x4*)(pe_row + c * 8);
            p1 = *(const f32x4*)(pe_row + c * 8 + 4);
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = (v[i][e] - mean) * rstd * gm[i][e] + bt[i][e] + (e < 4 ? p0[e & 3] : p1[e & 3]);
            o.e[e] = (f16)f;
          }
          *(uint4*)(y + (size_t)row * ldy + c * 8) = o.u;
        }
      }
    }
  }
}

// LayerNorm, row-group form (the one rcdm_layernorm launches): LPR lanes share a row (a power of two, 8..64), each lane
// owns CPL 16-byte chunks c = k * LPR + l, so a wave normalises 64 / LPR rows at once with EVERY lane loading (the
// wave-per-row form above has 40 of 64 lanes busy at C = 320) and each load instruction covering whole 128-byte lines.
// The two row reductions (mean, then sum of squared deviations: exact two-pass, in registers) are log2(LPR) DPP steps
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: no LDS crossbar) instead of six ds_bpermute round trips: the
// wave-per-row kernel spent its life in those dependent shuffles (3.65 TB/s where a plain copy of the same bytes runs at
// 6.2 TB/s on this part).
template <int LPR, int CPL, bool EARLY>
__global__ __launch_bounds__(256) void layernorm_grp_kernel(const f16* __restrict__ x, f16* __restrict__ y,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ pe, int M, int C, int ldx,
                                                            int ldy, float eps, int rows_per_frame, int frames) {
  constexpr int RW = 64 / LPR;  // rows per wave
  const int lane = threadIdx.x & 63, l = lane & (LPR - 1), rw = lane / LPR;
  const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW + rw;
  const bool live = row < M;
  const int nchunks = C >> 3;
  Pack16 raw[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int c = k * LPR + l;
    raw[k].u = make_uint4(0, 0, 0, 0);
    if (live && c < nchunks) raw[k].u = *(const uint4*)(x + (size_t)row * ldx + c * 8);
  }
  // EARLY (launches of <= 3 waves per SIMD, where registers are free): the affine parameters and the positional-encoding
  // row are requested right behind the data, one memory round trip instead of two on the critical path of a launch that
  // is a single shot of latency anyway
  f32x4 eg0[EARLY ? CPL : 1], eg1[EARLY ? CPL : 1], eb0[EARLY ? CPL : 1], eb1[EARLY ? CPL : 1];
  if constexpr (EARLY) {
    const float* pe_row = pe ? pe + (size_t)(((live ? row : 0) / rows_per_frame) % frames) * C : nullptr;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = min(k * LPR + l, nchunks - 1);
      eg0[k] = *(const f32x4*)(gamma + c * 8);
      eg1[k] = *(const f32x4*)(gamma + c * 8 + 4);
      eb0[k] = *(const f32x4*)(beta + c * 8);
      eb1[k] = *(const f32x4*)(beta + c * 8 + 4);
      if (pe_row) {
        eb0[k] += *(const f32x4*)(pe_row + c * 8);
        eb1[k] += *(const f32x4*)(pe_row + c * 8 + 4);
      }
    }
  }
  // fp32 pairs: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 do two lanes' worth per instruction (the kernel's ~400 scalar
  // VALU ops per wave were 3 us of the whole chip's issue capacity per 26-MB tensor)
  f32x2 v[CPL][4];
  f32x2 s2 = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < CPL; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[k][e] = f32x2{(float)raw[k].e[2 * e], (float)raw[k].e[2 * e + 1]};
      s2 += v[k][e];
    }
  const float invC = 1.0f / (float)C;
  const float mean = group_sum<LPR>(s2.x + s2.y) * invC;
  f32x2 q2 = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < CPL; ++k)
    if (k * LPR + l < nchunks) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[k][e] -= splat2(mean);
        q2 = __builtin_elementwise_fma(v[k][e], v[k][e], q2);
      }
    }
  const float rstd = rsqrtf(group_sum<LPR>(q2.x + q2.y) * invC + eps);
  if (!live) return;
  // affine parameters are fetched here, late, on purpose: held across the reductions they cost 56 VGPRs, i.e. three waves
  // per SIMD instead of six, and the 5120 waves of a 64x64-level tensor then need two rounds instead of one (measured:
  // 12.3 vs 10.9 us); they are L1/L2 hits
  const float* pe_row = pe ? pe + (size_t)((row / rows_per_frame) % frames) * C : nullptr;
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int c = k * LPR + l;
    if (c < nchunks) {
      f32x4 g0, g1, b0, b1;
      if constexpr (EARLY) {
        g0 = eg0[k]; g1 = eg1[k]; b0 = eb0[k]; b1 = eb1[k];
      } else {
        g0 = *(const f32x4*)(gamma + c * 8);
        g1 = *(const f32x4*)(gamma + c * 8 + 4);
        b0 = *(const f32x4*)(beta + c * 8);
        b1 = *(const f32x4*)(beta + c * 8 + 4);
        if (pe_row) {
          b0 += *(const f32x4*)(pe_row + c * 8);
          b1 += *(const f32x4*)(pe_row + c * 8 + 4);
        }
      }
      const f32x2 r2 = splat2(rstd);
      const f32x2 o0 = __builtin_elementwise_fma(v[k][0] * r2, f32x2{g0[0], g0[1]}, f32x2{b0[0], b0[1]});
      const f32x2 o1 = __builtin_elementwise_fma(v[k][1] * r2, f32x2{g0[2], g0[3]}, f32x2{b0[2], b0[3]});
      const f32x2 o2 = __builtin_elementwise_fma(v[k][2] * r2, f32x2{g1[0], g1[1]}, f32x2{b1[0], b1[1]});
      const f32x2 o3 = __builtin_elementwise_fma(v[k][3] * r2, f32x2{g1[2], g1[3]}, f32x2{b1[2], b1[3]});
      Pack16 o;
      o.e[0] = (f16)o0.x; o.e[1] = (f16)o0.y; o.e[2] = (f16)o1.x; o.e[3] = (f16)o1.y;
      o.e[4] = (f16)o2.x; o.e[5] = (f16)o2.y; o.e[6] = (f16)o3.x; o.e[7] = (f16)o3.y;
      *(uint4*)(y + (size_t)row * ldy + c * 8) = o.u;
    }
  }
}

// Row softmax over f16 rows (fp32 math): y[m][n] = softmax_n(scale * x[m][n]).  One wave per row, the row held in
// registers across the max / sum passes.  Used where a head dim is too wide for the flash kernel (the VAE mid-block
// attention: one head of 512 channels over 64 x 64 tokens), where scores are materialised by two GEMMs instead.
template <int NCH>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const f16* __restrict__ x, f16* __restrict__ y, int M, int N,
                                                           int ldx, int ldy, float scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nchunks = N >> 3;
  const float c = scale * 1.4426950408889634f;
  Pack16 raw[NCH];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nchunks) {
      raw[i].u = *(const uint4*)(x + (size_t)row * ldx + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)raw[i].e[e]);
    }
  }
  mx = wave_max(mx) * c;
  float v[NCH][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (lane + 64 * i < nchunks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = __builtin_amdgcn_exp2f(fmaf((float)raw[i].e[e], c, -mx));
        sum += v[i][e];
      }
    }
  }
  const float inv = __builtin_amdgcn_rcpf(wave_sum(sum));
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nchunks) {
      Pack16 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.e[e] = (f16)(v[i][e] * inv);
      *(uint4*)(y + (size_t)row * ldy + ch * 8) = o.u;
    }
  }
}

bool gn_single_launch(const GnArgs& a) {
  static int fused_mode = -1;  // RCDM_GN_FUSED=0: three-launch form everywhere (A/B switch)
  if (fused_mode < 0) {
    const char* e = getenv("RCDM_GN_FUSED");
    fused_mode = e ? atoi(e) : 1;
  }
  static int fused_rows = -1;  // RCDM_GN_FUSED_ROWS: rows per sample up to which the single-launch form is used
  if (fused_rows < 0) {
    const char* e = getenv("RCDM_GN_FUSED_ROWS");
    fused_rows = e ? atoi(e) : 512;
  }
  const int gb = gn_fused_bundle(a.G, a.cg);
  // one block per (sample, bundle) pays up to a few hundred rows per sample (measured: 15 -> 9 us at 320 rows, but
  // 17 -> 22 us at 1280 and 25 -> 64 us at 4096: a single block streams its slab too slowly)
  return fused_mode && gb && a.samples * (a.G / gb) >= 48 && a.P <= fused_rows;
}

// finalize + apply of the three-launch form (partials already in a.partial)
int gn_finalize_apply(GnArgs& a, hipStream_t stream) {
  const int threads = a.CH * a.RPB;
  // OFF by default: measured +0.05 ms per step (18.08 -> 18.13 ms, three interleaved pairs on one box) — the ~430 apply blocks
  // of a 32x32-level per-frame norm each pay a partials round trip + eight butterflies, more than the 4.9 us launch they replace.
  // RCDM_GN_FOLD=1 / rcdm_set_groupnorm_fold(1) turns it on (bit-identical results).
  int& fold_mode = g_gn_fold_mode;
  if (fold_mode < 0) {
    const char* e = getenv("RCDM_GN_FOLD");
    fold_mode = e ? atoi(e) : 0;
  }
  const int threads64 = (threads + 63) / 64 * 64;
  const bool fold = fold_mode && a.samples >= 4 && a.splits <= 128 && a.G <= 64 && threads64 <= 1024;
  if (!fold) {
    const int nsg = a.samples * a.G;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((nsg + 3) / 4), dim3(256), 0, stream, a);
    const int rc = rcdm_check_launch();
    if (rc) return rc;
  }
  int bps = (a.P + a.RPB * 8 - 1) / (a.RPB * 8);  // ~8 rows per thread
  const int cap = (2048 + a.samples - 1) / a.samples;
  if (bps > cap) bps = cap;
  if (bps < 1) bps = 1;
  if (fold) hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(bps, a.samples), dim3(threads64), 0, stream, a);
  else hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(bps, a.samples), dim3(threads), 0, stream, a);
  return rcdm_check_launch();
}

}  // namespace

int rcdm_gn_plan(const rcdm_groupnorm_desc* d, GnArgs& a) { return gn_plan(d, a); }
bool rcdm_gn_three_launch(const GnArgs& a) { return !gn_single_launch(a) && a.CH * a.RPB <= 1024; }

extern "C" {

int rcdm_set_groupnorm_fold(int32_t on) {
  g_gn_fold_mode = on < 0 ? -1 : (on ? 1 : 0);
  return RCDM_OK;
}

size_t rcdm_groupnorm_workspace_bytes(const rcdm_groupnorm_desc* d) {
  GnArgs a{};
  if (!d || gn_plan(d, a)) return 0;
  return ((size_t)a.samples * a.splits * a.G * 3 + (size_t)a.samples * a.G * 2) * sizeof(float);
}

int rcdm_groupnorm_silu(const rcdm_groupnorm_desc* d, const void* x, const float* gamma, const float* beta, void* y,
                        void* workspace, size_t workspace_bytes, void* stream_) {
  if (!d || !x || !gamma || !beta || !y) return RCDM_EINVAL;
  GnArgs a{};
  int rc = gn_plan(d, a);
  if (rc) return rc;
  const size_t need = ((size_t)a.samples * a.splits * a.G * 3 + (size_t)a.samples * a.G * 2) * sizeof(float);
  if (!workspace || workspace_bytes < need) return RCDM_EWORKSPACE;
  hipStream_t stream0 = (hipStream_t)stream_;
  {
    if (gn_single_launch(a)) {
      const int gb = gn_fused_bundle(a.G, a.cg);
      GnFusedArgs f;
      f.x = (const f16*)x; f.y = (f16*)y; f.gamma = gamma; f.beta = beta;
      f.P = a.P; f.cg = a.cg; f.GB = gb; f.NCHK = gb * a.cg / 8; f.RPB = 256 / f.NCHK;
      f.ldx = a.ldx; f.ldy = a.ldy; f.silu = a.silu; f.eps = a.eps;
      f.stat_out = nullptr; f.G = a.G;
      hipLaunchKernelGGL(gn_fused_kernel, dim3(a.G / gb, a.samples), dim3(256), 0, stream0, f);
      return rcdm_check_launch();
    }
  }
  a.x = (const f16*)x; a.y = (f16*)y; a.gamma = gamma; a.beta = beta;
  a.partial = (float*)workspace;
  a.stat = a.partial + (size_t)a.samples * a.splits * a.G * 3;
  hipStream_t stream = (hipStream_t)stream_;
  const int threads = a.CH * a.RPB;
  if (threads > 1024) return RCDM_ESHAPE;
  const size_t stats_lds = (size_t)(threads + a.CH) * 16 * sizeof(float);
  if (stats_lds > 64 * 1024) {  // C > 4096: beyond the default dynamic-LDS limit
    static bool attr_set[64] = {};
    if (rcdm_first_on_device(attr_set)) {
      (void)hipFuncSetAttribute((const void*)gn_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      }
  }
  hipLaunchKernelGGL(gn_stats_kernel, dim3(a.splits, a.samples), dim3(threads), stats_lds, stream, a);
  rc = rcdm_check_launch();
  if (rc) return rc;
  return gn_finalize_apply(a, stream);
}

int rcdm_groupnorm_prestat_ok(const rcdm_groupnorm_desc* d) {
  GnArgs a{};
  if (!d || gn_plan(d, a)) return 0;
  return rcdm_gn_three_launch(a) ? 1 : 0;
}

int rcdm_groupnorm_silu_prestat(const rcdm_groupnorm_desc* d, const void* x, const float* gamma, const float* beta, void* y,
                                void* workspace, size_t workspace_bytes, void* stream_) {
  if (!d || !x || !gamma || !beta || !y) return RCDM_EINVAL;
  GnArgs a{};
  int rc = gn_plan(d, a);
  if (rc) return rc;
  if (!rcdm_gn_three_launch(a)) return RCDM_ESHAPE;   // this descriptor's norm is a single launch: nobody can have left partials
  const size_t need = ((size_t)a.samples * a.splits * a.G * 3 + (size_t)a.samples * a.G * 2) * sizeof(float);
  if (!workspace || workspace_bytes < need) return RCDM_EWORKSPACE;
  a.x = (const f16*)x; a.y = (f16*)y; a.gamma = gamma; a.beta = beta;
  a.partial = (float*)workspace;
  a.stat = a.partial + (size_t)a.samples * a.splits * a.G * 3;
  return gn_finalize_apply(a, (hipStream_t)stream_);
}

int rcdm_groupnorm_stats(const rcdm_groupnorm_desc* d, const void* x, float* stat, void* workspace, size_t workspace_bytes,
                         void* stream_) {
  if (!d || !x || !stat) return RCDM_EINVAL;
  GnArgs a{};
  int rc = gn_plan(d, a);
  if (rc) return rc;
  const size_t need = (size_t)a.samples * a.splits * a.G * 3 * sizeof(float);
  if (!workspace || workspace_bytes < need) return RCDM_EWORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  if (gn_single_launch(a)) {   // small tensor (<= 512 rows per sample): one block per (sample, group bundle) finalises by itself
    const int gb = gn_fused_bundle(a.G, a.cg);
    GnFusedArgs f;
    f.x = (const f16*)x; f.y = nullptr; f.gamma = nullptr; f.beta = nullptr;
    f.P = a.P; f.cg = a.cg; f.GB = gb; f.NCHK = gb * a.cg / 8; f.RPB = 256 / f.NCHK;
    f.ldx = a.ldx; f.ldy = a.ldy; f.silu = 0; f.eps = a.eps;
    f.stat_out = stat; f.G = a.G;
    hipLaunchKernelGGL(gn_fused_kernel, dim3(a.G / gb, a.samples), dim3(256), 0, stream, f);
    return rcdm_check_launch();
  }
  a.x = (const f16*)x;
  a.partial = (float*)workspace;
  a.stat = stat;
  const int threads = a.CH * a.RPB;
  if (threads > 1024) return RCDM_ESHAPE;
  const size_t stats_lds = (size_t)(threads + a.CH) * 16 * sizeof(float);
  if (stats_lds > 64 * 1024) return RCDM_ESHAPE;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(a.splits, a.samples), dim3(threads), stats_lds, stream, a);
  rc = rcdm_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((a.samples * a.G + 3) / 4), dim3(256), 0, stream, a);
  return rcdm_check_launch();
}

int rcdm_groupnorm_stats_prestat(const rcdm_groupnorm_desc* d, float* stat, void* workspace, size_t workspace_bytes, void* stream_) {
  if (!d || !stat) return RCDM_EINVAL;
  GnArgs a{};
  int rc = gn_plan(d, a);
  if (rc) return rc;
  if (!rcdm_gn_three_launch(a)) return RCDM_ESHAPE;   // nobody can have left partials for a single-launch norm
  const size_t need = (size_t)a.samples * a.splits * a.G * 3 * sizeof(float);
  if (!workspace || workspace_bytes < need) return RCDM_EWORKSPACE;
  a.partial = (float*)workspace;
  a.stat = stat;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((a.samples * a.G + 3) / 4), dim3(256), 0, (hipStream_t)stream_, a);
  return rcdm_check_launch();
}

int rcdm_groupnorm_apply(const rcdm_groupnorm_desc* d, const void* x, const float* stat, const float* gamma, const float* beta,
                         void* y, void* stream_) {
  if (!d || !x || !stat || !gamma || !beta || !y) return RCDM_EINVAL;
  GnArgs a{};
  int rc = gn_plan(d, a);
  if (rc) return rc;
  const int threads = a.CH * a.RPB;
  if (threads > 1024) return RCDM_ESHAPE;
  a.x = (const f16*)x; a.y = (f16*)y; a.gamma = gamma; a.beta = beta;
  a.stat = const_cast<float*>(stat);
  int bps = (a.P + a.RPB * 8 - 1) / (a.RPB * 8);  // ~8 rows per thread (as gn_finalize_apply)
  const int cap = (2048 + a.samples - 1) / a.samples;
  if (bps > cap) bps = cap;
  if (bps < 1) bps = 1;
  hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(bps, a.samples), dim3(threads), 0, (hipStream_t)stream_, a);
  return rcdm_check_launch();
}

int rcdm_groupnorm_finalize(int32_t samples, int32_t groups, int32_t splits, float eps, const float* partial, float* stat,
                            void* stream_) {
  if (!partial || !stat || samples <= 0 || groups <= 0 || splits <= 0) return RCDM_EINVAL;
  GnArgs a{};
  a.samples = samples; a.G = groups; a.splits = splits; a.eps = eps;
  a.partial = const_cast<float*>(partial);
  a.stat = stat;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((samples * groups + 3) / 4), dim3(256), 0, (hipStream_t)stream_, a);
  return rcdm_check_launch();
}

int rcdm_layernorm(const rcdm_layernorm_desc* d, const void* x, const float* gamma, const float* beta, const float* pe,
                   void* y, void* stream_) {
  if (!d || !x || !gamma || !beta || !y) return RCDM_EINVAL;
  if (d->M <= 0 || d->C <= 0) return RCDM_EINVAL;
  if ((d->C & 7) || d->C > 2048 || (d->ldx & 7) || (d->ldy & 7)) return RCDM_ESHAPE;
  if (pe && (d->rows_per_frame <= 0 || d->frames <= 0)) return RCDM_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  const int nchunks = d->C >> 3;
  dim3 block(256);
  const int rpf = pe ? d->rows_per_frame : 1, fr = pe ? d->frames : 1;
  static int wave_rows = -1;  // RCDM_LN_WAVEROW=1: the round-1 wave-per-row kernel (A/B switch)
  if (wave_rows < 0) {
    const char* e = getenv("RCDM_LN_WAVEROW");
    wave_rows = e ? atoi(e) : 0;
  }
  if (!wave_rows) {
    const int lpr = nchunks <= 40 ? 8 : nchunks <= 80 ? 16 : nchunks <= 160 ? 32 : 64;
    const int cpl = (nchunks + lpr - 1) / lpr;
    const int rows_per_block = 4 * (64 / lpr);
    const int nblocks = (d->M + rows_per_block - 1) / rows_per_block;
    const bool early = nblocks <= 3 * 256;  // <= 3 waves per SIMD: registers are free, latency is everything
#define LNG_LAUNCH(L, K)                                                                                              \
  if (early)                                                                                                          \
    hipLaunchKernelGGL((layernorm_grp_kernel<L, K, true>), dim3(nblocks), block, 0, stream, (const f16*)x, (f16*)y,   \
                       gamma, beta, pe, d->M, d->C, d->ldx, d->ldy, d->eps, rpf, fr);                                 \
  else                                                                                                                \
    hipLaunchKernelGGL((layernorm_grp_kernel<L, K, false>), dim3(nblocks), block, 0, stream, (const f16*)x, (f16*)y,  \
                       gamma, beta, pe, d->M, d->C, d->ldx, d->ldy, d->eps, rpf, fr)
#define LNG_CPL(L)                  \
  switch (cpl) {                    \
    case 1: LNG_LAUNCH(L, 1); break; \
    case 2: LNG_LAUNCH(L, 2); break; \
    case 3: LNG_LAUNCH(L, 3); break; \
    case 4: LNG_LAUNCH(L, 4); break; \
    default: LNG_LAUNCH(L, 5); break; \
  }
    switch (lpr) {
      case 8: LNG_CPL(8); break;
      case 16: LNG_CPL(16); break;
      case 32: LNG_CPL(32); break;
      default: LNG_CPL(64); break;
    }
#undef LNG_CPL
#undef LNG_LAUNCH
    return rcdm_check_launch();
  }
  const int nch = (nchunks + 63) / 64;
#define LN_LAUNCH(N, R)                                                                                      \
  hipLaunchKernelGGL((layernorm_kernel<N, R>), dim3((d->M + 4 * R - 1) / (4 * R)), block, 0, stream,        \
                     (const f16*)x, (f16*)y, gamma, beta, pe, d->M, d->C, d->ldx, d->ldy, d->eps, rpf, fr)
  switch (nch) {
    case 1: LN_LAUNCH(1, 4); break;
    case 2: LN_LAUNCH(2, 2); break;
    case 3: LN_LAUNCH(3, 2); break;
    default: LN_LAUNCH(4, 1); break;
  }
#undef LN_LAUNCH
  return rcdm_check_launch();
}

int rcdm_softmax_rows(int32_t M, int32_t N, int32_t ldx, int32_t ldy, float scale, const void* x, void* y, void* stream_) {
  if (!x || !y || M <= 0 || N <= 0) return RCDM_EINVAL;
  if ((N & 7) || N > 4096 || (ldx & 7) || (ldy & 7) || scale <= 0.f) return RCDM_ESHAPE;  // scale > 0: max first
  hipStream_t stream = (hipStream_t)stream_;
  const int nch = ((N >> 3) + 63) / 64;
#define SM_LAUNCH(NCH_)                                                                                  \
  hipLaunchKernelGGL((softmax_rows_kernel<NCH_>), dim3((M + 3) / 4), dim3(256), 0, stream, (const f16*)x, \
                     (f16*)y, M, N, ldx, ldy, scale)
  switch (nch) {
    case 1: SM_LAUNCH(1); break;
    case 2: SM_LAUNCH(2); break;
    case 3: case 4: SM_LAUNCH(4); break;
    default: SM_LAUNCH(8); break;
  }
#undef SM_LAUNCH
  return rcdm_check_launch();
}

}  // extern "C"
