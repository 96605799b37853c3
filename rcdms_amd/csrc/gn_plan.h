// gn_plan.h — the launch geometry and the block reduction of the GroupNorm statistics pass, shared by norm.hip
// (gn_stats_kernel) and igemm.hip (splitk_reduce_gn_kernel: a split-K GEMM / conv whose reduce pass leaves the GroupNorm
// partial statistics of the rows it writes, so that the norm behind it needs no statistics launch of its own).
#pragma once
#include "common.h"

constexpr int GN_U = 8;  // 16-byte loads a thread of the stats / apply kernels keeps in flight

struct GnArgs {
  const f16* x;
  f16* y;
  const float* gamma;
  const float* beta;
  float* partial;  // [samples][groups][splits][3] = (count, mean, M2)
  float* stat;     // [samples][groups][2] = (mean, rstd)
  int samples, P, C, G, cg, CH, RPB, ldx, ldy, splits, rows_per_split;
  float eps;
  int silu;
};

// Geometry of the three-launch form for a descriptor (grid (splits, samples), CH * RPB threads: thread (rl, ch) owns 16-byte
// chunk ch of rows rl + k RPB of its split).  Defined in norm.hip.  gn_three_launch: whether rcdm_groupnorm_silu takes that
// form for this descriptor (not the single-launch kernel of the smallest tensors) — only then do partials exist.
int rcdm_gn_plan(const rcdm_groupnorm_desc* d, GnArgs& a);
bool rcdm_gn_three_launch(const GnArgs& a);

// Tail of a statistics block: per-thread column sums (sum[e], sq[e] of the thread's 8 columns over its rows) -> the G group
// partials (count, mean, M2) of split `sp` of sample `s`.  part: dynamic LDS, (threads + CH) * 16 floats.  Column sums first
// (CH * 16 values, each over the RPB row-threads, spread over the whole block), then the G groups: the one-step form (G
// threads walking RPB * cg entries each) was a 120-read serial tail on 32 threads per block.  (The [thread][16] layout puts
// a wave's ds accesses on two banks — SQ_LDS_BANK_CONFLICT several times SQ_ACTIVE_INST_LDS in the counters — but a
// conflict-free value-major layout measured the same kernel times and the same step time, round 5: the tail is not on the
// block's critical path, its one memory round trip is.)
__device__ __forceinline__ void gn_block_partials(const GnArgs& p, float* part, int t, const float (&sum)[8], const float (&sq)[8],
                                                  int s, int sp, int nrows) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    part[t * 16 + e] = sum[e];
    part[t * 16 + 8 + e] = sq[e];
  }
  __syncthreads();
  const float* colsum = part;  // one row-thread per chunk: the per-thread sums ARE the column sums
  if (p.RPB > 1) {
    float* cs = part + blockDim.x * 16;
    for (int o = t; o < p.CH * 16; o += blockDim.x) {
      const int c = o >> 4, k = o & 15;
      float a = 0.f;
      for (int r = 0; r < p.RPB; ++r) a += part[(r * p.CH + c) * 16 + k];
      cs[o] = a;
    }
    colsum = cs;
    __syncthreads();
  }
  if (t < p.G) {
    float gs = 0.f, gq = 0.f;
    for (int c = t * p.cg; c < (t + 1) * p.cg; ++c) {
      gs += colsum[(c >> 3) * 16 + (c & 7)];
      gq += colsum[(c >> 3) * 16 + 8 + (c & 7)];
    }
    const float n = (float)nrows * (float)p.cg;
    const float mean = n > 0.f ? gs / n : 0.f;
    float m2 = gq - gs * mean;
    if (m2 < 0.f) m2 = 0.f;
    float* o = p.partial + (((size_t)s * p.G + t) * p.splits + sp) * 3;
    o[0] = n;
    o[1] = mean;
    o[2] = m2;
  }
}
