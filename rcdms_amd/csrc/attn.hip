// attn.hip — flash-style spatial attention (self + cross) on MFMA, and the f-frame temporal
// attention, for gfx950.
//
// Replaces (reference): CrossAttention._attention src/models/attention.py:170-199 — baddbmm ->
// softmax -> bmm with the (B*heads, Lq, Lk) score tensor materialised (5.4 GB fp32 at 64x64 latents) —
// plus the head split/merge copies :93-105; and VersatileAttention.forward
// src/models/motion_module.py:294-354 (the "(b f) d c <-> (b d) f c" regroup + 5x5 attention).
//
// Spatial kernel: a block = 4 waves x 32 queries; K/V tiles of 64 keys staged row-major in ping-pong LDS images.
// Scores are computed TRANSPOSED, S^T = K Q^T (v_mfma_f32_32x32x16_f16, keys = rows, queries = cols),
// so a lane owns ONE query column: row max / sum / rescale are lane-local (+1 exchange with lane^32),
// and the exponentiated registers feed the second MFMA, O^T = V^T P^T, directly as its B operand.
// Its A operand (head-dim rows x keys) is read from the row-major V image with gfx950's LDS transpose read
// (ds_read_b64_tr_b16), in the key order the S^T accumulator registers hold P, so V needs no transposing
// scatter on the way in and no cross-lane shuffle on the way out.
#include <stdlib.h>
#include "common.h"

long long* g_attn_trace = nullptr;  // rcdm_debug_set_attn_trace (read by -DRCDM_ATTN_TRACE builds only)

namespace {

constexpr int KT = 64;        // keys per tile

struct AttnArgs {
  const f16* Q;
  const f16* K;
  const f16* V;
  f16* O;
  int batch, heads, Lq, Lk, d, dch;
  int ldq, ldk, ldv, ldo;
  float c;  // scale * log2(e)
  const unsigned char* kvalid;  // MASKED: [batch][Lk], 0 = key padded out (NULL = all valid)
  int causal;                   // MASKED: key k visible to query q only if k <= q
  int plain_order;              // RCDM_ATTN_XCD=0: blocks in plain (query block fastest) order, for A/B
  int wide;                     // RCDM_ATTN_WIDE_RANGE: the caller cannot bound |scaled score| < 2^15 -> never the MSUB kernel
  long long* trace;             // -DRCDM_ATTN_TRACE builds (tools/trace_attn.py): per-wave s_memtime sums of the loop's phases
};

// V row stride in LDS (halfs) for 32*DF padded columns: the smallest >= 64*DF bytes whose dword stride is 16 or 48
// (mod 64), so the four key rows one ds_read_b64_tr_b16 lane group touches fall in four different 16-bank slots.
__host__ __device__ constexpr int v_row_halfs(int DF) { return DF == 1 ? 32 : DF <= 3 ? 96 : 160; }

typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
// gfx950 LDS transpose read: every 16-lane group reads a [4 keys][16 columns] f16 block (lane i supplies the address
// of row i/4, columns 4(i%4)..+3) and lane i receives column i of that block, i.e. 4 keys of one head-dim column.
__device__ __forceinline__ s16x4 lds_read_tr16(const f16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// The kernel is VALU-bound at d = 40 (one exp per 4*d = 160 MFMA flops), so the softmax path is kept to
// ~4 VALU ops per score: raw v_exp_f32, v_max3 row max, packed RTZ f16 conversion, the row SUM taken from a
// ones-row appended to V^T (it falls out of the PV MFMA, consistently with the rounded P), the O rescale skipped
// while the running max does not move, and all K/V staging index math hoisted out of the key-tile loop.
// MASKED: causal and/or key-padding mask (the stage-1 prior transformer's additive -10000 mask, myprior_transformer.py:
// 389-393: a masked score contributes exp(-10000) = 0 in fp32, so masking hard to -inf is the same result).
// NW waves per block (4: two blocks per CU; 8: one block of 256 queries per CU — every K/V tile is fetched from L2 and
// staged once per 256 queries instead of once per 128: the self-attention of the 64x64 level re-reads 640 KB of K/V per
// query block, 1.6 GB per launch through the ~13 TB/s L2 -> CU path).
// MSUB (round 3; needs a spare QK^T column, d < 16 DS: the d = 40 self-attention of the 64x64 level, 85 % of the flash
// time): the softmax argument comes out of the MATRIX pipe.  Q is pre-multiplied by scale * log2(e), the spare column d
// of K holds 1 and the same column of Q holds -m (the running max, f16-exact), so the QK^T accumulator IS
// s * c - m and a score costs v_exp_f32 + half a v_cvt_pkrtz + half a v_max3 instead of those plus an fma.  The scores
// of tile kt + 1 are issued before the softmax of tile kt has decided its max, so they carry the max of one step earlier:
// the (per query, usually zero) difference is added only in tiles where some query's max moved — and it moves rarely,
// because the running max is only raised when a tile exceeds it by more than 2^MSUB_THR (deferred rescale: P <= 64 in
// f16, row sums and O in fp32).  Whatever value of m is used cancels in O / l, its f16 rounding included.
// Range: m lives in an f16 (the Q fragment), whose spacing is 2^(e-10) at magnitude 2^e, so P <= 2^(THR + ulp(m) / 2): 2^7 at
// |m| < 4096, 2^14 at |m| < 32768 — finite in f16 for every |scaled score| < 2^15 (documented in rcdm.h; beyond that the
// softmax is one-hot to ~10^4 digits and the caller's scale is wrong).  Accuracy: Q * c is re-rounded to f16, a relative
// 2^-12 on every score, i.e. a relative ln2 * 2^-12 * |score| on P (1 % at |score| = 50): tests/test_hip_kernels.py::
// test_flash_attn_msub_large_logits measures it against the fp32 oracle; RCDM_ATTN_MSUB=0 selects the fma-path kernel.
constexpr float MSUB_THR = 6.0f;

template <int DS, int QF, bool PIPE, bool MASKED, int NW, int MD = 0>  // MD: MSUB with head dim MD (0: off); d padded to 16*DS for QK^T and to 32*DF for PV; a wave owns QF fragments of 32 queries
__global__ __launch_bounds__(NW * 64, (MASKED && DS >= 5) ? 1 : 2) void flash_attn_kernel(const AttnArgs p) {   // (the masked wide-head forms need > 256 registers: one block per CU, the overflow in AGPRs instead of scratch)
  constexpr bool MSUB = MD > 0;
  static_assert(!MSUB || (PIPE && !MASKED && MD < 16 * DS), "MSUB: pipelined, unmasked kernel with a spare QK^T column only");
  constexpr int DF = (DS + 1) / 2;
  constexpr int KP = 16 * DS + 8;  // halfs per K row
  constexpr int NT = NW * 64;
  constexpr int NSLOT = (KT * 2 * DS + NT - 1) / NT;  // 16-B chunks a thread stages per tile (K and V each)
  constexpr int BQ = NW * 32 * QF;                  // queries per block
  constexpr int VR = v_row_halfs(DF);               // halfs per V row
  constexpr int SK = KT * KP, SV = KT * VR;         // halfs per K / V image; two of each (ping-pong)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f16* sK = (f16*)smem;           // [2][KT][KP]
  f16* sV = sK + 2 * SK;          // [2][KT][VR]  row-major, read transposed

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 31, hi = lane >> 5;
  // XCD-aware block order: block `lin` runs on XCD lin % 8 (each XCD has its own 4 MB L2), so every XCD is given a
  // contiguous run of the (batch, head, query block) sequence with the query block fastest — the 32 query blocks that
  // stream the same 640 KB of K/V then hit ONE L2 instead of pulling a copy into all eight (rocprofv3 FETCH_SIZE of the
  // 64x64 self-attention: 8x the K/V bytes with the plain order).
  const int nqb = (p.Lq + BQ - 1) / BQ;
  int item;
  {
    const int total = nqb * p.heads * p.batch, lin = blockIdx.x;
    const int xcd = lin & 7, q = total >> 3, r = total & 7;
    item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    if (p.plain_order) item = lin;
  }
  const int qb = item % nqb, bh = item / nqb;
  const int h = bh % p.heads, b = bh / p.heads;
  const int q0 = qb * BQ + wave * 32 * QF + lr;  // fragment j holds query q0 + 32 j
  // a spare V column exists: column d := 1 gives sum_k P[k][q] for free.  Always true for odd DS (d <= 16 DS < 32 DF).
  const bool ones_row = (DS & 1) ? true : p.d < 32 * DF;

  // zero the LDS images once: the pad columns of K and V stay zero afterwards
  for (int i = t; i < (2 * SK + 2 * SV) / 8; i += NT) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (ones_row && t < 2 * KT) sV[(t >> 6) * SV + (t & 63) * VR + p.d] = (f16)1.0f;
  if (MSUB && t < 2 * KT) sK[(t >> 6) * SK + (t & 63) * KP + p.d] = (f16)1.0f;   // K column d := 1 (never restaged: dch chunks only)
  constexpr int sd_frag = MD >> 4, sd_hi = (MD >> 3) & 1, sd_e = MD & 7;           // where column d sits in a Q fragment

  f16x8 qf[QF][DS];
#pragma unroll
  for (int j = 0; j < QF; ++j)
#pragma unroll
    for (int s = 0; s < DS; ++s) {
      const int dc = s * 16 + hi * 8;
      Pack16 v;
      v.u = make_uint4(0, 0, 0, 0);
      if (q0 + 32 * j < p.Lq && dc < p.d)
        v.u = *(const uint4*)(p.Q + ((size_t)b * p.Lq + q0 + 32 * j) * p.ldq + h * p.d + dc);
      if (MSUB) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v.e[e] = (f16)((float)v.e[e] * p.c);
      }
      qf[j][s] = v.h;
    }
  float m_sub[QF], m_issue[QF];   // MSUB: the max inside the Q fragment now / when the scores being consumed were issued
#pragma unroll
  for (int j = 0; j < QF; ++j) m_sub[j] = m_issue[j] = 0.f;

  f32x16 oacc[QF][DF];
  float m_run[QF], l_run[QF];
#pragma unroll
  for (int j = 0; j < QF; ++j) {
    m_run[j] = -INFINITY;
    l_run[j] = 0.f;
#pragma unroll
    for (int f = 0; f < DF; ++f)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[j][f][e] = 0.f;
  }

  const f16* Kb = p.K + (size_t)b * p.Lk * p.ldk + h * p.d;
  const f16* Vb = p.V + (size_t)b * p.Lk * p.ldv + h * p.d;
  const int ntiles = (p.Lk + KT - 1) / KT;
  const int nchunks = KT * p.dch;

  // ---- staging slots: loop-invariant (key, chunk) of the 16-B pieces this thread moves every tile ----------
  // Loads are raw buffer loads: a slot this thread does not own, and keys past Lk (last tile and the tiles the
  // pipeline over-fetches past the end), fall outside num_records and read as zero without branches.
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(
      (void*)Kb, 0, (int)(((size_t)(p.Lk - 1) * p.ldk + p.dch * 8) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(
      (void*)Vb, 0, (int)(((size_t)(p.Lk - 1) * p.ldv + p.dch * 8) * 2), 0x00020000);
  int k_lds[NSLOT], v_lds[NSLOT];
  unsigned k_off[NSLOT], v_off[NSLOT];  // byte offsets of the NEXT K / V tile to fetch
  bool s_ok[NSLOT];
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) {
    const int idx = t + NT * sl;
    s_ok[sl] = idx < nchunks;
    const int key = idx / p.dch, c = idx - key * p.dch;  // chunk fastest: contiguous in HBM and in LDS
    k_lds[sl] = key * KP + c * 8;
    v_lds[sl] = key * VR + c * 8;
    k_off[sl] = s_ok[sl] ? (unsigned)(key * p.ldk + c * 8) * 2u : OOB;
    v_off[sl] = s_ok[sl] ? (unsigned)(key * p.ldv + c * 8) * 2u : OOB;
  }
  // an OOB slot stays >= 2^31 and a live one < 2^31 while stepping: (Lk + 4 KT) * ld * 2 < 2^31 (launcher)
  const unsigned k_step = (unsigned)(KT * p.ldk) * 2u, v_step = (unsigned)(KT * p.ldv) * 2u;

  Pack16 kreg[NSLOT], vreg[NSLOT];
  auto fetch_k = [&]() {
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
      kreg[sl].v = __builtin_amdgcn_raw_buffer_load_b128(rK, k_off[sl], 0, 0);
      k_off[sl] += k_step;
    }
  };
  auto fetch_v = [&]() {
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
      vreg[sl].v = __builtin_amdgcn_raw_buffer_load_b128(rV, v_off[sl], 0, 0);
      v_off[sl] += v_step;
    }
  };
  auto put_k = [&](int par) {
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl)
      if (s_ok[sl]) *(uint4*)(sK + par * SK + k_lds[sl]) = kreg[sl].u;
  };
  auto put_v = [&](int par) {
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl)
      if (s_ok[sl]) *(uint4*)(sV + par * SV + v_lds[sl]) = vreg[sl].u;
  };
  // transposed V fragment base: lane (hi, column half ch, i) addresses key row 4 hi + i/4, columns 16 ch + 4 (i%4)
  const int v_rd = (4 * hi + ((lane & 15) >> 2)) * VR + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  // S^T = K Q^T for one key tile: two 32-key fragments, each K fragment read once for all QF query fragments
  auto qk = [&](int par, f32x16 (&sacc)[QF][2]) {
    constexpr f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int s = 0; s < DS; ++s) {
        const f16x8 kf = *(const f16x8*)(sK + par * SK + (f * 32 + lr) * KP + s * 16 + hi * 8);
#pragma unroll
        for (int j = 0; j < QF; ++j)
          sacc[j][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[j][s], s == 0 ? zero : sacc[j][f], 0, 0, 0);
      }
  };

  // ---- software pipeline ------------------------------------------------------------------------------------
  // Iteration kt consumes S(kt) (registers, computed one iteration earlier) and V(kt) (LDS, written one iteration
  // earlier), and meanwhile issues the QK^T MFMAs of tile kt+1, so the matrix pipe works under the softmax VALU
  // stream of the same wave.  K tiles are staged two ahead and V tiles one ahead into ping-pong LDS images, which
  // leaves ONE barrier per tile: everything written in iteration kt is first read in iteration kt+1, and
  // everything read in iteration kt was written before the barrier that opens it.
  fetch_k();  // K0
  fetch_v();  // V0
  put_k(0);
  put_v(0);
  if constexpr (PIPE) {
    fetch_k();  // K1
    put_k(1);
  }
  fetch_k();  // K2 (PIPE) or K1 -> registers
  fetch_v();  // V1 -> registers
  __syncthreads();
  f32x16 s_even[QF][2], s_odd[PIPE ? QF : 1][2];
  if constexpr (PIPE) qk(0, s_even);

#ifdef RCDM_ATTN_TRACE
  long long tr_sync = 0, tr_qk = 0, tr_sm = 0, tr_pv = 0, tr_q;
#define RCDM_ATTN_STAMP(acc_) do { const long long n_ = __builtin_amdgcn_s_memtime(); acc_ += n_ - tr_q; tr_q = n_; } while (0)
  const long long tr_begin = __builtin_amdgcn_s_memtime();
#else
#define RCDM_ATTN_STAMP(acc_) do {} while (0)
#endif
  auto step = [&](int kt, f32x16 (&sacc)[QF][2], f32x16 (&snext)[QF][2]) {  // !PIPE: snext aliases sacc, unused
    const int kbase = kt * KT, par = kt & 1;
#ifdef RCDM_ATTN_TRACE
    tr_q = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
    put_k(PIPE ? par : par ^ 1);  // PIPE: K(kt+2) over K(kt), last read in iteration kt-1; else K(kt+1)
    put_v(par ^ 1);   // V(kt+1): image `par^1` held V(kt-1), last read in iteration kt-1
    fetch_k();        // K(kt+3)
    fetch_v();        // V(kt+2)
    float m_next[QF];
#pragma unroll
    for (int j = 0; j < QF; ++j) m_next[j] = m_sub[j];
    RCDM_ATTN_STAMP(tr_sync);
    if constexpr (PIPE) {
      if (kt + 1 < ntiles) qk(par ^ 1, snext);
    } else {
      qk(par, sacc);  // d > 80: a second score set does not fit the register file
    }

    RCDM_ATTN_STAMP(tr_qk);
    f16x8 pf[QF][4];
#pragma unroll
    for (int j = 0; j < QF; ++j) {
      // lane holds, for query lr of fragment j, keys  f*32 + (r&3) + 8*(r>>2) + 4*hi
      if constexpr (MASKED) {
        // validity of the tile's 64 keys as one wave-uniform 64-bit mask (lane i looks at key kbase + i)
        bool kv = kbase + lane < p.Lk;
        if (kv && p.kvalid) kv = p.kvalid[(size_t)b * p.Lk + kbase + lane] != 0;
        const unsigned long long vmask = __ballot(kv);
        const int qq = q0 + 32 * j;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kl = f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const bool vis = ((vmask >> kl) & 1ull) && !(p.causal && kbase + kl > qq);
            if (!vis) sacc[j][f][r] = -INFINITY;
          }
      } else if (kbase + KT > p.Lk) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kbase + f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= p.Lk) sacc[j][f][r] = -INFINITY;
          }
      }
      float mx = max3f(sacc[j][0][0], sacc[j][0][1], sacc[j][0][2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) mx = max3f(mx, sacc[j][0][r], sacc[j][0][r + 1]);
      mx = max3f(mx, sacc[j][0][15], sacc[j][1][0]);
#pragma unroll
      for (int r = 1; r < 15; r += 2) mx = max3f(mx, sacc[j][1][r], sacc[j][1][r + 1]);
      mx = fmaxf(mx, sacc[j][1][15]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if constexpr (MSUB) {
        const float cand = mx + m_issue[j];   // this tile's (scaled) row max
        float m_new = m_run[j];
        if (cand > m_run[j] + MSUB_THR) m_new = (float)(f16)cand;   // first tile: m_run = -inf
        const bool moved = m_new != m_run[j];
        const float alpha = __builtin_amdgcn_exp2f(m_run[j] - m_new);
        m_run[j] = m_new;
        const float delta = m_issue[j] - m_new;   // the scores carry -m_issue
        if (__any(delta != 0.f)) {
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const auto pk = __builtin_amdgcn_cvt_pkrtz(__builtin_amdgcn_exp2f(sacc[j][f][r] + delta),
                                                         __builtin_amdgcn_exp2f(sacc[j][f][r + 1] + delta));
              pf[j][f * 2 + (r >> 3)][r & 7] = (f16)pk[0];
              pf[j][f * 2 + (r >> 3)][(r & 7) + 1] = (f16)pk[1];
            }
        } else {
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const auto pk = __builtin_amdgcn_cvt_pkrtz(__builtin_amdgcn_exp2f(sacc[j][f][r]),
                                                         __builtin_amdgcn_exp2f(sacc[j][f][r + 1]));
              pf[j][f * 2 + (r >> 3)][r & 7] = (f16)pk[0];
              pf[j][f * 2 + (r >> 3)][(r & 7) + 1] = (f16)pk[1];
            }
        }
        if (__any(moved)) {
          l_run[j] *= alpha;
#pragma unroll
          for (int f = 0; f < DF; ++f)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[j][f][e] *= alpha;
          // the new max into column d of this query's Q fragment (lanes of the half that holds it)
          m_sub[j] = m_new;
          if (hi == sd_hi) qf[j][sd_frag < DS ? sd_frag : 0][sd_e] = (f16)(-m_new);
        }
        m_issue[j] = m_next[j];
        continue;
      }
      float m_new = fmaxf(m_run[j], mx * p.c);  // unmasked: every tile has >= 1 valid key, so m_new is finite
      if constexpr (MASKED) {
        if (m_new == -INFINITY) m_new = 0.f;  // nothing visible to this query so far: P = exp2(-inf - 0) = 0, not NaN
      }
      const bool moved = m_new != m_run[j];
      const float alpha = __builtin_amdgcn_exp2f(m_run[j] - m_new);  // first tile: exp2(-inf) = 0
      m_run[j] = m_new;
      float lsum = 0.f;
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = __builtin_amdgcn_exp2f(fmaf(sacc[j][f][r], p.c, -m_new));
          const float p1 = __builtin_amdgcn_exp2f(fmaf(sacc[j][f][r + 1], p.c, -m_new));
          if (!ones_row) lsum += p0 + p1;
          const auto pk = __builtin_amdgcn_cvt_pkrtz(p0, p1);  // v_cvt_pkrtz_f16_f32: two f16 in one VALU op
          pf[j][f * 2 + (r >> 3)][r & 7] = (f16)pk[0];
          pf[j][f * 2 + (r >> 3)][(r & 7) + 1] = (f16)pk[1];
        }
      if (__any(moved)) {  // wave-uniform: once the running max has settled the accumulators are left alone
        l_run[j] *= alpha;
#pragma unroll
        for (int f = 0; f < DF; ++f)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[j][f][e] *= alpha;
      }
      l_run[j] += lsum;
    }

    // ---- O^T += V^T P^T : 4 steps of 16 keys.  The A fragment (head-dim rows x 8 keys per lane) is two
    // transpose reads of the row-major V image: keys 16 st + 4 hi + {0..3} and + 8 + {0..3}, the order P holds.
    RCDM_ATTN_STAMP(tr_sm);
#pragma unroll
    for (int f = 0; f < DF; ++f)
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const f16* vp = sV + par * SV + v_rd + (16 * st) * VR + 32 * f;
        union { s16x4 h[2]; f16x8 v; } vf;
        vf.h[0] = lds_read_tr16(vp);
        vf.h[1] = lds_read_tr16(vp + 8 * VR);
#pragma unroll
        for (int j = 0; j < QF; ++j) oacc[j][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf.v, pf[j][st], oacc[j][f], 0, 0, 0);
      }
    RCDM_ATTN_STAMP(tr_pv);
  };

  if constexpr (PIPE) {
    for (int kt = 0; kt < ntiles; kt += 2) {
      step(kt, s_even, s_odd);
      if (kt + 1 < ntiles) step(kt + 1, s_odd, s_even);
    }
  } else {
    for (int kt = 0; kt < ntiles; ++kt) step(kt, s_even, s_even);
  }
#ifdef RCDM_ATTN_TRACE
  if (p.trace && lane == 0) {
    long long* o = p.trace + ((size_t)blockIdx.x * NW + (threadIdx.x >> 6)) * 8;
    o[0] = __builtin_amdgcn_s_memtime() - tr_begin; o[1] = tr_sync; o[2] = tr_qk; o[3] = tr_sm; o[4] = tr_pv; o[5] = ntiles;
  }
#endif

#pragma unroll
  for (int j = 0; j < QF; ++j) {
    float l_tot;
    if (ones_row) {
      // row d of O^T: fragment d/32, register ((d%32)/8)*4, in the hi = 0 half of the wave
      float l0 = 0.f;
#pragma unroll
      for (int f = 0; f < DF; ++f)
#pragma unroll
        for (int rr = 0; rr < 16; rr += 4)
          if (f * 32 + rr * 2 == p.d) l0 = oacc[j][f][rr];
      l_tot = __shfl(l0, lr, 64);
    } else {
      l_tot = l_run[j] + __shfl_xor(l_run[j], 32, 64);
    }
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;  // MASKED: a query with no visible key gives 0, not NaN
    const int q = q0 + 32 * j;
    if (q < p.Lq) {
      f16* ob = p.O + ((size_t)b * p.Lq + q) * p.ldo + h * p.d;
#pragma unroll
      for (int f = 0; f < DF; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = f * 32 + 8 * g + 4 * hi;
          if (dd < p.d) {
            f16x4 o = {(f16)(oacc[j][f][4 * g] * inv), (f16)(oacc[j][f][4 * g + 1] * inv),
                       (f16)(oacc[j][f][4 * g + 2] * inv), (f16)(oacc[j][f][4 * g + 3] * inv)};
            *(f16x4*)(ob + dd) = o;
          }
        }
    }
  }
}

template <int DS>
int launch_flash(const AttnArgs& a_in, hipStream_t stream) {
  constexpr int DF = (DS + 1) / 2;
  constexpr int KP = 16 * DS + 8;
  const size_t lds = (size_t)2 * (KT * KP + KT * v_row_halfs(DF)) * sizeof(f16);  // ping-pong K and V images
  // (256-query, 8-wave blocks were measured 3-30 % SLOWER on every shape — the K/V stream is not what binds, an 8-wave barrier
  // per key tile costs more — and are not instantiated any more)
  static int xcd_mode = -1;
  if (xcd_mode < 0) {
    const char* e = getenv("RCDM_ATTN_XCD");
    xcd_mode = e ? atoi(e) : 1;
  }
  AttnArgs a = a_in;
  a.plain_order = xcd_mode ? 0 : 1;
  if (a.kvalid || a.causal) {
    dim3 grid(((a.Lq + 127) / 128) * a.heads * a.batch);
    hipLaunchKernelGGL((flash_attn_kernel<DS, 1, (DS <= 5), true, 4>), grid, dim3(256), lds, stream, a);
  } else {
    dim3 grid(((a.Lq + 127) / 128) * a.heads * a.batch);
    static int msub_mode = -1;  // RCDM_ATTN_MSUB=0: the fma-based softmax everywhere (A/B switch)
    if (msub_mode < 0) {
      const char* e = getenv("RCDM_ATTN_MSUB");
      msub_mode = e ? atoi(e) : 1;
    }
    {
      // d = 40: a spare QK^T column (40 of 48), a V ones-row (row sums in fp32 out of the PV MFMA) and a long key loop
      if constexpr (DS == 3) {
        if (msub_mode && !a.wide && a.d == 40 && a.Lk >= 4 * KT) {
          hipLaunchKernelGGL((flash_attn_kernel<DS, 1, true, false, 4, 40>), grid, dim3(256), lds, stream, a);
          return rcdm_check_launch();
        }
      }
    }
    hipLaunchKernelGGL((flash_attn_kernel<DS, 1, (DS <= 5), false, 4>), grid, dim3(256), lds, stream, a);
  }
  return rcdm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Cross-attention with a short key sequence (Lk <= 96: the 85 / 91 context rows of the stage-2 UNet, attention.py:
// 139-168 with encoder_hidden_states).  The flash kernel above is built for long key loops (LDS ping-pong images, one
// barrier per 64-key tile, online rescale); on two tiles it is a chain of latencies.  Here nothing is staged and
// nothing is shared: a wave owns 32 queries of ONE head and holds all 96 (padded) scores in registers —
//   S^T = K Q^T     K fragments straight from global memory,
//   softmax         one pass, lane-local + one exchange with lane ^ 32, no running max,
//   O^T = V^T P^T   V^T fragments straight from global memory, a row of ones behind the head's last dim so the row sum
//                   falls out of the same MFMA (consistent with the rounded P).
// K and V of the context do not change over the denoising steps, so rcdm_xattn_pack_kv writes them ONCE per context as
// a FRAGMENT-MAJOR image: every MFMA operand fragment is one contiguous 1-KiB block in lane order (lane l's eight halfs
// at byte 16 l), i.e. one fully coalesced load instruction.  (A first version read the fragments out of the row-major
// K and a [dim][key] V^T: 32 rows x 32 bytes per instruction — 25 % of every cache line used, the L1 thrashing — and was
// SLOWER than the flash kernel at every level: 37 / 27 / 25 us against 29 / 18 / 12.)  The V^T fragments hold the keys
// of every 16-group in the order the S^T accumulator registers hold P.
// One barrier: a block is 8 waves x 32 queries of one (batch, head); the head's fragment image (21 KB at d = 40) is copied
// into LDS once per block and every wave reads its operands from there with conflict-free lane-linear ds_read_b128.
// (Measured and dropped: one head per WAVE with each wave fetching the image itself — 215 MB of L2 -> CU traffic per
// launch at the 64x64 level, 25 us; the block's Q / O rows staged through LDS for whole-row accesses — two more
// barriers, 27 us.)
struct XAttnArgs {
  const f16* Q;
  const f16* KF;  // [batch][heads][3 key tiles][DS][64 lanes][8]
  const f16* VF;  // [batch][heads][DF dim tiles][6 key steps][64 lanes][8]
  f16* O;
  int batch, heads, Lq, Lk, d;
  int ldq, ldo;
  float c;  // scale * log2(e)
  int qi;   // chunks of NW * 32 queries a block walks through (same batch and head: one copy of the image serves them all)
};

template <int DS, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 8 && DS <= 3) ? 4 : 2) void xattn_kernel(const XAttnArgs p) {   // (second bound = waves per SIMD: two blocks of 8 waves per CU — 128 registers — for every head width but the 160-wide one)
  constexpr int DF = (DS + 1) / 2;
  constexpr int NFRAG = 3 * DS + 6 * DF;  // 1-KiB operand fragments of one (batch, head): K then V^T
  extern __shared__ __attribute__((aligned(16))) char xa_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, hi = lane >> 5;
  // a block = 8 waves x 32 queries of ONE (batch, head): its fragment image is copied to LDS once and read by all
  // eight (every wave fetching its own copy from L2 — 21 KB per 32 queries at d = 40 — ran at the L2 -> CU delivery limit)
  // A block walks through p.qi such chunks (round 5): the image copy — 80 bytes per query against the 160 of Q in + O out —
  // and the launch's latency chain (image, Q, one barrier, 21 MFMAs, 48 exps, store) are paid once per qi chunks, and the Q
  // rows of chunk i + 1 are requested before chunk i is computed.
  constexpr int BQ = NW * 32;
  const int nqb = (p.Lq + BQ - 1) / BQ;
  constexpr bool PF = DS <= 3;   // the chunk loop (with the next chunk's Q rows prefetched) for the narrow heads only: the wide ones have
                                 // no registers to spare at 4 waves per SIMD and few query chunks per head anyway (host: qi = 1)
  const int qi = PF ? p.qi : 1;
  const int npb = (nqb + qi - 1) / qi;          // blocks per (batch, head)
  const int bh = blockIdx.x / npb, qb0 = (blockIdx.x - bh * npb) * qi;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const bool ones_row = p.d < 32 * DF;
  constexpr f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  {
    const uint4* srck = (const uint4*)(p.KF + (size_t)bh * 3 * DS * 512);
    const uint4* srcv = (const uint4*)(p.VF + (size_t)bh * 6 * DF * 512);
    // (all of a thread's chunks requested before the first LDS write: one memory round trip for the image, not one per pass)
    constexpr int NIT = (NFRAG * 64 + NW * 64 - 1) / (NW * 64);
    uint4 tmp[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int it = threadIdx.x + k * NW * 64;
      tmp[k] = make_uint4(0, 0, 0, 0);
      if (it < NFRAG * 64) tmp[k] = *(it < 3 * DS * 64 ? srck + it : srcv + (it - 3 * DS * 64));
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int it = threadIdx.x + k * NW * 64;
      if (it < NFRAG * 64) ((uint4*)xa_smem)[it] = tmp[k];
    }
  }
  f16x8 qf[DS], qn[PF ? DS : 1];
  auto load_q = [&](int qb, f16x8 (&dst)[DS]) __attribute__((always_inline)) {
    const int q = qb * BQ + wave * 32 + lr;
    const bool ok = q < p.Lq;
    const f16* qrow = p.Q + ((size_t)b * p.Lq + (ok ? q : 0)) * p.ldq + h * p.d;
#pragma unroll
    for (int s = 0; s < DS; ++s) {
      const int dc = s * 16 + hi * 8;
      Pack16 v;
      v.u = make_uint4(0, 0, 0, 0);
      if (ok && dc < p.d) v.u = *(const uint4*)(qrow + dc);
      dst[s] = v.h;
    }
  };
  load_q(qb0, qf);
  __syncthreads();
  const char* frag = xa_smem + lane * 16;
  for (int it = 0; it < qi; ++it) {
  const int qb = qb0 + it;
  if (qb * BQ + wave * 32 >= p.Lq) return;  // a wave past the last query (no barrier follows)
  const int q = qb * BQ + wave * 32 + lr;
  const bool q_ok = q < p.Lq;
  if constexpr (PF) {
    if (it + 1 < qi) load_q(qb + 1, qn);    // (rows past Lq are not dereferenced)
  }

  // ---- S^T: three 32-key tiles
  f32x16 sacc[3];
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int s = 0; s < DS; ++s)
      sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(frag + (kt * DS + s) * 1024), qf[s],
                                                        s == 0 ? zero : sacc[kt], 0, 0, 0);

  // ---- softmax over the <= 96 keys of this lane's query: register r of tile kt is key 32 kt + (r&3) + 8 (r>>2) + 4 hi
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (key >= p.Lk) sacc[kt][r] = -INFINITY;
      mx = fmaxf(mx, sacc[kt][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float m = mx * p.c;  // Lk >= 1: at least one key is visible, m is finite
  f16x8 pf[6];
  float lsum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float p0 = __builtin_amdgcn_exp2f(fmaf(sacc[kt][r], p.c, -m));
      const float p1 = __builtin_amdgcn_exp2f(fmaf(sacc[kt][r + 1], p.c, -m));
      if (!ones_row) lsum += p0 + p1;
      const auto pk = __builtin_amdgcn_cvt_pkrtz(p0, p1);
      pf[kt * 2 + (r >> 3)][r & 7] = (f16)pk[0];
      pf[kt * 2 + (r >> 3)][(r & 7) + 1] = (f16)pk[1];
    }

  // ---- O^T = V^T P^T: six steps of 16 keys per dim tile
  f32x16 oacc[DF];
#pragma unroll
  for (int f = 0; f < DF; ++f)
#pragma unroll
    for (int st = 0; st < 6; ++st)
      oacc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(frag + (3 * DS + f * 6 + st) * 1024), pf[st],
                                                       st == 0 ? zero : oacc[f], 0, 0, 0);

  float l_tot;
  if (ones_row) {
    float l0 = 0.f;
#pragma unroll
    for (int f = 0; f < DF; ++f)
#pragma unroll
      for (int rr = 0; rr < 16; rr += 4)
        if (f * 32 + rr * 2 == p.d) l0 = oacc[f][rr];
    l_tot = __shfl(l0, lr, 64);
  } else {
    l_tot = lsum + __shfl_xor(lsum, 32, 64);
  }
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (q_ok) {
    f16* ob = p.O + ((size_t)b * p.Lq + q) * p.ldo + h * p.d;
#pragma unroll
    for (int f = 0; f < DF; ++f)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = f * 32 + g * 8 + hi * 4;
        if (dd < p.d) {
          f16x4 o = {(f16)(oacc[f][4 * g] * inv), (f16)(oacc[f][4 * g + 1] * inv), (f16)(oacc[f][4 * g + 2] * inv),
                     (f16)(oacc[f][4 * g + 3] * inv)};
          *(f16x4*)(ob + dd) = o;
        }
      }
  }
  if constexpr (PF) {
#pragma unroll
    for (int s = 0; s < DS; ++s) qf[s] = qn[s];
  }
  }
}

// K, V [batch*Lk][ld] (head h at columns h*d ..) -> the fragment-major image: thread = one lane slot (16 bytes) of one
// fragment; fragments of a (batch, head): 3*DS of K then 6*DF of V^T
__global__ void xattn_pack_kv_kernel(const f16* __restrict__ K, const f16* __restrict__ V, int batch, int Lk, int heads, int d,
                                     int ldk, int ldv, int DS, int DF, f16* __restrict__ img_k, f16* __restrict__ img_v) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_bh = (3 * DS + 6 * DF) * 64;
  if (idx >= (size_t)batch * heads * per_bh) return;
  const int bh = (int)(idx / per_bh), r = (int)(idx - (size_t)bh * per_bh);
  const int frag = r >> 6, lane = r & 63, lr = lane & 31, hi = lane >> 5;
  const int h = bh % heads, b = bh / heads;
  union { f16 e[8]; uint4 u; } o;
  if (frag < 3 * DS) {
    const int kt = frag / DS, s = frag - kt * DS;
    const int key = kt * 32 + lr, dc = s * 16 + hi * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      o.e[i] = (key < Lk && dc + i < d) ? K[((size_t)b * Lk + key) * ldk + h * d + dc + i] : (f16)0.f;
    *(uint4*)(img_k + ((size_t)bh * 3 * DS + frag) * 512 + lane * 8) = o.u;
  } else {
    const int fv = frag - 3 * DS, f = fv / 6, st = fv - f * 6;
    const int dd = f * 32 + lr;
    const bool ones = d < 32 * DF && dd == d;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int key = 16 * st + (i < 4 ? 4 * hi + i : 8 + 4 * hi + (i - 4));
      float v = ones ? 1.0f : 0.0f;
      if (dd < d && key < Lk) v = (float)V[((size_t)b * Lk + key) * ldv + h * d + dd];
      o.e[i] = (f16)v;
    }
    *(uint4*)(img_v + ((size_t)bh * 6 * DF + fv) * 512 + lane * 8) = o.u;
  }
}

// ---------------------------------------------------------------------------------------------
// temporal attention.  A block stages the [q|k|v] rows of TPB pixels x F frames in LDS with fully coalesced
// 16-byte loads (rows are 3C halfs contiguous), then lane (pixel, head, query frame i) computes its F scores,
// softmax and output row from LDS, overwrites its own q segment with the result, and the block writes the
// [0, C) columns of every staged row back as whole rows.  HBM sees each byte once, coalesced.
struct TAttnArgs {
  const f16* qkv;
  f16* out;
  int samples, pixels, heads, d, ldqkv, ldo, tpb;
  int g;  // lanes per (pixel, head, query frame) item: 1, 2, 4 or 8 adjacent lanes split the head dim
  float scale;
};

template <int F>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const TAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f16* tile = (f16*)smem;  // [tpb][F][3C]
  const int t = threadIdx.x;
  const int C = p.heads * p.d, C3 = 3 * C;
  const int rowc = C3 / 8;  // 16-B chunks per staged row
  const long total_px = (long)p.samples * p.pixels;
  const long px0 = (long)blockIdx.x * p.tpb;

  // ---- stage: a wave takes whole rows (pl, f), its lanes the row's 16-B chunks: the row -> address arithmetic
  // (divisions by runtime values) happens once per row on wave-uniform values, not once per chunk
  const int lane = t & 63, wave = t >> 6;
  // Four rows per wave and pass, all their loads issued before the first LDS write (round 5): with one row at a time a wave
  // had one or two 16-byte loads in flight and the kernel lived on occupancy alone.
  const int nrows = p.tpb * F;
  constexpr int SU = 4;
  for (int r0 = wave; r0 < nrows; r0 += 4 * SU) {
    const f16* src[SU];
    f16* dst[SU];
    bool ok[SU], live[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int r = r0 + 4 * u;
      live[u] = r < nrows;
      const int f = r % F, pl = r / F;
      const int px = (int)px0 + pl;
      ok[u] = live[u] && px < (int)total_px;
      const int b = ok[u] ? px / p.pixels : 0, pix = px - b * p.pixels;
      src[u] = p.qkv + ((long)(b * F + f) * p.pixels + (ok[u] ? pix : 0)) * (long)p.ldqkv;
      dst[u] = tile + (size_t)(live[u] ? r : 0) * C3;
    }
    for (int c = lane; c < rowc; c += 64) {
      uint4 v[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        v[u] = make_uint4(0, 0, 0, 0);
        if (ok[u]) v[u] = *(const uint4*)(src[u] + c * 8);
      }
#pragma unroll
      for (int u = 0; u < SU; ++u)
        if (live[u]) *(uint4*)(dst[u] + c * 8) = v[u];
    }
  }
  __syncthreads();

  // ---- compute: item = (pl, head, i); its p.g adjacent lanes take every p.g-th 16-B chunk of the head dim (wide heads
  // with one pixel per block would otherwise leave most of the block idle), partial scores meet in a xor-butterfly
  const int items = p.tpb * p.heads * F;
  const int sub = t % p.g, item = t / p.g;
  if (item < items) {
    const int i = item % F, ph = item / F;
    const int head = ph % p.heads, pl = ph / p.heads;
    f16* base = tile + (size_t)pl * F * C3 + head * p.d;
    f16* qrow = base + (size_t)i * C3;
    float s[F];
#pragma unroll
    for (int j = 0; j < F; ++j) s[j] = 0.f;
    for (int c = sub * 8; c < p.d; c += 8 * p.g) {
      Pack16 qv;
      qv.u = *(const uint4*)(qrow + c);
#pragma unroll
      for (int j = 0; j < F; ++j) {
        Pack16 kv;
        kv.u = *(const uint4*)(base + (size_t)j * C3 + C + c);
        float acc = s[j];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf((float)qv.e[e], (float)kv.e[e], acc);
        s[j] = acc;
      }
    }
    for (int off = 1; off < p.g; off <<= 1) {
#pragma unroll
      for (int j = 0; j < F; ++j) s[j] += __shfl_xor(s[j], off, 64);
    }
    float mx = s[0] * p.scale;
#pragma unroll
    for (int j = 0; j < F; ++j) {
      s[j] *= p.scale;
      mx = fmaxf(mx, s[j]);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < F; ++j) {
      s[j] = __expf(s[j] - mx);
      sum += s[j];
    }
    const float inv = __builtin_amdgcn_rcpf(sum);
    // every lane of the item has finished reading q (the butterfly above is a wave-level rendezvous), so the item's
    // q segment can take the output
    for (int c = sub * 8; c < p.d; c += 8 * p.g) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
      for (int j = 0; j < F; ++j) {
        Pack16 vv;
        vv.u = *(const uint4*)(base + (size_t)j * C3 + 2 * C + c);
        const float pj = s[j] * inv;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(pj, (float)vv.e[e], o[e]);
      }
      Pack16 ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) ov.e[e] = (f16)o[e];
      *(uint4*)(qrow + c) = ov.u;  // this item's own q segment: no other item reads it
    }
  }
  __syncthreads();

  // ---- store the [0, C) columns of every staged row as whole rows -----------------------------------------
  const int outc = C / 8;
  for (int r = wave; r < nrows; r += 4) {
    const int f = r % F, pl = r / F;
    const int px = (int)px0 + pl;
    if (px < (int)total_px) {
      const int b = px / p.pixels, pix = px - b * p.pixels;
      f16* dst = p.out + ((long)(b * F + f) * p.pixels + pix) * (long)p.ldo;
      const f16* srow = tile + (size_t)r * C3;
      for (int c = lane; c < outc; c += 64) *(uint4*)(dst + c * 8) = *(const uint4*)(srow + c * 8);
    }
  }
}

}  // namespace

extern "C" {

int rcdm_flash_attn_masked(const rcdm_attn_desc* d, const void* Q, const void* K, const void* V,
                           const unsigned char* key_valid, int32_t causal, void* out, void* stream_) {
  if (!d || !Q || !K || !V || !out) return RCDM_EINVAL;
  if (d->batch <= 0 || d->heads <= 0 || d->Lq <= 0 || d->Lk <= 0 || d->d <= 0) return RCDM_EINVAL;
  if ((d->d & 7) || d->d > 160) return RCDM_ESHAPE;
  if ((d->ldq & 7) || (d->ldk & 7) || (d->ldv & 7) || (d->ldo & 3)) return RCDM_ESHAPE;
  // K/V rows of one (batch, head) are addressed with 32-bit byte offsets (raw buffer loads)
  if ((size_t)(d->Lk + 4 * 64) * (size_t)(d->ldk > d->ldv ? d->ldk : d->ldv) * 2 >= 0x7FFFFFFFull) return RCDM_ESHAPE;
  AttnArgs a;
  a.Q = (const f16*)Q; a.K = (const f16*)K; a.V = (const f16*)V; a.O = (f16*)out;
  a.batch = d->batch; a.heads = d->heads; a.Lq = d->Lq; a.Lk = d->Lk; a.d = d->d; a.dch = d->d / 8;
  a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.ldo = d->ldo;
  a.c = d->scale * 1.4426950408889634f;
  a.kvalid = key_valid;
  a.wide = (d->flags & RCDM_ATTN_WIDE_RANGE) != 0;
  a.causal = causal ? 1 : 0;
  a.trace = g_attn_trace;
  hipStream_t stream = (hipStream_t)stream_;
  const int ds = (d->d + 15) / 16;
  if (ds <= 1) return launch_flash<1>(a, stream);
  if (ds <= 2) return launch_flash<2>(a, stream);
  if (ds <= 3) return launch_flash<3>(a, stream);
  if (ds <= 5) return launch_flash<5>(a, stream);
  return launch_flash<10>(a, stream);
}

int rcdm_debug_set_attn_trace(void* device_buffer) {
  g_attn_trace = (long long*)device_buffer;
  return RCDM_OK;
}

int rcdm_flash_attn(const rcdm_attn_desc* d, const void* Q, const void* K, const void* V, void* out, void* stream_) {
  return rcdm_flash_attn_masked(d, Q, K, V, nullptr, 0, out, stream_);
}

static int xattn_ds(int d) {
  const int ds = (d + 15) / 16;
  return ds <= 3 ? ds : ds <= 5 ? 5 : 10;
}

size_t rcdm_xattn_image_bytes(int32_t batch, int32_t heads, int32_t d) {
  if (batch <= 0 || heads <= 0 || d <= 0 || (d & 7) || d > 160) return 0;
  const int DS = xattn_ds(d), DF = (DS + 1) / 2;
  return (size_t)batch * heads * (3 * DS + 6 * DF) * 1024;
}

int rcdm_xattn_pack_kv(const void* K, const void* V, int32_t batch, int32_t Lk, int32_t heads, int32_t d, int32_t ldk,
                       int32_t ldv, void* image, void* stream_) {
  if (!K || !V || !image || batch <= 0 || heads <= 0 || Lk <= 0 || d <= 0) return RCDM_EINVAL;
  if ((d & 7) || d > 160 || Lk > 96 || ldk < heads * d || ldv < heads * d) return RCDM_ESHAPE;
  const int DS = xattn_ds(d), DF = (DS + 1) / 2;
  const size_t total = (size_t)batch * heads * (3 * DS + 6 * DF) * 64;
  f16* img_k = (f16*)image;
  f16* img_v = img_k + (size_t)batch * heads * 3 * DS * 512;
  hipLaunchKernelGGL(xattn_pack_kv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                     (const f16*)K, (const f16*)V, batch, Lk, heads, d, ldk, ldv, DS, DF, img_k, img_v);
  return rcdm_check_launch();
}

int rcdm_xattn(const rcdm_attn_desc* d, const void* Q, const void* image, void* out, void* stream_) {
  if (!d || !Q || !image || !out) return RCDM_EINVAL;
  if (d->batch <= 0 || d->heads <= 0 || d->Lq <= 0 || d->Lk <= 0 || d->d <= 0) return RCDM_EINVAL;
  if ((d->d & 7) || d->d > 160 || d->Lk > 96) return RCDM_ESHAPE;
  if ((d->ldq & 7) || (d->ldo & 3)) return RCDM_ESHAPE;
  const int DS = xattn_ds(d->d);
  XAttnArgs a;
  a.Q = (const f16*)Q; a.O = (f16*)out;
  a.KF = (const f16*)image;
  a.VF = a.KF + (size_t)d->batch * d->heads * 3 * DS * 512;
  a.batch = d->batch; a.heads = d->heads; a.Lq = d->Lq; a.Lk = d->Lk; a.d = d->d;
  a.ldq = d->ldq; a.ldo = d->ldo;
  a.c = d->scale * 1.4426950408889634f;
  const int DF_ = (DS + 1) / 2;
  static int nw_mode = -1;  // RCDM_XATTN_WAVES=4|8: waves (x 32 queries) per block (A/B switch)
  if (nw_mode < 0) {
    const char* e = getenv("RCDM_XATTN_WAVES");
    nw_mode = e ? atoi(e) : 4;
  }
  // Back to back (round 2) 8 waves won at the 64x64 level (21.2 us against 26.7 for 4, 23.5 for 16); in the replayed step
  // graph 4 waves are -0.07 ms per step over the sixteen launches (round 5, five same-box pairs: twice the blocks for the
  // 16x16 / 8x8 levels' 80 heads, and a cold image reaches 4 waves sooner than 8)
  const int nw = nw_mode == 8 ? 8 : 4;
  // query chunks per block: as many (<= 4 with 8 waves, <= 2 with 4) as leave one block per CU (RCDM_XATTN_QI overrides: A/B
  // switch).  Same-box A/B at the 64x64 level's five launches, 8 waves: 1 / 2 / 4 chunks = 17.685 / 17.645 / 17.638 ms per step;
  // with 4 waves the chunk count is within the noise (1 / 2 / 4: 17.28 / 17.27 / 17.28)
  static int qi_mode = -1;
  if (qi_mode < 0) {
    const char* e = getenv("RCDM_XATTN_QI");
    qi_mode = e ? atoi(e) : 0;
  }
  const int nqb = (d->Lq + nw * 32 - 1) / (nw * 32);
  int qi = qi_mode > 0 ? qi_mode : 1;
  if (qi_mode <= 0) {
    static int cus = 0;
    if (cus <= 0) {
      int dev = 0, n = 0;
      cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    const int cap = nw == 8 ? 4 : 2;
    while (qi < cap && d->batch * d->heads * ((nqb + 2 * qi - 1) / (2 * qi)) >= cus) qi *= 2;
  }
  if (qi > nqb) qi = nqb;
  if (DS > 3) qi = 1;   // (the wide-head instantiations have no chunk loop)
  a.qi = qi;
  const dim3 grid((unsigned)(d->batch * d->heads * ((nqb + qi - 1) / qi))), block(nw * 64);
  const size_t lds = (size_t)(3 * DS + 6 * DF_) * 1024;
  hipStream_t stream = (hipStream_t)stream_;
#define XA_LAUNCH(DS_)                                                                \
  if (nw == 4)                                                                        \
    hipLaunchKernelGGL((xattn_kernel<DS_, 4>), grid, block, lds, stream, a);          \
  else                                                                                \
    hipLaunchKernelGGL((xattn_kernel<DS_, 8>), grid, block, lds, stream, a)
  switch (DS) {
    case 1: XA_LAUNCH(1); break;
    case 2: XA_LAUNCH(2); break;
    case 3: XA_LAUNCH(3); break;
    case 5: XA_LAUNCH(5); break;
    default: XA_LAUNCH(10); break;
  }
#undef XA_LAUNCH
  return rcdm_check_launch();
}

int rcdm_temporal_attn(const rcdm_temporal_attn_desc* d, const void* qkv, void* out, void* stream_) {
  if (!d || !qkv || !out) return RCDM_EINVAL;
  if (d->samples <= 0 || d->pixels <= 0 || d->heads <= 0 || d->d <= 0) return RCDM_EINVAL;
  if (d->frames < 1 || d->frames > 8 || (d->d & 7) || (d->ldqkv & 7) || (d->ldo & 7)) return RCDM_ESHAPE;
  TAttnArgs a;
  a.qkv = (const f16*)qkv; a.out = (f16*)out; a.samples = d->samples; a.pixels = d->pixels;
  a.heads = d->heads; a.d = d->d; a.ldqkv = d->ldqkv; a.ldo = d->ldo; a.scale = d->scale;
  const int C = d->heads * d->d;
  const size_t px_bytes = (size_t)d->frames * 3 * C * sizeof(f16);   // LDS per pixel
  // measured (tools/kbench.py attn --only temporal): 32 KB blocks run the 64x64 / 32x32 levels at 4.1 / 3.5 TB/s, 60 KB
  // blocks (two per CU) at 2.9 / 2.5 — the three phases (stage, compute, store) of a block do not overlap, more blocks do
  static int lds_cap_kb = -1;  // RCDM_TATTN_KB: LDS per block (A/B switch)
  if (lds_cap_kb < 0) {
    const char* e = getenv("RCDM_TATTN_KB");
    lds_cap_kb = e ? atoi(e) : 32;
  }
  int tpb = (int)(((size_t)lds_cap_kb * 1024) / px_bytes);
  const int by_threads = 256 / (d->heads * d->frames);
  if (tpb > by_threads) tpb = by_threads;
  if (tpb < 1) tpb = 1;
  if ((size_t)tpb * px_bytes > 160 * 1024 || d->heads * d->frames > 256) return RCDM_ESHAPE;
  a.tpb = tpb;
  int g = 8;  // as many lanes per item as the block and the head dim allow
  while (g > 1 && (tpb * d->heads * d->frames * g > 256 || d->d / 8 < g)) g >>= 1;
  a.g = g;
  const size_t lds = (size_t)tpb * px_bytes;
  hipStream_t stream = (hipStream_t)stream_;
  const long total_px = (long)d->samples * d->pixels;
  dim3 grid((unsigned)((total_px + tpb - 1) / tpb)), block(256);
#define TA_LAUNCH(F)                                                                                              \
  {                                                                                                               \
    static size_t lds_set = 0;                                                                                    \
    if (lds > lds_set) {                                                                                          \
      (void)hipFuncSetAttribute((const void*)temporal_attn_kernel<F>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                        \
      lds_set = lds;                                                                                              \
    }                                                                                                             \
    hipLaunchKernelGGL(temporal_attn_kernel<F>, grid, block, lds, stream, a);                                     \
  }
  switch (d->frames) {
    case 1: TA_LAUNCH(1); break;
    case 2: TA_LAUNCH(2); break;
    case 3: TA_LAUNCH(3); break;
    case 4: TA_LAUNCH(4); break;
    case 5: TA_LAUNCH(5); break;
    case 6: TA_LAUNCH(6); break;
    case 7: TA_LAUNCH(7); break;
    default: TA_LAUNCH(8); break;
  }
#undef TA_LAUNCH
  return rcdm_check_launch();
}

}  // extern "C"
