// attn.hip — flash-style spatial attention (self + cross) on MFMA, and the f-frame temporal
// attention, for gfx950.
//
// Replaces (reference): CrossAttention._attention src/models/attention.py:170-199 — baddbmm ->
// softmax -> bmm with the (B*heads, Lq, Lk) score tensor materialised (5.4 GB fp32 at 64x64 latents) —
// plus the head split/merge copies :93-105; and VersatileAttention.forward
// src/models/motion_module.py:294-354 (the "(b f) d c <-> (b d) f c" regroup + 5x5 attention).
//
// Spatial kernel: a block = 4 waves x 32 queries; K/V tiles of 64 keys staged in LDS.
// Scores are computed TRANSPOSED, S^T = K Q^T (v_mfma_f32_32x32x16_f16, keys = rows, queries = cols),
// so a lane owns ONE query column: row max / sum / rescale are lane-local (+1 exchange with lane^32),
// and the exponentiated registers feed the second MFMA, O^T = V^T P^T, directly as its B operand:
// the key order inside each 16-key step only has to agree between P (registers) and V^T (LDS image),
// so V is transposed into LDS with the matching key permutation and no cross-lane shuffle is needed.
#include "common.h"

namespace {

constexpr int KT = 64;        // keys per tile
constexpr int VP = KT + 8;    // halfs per V^T row (144 B)

struct AttnArgs {
  const f16* Q;
  const f16* K;
  const f16* V;
  f16* O;
  int batch, heads, Lq, Lk, d, dch;
  int ldq, ldk, ldv, ldo;
  float c;  // scale * log2(e)
};

// position of key k (0..63) inside a V^T row: within each 16-key step the lane group `hi` must find
// its 8 keys contiguous, in the order the S^T accumulator registers hold them.
__device__ __forceinline__ int vt_pos(int k) {
  return (k & ~15) + 8 * ((k >> 2) & 1) + (k & 3) + 4 * ((k >> 3) & 1);
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
template <int DS>  // d padded to 16*DS for QK^T and to 32*DF for PV
__global__ __launch_bounds__(256, 2) void flash_attn_kernel(const AttnArgs p) {
  constexpr int DF = (DS + 1) / 2;
  constexpr int KP = 16 * DS + 8;  // halfs per K row
  constexpr int NSLOT = (KT * 2 * DS + 255) / 256;  // 16-B chunks a thread stages per tile (K and V each)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f16* sK = (f16*)smem;           // [KT][KP]
  f16* sVt = sK + KT * KP;        // [32*DF][VP]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 31, hi = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q = blockIdx.x * 128 + wave * 32 + lr;
  const bool q_ok = q < p.Lq;
  const bool ones_row = p.d < 32 * DF;  // a spare V^T row exists: row d := 1 gives sum_k P[k][q] for free

  // zero both LDS images once: pad columns of K and pad rows of V^T stay zero afterwards
  for (int i = t; i < (KT * KP + 32 * DF * VP) / 8; i += 256) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (ones_row && t < KT) sVt[p.d * VP + t] = (f16)1.0f;

  f16x8 qf[DS];
#pragma unroll
  for (int s = 0; s < DS; ++s) {
    const int dc = s * 16 + hi * 8;
    Pack16 v;
    v.u = make_uint4(0, 0, 0, 0);
    if (q_ok && dc < p.d) v.u = *(const uint4*)(p.Q + ((size_t)b * p.Lq + q) * p.ldq + h * p.d + dc);
    qf[s] = v.h;
  }

  f32x16 oacc[DF];
#pragma unroll
  for (int f = 0; f < DF; ++f)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[f][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const f16* Kb = p.K + (size_t)b * p.Lk * p.ldk + h * p.d;
  const f16* Vb = p.V + (size_t)b * p.Lk * p.ldv + h * p.d;
  const int ntiles = (p.Lk + KT - 1) / KT;
  const int nchunks = KT * p.dch;

  // ---- staging slots: loop-invariant (key, chunk) of the 16-B pieces this thread moves every tile ----------
  int k_key[NSLOT], k_lds[NSLOT], v_key[NSLOT], v_lds[NSLOT];
  size_t k_src[NSLOT], v_src[NSLOT];
  bool s_ok[NSLOT];
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) {
    const int idx = t + 256 * sl;
    s_ok[sl] = idx < nchunks;
    const int key = idx / p.dch, c = idx - key * p.dch;  // K: chunk fastest (contiguous in HBM and LDS)
    k_key[sl] = key;
    k_lds[sl] = key * KP + c * 8;
    k_src[sl] = (size_t)key * p.ldk + c * 8;
    const int vkey = idx & (KT - 1), vc = idx >> 6;      // V: key fastest (one V^T row per transposing write)
    v_key[sl] = vkey;
    v_lds[sl] = vc * 8 * VP + vt_pos(vkey);
    v_src[sl] = (size_t)vkey * p.ldv + vc * 8;
  }

  for (int kt = 0; kt < ntiles; ++kt) {
    const int kbase = kt * KT;
    Pack16 kreg[NSLOT], vreg[NSLOT];
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
      kreg[sl].u = vreg[sl].u = make_uint4(0, 0, 0, 0);
      if (s_ok[sl] && kbase + k_key[sl] < p.Lk) kreg[sl].u = *(const uint4*)(Kb + (size_t)kbase * p.ldk + k_src[sl]);
      if (s_ok[sl] && kbase + v_key[sl] < p.Lk) vreg[sl].u = *(const uint4*)(Vb + (size_t)kbase * p.ldv + v_src[sl]);
    }
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
      if (s_ok[sl]) {
        *(uint4*)(sK + k_lds[sl]) = kreg[sl].u;
#pragma unroll
        for (int e = 0; e < 8; ++e) sVt[v_lds[sl] + e * VP] = vreg[sl].e[e];
      }
    }
    __syncthreads();

    // ---- S^T = K Q^T : two 32-key fragments ---------------------------------------------------
    f32x16 sacc[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[f][e] = 0.f;
#pragma unroll
      for (int s = 0; s < DS; ++s) {
        const f16x8 kf = *(const f16x8*)(sK + (f * 32 + lr) * KP + s * 16 + hi * 8);
        sacc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc[f], 0, 0, 0);
      }
    }
    // lane holds, for query lr, keys  f*32 + (r&3) + 8*(r>>2) + 4*hi
    if (kbase + KT > p.Lk) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= p.Lk) sacc[f][r] = -INFINITY;
        }
    }
    float mx = max3f(sacc[0][0], sacc[0][1], sacc[0][2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = max3f(mx, sacc[0][r], sacc[0][r + 1]);
    mx = max3f(mx, sacc[0][15], sacc[1][0]);
#pragma unroll
    for (int r = 1; r < 15; r += 2) mx = max3f(mx, sacc[1][r], sacc[1][r + 1]);
    mx = fmaxf(mx, sacc[1][15]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * p.c);  // every tile has >= 1 valid key, so m_new is finite
    const bool moved = m_new != m_run;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // first tile: exp2(-inf) = 0
    m_run = m_new;
    float lsum = 0.f;
    f16x8 pf[4];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(fmaf(sacc[f][r], p.c, -m_new));
        const float p1 = __builtin_amdgcn_exp2f(fmaf(sacc[f][r + 1], p.c, -m_new));
        if (!ones_row) lsum += p0 + p1;
        const auto pk = __builtin_amdgcn_cvt_pkrtz(p0, p1);  // v_cvt_pkrtz_f16_f32: two f16 in one VALU op
        pf[f * 2 + (r >> 3)][r & 7] = (f16)pk[0];
        pf[f * 2 + (r >> 3)][(r & 7) + 1] = (f16)pk[1];
      }
    if (__any(moved)) {  // wave-uniform: once the running max has settled the accumulators are left alone
      l_run *= alpha;
#pragma unroll
      for (int f = 0; f < DF; ++f)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[f][e] *= alpha;
    }
    l_run += lsum;

    // ---- O^T += V^T P^T : 4 steps of 16 keys ---------------------------------------------------
#pragma unroll
    for (int f = 0; f < DF; ++f)
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const f16x8 vf = *(const f16x8*)(sVt + (f * 32 + lr) * VP + st * 16 + hi * 8);
        oacc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[st], oacc[f], 0, 0, 0);
      }
  }

  float l_tot;
  if (ones_row) {
    // row d of O^T: fragment d/32, register ((d%32)/8)*4, in the hi = 0 half of the wave
    float l0 = 0.f;
#pragma unroll
    for (int f = 0; f < DF; ++f)
#pragma unroll
      for (int rr = 0; rr < 16; rr += 4)
        if (f * 32 + rr * 2 == p.d) l0 = oacc[f][rr];
    l_tot = __shfl(l0, lr, 64);
  } else {
    l_tot = l_run + __shfl_xor(l_run, 32, 64);
  }
  const float inv = 1.f / l_tot;
  if (q_ok) {
    f16* ob = p.O + ((size_t)b * p.Lq + q) * p.ldo + h * p.d;
#pragma unroll
    for (int f = 0; f < DF; ++f)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = f * 32 + 8 * g + 4 * hi;
        if (dd < p.d) {
          f16x4 o = {(f16)(oacc[f][4 * g] * inv), (f16)(oacc[f][4 * g + 1] * inv),
                     (f16)(oacc[f][4 * g + 2] * inv), (f16)(oacc[f][4 * g + 3] * inv)};
          *(f16x4*)(ob + dd) = o;
        }
      }
  }
}

template <int DS>
int launch_flash(const AttnArgs& a, hipStream_t stream) {
  constexpr int DF = (DS + 1) / 2;
  constexpr int KP = 16 * DS + 8;
  const size_t lds = (size_t)(KT * KP + 32 * DF * VP) * sizeof(f16);
  dim3 grid((a.Lq + 127) / 128, a.heads, a.batch);
  hipLaunchKernelGGL(flash_attn_kernel<DS>, grid, dim3(256), lds, stream, a);
  return rcdm_check_launch();
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

  // ---- stage: chunk id -> (pl, f, c) ------------------------------------------------------------------
  const int nchunks = p.tpb * F * rowc;
  for (int id = t; id < nchunks; id += 256) {
    const int c = id % rowc, r = id / rowc;
    const int f = r % F, pl = r / F;
    const long px = px0 + pl;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (px < total_px) {
      const long b = px / p.pixels, pix = px - b * p.pixels;
      v = *(const uint4*)(p.qkv + ((b * F + f) * p.pixels + pix) * (long)p.ldqkv + c * 8);
    }
    *(uint4*)(tile + (size_t)r * C3 + c * 8) = v;
  }
  __syncthreads();

  // ---- compute: lane = (pl, head, i) ----------------------------------------------------------------------
  const int items = p.tpb * p.heads * F;
  if (t < items) {
    const int i = t % F, ph = t / F;
    const int head = ph % p.heads, pl = ph / p.heads;
    f16* base = tile + (size_t)pl * F * C3 + head * p.d;
    f16* qrow = base + (size_t)i * C3;
    float s[F];
#pragma unroll
    for (int j = 0; j < F; ++j) s[j] = 0.f;
    for (int c = 0; c < p.d; c += 8) {
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
    const float inv = 1.f / sum;
    for (int c = 0; c < p.d; c += 8) {
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
      *(uint4*)(qrow + c) = ov.u;  // this lane's own q segment: nobody else reads it
    }
  }
  __syncthreads();

  // ---- store the [0, C) columns of every staged row as whole rows -----------------------------------------
  const int outc = C / 8;
  const int nout = p.tpb * F * outc;
  for (int id = t; id < nout; id += 256) {
    const int c = id % outc, r = id / outc;
    const int f = r % F, pl = r / F;
    const long px = px0 + pl;
    if (px < total_px) {
      const long b = px / p.pixels, pix = px - b * p.pixels;
      *(uint4*)(p.out + ((b * F + f) * p.pixels + pix) * (long)p.ldo + c * 8) =
          *(const uint4*)(tile + (size_t)r * C3 + c * 8);
    }
  }
}

}  // namespace

extern "C" {

int rcdm_flash_attn(const rcdm_attn_desc* d, const void* Q, const void* K, const void* V, void* out, void* stream_) {
  if (!d || !Q || !K || !V || !out) return RCDM_EINVAL;
  if (d->batch <= 0 || d->heads <= 0 || d->Lq <= 0 || d->Lk <= 0 || d->d <= 0) return RCDM_EINVAL;
  if ((d->d & 7) || d->d > 160) return RCDM_ESHAPE;
  if ((d->ldq & 7) || (d->ldk & 7) || (d->ldv & 7) || (d->ldo & 3)) return RCDM_ESHAPE;
  AttnArgs a;
  a.Q = (const f16*)Q; a.K = (const f16*)K; a.V = (const f16*)V; a.O = (f16*)out;
  a.batch = d->batch; a.heads = d->heads; a.Lq = d->Lq; a.Lk = d->Lk; a.d = d->d; a.dch = d->d / 8;
  a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.ldo = d->ldo;
  a.c = d->scale * 1.4426950408889634f;
  hipStream_t stream = (hipStream_t)stream_;
  const int ds = (d->d + 15) / 16;
  if (ds <= 1) return launch_flash<1>(a, stream);
  if (ds <= 2) return launch_flash<2>(a, stream);
  if (ds <= 3) return launch_flash<3>(a, stream);
  if (ds <= 5) return launch_flash<5>(a, stream);
  return launch_flash<10>(a, stream);
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
  int tpb = (int)((60 * 1024) / px_bytes);                            // ~60 KB: two blocks per CU
  const int by_threads = 256 / (d->heads * d->frames);
  if (tpb > by_threads) tpb = by_threads;
  if (tpb < 1) tpb = 1;
  if ((size_t)tpb * px_bytes > 160 * 1024 || d->heads * d->frames > 256) return RCDM_ESHAPE;
  a.tpb = tpb;
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
