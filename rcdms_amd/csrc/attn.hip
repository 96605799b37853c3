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

template <int DS>  // d padded to 16*DS for QK^T and to 32*DF for PV
__global__ __launch_bounds__(256, 2) void flash_attn_kernel(const AttnArgs p) {
  constexpr int DF = (DS + 1) / 2;
  constexpr int KP = 16 * DS + 8;  // halfs per K row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f16* sK = (f16*)smem;           // [KT][KP]
  f16* sVt = sK + KT * KP;        // [32*DF][VP]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 31, hi = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q = blockIdx.x * 128 + wave * 32 + lr;
  const bool q_ok = q < p.Lq;

  // zero both LDS images once: pad columns of K and pad rows of V^T stay zero afterwards
  for (int i = t; i < (KT * KP + 32 * DF * VP) / 8; i += 256) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);

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

  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();  // previous tile fully consumed (also orders the initial zero fill)
    for (int idx = t; idx < nchunks; idx += 256) {
      const int key = idx / p.dch, c = idx - key * p.dch;
      const int kg = kt * KT + key;
      Pack16 kv, vv;
      kv.u = vv.u = make_uint4(0, 0, 0, 0);
      if (kg < p.Lk) {
        kv.u = *(const uint4*)(Kb + (size_t)kg * p.ldk + c * 8);
        vv.u = *(const uint4*)(Vb + (size_t)kg * p.ldv + c * 8);
      }
      *(uint4*)(sK + key * KP + c * 8) = kv.u;
      const int pos = vt_pos(key);
#pragma unroll
      for (int e = 0; e < 8; ++e) sVt[(c * 8 + e) * VP + pos] = vv.e[e];
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
    const int kbase = kt * KT;
    if (kbase + KT > p.Lk) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= p.Lk) sacc[f][r] = -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[f][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * p.c);  // every tile has >= 1 valid key, so m_new is finite
    const float alpha = exp2f(m_run - m_new);    // first tile: exp2(-inf) = 0
    m_run = m_new;
    float lsum = 0.f;
    f16x8 pf[4];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = exp2f(fmaf(sacc[f][r], p.c, -m_new));
        lsum += pv;
        pf[f * 2 + (r >> 3)][r & 7] = (f16)pv;
      }
    l_run = l_run * alpha + lsum;
#pragma unroll
    for (int f = 0; f < DF; ++f)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[f][e] *= alpha;

    // ---- O^T += V^T P^T : 4 steps of 16 keys ---------------------------------------------------
#pragma unroll
    for (int f = 0; f < DF; ++f)
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const f16x8 vf = *(const f16x8*)(sVt + (f * 32 + lr) * VP + st * 16 + hi * 8);
        oacc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[st], oacc[f], 0, 0, 0);
      }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
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
// temporal attention: one lane per (sample, pixel, head); F x F scores in registers.
struct TAttnArgs {
  const f16* qkv;
  f16* out;
  int samples, pixels, heads, d, ldqkv, ldo;
  float scale;
};

template <int F>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const TAttnArgs p) {
  const size_t total = (size_t)p.samples * p.pixels * p.heads;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int head = (int)(idx % p.heads);
  const size_t bp = idx / p.heads;
  const int pix = (int)(bp % p.pixels);
  const int b = (int)(bp / p.pixels);
  const int C = p.heads * p.d;
  const f16* base = p.qkv + ((size_t)b * F * p.pixels + pix) * p.ldqkv + head * p.d;
  const size_t fstride = (size_t)p.pixels * p.ldqkv;

  float s[F][F];
#pragma unroll
  for (int i = 0; i < F; ++i)
#pragma unroll
    for (int j = 0; j < F; ++j) s[i][j] = 0.f;

  for (int c = 0; c < p.d; c += 8) {
    Pack16 qv[F], kv[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
      qv[f].u = *(const uint4*)(base + f * fstride + c);
      kv[f].u = *(const uint4*)(base + f * fstride + C + c);
    }
#pragma unroll
    for (int i = 0; i < F; ++i)
#pragma unroll
      for (int j = 0; j < F; ++j) {
        float acc = s[i][j];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf((float)qv[i].e[e], (float)kv[j].e[e], acc);
        s[i][j] = acc;
      }
  }
#pragma unroll
  for (int i = 0; i < F; ++i) {
    float mx = s[i][0] * p.scale;
#pragma unroll
    for (int j = 0; j < F; ++j) {
      s[i][j] *= p.scale;
      mx = fmaxf(mx, s[i][j]);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < F; ++j) {
      s[i][j] = __expf(s[i][j] - mx);
      sum += s[i][j];
    }
    const float inv = 1.f / sum;
#pragma unroll
    for (int j = 0; j < F; ++j) s[i][j] *= inv;
  }
  f16* ob = p.out + ((size_t)b * F * p.pixels + pix) * p.ldo + head * p.d;
  const size_t ostride = (size_t)p.pixels * p.ldo;
  for (int c = 0; c < p.d; c += 8) {
    Pack16 vv[F];
#pragma unroll
    for (int f = 0; f < F; ++f) vv[f].u = *(const uint4*)(base + f * fstride + 2 * C + c);
#pragma unroll
    for (int i = 0; i < F; ++i) {
      Pack16 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < F; ++j) acc = fmaf(s[i][j], (float)vv[j].e[e], acc);
        o.e[e] = (f16)acc;
      }
      *(uint4*)(ob + i * ostride + c) = o.u;
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
  hipStream_t stream = (hipStream_t)stream_;
  const size_t total = (size_t)d->samples * d->pixels * d->heads;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define TA_LAUNCH(F) hipLaunchKernelGGL(temporal_attn_kernel<F>, grid, block, 0, stream, a)
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
