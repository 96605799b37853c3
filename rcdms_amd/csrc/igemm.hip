// igemm.hip — MFMA implicit-GEMM for gfx950: Linear / 1x1 conv (taps = 1) and conv3x3 (taps = 9)
// over channels-last f16 rows, fp32 accumulate, fused epilogues.
//
// Replaces (reference, muzishen/RCDMs): InflatedConv3d.forward src/models/resnet.py:10-18, the
// nn.Linear calls of CrossAttention src/models/attention.py:121,140-141,164, Transformer3DModel
// proj_in/proj_out :330,:352, TemporalTransformer3DModel proj_in/out motion_module.py:166,170,
// diffusers FeedForward (GEGLU) and ResnetBlock3D conv_shortcut resnet.py:208.
//
// Tile: 128 pixels x 128 channels x 64 k, 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_f16.
// The WEIGHT tile is the MFMA A operand and the ACTIVATION tile the B operand, so D[channel][pixel]:
// a lane owns one pixel and 4 consecutive channels per register quad -> 16-byte LDS staging of the
// fp32 tile, then a coalesced 16-byte-per-lane epilogue (bias / per-sample row vector / residual /
// GEGLU / scale) with one rounding to f16.
// LDS rows are padded to 144 B so every ds_read_b128 lane group hits 16 distinct 16-B slots.
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int PITCH = BK + 8;   // halfs per LDS operand row (144 B)
constexpr int CPITCH = BN + 4;  // floats per staged C row (528 B)
constexpr int NTHREADS = 256;
constexpr int LDS_AB = 2 * 2 * BM * PITCH * 2;  // 73728
constexpr int LDS_C = BM * CPITCH * 4;          // 67584
constexpr int LDS_BYTES = LDS_AB > LDS_C ? LDS_AB : LDS_C;

struct IgemmArgs {
  const f16* A;
  const f16* W;
  const float* bias;
  const float* rowvec;
  const f16* res;
  f16* out;
  float* partial;
  int M, N, Cin, Ktot;
  int Hi, Wi, Ho, Wo, stride, up;
  int lda, ldc, ldr, ldt, rows_per_sample;
  int epi;
  float out_scale;
  int tilesM, tilesN, kc, nk, splits, nk_per_split;
};

// v: 8 accumulated values of row m at packed columns n..n+7 (GEGLU: g = the matching gate columns).
__device__ __forceinline__ void epilogue_store(const IgemmArgs& p, int m, int n, float (&v)[8], float (&g)[8]) {
  int oc = n;
  if (p.epi & RCDM_EPI_GEGLU) {
    if (p.epi & RCDM_EPI_BIAS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] += p.bias[n + e];
        g[e] += p.bias[n + 64 + e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * gelu_f(g[e]);
    oc = (n >> 7) * 64 + (n & 63);
  } else if (p.epi & RCDM_EPI_BIAS) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += p.bias[n + e];
  }
  if (p.epi & RCDM_EPI_ROWVEC) {
    const float* rv = p.rowvec + (size_t)(m / p.rows_per_sample) * p.ldt + oc;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += rv[e];
  }
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
}

template <int TAPS>
__global__ __launch_bounds__(NTHREADS, 2) void igemm_kernel(const IgemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f16* sA = (f16*)smem;           // [2][BM][PITCH] activation rows (pixels)
  f16* sB = sA + 2 * BM * PITCH;  // [2][BN][PITCH] weight rows (output channels)
  float* sC = (float*)smem;       // [BM][CPITCH] after the k loop

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

  // XCD-aware tile order: block b runs on XCD b%8; give each XCD a contiguous run of tiles so the
  // tiles sharing an activation row-panel hit the same L2.
  const int ntiles = p.tilesM * p.tilesN;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = ntiles >> 3, r = ntiles & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_n = bid % p.tilesN, tile_m = bid / p.tilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int ks_begin = blockIdx.y * p.nk_per_split;
  const int ks_end = min(p.nk, ks_begin + p.nk_per_split);

  // ---- loader state: thread owns 16-B chunk cc of rows r0 + 32 j -------------------------------
  // All global reads are raw buffer loads: a lane whose row / tap / channel chunk is out of range gets
  // the offset 0x80000000 (>= num_records) and the hardware returns zeros — no branches, no scratch.
  const int cc = t & 7, r0 = t >> 3;
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7FFFFFFF, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  unsigned a_off0, a_off1, a_off2, a_off3;      // TAPS==1: byte offset of the row
  int a_img0, a_img1, a_img2, a_img3, a_iy0, a_iy1, a_iy2, a_iy3, a_ix0, a_ix1, a_ix2, a_ix3;
  unsigned w_off0, w_off1, w_off2, w_off3;
#define ROW_SETUP(J)                                                              \
  {                                                                               \
    const int m = m0 + r0 + 32 * J;                                               \
    a_img##J = a_iy##J = a_ix##J = 0;                                             \
    a_off##J = OOB;                                                               \
    if (TAPS == 1) {                                                              \
      if (m < p.M) a_off##J = (unsigned)m * (unsigned)p.lda * 2u;                 \
    } else {                                                                      \
      const int hw = p.Ho * p.Wo;                                                 \
      const int img = m / hw, rem = m - img * hw;                                 \
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;                            \
      a_img##J = img;                                                             \
      a_iy##J = m < p.M ? oy * p.stride - 1 : -(1 << 20);                         \
      a_ix##J = ox * p.stride - 1;                                                \
    }                                                                             \
    const int n = n0 + r0 + 32 * J;                                               \
    w_off##J = n < p.N ? (unsigned)n * (unsigned)p.Ktot * 2u : OOB;               \
  }
  ROW_SETUP(0) ROW_SETUP(1) ROW_SETUP(2) ROW_SETUP(3)
#undef ROW_SETUP
  const int Hv = p.Hi << p.up, Wv = p.Wi << p.up;

  u32x4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define LOAD_ROW(J, tap, c, c_ok, dy, dx)                                                                  \
  {                                                                                                        \
    unsigned ao;                                                                                           \
    if (TAPS == 1) {                                                                                       \
      ao = (c_ok && a_off##J != OOB) ? a_off##J + (unsigned)(c)*2u : OOB;                                  \
    } else {                                                                                               \
      const int iy = a_iy##J + dy, ix = a_ix##J + dx;                                                      \
      const bool ok = c_ok && ((unsigned)iy < (unsigned)Hv) && ((unsigned)ix < (unsigned)Wv);              \
      const int sy = iy >> p.up, sx = ix >> p.up;                                                          \
      ao = ok ? ((unsigned)((a_img##J * p.Hi + sy) * p.Wi + sx) * (unsigned)p.lda + (unsigned)(c)) * 2u : OOB; \
    }                                                                                                      \
    ra##J = __builtin_amdgcn_raw_buffer_load_b128(rsrcA, ao, 0, 0);                                        \
    const unsigned wo = (c_ok && w_off##J != OOB) ? w_off##J + ((unsigned)(tap) * (unsigned)p.Cin + (unsigned)(c)) * 2u : OOB; \
    rb##J = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, wo, 0, 0);                                        \
  }
#define LOAD_TILE(ks)                                     \
  {                                                       \
    int tap = 0, kci = (ks);                              \
    if (TAPS != 1) {                                      \
      tap = (ks) / p.kc;                                  \
      kci = (ks)-tap * p.kc;                              \
    }                                                     \
    const int c = kci * BK + cc * 8;                      \
    const bool c_ok = c < p.Cin;                          \
    const int dy = tap / 3, dx = tap - dy * 3;            \
    LOAD_ROW(0, tap, c, c_ok, dy, dx)                     \
    LOAD_ROW(1, tap, c, c_ok, dy, dx)                     \
    LOAD_ROW(2, tap, c, c_ok, dy, dx)                     \
    LOAD_ROW(3, tap, c, c_ok, dy, dx)                     \
  }
#define STORE_TILE(buf)                                                     \
  {                                                                         \
    const int off = (buf)*BM * PITCH + r0 * PITCH + cc * 8;                 \
    *(u32x4*)(sA + off) = ra0;                                              \
    *(u32x4*)(sB + off) = rb0;                                              \
    *(u32x4*)(sA + off + 32 * PITCH) = ra1;                                 \
    *(u32x4*)(sB + off + 32 * PITCH) = rb1;                                 \
    *(u32x4*)(sA + off + 64 * PITCH) = ra2;                                 \
    *(u32x4*)(sB + off + 64 * PITCH) = rb2;                                 \
    *(u32x4*)(sA + off + 96 * PITCH) = ra3;                                 \
    *(u32x4*)(sB + off + 96 * PITCH) = rb3;                                 \
  }

  // ---- main loop -------------------------------------------------------------------------------
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, hi = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (ks_begin < ks_end) {
    LOAD_TILE(ks_begin)
    STORE_TILE(0)
  }
  __syncthreads();
  for (int ks = ks_begin; ks < ks_end; ++ks) {
    const int buf = (ks - ks_begin) & 1;
    const bool more = ks + 1 < ks_end;
    if (more) LOAD_TILE(ks + 1)  // global loads in flight under the MFMAs below
    const f16* bA = sA + buf * BM * PITCH + (wm * 64 + lr) * PITCH + hi * 8;
    const f16* bB = sB + buf * BM * PITCH + (wn * 64 + lr) * PITCH + hi * 8;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      f16x8 wf[2], xf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = *(const f16x8*)(bB + i * 32 * PITCH + kk * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j) xf[j] = *(const f16x8*)(bA + j * 32 * PITCH + kk * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    if (more) STORE_TILE(buf ^ 1)
    __syncthreads();
  }

#undef LOAD_ROW
#undef LOAD_TILE
#undef STORE_TILE
  // ---- stage the fp32 tile through LDS: sC[pixel][channel] -------------------------------------
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int pix = wm * 64 + j * 32 + lr;
        const int ch = wn * 64 + i * 32 + 8 * q + 4 * hi;
        f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        *(f32x4*)(sC + pix * CPITCH + ch) = v;
      }
  __syncthreads();

  if (p.splits > 1) {
    float* dst = p.partial + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int row = (t >> 4) + 16 * jj, c8 = (t & 15) * 8;
      const int m = m0 + row, n = n0 + c8;
      if (m < p.M && n < p.N) {
        const f32x4 v0 = *(const f32x4*)(sC + row * CPITCH + c8);
        const f32x4 v1 = *(const f32x4*)(sC + row * CPITCH + c8 + 4);
        *(f32x4*)(dst + (size_t)m * p.N + n) = v0;
        *(f32x4*)(dst + (size_t)m * p.N + n + 4) = v1;
      }
    }
    return;
  }

  if (p.epi & RCDM_EPI_GEGLU) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int row = (t >> 3) + 32 * jj, c8 = (t & 7) * 8;
      const int m = m0 + row;
      if (m < p.M) {
        float v[8], g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = sC[row * CPITCH + c8 + e];
          g[e] = sC[row * CPITCH + 64 + c8 + e];
        }
        epilogue_store(p, m, n0 + c8, v, g);
      }
    }
  } else {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int row = (t >> 4) + 16 * jj, c8 = (t & 15) * 8;
      const int m = m0 + row, n = n0 + c8;
      if (m < p.M && n < p.N) {
        float v[8], g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = sC[row * CPITCH + c8 + e];
          g[e] = 0.f;
        }
        epilogue_store(p, m, n, v, g);
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// LDS-DMA variant (the default): global -> LDS with buffer_load ... lds (no VGPR staging, no ds_write),
// STAGES-deep ring with counted s_waitcnt vmcnt + one raw s_barrier per k-step.
//   BM = 256: 8 waves (4 x 2), 3 stages x 48 KB, one block per CU  — the large-M levels
//   BM = 128: 4 waves (2 x 2), 2 stages x 32 KB, two blocks per CU — small M (+ split-K)
// The DMA writes LDS lane-linearly (wave-uniform base + lane*16), so rows are unpadded 128 B and the
// bank-conflict fix is an XOR swizzle applied on the SOURCE side: LDS slot (row r, 16-B chunk c') holds
// global chunk c' ^ ((r>>1)&7); a fragment read of chunk kc of row r reads slot kc ^ ((r>>1)&7).  For the
// 32x32x16 operand pattern (lanes 0-31 = rows, lane>>5 = chunk parity) every ds_read_b128 lane group then
// touches 16 distinct 16-B slots.
template <int TAPS, int BM, int STAGES>
__global__ __launch_bounds__(BM * 2, 2) void igemm_dma_kernel(const IgemmArgs p) {
  constexpr int NW = BM / 32;             // waves
  constexpr int NT = NW * 64;             // threads
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int AI = BM / 8 / NW;         // 1-KiB DMA pieces per wave per stage, activations (= 4)
  constexpr int BI = BN / 8 / NW;         // weights (2 or 4)
  constexpr int LPW = AI + BI;
  static_assert(STAGES == 2 || STAGES == 3, "ring depth");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sC = (float*)smem;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

  const int ntiles = p.tilesM * p.tilesN;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = ntiles >> 3, r = ntiles & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_n = bid % p.tilesN, tile_m = bid / p.tilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int ks_begin = blockIdx.y * p.nk_per_split;
  const int ks_end = min(p.nk, ks_begin + p.nk_per_split);
  const int nkl = ks_end - ks_begin;

  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7FFFFFFF, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const int Hv = p.Hi << p.up, Wv = p.Wi << p.up;

  // ---- per-lane loader state (static-indexed arrays: fully unrolled) -----------------------------
  const int lrow = lane >> 3, lch = lane & 7;
  unsigned a_off[AI], w_off[BI];
  int a_img[AI], a_iy[AI], a_ix[AI], a_c[AI], w_c[BI];
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
      a_iy[i] = m < p.M ? oy * p.stride - 1 : -(1 << 20);
      a_ix[i] = ox * p.stride - 1;
    }
  }
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int row = (wave * BI + i) * 8 + lrow;
    const int n = n0 + row;
    w_c[i] = (lch ^ ((row >> 1) & 7)) * 8;
    w_off[i] = n < p.N ? (unsigned)n * (unsigned)p.Ktot * 2u : OOB;
  }

  auto issue = [&](int ks, int stage) __attribute__((always_inline)) {
    int tap = 0, kci = ks;
    if (TAPS != 1) {
      tap = ks / p.kc;
      kci = ks - tap * p.kc;
    }
    const int c0 = kci * BK;
    const int dy = tap / 3, dx = tap - dy * 3;
    char* sbase = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int c = c0 + a_c[i];
      unsigned vo;
      if (TAPS == 1) {
        vo = (c < p.Cin && a_off[i] != OOB) ? a_off[i] + (unsigned)c * 2u : OOB;
      } else {
        const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
        const bool ok = (c < p.Cin) && ((unsigned)iy < (unsigned)Hv) && ((unsigned)ix < (unsigned)Wv);
        const int sy = iy >> p.up, sx = ix >> p.up;
        vo = ok ? ((unsigned)((a_img[i] * p.Hi + sy) * p.Wi + sx) * (unsigned)p.lda + (unsigned)c) * 2u : OOB;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsrcA, (__attribute__((address_space(3))) void*)(sbase + (wave * AI + i) * 1024), 16, vo, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int c = c0 + w_c[i];
      const unsigned vo = (c < p.Cin && w_off[i] != OOB)
                              ? w_off[i] + ((unsigned)tap * (unsigned)p.Cin + (unsigned)c) * 2u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsrcW, (__attribute__((address_space(3))) void*)(sbase + A_BYTES + (wave * BI + i) * 1024), 16, vo, 0, 0, 0);
    }
  };

  // ---- compute mapping ---------------------------------------------------------------------------
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, hi = lane >> 5;
  const int sw = (lr >> 1) & 7;
  int koff[4];  // byte offset of k-chunk (kk*2 + hi) inside a swizzled 128-B row
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) koff[kk] = (((kk * 2 + hi) ^ sw) << 4);
  const int rowA = (wm * 64 + lr) * 128, rowB = A_BYTES + (wn * 64 + lr) * 128;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#pragma unroll
  for (int s0 = 0; s0 < STAGES - 1; ++s0)
    if (s0 < nkl) issue(ks_begin + s0, s0);

  int stage = 0;
  for (int it = 0; it < nkl; ++it) {
    // retire this wave's DMA pieces of stage `it`, then make every wave's pieces visible
    if (STAGES == 3 && it + 1 < nkl) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    // refill the stage everybody finished reading one iteration ago
    if (it + STAGES - 1 < nkl) {
      int st2 = stage + STAGES - 1;
      if (st2 >= STAGES) st2 -= STAGES;
      issue(ks_begin + it + STAGES - 1, st2);
    }
    const char* sb = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f16x8 wf[2], xf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = *(const f16x8*)(sb + rowB + i * 32 * 128 + koff[kk]);
#pragma unroll
      for (int j = 0; j < 2; ++j) xf[j] = *(const f16x8*)(sb + rowA + j * 32 * 128 + koff[kk]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    stage = stage + 1 == STAGES ? 0 : stage + 1;
  }
  __syncthreads();  // all operand reads done (no DMA outstanding): the ring becomes the fp32 staging tile

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int pix = wm * 64 + j * 32 + lr;
        const int ch = wn * 64 + i * 32 + 8 * q + 4 * hi;
        f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        *(f32x4*)(sC + pix * CPITCH + ch) = v;
      }
  __syncthreads();

  if (p.splits > 1) {
    float* dst = p.partial + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int jj = 0; jj < BM * 16 / NT; ++jj) {
      const int row = (t >> 4) + (NT / 16) * jj, c8 = (t & 15) * 8;
      const int m = m0 + row, n = n0 + c8;
      if (m < p.M && n < p.N) {
        const f32x4 v0 = *(const f32x4*)(sC + row * CPITCH + c8);
        const f32x4 v1 = *(const f32x4*)(sC + row * CPITCH + c8 + 4);
        *(f32x4*)(dst + (size_t)m * p.N + n) = v0;
        *(f32x4*)(dst + (size_t)m * p.N + n + 4) = v1;
      }
    }
    return;
  }
  if (p.epi & RCDM_EPI_GEGLU) {
#pragma unroll
    for (int jj = 0; jj < BM * 8 / NT; ++jj) {
      const int row = (t >> 3) + (NT / 8) * jj, c8 = (t & 7) * 8;
      const int m = m0 + row;
      if (m < p.M) {
        float v[8], g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = sC[row * CPITCH + c8 + e];
          g[e] = sC[row * CPITCH + 64 + c8 + e];
        }
        epilogue_store(p, m, n0 + c8, v, g);
      }
    }
  } else {
#pragma unroll
    for (int jj = 0; jj < BM * 16 / NT; ++jj) {
      const int row = (t >> 4) + (NT / 16) * jj, c8 = (t & 15) * 8;
      const int m = m0 + row, n = n0 + c8;
      if (m < p.M && n < p.N) {
        float v[8], g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = sC[row * CPITCH + c8 + e];
          g[e] = 0.f;
        }
        epilogue_store(p, m, n, v, g);
      }
    }
  }
}

// split-K second pass: fixed-order sum of the fp32 slabs + the same epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const IgemmArgs p) {
  const bool geglu = (p.epi & RCDM_EPI_GEGLU) != 0;
  const int nout8 = (geglu ? p.N / 2 : p.N) / 8;
  const size_t total = (size_t)p.M * nout8;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(idx / nout8);
    const int oc = (int)(idx - (size_t)m * nout8) * 8;
    const int n = geglu ? (oc >> 6) * 128 + (oc & 63) : oc;
    float v[8], g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = g[e] = 0.f;
    for (int s = 0; s < p.splits; ++s) {
      const float* src = p.partial + ((size_t)s * p.M + m) * p.N + n;
      const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += a[e];
        v[4 + e] += b[e];
      }
      if (geglu) {
        const f32x4 c = *(const f32x4*)(src + 64), d = *(const f32x4*)(src + 68);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          g[e] += c[e];
          g[4 + e] += d[e];
        }
      }
    }
    epilogue_store(p, m, n, v, g);
  }
}

int plan_splits(int tilesM, int tilesN, int nk, int requested) {
  if (requested == 1) return 1;
  if (requested > 1) return requested < nk ? requested : nk;
  const int tiles = tilesM * tilesN;
  int s = 1;
  if (tiles < 384 && nk >= 8) {
    s = (512 + tiles - 1) / tiles;
    if (s > nk / 4) s = nk / 4;
    if (s > 16) s = 16;
    if (s < 1) s = 1;
  }
  return s;
}

// variant: 0 = register-staged 128x128 (legacy, RCDM_IGEMM=legacy), 1 = DMA 128x128x2-stage, 2 = DMA 256x128x3-stage
int g_force_variant = -1;

int pick_variant(const IgemmArgs& a) {
  if (g_force_variant < 0) {
    const char* e = getenv("RCDM_IGEMM");
    g_force_variant = 99;
    if (e && !strcmp(e, "legacy")) g_force_variant = 0;
    if (e && !strcmp(e, "dma128")) g_force_variant = 1;
    if (e && !strcmp(e, "dma256")) g_force_variant = 2;
  }
  if (g_force_variant != 99) return g_force_variant;
  const long tiles256 = (long)((a.M + 255) / 256) * ((a.N + BN - 1) / BN);
  return tiles256 >= 192 ? 2 : 1;
}

int fill_common(IgemmArgs& a, int requested_split, int* variant_out = nullptr) {
  const int variant = pick_variant(a);
  if (variant_out) *variant_out = variant;
  const int bm = variant == 2 ? 256 : 128;
  a.tilesM = (a.M + bm - 1) / bm;
  a.tilesN = (a.N + BN - 1) / BN;
  a.kc = (a.Cin + BK - 1) / BK;
  const int taps = a.Ktot / a.Cin;
  a.nk = taps * a.kc;
  int s = variant == 2 ? (requested_split > 1 ? requested_split : 1) : plan_splits(a.tilesM, a.tilesN, a.nk, requested_split);
  if (s > a.nk) s = a.nk;
  a.nk_per_split = (a.nk + s - 1) / s;
  a.splits = (a.nk + a.nk_per_split - 1) / a.nk_per_split;
  return RCDM_OK;
}

int check_common(const IgemmArgs& a) {
  if (!a.A || !a.W || !a.out) return RCDM_EINVAL;
  if (a.M <= 0 || a.N <= 0 || a.Cin <= 0) return RCDM_EINVAL;
  if ((a.Cin & 7) || (a.N & 7) || (a.lda & 7) || (a.ldc & 7)) return RCDM_ESHAPE;
  if ((a.epi & RCDM_EPI_BIAS) && !a.bias) return RCDM_EINVAL;
  if ((a.epi & RCDM_EPI_ROWVEC) && (!a.rowvec || a.rows_per_sample <= 0)) return RCDM_EINVAL;
  if ((a.epi & RCDM_EPI_RESIDUAL) && (!a.res || (a.ldr & 7))) return RCDM_EINVAL;
  if ((a.epi & RCDM_EPI_GEGLU) && (a.N % 128)) return RCDM_ESHAPE;
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
  constexpr int LDS_DMA128 = 2 * (128 * 128 + BN * 128) > 128 * CPITCH * 4 ? 2 * (128 * 128 + BN * 128) : 128 * CPITCH * 4;
  constexpr int LDS_DMA256 = 3 * (256 * 128 + BN * 128);
  static bool attr_set = false;
  if (!attr_set) {
    set_lds(igemm_kernel<TAPS>, LDS_BYTES);
    set_lds(igemm_dma_kernel<TAPS, 128, 2>, LDS_DMA128);
    set_lds(igemm_dma_kernel<TAPS, 256, 3>, LDS_DMA256);
    attr_set = true;
  }
  if (a.splits > 1) {
    const size_t need = (size_t)a.splits * a.M * a.N * sizeof(float);
    if (!workspace || workspace_bytes < need) return RCDM_EWORKSPACE;
    a.partial = (float*)workspace;
  } else {
    a.partial = nullptr;
  }
  dim3 grid(a.tilesM * a.tilesN, a.splits);
  if (variant == 2)
    hipLaunchKernelGGL((igemm_dma_kernel<TAPS, 256, 3>), grid, dim3(512), LDS_DMA256, stream, a);
  else if (variant == 1)
    hipLaunchKernelGGL((igemm_dma_kernel<TAPS, 128, 2>), grid, dim3(256), LDS_DMA128, stream, a);
  else
    hipLaunchKernelGGL(igemm_kernel<TAPS>, grid, dim3(NTHREADS), LDS_BYTES, stream, a);
  int rc = rcdm_check_launch();
  if (rc) return rc;
  if (a.splits > 1) {
    const bool geglu = (a.epi & RCDM_EPI_GEGLU) != 0;
    const size_t total = (size_t)a.M * ((geglu ? a.N / 2 : a.N) / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, a);
    rc = rcdm_check_launch();
  }
  return rc;
}

void from_gemm(const rcdm_gemm_desc* d, IgemmArgs& a) {
  a.M = d->M; a.N = d->N; a.Cin = d->K; a.Ktot = d->K;
  a.Hi = a.Wi = a.Ho = a.Wo = 1; a.stride = 1; a.up = 0;
  a.lda = d->lda; a.ldc = d->ldc; a.ldr = d->ldr; a.ldt = d->ldt;
  a.rows_per_sample = d->rows_per_sample; a.epi = d->epilogue; a.out_scale = d->out_scale;
}

int from_conv(const rcdm_conv3x3_desc* d, IgemmArgs& a) {
  if (d->stride != 1 && d->stride != 2) return RCDM_ESHAPE;
  if (d->upsample != 0 && d->upsample != 1) return RCDM_ESHAPE;
  if (d->n_img <= 0 || d->h_in <= 0 || d->w_in <= 0) return RCDM_EINVAL;
  const int hv = d->h_in << d->upsample, wv = d->w_in << d->upsample;
  a.Ho = (hv - 1) / d->stride + 1;
  a.Wo = (wv - 1) / d->stride + 1;
  a.Hi = d->h_in; a.Wi = d->w_in; a.stride = d->stride; a.up = d->upsample;
  a.M = d->n_img * a.Ho * a.Wo; a.N = d->c_out; a.Cin = d->c_in; a.Ktot = 9 * d->c_in;
  a.lda = d->lda; a.ldc = d->ldc; a.ldr = d->ldr; a.ldt = d->ldt;
  a.rows_per_sample = d->rows_per_sample; a.epi = d->epilogue; a.out_scale = d->out_scale;
  return RCDM_OK;
}

}  // namespace

extern "C" {

int rcdm_set_igemm_variant(int32_t v) {
  if (v < -1 || v > 2) return RCDM_EINVAL;
  g_force_variant = v < 0 ? 99 : v;
  return RCDM_OK;
}

size_t rcdm_gemm_workspace_bytes(const rcdm_gemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
  IgemmArgs a{};
  from_gemm(d, a);
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

size_t rcdm_conv3x3_workspace_bytes(const rcdm_conv3x3_desc* d) {
  if (!d) return 0;
  IgemmArgs a{};
  if (from_conv(d, a) || a.Cin <= 0 || a.N <= 0) return 0;
  fill_common(a, d->split_k);
  return a.splits > 1 ? (size_t)a.splits * a.M * a.N * sizeof(float) : 0;
}

int rcdm_conv3x3(const rcdm_conv3x3_desc* d, const void* in, const void* W, const float* bias, const float* rowvec,
                 const void* residual, void* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d) return RCDM_EINVAL;
  IgemmArgs a{};
  int rc = from_conv(d, a);
  if (rc) return rc;
  a.A = (const f16*)in; a.W = (const f16*)W; a.bias = bias; a.rowvec = rowvec;
  a.res = (const f16*)residual; a.out = (f16*)out;
  rc = check_common(a);
  if (rc) return rc;
  if (a.epi & RCDM_EPI_GEGLU) return RCDM_ESHAPE;
  int variant = 0;
  fill_common(a, d->split_k, &variant);
  return launch<9>(a, variant, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
