// common.h — shared types/helpers for the gfx950 kernels of librcdm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rcdm.h"

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

union Pack16 {  // one 16-byte global/LDS transaction seen as 8 halves
  uint4 u;
  u32x4 v;
  f16x8 h;
  f16 e[8];
};

extern thread_local int g_rcdm_last_hip_error;

static inline int rcdm_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_rcdm_last_hip_error = (int)e;
    return RCDM_ELAUNCH;
  }
  return RCDM_OK;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is PER DEVICE: a process that drives a second GPU must set it there too.
// true the first time it is called on the current device for `done` (a benign race: setting it twice is harmless).
static inline bool rcdm_first_on_device(bool (&done)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  if (done[dev]) return false;
  done[dev] = true;
  return true;
}

// v_rcp_f32 (1 ulp) instead of an IEEE division: `a / b` and __frcp_rn expand to ~12 VALU ops on gfx950
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// exact-form (erf) GELU, as diffusers GEGLU.gelu -> F.gelu(approximate="none"): x * Phi(x) with erf by Abramowitz &
// Stegun 7.1.25 (three terms, |error of erf| <= 2.5e-5, so |error of the GELU| <= 1.25e-5 |x|: a fortieth of the f16 rounding
// of the value it is about to be stored as; round 4 — rounds 1-3 used the five-term 7.1.26 at 1.5e-7, two more packed FMAs per
// pair; libm erff's ~40 VALU ops made the GEGLU epilogue VALU-bound).  With z = |x| / sqrt(2), t = 1 / (1 + p z),
// q = (1 - erf(z)) / 2 = poly(t) * exp(-z^2) / 2:   gelu(x) = max(x, 0) - |x| * q
// (sqrt(2) and the 1/2 are folded into the constants).  Written on pairs: v_pk_fma_f32 / v_pk_mul_f32 do two fp32
// lanes per instruction, so a pair costs 4 packed ops + 2 rcp + 2 exp + 2 max + 3 mul; the epilogues of the K = 320 GEMMs
// (50 GELUs per thread against 250 MFMAs per wave) are VALU-bound, DESIGN.md section 4a.  -DRCDM_GELU_5TERM: the old form.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float a) { return f32x2{a, a}; }
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
#ifdef RCDM_GELU_ABLATE   // timing bound of a free GELU (wrong results)
  return x * splat2(0.5f);
#endif
#ifdef RCDM_LIBM_ERF
  return f32x2{0.5f * x.x * (1.0f + erff(x.x * 0.70710678118654752f)), 0.5f * x.y * (1.0f + erff(x.y * 0.70710678118654752f))};
#else
  const f32x2 ax = __builtin_elementwise_abs(x);
#ifdef RCDM_GELU_5TERM
  const f32x2 d = __builtin_elementwise_fma(ax, splat2(0.2316418882f), splat2(1.0f));  // 0.3275911 / sqrt(2)
  const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  f32x2 p = __builtin_elementwise_fma(t, splat2(0.5307027145f), splat2(-0.7265760135f));  // A&S 7.1.26 coefficients / 2
  p = __builtin_elementwise_fma(p, t, splat2(0.7107068705f));
  p = __builtin_elementwise_fma(p, t, splat2(-0.142248368f));
  p = __builtin_elementwise_fma(p, t, splat2(0.127414796f));
#else
  const f32x2 d = __builtin_elementwise_fma(ax, splat2(0.3326725f), splat2(1.0f));  // 0.47047 / sqrt(2)
  const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  f32x2 p = __builtin_elementwise_fma(t, splat2(0.3739278f), splat2(-0.0479399f));  // A&S 7.1.25 coefficients / 2
  p = __builtin_elementwise_fma(p, t, splat2(0.1740121f));
#endif
  const f32x2 xx = x * x * splat2(-0.72134752044f);  // -z^2 * log2(e)
  const f32x2 e = {__builtin_amdgcn_exp2f(xx.x), __builtin_amdgcn_exp2f(xx.y)};
  const f32x2 q = p * t * e;
  return __builtin_elementwise_fma(-ax, q, __builtin_elementwise_max(x, splat2(0.0f)));
#endif
}
__device__ __forceinline__ float gelu_f(float x) { return gelu2(splat2(x)).x; }
__device__ __forceinline__ void gelu8(float (&v)[8]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x2 r = gelu2(f32x2{v[2 * k], v[2 * k + 1]});
    v[2 * k] = r.x;
    v[2 * k + 1] = r.y;
  }
}
// GEGLU on eight packed halfs: (h + bh) * gelu(g + bg) * sc -> eight halfs
__device__ __forceinline__ uint4 geglu8(uint4 h, uint4 g, const float (&bh)[8], const float (&bg)[8], float sc) {
  union P { uint4 u; f16 e[8]; } hh, gg, o;
  hh.u = h;
  gg.u = g;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x2 hv = f32x2{(float)hh.e[2 * k], (float)hh.e[2 * k + 1]} + f32x2{bh[2 * k], bh[2 * k + 1]};
    const f32x2 gv = f32x2{(float)gg.e[2 * k], (float)gg.e[2 * k + 1]} + f32x2{bg[2 * k], bg[2 * k + 1]};
    const f32x2 r = hv * gelu2(gv) * splat2(sc);
    o.e[2 * k] = (f16)r.x;
    o.e[2 * k + 1] = (f16)r.y;
  }
  return o.u;
}

// fma(f16 half `sel` of the dword h, s, c) in fp32: v_fma_mix_f32 converts the f16 source on the fly
__device__ __forceinline__ float mix_f16_f32(unsigned h, int sel, float s, float c) {
  float t;
  if (sel) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t) : "v"(h), "v"(s), "v"(c));
  else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(t) : "v"(h), "v"(s), "v"(c));
  return t;
}
// {f16(fma(r.lo, s, t0)), f16(fma(r.hi, s, t1))}: fp32 fma of the f16 halves of r, one rounding each, packed
__device__ __forceinline__ unsigned mix_f16_pack(unsigned r, float s, float t0, float t1) {
  unsigned o;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(o) : "v"(r), "v"(s), "v"(t0), "v"(t1));
  return o;
}

// sum over groups of LPR consecutive lanes (LPR a power of two, 8..64), result in every lane of the group: DPP steps
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: no LDS crossbar), ds_bpermute only across 16-lane rows
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);  // row_half_mirror: quads are uniform by now, so the mirrored partner is the other quad's sum
  if constexpr (LPR >= 16) v += dpp_f32<0x140>(v);  // row_mirror: the other 8-lane group of the 16-lane row
  if constexpr (LPR >= 32) v += __shfl_xor(v, 16, 64);
  if constexpr (LPR >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// Deferred LayerNorm, consumer side: TPR consecutive lanes (1, 2 or 4) share a row and split its <= MAXP partial
// (sum, sum of squares) slots; finish() gives (rstd, mean * rstd) of the row in every lane of the group.
template <int TPR, int MAXP>
struct LnxRow {
  static constexpr int NL = (MAXP + TPR - 1) / TPR;
  f32x2 v[NL];
  // layout: slot-major, stat[slot][ld rows] of (sum, sum of squares) — the lanes of a load instruction (consecutive rows of
  // one slot) read consecutive 8-byte pairs.  Loads go through a buffer descriptor of exactly parts * ld pairs: a slot past
  // `parts` (and a dead row, sent to offset 2^31) reads as zero in hardware, so there is neither a branch per element (which
  // hipcc turns into a wait per element) nor a clamp / select / 64-bit address per element — issuing the ten loads of the
  // first version cost a wave 1270 ticks at kernel start (tools/trace_lnx.py), a buffer load is one VALU add.
  __device__ __forceinline__ void load(const float* stat, int ld, int m, bool live, int parts, int sub) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)stat, 0, parts * ld * 8, 0x00020000);
    const unsigned stride = (unsigned)(TPR * ld) * 8u;
    unsigned off = live ? (unsigned)(sub * ld + m) * 8u : 0x80000000u;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const u32x2_t r = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
      v[j] = f32x2{__uint_as_float(r.x), __uint_as_float(r.y)};
      off += stride;
    }
  }
  __device__ __forceinline__ void finish(float invC, float eps, float& rstd, float& mr) const {
    f32x2 s = v[0];
#pragma unroll
    for (int j = 1; j < NL; ++j) s += v[j];
    if constexpr (TPR >= 2) {
      s.x += dpp_f32<0xB1>(s.x);
      s.y += dpp_f32<0xB1>(s.y);
    }
    if constexpr (TPR >= 4) {
      s.x += dpp_f32<0x4E>(s.x);
      s.y += dpp_f32<0x4E>(s.y);
    }
    const float mean = s.x * invC;
    float var = __builtin_fmaf(-mean, mean, s.y * invC);
    if (var < 0.f) var = 0.f;
    rstd = rsqrtf(var + eps);
    mr = mean * rstd;
  }
};
// (rstd, mean rstd) of row m with the load count sized to the slot count: <= 8 slots (C <= 1024 behind 128-wide producer
// tiles) take the short form
template <int TPR, int MAXP>
__device__ __forceinline__ void lnx_row(const float* stat, int ld, int m, bool live, int parts, int sub, float invC, float eps,
                                        float& rstd, float& mr) {
  if (parts <= 8) {   // wave-uniform
    LnxRow<TPR, 8> r;
    r.load(stat, ld, m, live, parts, sub);
    r.finish(invC, eps, rstd, mr);
  } else {
    LnxRow<TPR, MAXP> r;
    r.load(stat, ld, m, live, parts, sub);
    r.finish(invC, eps, rstd, mr);
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
