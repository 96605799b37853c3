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

// v_rcp_f32 (1 ulp) instead of an IEEE division: `a / b` and __frcp_rn expand to ~12 VALU ops on gfx950
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, i.e. at fp32 round-off level and three orders of
// magnitude below the f16 output rounding): 1 rcp + 1 exp + 7 fma instead of libm erff's ~40 VALU ops, which made
// the GEGLU epilogue VALU-bound.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}
// exact-form (erf) GELU, as diffusers GEGLU.gelu -> F.gelu(approximate="none")
__device__ __forceinline__ float gelu_f(float x) {
#ifdef RCDM_LIBM_ERF
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
#else
  return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f));
#endif
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
