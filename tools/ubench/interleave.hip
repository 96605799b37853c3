// interleave.hip — a flash-attention-like instruction mix per "tile" (14 x v_mfma_f32_32x32x16_f16, 62 plain VALU, 32 v_exp_f32)
// on W waves per SIMD, (a) in phases: 6 MFMA | all VALU | 8 MFMA, (b) interleaved: every MFMA followed by its share of the VALU
// work.  Reports clocks per tile per SIMD.  Waves are started with different phase offsets (wave w skips w/W of a tile of VALU).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define FMA4 asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
#define FMA2 asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1" : "+v"(a4), "+v"(a5));
#define EXP2 asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(a6), "+v"(a7));
#define EXP4 asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9));
#define MF(c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); __builtin_amdgcn_sched_barrier(0);
template <int MODE, int PRIO>
__global__ void k(long long* out, float seed, int tiles) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 * 1e-3f, a7 = a6 + 1e-3f, a8 = a7, a9 = a6;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(seed + e); b[e] = (_Float16)(seed - e); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  const int wave = threadIdx.x >> 6, grp = wave >> 2;
  // de-phase the waves of a SIMD
  for (int i = 0; i < grp * 40; ++i) { FMA4 }
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < tiles; ++it) {
    if (MODE == 0) {   // phases
      MF(c0) MF(c1) MF(c0) MF(c1) MF(c0) MF(c1)
      if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
      for (int r = 0; r < 8; ++r) { FMA4 FMA2 EXP4 __builtin_amdgcn_sched_barrier(0); }   // 48 fma + 32 exp
      FMA4 FMA4 FMA4 FMA2     // + 14 fma = 62
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      MF(c2) MF(c3) MF(c2) MF(c3) MF(c2) MF(c3) MF(c2) MF(c3)
    } else {           // interleaved: 14 x [MFMA, 4-5 fma, 2-3 exp]
#define STEP(c) MF(c) FMA4 EXP2 __builtin_amdgcn_sched_barrier(0);
      STEP(c0) STEP(c1) STEP(c0) STEP(c1) STEP(c0) STEP(c1)
      STEP(c2) STEP(c3) STEP(c2) STEP(c3) STEP(c2) STEP(c3) STEP(c2) STEP(c3)
      FMA4 FMA2 EXP4   // 14*4 + 6 = 62 fma, 14*2 + 4 = 32 exp
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + c0[0] + c1[0] + c2[0] + c3[0] == 1.2345e30f) out[0] = 0;
}
template <int MODE, int PRIO> void run(const char* name, long long* d, int W) {
  const int tiles = 400;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE, PRIO>), dim3(1), dim3(256 * W), 0, 0, d, 1.0f, tiles);
  hipDeviceSynchronize();
  long long h[16]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  long long mx = 0; for (int g = 0; g < W; ++g) if (h[4 * g] > mx) mx = h[4 * g];
  printf("%-34s %d waves/SIMD: %7.1f clocks per tile per wave, %7.1f per tile per SIMD\n", name, W, (double)mx / tiles, (double)mx / tiles / W);
}
int main() {
  long long* d; hipMalloc(&d, 1024);
  for (int W = 1; W <= 4; ++W) {
    run<0, 0>("phases", d, W); run<0, 3>("phases, VALU at prio 3", d, W); run<1, 0>("interleaved", d, W);
  }
  return 0;
}
