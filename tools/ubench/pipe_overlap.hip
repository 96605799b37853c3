// pipe_overlap.hip — do the matrix pipe and the vector ALU of one gfx950 SIMD run at the same time?  One block, W waves
// per SIMD; waves with (wave / 4) % 2 == 0 run an independent-accumulator MFMA stream, the others a VALU stream
// (v_fma_f32 or v_exp_f32); each wave reports its own s_memtime span.  Modes: MFMA alone, VALU alone, both.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP8(x) x x x x x x x x
template <int MF, int VK, int PRIO = 0>   // PRIO: s_setprio of the VALU waves;  MF: 0 none, 1 = 32x32x16, 2 = 16x16x32 ;  VK: 0 none, 1 = v_fma_f32, 2 = v_exp_f32
__global__ void k(long long* out, float seed, int mfma_waves_mask) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = MF && ((mfma_waves_mask >> (wave / 4)) & 1);
  long long t0 = __builtin_amdgcn_s_memtime();
  if (do_mfma) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(seed + e); b[e] = (_Float16)(seed - e); }
    if (MF == 1) {
      f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
      for (int it = 0; it < 512; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
      }
      if (c0[0] + c1[0] + c2[0] + c3[0] == 1.2345e30f) out[0] = 0;
    } else {
      f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
      for (int it = 0; it < 512; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c7, 0, 0, 0);
      }
      if (c0[0] + c1[0] + c2[0] + c3[0] + c4[0] + c5[0] + c6[0] + c7[0] == 1.2345e30f) out[0] = 0;
    }
  } else if (VK) {
    if (PRIO == 3) __builtin_amdgcn_s_setprio(3);
    if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int it = 0; it < 64; ++it) {
      if (VK == 1) { REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
      if (VK == 2) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345e30f) out[0] = 0;
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[1 + wave] = t1 - t0;
}
template <int MF, int VK, int PRIO = 0> void run(const char* name, long long* d, int groups, int mask) {
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MF, VK, PRIO>), dim3(1), dim3(256 * groups), 0, 0, d, 1.0f, mask);
  hipDeviceSynchronize();
  long long h[17]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("%-44s", name);
  const int n_mfma = MF == 1 ? 2048 : 4096, n_valu = 64 * 64;
  for (int g = 0; g < groups; ++g) {
    const bool m = MF && ((mask >> g) & 1);
    if (!m && !VK) continue;
    printf("  [%s wave: %.1f clk/instr]", m ? "MFMA" : "VALU", (double)h[1 + 4 * g] / (m ? n_mfma : n_valu));
  }
  printf("\n");
}
int main() {
  long long* d; hipMalloc(&d, 1024);
  run<1, 0>("32x32x16 alone, 1 wave/SIMD", d, 1, 1);
  run<1, 0>("32x32x16 alone, 2 waves/SIMD", d, 2, 3);
  run<2, 0>("16x16x32 alone, 1 wave/SIMD", d, 1, 1);
  run<2, 0>("16x16x32 alone, 2 waves/SIMD", d, 2, 3);
  run<0, 1>("v_fma alone, 1 wave/SIMD", d, 1, 0);
  run<0, 1>("v_fma alone, 2 waves/SIMD", d, 2, 0);
  run<0, 1>("v_fma alone, 3 waves/SIMD", d, 3, 0);
  run<0, 1>("v_fma alone, 4 waves/SIMD", d, 4, 0);
  run<0, 2>("v_exp alone, 3 waves/SIMD", d, 3, 0);
  run<0, 2>("v_exp alone, 4 waves/SIMD", d, 4, 0);
  run<1, 1>("32x32x16 wave + v_fma wave", d, 2, 1);
  run<1, 2>("32x32x16 wave + v_exp wave", d, 2, 1);
  run<2, 1>("16x16x32 wave + v_fma wave", d, 2, 1);
  run<1, 1>("32x32x16 wave + 2 v_fma waves", d, 3, 1);
  run<1, 1>("2 x 32x32x16 waves + v_fma wave", d, 3, 3);
  run<1, 2>("2 x 32x32x16 waves + v_exp wave", d, 3, 3);
  run<1, 1, 3>("2 x 32x32x16 waves + v_fma wave at prio 3", d, 3, 3);
  run<1, 2, 3>("2 x 32x32x16 waves + v_exp wave at prio 3", d, 3, 3);
  run<1, 1, 3>("32x32x16 wave + 2 v_fma waves at prio 3", d, 3, 1);
  run<1, 1, 3>("VALU wave FIRST (prio 3) + 2 MFMA waves (mask 6)", d, 3, 6);
  run<1, 1, 0>("VALU wave FIRST (prio 0) + 2 MFMA waves (mask 6)", d, 3, 6);
  run<2, 1, 0>("VALU wave FIRST + 2 x 16x16x32 waves (mask 6)", d, 3, 6);
  run<2, 1, 3>("VALU wave FIRST (prio 3) + 2 x 16x16x32 waves", d, 3, 6);
  return 0;
}
