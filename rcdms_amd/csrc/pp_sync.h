// pp_sync.h — wait / barrier helpers shared by the phase-structured kernels (igemm8.hip, rowpanel.hip).
#pragma once
#include "common.h"

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// one tick boundary: nothing (ds_read, DMA issue, MFMA) may be scheduled across it
__device__ __forceinline__ void tick_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
// s_waitcnt vmcnt(n) for a wave-uniform runtime n in [0, 12]
__device__ __forceinline__ void wait_vm_n(int n) {
  switch (n) {
    case 0: wait_vm<0>(); break;
    case 1: wait_vm<1>(); break;
    case 2: wait_vm<2>(); break;
    case 3: wait_vm<3>(); break;
    case 4: wait_vm<4>(); break;
    case 5: wait_vm<5>(); break;
    case 6: wait_vm<6>(); break;
    case 7: wait_vm<7>(); break;
    case 8: wait_vm<8>(); break;
    case 9: wait_vm<9>(); break;
    case 10: wait_vm<10>(); break;
    case 11: wait_vm<11>(); break;
    default: wait_vm<12>(); break;
  }
}
