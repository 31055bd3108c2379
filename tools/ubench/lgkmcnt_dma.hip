#include <hip/hip_runtime.h>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
#ifndef DMA
#define DMA 0
#endif
__global__ void k(const unsigned* __restrict__ g, float* out, int n) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  const int lane = threadIdx.x & 63;
  f32x4 acc = {0, 0, 0, 0};
  u32x4 b = {1, 2, 3, 4};
  for (int it = 0; it < n; ++it) {
    const unsigned* buf = lds + (it & 1) * 8192;
#if DMA
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) char*)g + lane * 16 + it * 1024,
                                     (__attribute__((address_space(3))) void*)(lds + ((it + 1) & 1) * 8192), 16, 0, 0);
#endif
    u32x4 A[3];
    A[0] = *(const u32x4*)(buf + lane * 4);
    A[1] = *(const u32x4*)(buf + 256 + lane * 4);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u + 2 < 8) A[(u + 2) % 3] = *(const u32x4*)(buf + (u + 2) * 256 + lane * 4);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, A[u % 3]), __builtin_bit_cast(h8, b), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, A[u % 3]), __builtin_bit_cast(h8, b), acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#if DMA
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
  }
  out[threadIdx.x] = acc[0] + acc[1];
}
