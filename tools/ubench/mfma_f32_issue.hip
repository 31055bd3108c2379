// Microbenchmark: does VALU work hide in the shadow of v_mfma_f32_16x16x4_f32 on gfx950?
// One wave; s_memtime around an unrolled loop of MFMAs with K VALU ops interleaved per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int NACC, int NVALU, int TRANS>
__global__ void __launch_bounds__(64) k(float* out, unsigned long long* cyc, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
  float av[4], bv[4];   // distinct operands per accumulator so the chains cannot be CSE'd together
  for (int i = 0; i < 4; ++i) { av[i] = a * (i + 1) + threadIdx.x; bv[i] = b - i * 0.25f + threadIdx.x; }
  unsigned long long t0, t1;
  FENCE();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  FENCE();
  for (int it = 0; it < 16; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m % NACC], bv[m % NACC], acc[m % NACC], 0, 0, 0);
      FENCE();
#pragma unroll
      for (int q = 0; q < NVALU; ++q) {
        if (TRANS) v[q % 8] = __builtin_amdgcn_exp2f(v[q % 8]);
        else v[q % 8] = __builtin_fmaf(v[q % 8], a, b);
        FENCE();
      }
    }
  }
  FENCE();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  FENCE();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, int NVALU, int TRANS>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 8);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<NACC, NVALU, TRANS>), dim3(1), dim3(64), 0, 0, out, cyc, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-34s  %6.1f cycles per MFMA (256 MFMAs, %d VALU each)\n", name, c / 256.0, NVALU);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<4, 0, 0>("4 acc, no VALU");
  run<2, 0, 0>("2 acc, no VALU");
  run<1, 0, 0>("1 acc, no VALU");
  run<4, 1, 0>("4 acc, 1 fma");
  run<4, 2, 0>("4 acc, 2 fma");
  run<4, 4, 0>("4 acc, 4 fma");
  run<4, 6, 0>("4 acc, 6 fma");
  run<4, 8, 0>("4 acc, 8 fma");
  run<4, 1, 1>("4 acc, 1 exp");
  run<4, 2, 1>("4 acc, 2 exp");
  run<4, 3, 1>("4 acc, 3 exp");
  run<2, 2, 0>("2 acc, 2 fma");
  run<2, 4, 0>("2 acc, 4 fma");
  return 0;
}
