// Microbenchmark: VALU co-issue under v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16 (one wave, gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using i32x4 = __attribute__((ext_vector_type(4))) int;
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int BIG, int NVALU, int TRANS>
__global__ void __launch_bounds__(64) k(float* out, unsigned long long* cyc, float a, float b, int seed) {
  f32x16 accB[2];
  f32x4 accS[2];
  for (int i = 0; i < 2; ++i) { for (int j = 0; j < 16; ++j) accB[i][j] = 0; accS[i] = f32x4{0, 0, 0, 0}; }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
  f16x8 av[2], bv[2];
  for (int i = 0; i < 2; ++i) {
    i32x4 t = i32x4{seed + i, seed * (i + 2), (int)threadIdx.x * (i + 1), seed ^ i};
    av[i] = __builtin_bit_cast(f16x8, t);
    t[0] += 17;
    bv[i] = __builtin_bit_cast(f16x8, t);
  }
  unsigned long long t0, t1;
  FENCE();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  FENCE();
  for (int it = 0; it < 16; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (BIG) accB[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[m & 1], bv[m & 1], accB[m & 1], 0, 0, 0);
      else accS[m & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[m & 1], bv[m & 1], accS[m & 1], 0, 0, 0);
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
  for (int i = 0; i < 2; ++i) { for (int j = 0; j < 16; ++j) s += accB[i][j]; s += accS[i][0] + accS[i][3]; }
  for (int i = 0; i < 8; ++i) s += v[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int BIG, int NVALU, int TRANS>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 8);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<BIG, NVALU, TRANS>), dim3(1), dim3(64), 0, 0, out, cyc, 1.0001f, 0.5f, 12345);
  hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-36s  %6.1f cycles per MFMA  (%5.2f cycles per 16384 FLOP)\n", name, c / 256.0, c / 256.0 / (BIG ? 2.0 : 1.0));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0, 0, 0>("16x16x32: alone");
  run<0, 2, 0>("16x16x32: + 2 fma");
  run<0, 4, 0>("16x16x32: + 4 fma");
  run<1, 0, 0>("32x32x16: alone");
  run<1, 2, 0>("32x32x16: + 2 fma");
  run<1, 4, 0>("32x32x16: + 4 fma");
  run<1, 6, 0>("32x32x16: + 6 fma");
  run<1, 8, 0>("32x32x16: + 8 fma");
  run<1, 12, 0>("32x32x16: + 12 fma");
  run<1, 2, 1>("32x32x16: + 2 exp");
  run<1, 4, 1>("32x32x16: + 4 exp");
  return 0;
}
