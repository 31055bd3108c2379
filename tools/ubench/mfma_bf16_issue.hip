// Microbenchmark: VALU / LDS co-issue under v_mfma_f32_16x16x32_bf16 on gfx950 (one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using i32x4 = __attribute__((ext_vector_type(4))) int;
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int NACC, int NVALU, int TRANS, int NLDS>
__global__ void __launch_bounds__(64) k(float* out, unsigned long long* cyc, float a, float b, int seed) {
  __shared__ i32x4 L[64 * 8];
  for (int q = 0; q < 8; ++q) L[q * 64 + threadIdx.x] = i32x4{seed + q, seed * 3, q, (int)threadIdx.x};
  __syncthreads();
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
  bf16x8 av[4], bv[4];
  for (int i = 0; i < 4; ++i) {
    i32x4 t = i32x4{seed + i, seed * (i + 2), (int)threadIdx.x * (i + 1), seed ^ i};
    av[i] = __builtin_bit_cast(bf16x8, t);
    t[0] += 17;
    bv[i] = __builtin_bit_cast(bf16x8, t);
  }
  i32x4 ld[4] = {};
  unsigned long long t0, t1;
  FENCE();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  FENCE();
  for (int it = 0; it < 16; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[m % 4], bv[m % 4], acc[m % NACC], 0, 0, 0);
      FENCE();
#pragma unroll
      for (int q = 0; q < NVALU; ++q) {
        if (TRANS) v[q % 8] = __builtin_amdgcn_exp2f(v[q % 8]);
        else v[q % 8] = __builtin_fmaf(v[q % 8], a, b);
        FENCE();
      }
#pragma unroll
      for (int q = 0; q < NLDS; ++q) {
        ld[q % 4] = ld[q % 4] + L[((m + q) % 8) * 64 + threadIdx.x];
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
  for (int i = 0; i < 4; ++i) s += ld[i][0] + ld[i][3];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, int NVALU, int TRANS, int NLDS>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 8);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<NACC, NVALU, TRANS, NLDS>), dim3(1), dim3(64), 0, 0, out, cyc, 1.0001f, 0.5f, 12345);
  hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-40s  %6.1f cycles per MFMA\n", name, c / 256.0);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<4, 0, 0, 0>("bf16 16x16x32: 4 acc, nothing else");
  run<2, 0, 0, 0>("2 acc");
  run<1, 0, 0, 0>("1 acc (dependent chain)");
  run<2, 1, 0, 0>("2 acc + 1 fma");
  run<2, 2, 0, 0>("2 acc + 2 fma");
  run<2, 3, 0, 0>("2 acc + 3 fma");
  run<2, 4, 0, 0>("2 acc + 4 fma");
  run<2, 6, 0, 0>("2 acc + 6 fma");
  run<2, 1, 1, 0>("2 acc + 1 exp");
  run<2, 2, 1, 0>("2 acc + 2 exp");
  run<2, 0, 0, 1>("2 acc + 1 ds_read_b128(+add)");
  run<2, 2, 0, 1>("2 acc + 2 fma + 1 ds_read_b128(+add)");
  return 0;
}
