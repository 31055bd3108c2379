// Microbenchmark: two waves on ONE SIMD (512-thread workgroup: waves w and w+4 share SIMD w): does the VALU
// stream of one wave run under the f16 / f32 MFMA stream of the other?
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using i32x4 = __attribute__((ext_vector_type(4))) int;
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// role: 0 idle, 1 f16 MFMA loop, 2 VALU fma loop, 3 VALU exp loop, 4 f32 MFMA loop, 5 mixed (1 MFMA + 2 fma)
__device__ __forceinline__ unsigned long long work(int role, float a, float b, int seed, float* sink) {
  f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
  i32x4 t = i32x4{seed, seed * 3, (int)threadIdx.x, seed ^ 5};
  f16x8 av = __builtin_bit_cast(f16x8, t);
  t[0] += 17;
  f16x8 bv = __builtin_bit_cast(f16x8, t);
  unsigned long long t0, t1;
  FENCE();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  FENCE();
  if (role == 1) {
    for (int it = 0; it < 64; ++it) {
#pragma unroll
      for (int m = 0; m < 16; ++m) { acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[m & 1], 0, 0, 0); FENCE(); }
    }
  } else if (role == 2) {
    for (int it = 0; it < 64; ++it) {
#pragma unroll
      for (int m = 0; m < 64; ++m) { v[m & 7] = __builtin_fmaf(v[m & 7], a, b); FENCE(); }
    }
  } else if (role == 3) {
    for (int it = 0; it < 64; ++it) {
#pragma unroll
      for (int m = 0; m < 32; ++m) { v[m & 7] = __builtin_amdgcn_exp2f(v[m & 7]); FENCE(); }
    }
  } else if (role == 4) {
    for (int it = 0; it < 64; ++it) {
#pragma unroll
      for (int m = 0; m < 16; ++m) { acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + m, b, acc[m & 1], 0, 0, 0); FENCE(); }
    }
  } else if (role == 5) {
    for (int it = 0; it < 64; ++it) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[m & 1], 0, 0, 0); FENCE();
        v[m & 7] = __builtin_fmaf(v[m & 7], a, b); FENCE();
        v[(m + 1) & 7] = __builtin_fmaf(v[(m + 1) & 7], a, b); FENCE();
      }
    }
  }
  FENCE();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  FENCE();
  float s = acc[0][0] + acc[1][1];
  for (int i = 0; i < 8; ++i) s += v[i];
  sink[threadIdx.x] = s;
  return t1 - t0;
}

__global__ void __launch_bounds__(512) k(float* sink, unsigned long long* cyc, float a, float b, int seed, int roleA, int roleB) {
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? roleA : roleB;     // waves 0-3 and 4-7 pair up on SIMDs 0-3
  __syncthreads();
  unsigned long long c = work(role, a, b, seed, sink);
  if ((threadIdx.x & 63) == 0) cyc[wave] = c;
}

int main() {
  float* sink; unsigned long long* cyc;
  hipMalloc(&sink, 512 * 4); hipMalloc(&cyc, 64);
  const char* names[] = {"idle", "f16 MFMA x1024", "fma x4096", "exp x2048", "f32 MFMA x1024", "f16 MFMA x1024 + 2 fma each"};
  int combos[][2] = {{1, 0}, {2, 0}, {3, 0}, {4, 0}, {5, 0}, {1, 2}, {1, 3}, {4, 2}, {1, 1}, {2, 2}, {5, 5}, {1, 5}};
  for (auto& c : combos) {
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, sink, cyc, 1.0001f, 0.5f, 12345, c[0], c[1]);
    hipDeviceSynchronize();
    unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("A(waves0-3)=%-28s B(waves4-7)=%-28s  cycles A %7llu  B %7llu\n", names[c[0]], names[c[1]], h[0], h[4]);
  }
  return 0;
}
