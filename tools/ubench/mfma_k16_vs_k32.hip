// Microbenchmark (round 5): cycles of the k = 16 f16 MFMA shapes against v_mfma_f32_16x16x32_f16 on gfx950 -- is a
// trailing half chunk (k = 16) half the price of a k = 32 one?  One wave, independent accumulators; also two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 mfma_k16_vs_k32.hip -o mfma_k16_vs_k32 && ./mfma_k16_vs_k32
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x2 = __attribute__((ext_vector_type(2))) int;
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// SHAPE 0: 16x16x32_f16; 1: 16x16x16_f16 (legacy shape); 2: alternating 3 x k32 + 1 x k16 (a 112-wide contraction)
template <int SHAPE, int NACC>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, int seed) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  f16x8 a8[4], b8[4];
  f16x4 a4[4], b4[4];
  for (int i = 0; i < 4; ++i) {
    i32x4 t = i32x4{seed + i, seed * (i + 2), (int)threadIdx.x * (i + 1), seed ^ i};
    a8[i] = __builtin_bit_cast(f16x8, t);
    t[0] += 17;
    b8[i] = __builtin_bit_cast(f16x8, t);
    a4[i] = __builtin_bit_cast(f16x4, i32x2{t[1], t[2]});
    b4[i] = __builtin_bit_cast(f16x4, i32x2{t[3], t[0]});
  }
  unsigned long long t0, t1;
  __syncthreads();
  FENCE();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  FENCE();
  for (int it = 0; it < 16; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const bool k32 = SHAPE == 0 || (SHAPE == 2 && (m & 3) != 3);
      if (k32) acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8[m % 4], b8[m % 4], acc[m % NACC], 0, 0, 0);
      else acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4[m % 4], b4[m % 4], acc[m % NACC], 0, 0, 0);
      FENCE();
    }
  }
  FENCE();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  FENCE();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int SHAPE, int NACC>
void run(const char* name, int waves) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 64);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(1), dim3(64 * waves), 0, 0, out, cyc, 12345);
  hipDeviceSynchronize();
  unsigned long long c[8]; hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
  unsigned long long mx = 0; for (int w = 0; w < waves; ++w) mx = c[w] > mx ? c[w] : mx;
  printf("%-44s %d wave(s) in the workgroup (%d per SIMD): %6.2f cycles per MFMA of a wave, %6.2f per MFMA of the SIMD\n", name, waves, (waves + 3) / 4,
         mx / 256.0, mx / 256.0 / ((waves + 3) / 4));
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int waves : {1, 4, 8}) {
    run<0, 4>("v_mfma_f32_16x16x32_f16, 4 accumulators", waves);
    run<1, 4>("v_mfma_f32_16x16x16_f16, 4 accumulators", waves);
    run<2, 4>("3 x k32 + 1 x k16 (k = 112), 4 accumulators", waves);
    run<0, 2>("v_mfma_f32_16x16x32_f16, 2 accumulators", waves);
    run<1, 2>("v_mfma_f32_16x16x16_f16, 2 accumulators", waves);
    run<1, 1>("v_mfma_f32_16x16x16_f16, 1 accumulator (chain)", waves);
    run<0, 1>("v_mfma_f32_16x16x32_f16, 1 accumulator (chain)", waves);
  }
  return 0;
}
