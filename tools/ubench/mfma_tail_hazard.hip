// Microbenchmark (gfx950): how many wait states does a VALU write need behind a dependent, untied v_mfma chain whose
// source-C registers it overwrites?  (The hazard behind gbnf_flow_kernel_hx3's mfma_tail_guard.)
//
//   barrier
//   v_mfma c1 = A.B + c0          (c0 = 1)           A = B = 1.0 (f16), so each product adds k = 32
//   v_mfma c2 = A.B + c1          vDst != srcC: c1 is dead behind this instruction
//   s_nop NOPS
//   v_mov c1[0..3] = 1e9          the reuse of the dead registers
//   -> c2 must be 65 everywhere; a c2 that picked up 1e9 read source C after the v_mov
//
// One workgroup per CU; 256 threads = one wave per SIMD, 512 = two per SIMD, both leaving the barrier together.
// Reports, per NOPS, the number of (launch, wave) pairs with a wrong c2.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_tail_hazard mfma_tail_hazard.hip && ./mfma_tail_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <int NOPS, int CHAIN>
__global__ void __launch_bounds__(512) k(unsigned* bad, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)1.0f; b[i] = (_Float16)1.0f; }
  unsigned wrong = 0;
  for (int it = 0; it < iters; ++it) {
    f32x4 c0 = {1.0f, 1.0f, 1.0f, 1.0f};
    float r0, r1, r2, r3;
    asm volatile("" : "+v"(c0), "+v"(a), "+v"(b));
    // fixed registers: c1 = v[40:43], c2 = v[44:47], c3 = v[48:51]
    if constexpr (CHAIN == 2) {
      asm volatile(
          "s_barrier\n\t"
          "v_mfma_f32_16x16x32_f16 v[40:43], %5, %6, %4\n\t"
          "v_mfma_f32_16x16x32_f16 v[44:47], %5, %6, v[40:43]\n\t"
          "s_nop %7\n\t"
          "v_mov_b32 v40, 0x4e6e6b28\n\tv_mov_b32 v41, 0x4e6e6b28\n\tv_mov_b32 v42, 0x4e6e6b28\n\tv_mov_b32 v43, 0x4e6e6b28\n\t"
          "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
          "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v46\n\tv_mov_b32 %3, v47"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
          : "v"(c0), "v"(a), "v"(b), "n"(NOPS)
          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
      wrong += (r0 != 65.0f) | (r1 != 65.0f) | (r2 != 65.0f) | (r3 != 65.0f);
    } else {
      asm volatile(
          "s_barrier\n\t"
          "v_mfma_f32_16x16x32_f16 v[40:43], %5, %6, %4\n\t"
          "v_mfma_f32_16x16x32_f16 v[44:47], %5, %6, v[40:43]\n\t"
          "v_mfma_f32_16x16x32_f16 v[48:51], %5, %6, v[44:47]\n\t"
          "s_nop %7\n\t"
          "v_mov_b32 v44, 0x4e6e6b28\n\tv_mov_b32 v45, 0x4e6e6b28\n\tv_mov_b32 v46, 0x4e6e6b28\n\tv_mov_b32 v47, 0x4e6e6b28\n\t"
          "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
          "v_mov_b32 %0, v48\n\tv_mov_b32 %1, v49\n\tv_mov_b32 %2, v50\n\tv_mov_b32 %3, v51"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
          : "v"(c0), "v"(a), "v"(b), "n"(NOPS)
          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
      wrong += (r0 != 97.0f) | (r1 != 97.0f) | (r2 != 97.0f) | (r3 != 97.0f);
    }
  }
  if (__any(wrong != 0) && (threadIdx.x & 63) == 0) atomicAdd(bad, 1u);
}

template <int NOPS, int CHAIN>
static void run(unsigned* bad, int threads) {
  hipMemset(bad, 0, 4);
  k<NOPS, CHAIN><<<256, threads>>>(bad, 2000);
  unsigned h = 0;
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("  chain %d  s_nop %2d  %d waves/SIMD: %u of %d waves saw a wrong accumulator\n", CHAIN, NOPS, threads / 256, h, 256 * threads / 64);
}

template <int CHAIN>
static void sweep(unsigned* bad) {
  for (int threads : {256, 512}) {
    run<0, CHAIN>(bad, threads);
    run<1, CHAIN>(bad, threads);
    run<2, CHAIN>(bad, threads);
    run<3, CHAIN>(bad, threads);
    run<4, CHAIN>(bad, threads);
    run<5, CHAIN>(bad, threads);
    run<6, CHAIN>(bad, threads);
    run<7, CHAIN>(bad, threads);
    run<9, CHAIN>(bad, threads);
    run<11, CHAIN>(bad, threads);
    run<15, CHAIN>(bad, threads);
  }
}

int main() {
  unsigned* bad;
  hipMalloc(&bad, 4);
  printf("v_mov of the dead source-C register of the chain's last v_mfma, s_nop N in between (N+1 wait states):\n");
  sweep<2>(bad);
  sweep<3>(bad);
  return 0;
}
