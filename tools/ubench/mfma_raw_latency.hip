// Microbenchmark (gfx950), round 3: how many wait states does a VALU READ of a v_mfma_f32_16x16x32_f16 result need?
//
// hipcc (ROCm 7.2) pads 8 wait states between a v_mfma_f32_16x16x32_f16 and a VALU instruction that reads its result
// (its scheduling model counts the instruction as 4 passes).  The hardware has no interlock for this dependency (the ISA
// leaves it to software).  This measures the real requirement, for a lone v_mfma and for the last link of a chain of
// DEPENDENT v_mfma issued back to back (srcC = the previous result: issued with 0 wait states, executed one after the
// other), with one and with two waves per SIMD leaving a barrier together (MODE 0) or with the SIMD's other wave
// flooding the matrix pipe (MODE 1).
//
//   s_barrier
//   CHAIN x v_mfma acc = A.B + acc        (acc = v[40:43], starts at 1; A = B = 1: every link adds 32)
//   s_nop ... (WS wait states in all)
//   v_mov r, acc[0..3]                    -> must be 1 + 32 CHAIN; a smaller value was read before the last link wrote it
//
//   hipcc --offload-arch=gfx950 -O2 -o mfma_raw_latency mfma_raw_latency.hip && ./mfma_raw_latency
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

#define M "v_mfma_f32_16x16x32_f16 v[40:43], v[32:35], v[36:39], v[40:43]\n\t"
#define PRO                                                                                                             \
  "v_mov_b32 v32, 0x3c003c00\n\tv_mov_b32 v33, 0x3c003c00\n\tv_mov_b32 v34, 0x3c003c00\n\tv_mov_b32 v35, 0x3c003c00\n\t" \
  "v_mov_b32 v36, 0x3c003c00\n\tv_mov_b32 v37, 0x3c003c00\n\tv_mov_b32 v38, 0x3c003c00\n\tv_mov_b32 v39, 0x3c003c00\n\t" \
  "v_mov_b32 v40, 1.0\n\tv_mov_b32 v41, 1.0\n\tv_mov_b32 v42, 1.0\n\tv_mov_b32 v43, 1.0\n\t"                             \
  "s_nop 7\n\ts_barrier\n\t"
#define EPI "v_mov_b32 %0, v40\n\tv_mov_b32 %1, v41\n\tv_mov_b32 %2, v42\n\tv_mov_b32 %3, v43\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
#define REGS "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43"

template <int CHAIN, int WS>
__device__ __forceinline__ void body(float& r0, float& r1, float& r2, float& r3) {
#define RUN(SEQ)                                                                                                         \
  if constexpr (WS == 0) asm volatile(PRO SEQ EPI : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : : REGS);                \
  else asm volatile(PRO SEQ "s_nop %4\n\t" EPI : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "n"(WS > 0 ? WS - 1 : 0) : REGS);
  if constexpr (CHAIN == 1) { RUN(M) }
  else if constexpr (CHAIN == 2) { RUN(M M) }
  else if constexpr (CHAIN == 3) { RUN(M M M) }
  else { RUN(M M M M M M) }
#undef RUN
}

template <int CHAIN, int WS, int MODE>
__global__ void __launch_bounds__(512) k(unsigned* bad, int iters) {
  const int wave = threadIdx.x >> 6;
  unsigned wrong = 0;
  const float want = 1.0f + 32.0f * CHAIN;
  if (MODE == 1 && wave >= 4) {          // the second wave of every SIMD floods the matrix pipe (it still meets the barriers)
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)1.0f; b[i] = (_Float16)1.0f; }
    f32x4 x0 = {0, 0, 0, 0}, x1 = x0, x2 = x0, x3 = x0;
    for (int it = 0; it < iters; ++it) {
      asm volatile("s_barrier");
      for (int q = 0; q < 4; ++q) {
        x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, x0, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, x1, 0, 0, 0);
        x2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, x2, 0, 0, 0);
        x3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, x3, 0, 0, 0);
      }
    }
    if (x0[0] + x1[0] + x2[0] + x3[0] == -1.0f) atomicAdd(bad + 1, 1u);
    return;
  }
  for (int it = 0; it < iters; ++it) {
    float r0, r1, r2, r3;
    body<CHAIN, WS>(r0, r1, r2, r3);
    wrong += (r0 != want) | (r1 != want) | (r2 != want) | (r3 != want);
  }
  if ((threadIdx.x & 63) == 0 && wrong) atomicAdd(bad, wrong);
}

template <int CHAIN, int WS, int MODE>
static void run(unsigned* bad, int threads) {
  (void)hipMemset(bad, 0, 8);
  k<CHAIN, WS, MODE><<<256, threads>>>(bad, 2000);
  unsigned h = 0;
  (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf(" %7u", h);
}

template <int CHAIN, int MODE>
static void sweep(unsigned* bad, int threads) {
  printf("  chain %d, %d wave(s)/SIMD, mode %d:", CHAIN, threads / 256, MODE);
  run<CHAIN, 0, MODE>(bad, threads); run<CHAIN, 2, MODE>(bad, threads); run<CHAIN, 4, MODE>(bad, threads); run<CHAIN, 5, MODE>(bad, threads);
  run<CHAIN, 6, MODE>(bad, threads); run<CHAIN, 7, MODE>(bad, threads); run<CHAIN, 8, MODE>(bad, threads); run<CHAIN, 9, MODE>(bad, threads);
  run<CHAIN, 10, MODE>(bad, threads); run<CHAIN, 11, MODE>(bad, threads); run<CHAIN, 12, MODE>(bad, threads); run<CHAIN, 14, MODE>(bad, threads);
  run<CHAIN, 16, MODE>(bad, threads); run<CHAIN, 20, MODE>(bad, threads); run<CHAIN, 24, MODE>(bad, threads); run<CHAIN, 32, MODE>(bad, threads);
  printf("\n");
}

int main() {
  unsigned* bad;
  (void)hipMalloc(&bad, 8);
  printf("stale reads (lane 0 of every testing wave x 2000 runs) by wait states between the last v_mfma of the chain and the VALU read\n"
         "of its result (hipcc pads 8):                WS =  0 2 4 5 6 7 8 9 10 11 12 14 16 20 24 32\n");
  sweep<1, 0>(bad, 256); sweep<2, 0>(bad, 256); sweep<3, 0>(bad, 256); sweep<6, 0>(bad, 256);
  sweep<1, 0>(bad, 512); sweep<2, 0>(bad, 512); sweep<3, 0>(bad, 512); sweep<6, 0>(bad, 512);
  sweep<1, 1>(bad, 512); sweep<2, 1>(bad, 512); sweep<3, 1>(bad, 512); sweep<6, 1>(bad, 512);
  return 0;
}
