// Microbenchmark (gfx950), round 3: the hazard behind gbnf_flow_kernel_hx3's mfma_tail_guard, this time with the
// matrix pipe BUSY when the tail is issued.  Round 2's replay (mfma_tail_hazard.hip) issued the two-instruction tail
// into an idle pipe right behind a barrier and never failed.  In the kernel the tail sits at the end of a stream of
// dependent v_mfma (one accumulator, back to back), issued faster (8 cycles each) than they execute (16 cycles
// each), with a second wave on the same SIMD doing the same.
//
//   v_mfma acc = A.B + c0                      (acc = v[40:43])
//   (CHAIN - 1) x  v_mfma acc = A.B + acc      tied, back to back: what a pass of the flow kernel looks like
//   v_mfma out = A.B + acc                     UNTIED tail, and out = v[44:47] IS the A operand's registers (vDst == srcA), as hipcc
//                                              allocates it in the failing kernel; acc is dead behind it
//   s_nop ...                                  WS wait states in all
//   v_mov acc[0..3] = 1e9                      the compiler's reuse of the dead registers
//   -> out must be 1 + 32 (CHAIN + 1); an `out` that picked up 1e9 read srcC after the v_mov.
//
// MODE 0: every wave runs the test, the waves of a SIMD drift against each other (per-wave start delays, no barrier);
// MODE 1: the second wave of every SIMD floods the pipe with independent v_mfma instead.
// Prints, per (CHAIN, WS), how many of the 1024 (MODE 1: 512) testing waves x ITERS tails were wrong.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_srcc_war mfma_srcc_war.hip && ./mfma_srcc_war
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

#define TIED "v_mfma_f32_16x16x32_f16 v[40:43], v[44:47], %6, v[40:43]\n\t"
#define REP1(x) x
#define REP3(x) x x x
#define REP7(x) x x x x x x x
#define REP13(x) x x x x x x x x x x x x x

template <int CHAIN> struct Body;
#define BODY(N, MID)                                                                                                 \
  template <> struct Body<N> {                                                                                       \
    template <int WS>                                                                                                \
    static __device__ __forceinline__ void run(f32x4 c0, f16x8 a, f16x8 b, float& r0, float& r1, float& r2, float& r3) { \
      if constexpr (WS == 0)                                                                                         \
        asm volatile("v_mov_b32 v44, 0x3c003c00\n\tv_mov_b32 v45, 0x3c003c00\n\tv_mov_b32 v46, 0x3c003c00\n\tv_mov_b32 v47, 0x3c003c00\n\ts_nop 1\n\tv_mfma_f32_16x16x32_f16 v[40:43], v[44:47], %6, %4\n\t" MID                                          \
                     "v_mfma_f32_16x16x32_f16 v[44:47], v[44:47], %6, v[40:43]\n\t"                                        \
                     "v_mov_b32 v40, 0x4e6e6b28\n\tv_mov_b32 v41, 0x4e6e6b28\n\tv_mov_b32 v42, 0x4e6e6b28\n\tv_mov_b32 v43, 0x4e6e6b28\n\t" \
                     "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"                       \
                     "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v46\n\tv_mov_b32 %3, v47"              \
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)                                                    \
                     : "v"(c0), "v"(a), "v"(b)                                                                       \
                     : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");                                      \
      else                                                                                                           \
        asm volatile("v_mov_b32 v44, 0x3c003c00\n\tv_mov_b32 v45, 0x3c003c00\n\tv_mov_b32 v46, 0x3c003c00\n\tv_mov_b32 v47, 0x3c003c00\n\ts_nop 1\n\tv_mfma_f32_16x16x32_f16 v[40:43], v[44:47], %6, %4\n\t" MID                                          \
                     "v_mfma_f32_16x16x32_f16 v[44:47], v[44:47], %6, v[40:43]\n\t"                                        \
                     "s_nop %7\n\t"                                                                                  \
                     "v_mov_b32 v40, 0x4e6e6b28\n\tv_mov_b32 v41, 0x4e6e6b28\n\tv_mov_b32 v42, 0x4e6e6b28\n\tv_mov_b32 v43, 0x4e6e6b28\n\t" \
                     "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"                       \
                     "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v46\n\tv_mov_b32 %3, v47"              \
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)                                                    \
                     : "v"(c0), "v"(a), "v"(b), "n"(WS > 0 ? WS - 1 : 0)                                             \
                     : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");                                      \
    }                                                                                                                \
  };
BODY(1, "")
BODY(2, REP1(TIED))
BODY(4, REP3(TIED))
BODY(8, REP7(TIED))
BODY(14, REP13(TIED))

template <int CHAIN, int WS, int MODE>
__global__ void __launch_bounds__(512) k(unsigned* bad, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)1.0f; b[i] = (_Float16)1.0f; }
  const int wave = threadIdx.x >> 6;
  unsigned wrong = 0;
  const float want = 1.0f + 32.0f * (CHAIN + 1);
  if (MODE == 1 && wave >= 4) {
    // flood: independent accumulators, back to back, for about as long as the testers run
    f32x4 x0 = {0, 0, 0, 0}, x1 = x0, x2 = x0, x3 = x0;
    for (int it = 0; it < iters * (CHAIN + 8) / 4; ++it) {
      x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, x0, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, x1, 0, 0, 0);
      x2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, x2, 0, 0, 0);
      x3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, x3, 0, 0, 0);
    }
    if (x0[0] + x1[0] + x2[0] + x3[0] == -1.0f) atomicAdd(bad + 1, 1u);
    return;
  }
  for (int it = 0; it < iters; ++it) {
    f32x4 c0 = {1.0f, 1.0f, 1.0f, 1.0f};
    asm volatile("" : "+v"(c0), "+v"(a), "+v"(b));
    // drift: a start delay that differs between the two waves of a SIMD and from iteration to iteration
    const int dly = ((wave >> 2) * 5 + it * 3 + wave) & 15;
    for (int q = 0; q < dly; ++q) asm volatile("s_nop 3");
    float r0, r1, r2, r3;
    Body<CHAIN>::template run<WS>(c0, a, b, r0, r1, r2, r3);
    wrong += (r0 != want) | (r1 != want) | (r2 != want) | (r3 != want);
  }
  const unsigned any = __builtin_amdgcn_readfirstlane(__popcll(__ballot(wrong != 0)) ? 1 : 0);
  unsigned tot = wrong;       // lanes agree in practice; count tails once per wave via lane 0
  if ((threadIdx.x & 63) == 0 && any) atomicAdd(bad, tot ? tot : 1u);
}

template <int CHAIN, int WS, int MODE>
static void run(unsigned* bad) {
  hipMemset(bad, 0, 8);
  const int iters = 4000;
  k<CHAIN, WS, MODE><<<256, 512>>>(bad, iters);
  unsigned h = 0;
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf(" %8u", h);
}

template <int CHAIN, int MODE>
static void sweep(unsigned* bad) {
  printf("  chain %2d mode %d:", CHAIN, MODE);
  run<CHAIN, 0, MODE>(bad); run<CHAIN, 1, MODE>(bad); run<CHAIN, 2, MODE>(bad); run<CHAIN, 3, MODE>(bad);
  run<CHAIN, 4, MODE>(bad); run<CHAIN, 5, MODE>(bad); run<CHAIN, 6, MODE>(bad); run<CHAIN, 7, MODE>(bad);
  run<CHAIN, 8, MODE>(bad); run<CHAIN, 9, MODE>(bad); run<CHAIN, 10, MODE>(bad); run<CHAIN, 11, MODE>(bad);
  run<CHAIN, 12, MODE>(bad); run<CHAIN, 14, MODE>(bad); run<CHAIN, 16, MODE>(bad);
  printf("\n");
}

int main() {
  unsigned* bad;
  hipMalloc(&bad, 8);
  printf("wrong tails (lane-0 count per wave, 4000 tails per wave) by wait states between the untied tail v_mfma and the v_mov\n"
         "into its srcC registers:   WS =  0 1 2 3 4 5 6 7 8 9 10 11 12 14 16\n");
  sweep<1, 0>(bad); sweep<2, 0>(bad); sweep<4, 0>(bad); sweep<8, 0>(bad); sweep<14, 0>(bad);
  sweep<1, 1>(bad); sweep<2, 1>(bad); sweep<4, 1>(bad); sweep<8, 1>(bad); sweep<14, 1>(bad);
  return 0;
}
