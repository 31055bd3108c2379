// Microbenchmark (gfx950), round 3: is a VALU write to the srcA / srcB registers of a just-issued v_mfma safe?
//
// hipcc's hazard recognizer has no rule for "XDL reads srcA/srcB -> VALU write" (A and B are taken to be read at issue).
// In flow_kernel_hx3 the last consumption unit of a pass with an output-layer chunk multiplies the chunk's B operand hO
// (the activated hidden tiles, registers) into the output accumulators, and the NEXT pass starts re-writing hO with the
// next tile's activation + split a dozen instructions later.  If the matrix pipe is backed up (two waves per SIMD, 12
// v_mfma issued in 96 cycles that take 192 to execute) the last v_mfma may read B after that write.
//
//   N x v_mfma acc[k % NACC] += A . B          (NACC accumulators round robin; N = CHAIN)
//   s_nop ...                                   WS wait states in all
//   v_mov B[0..3] = 1e4 (f16 pairs)             (MODE A: the same for the A registers)
//   -> every accumulator must hold 1 + 32 * (its share of N); a larger value read an operand after the v_mov.
//
//   hipcc --offload-arch=gfx950 -O2 -o mfma_srcab_war mfma_srcab_war.hip && ./mfma_srcab_war
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

// fixed registers: A = v[32:35], B = v[36:39], accumulators v[40:43], v[44:47], v[48:51], v[52:55]
#define M0_ "v_mfma_f32_16x16x32_f16 v[40:43], v[32:35], v[36:39], v[40:43]\n\t"
#define M1_ "v_mfma_f32_16x16x32_f16 v[44:47], v[32:35], v[36:39], v[44:47]\n\t"
#define M2_ "v_mfma_f32_16x16x32_f16 v[48:51], v[32:35], v[36:39], v[48:51]\n\t"
#define M3_ "v_mfma_f32_16x16x32_f16 v[52:55], v[32:35], v[36:39], v[52:55]\n\t"
// the kernel's last unit: 4 output tiles x 3 products = 3 dependent v_mfma per accumulator, accumulator after accumulator
#define UNIT12 M0_ M0_ M0_ M1_ M1_ M1_ M2_ M2_ M2_ M3_ M3_ M3_
// independent round robin
#define RR12 M0_ M1_ M2_ M3_ M0_ M1_ M2_ M3_ M0_ M1_ M2_ M3_
#define ONE M3_

#define CLOBBER_B "v_mov_b32 v36, 0x70e270e2\n\tv_mov_b32 v37, 0x70e270e2\n\tv_mov_b32 v38, 0x70e270e2\n\tv_mov_b32 v39, 0x70e270e2\n\t"
#define CLOBBER_A "v_mov_b32 v32, 0x70e270e2\n\tv_mov_b32 v33, 0x70e270e2\n\tv_mov_b32 v34, 0x70e270e2\n\tv_mov_b32 v35, 0x70e270e2\n\t"

#define PROLOGUE                                                                                                      \
  "v_mov_b32 v32, 0x3c003c00\n\tv_mov_b32 v33, 0x3c003c00\n\tv_mov_b32 v34, 0x3c003c00\n\tv_mov_b32 v35, 0x3c003c00\n\t" \
  "v_mov_b32 v36, 0x3c003c00\n\tv_mov_b32 v37, 0x3c003c00\n\tv_mov_b32 v38, 0x3c003c00\n\tv_mov_b32 v39, 0x3c003c00\n\t" \
  "v_mov_b32 v40, 1.0\n\tv_mov_b32 v41, 1.0\n\tv_mov_b32 v42, 1.0\n\tv_mov_b32 v43, 1.0\n\t"                           \
  "v_mov_b32 v44, 1.0\n\tv_mov_b32 v45, 1.0\n\tv_mov_b32 v46, 1.0\n\tv_mov_b32 v47, 1.0\n\t"                           \
  "v_mov_b32 v48, 1.0\n\tv_mov_b32 v49, 1.0\n\tv_mov_b32 v50, 1.0\n\tv_mov_b32 v51, 1.0\n\t"                           \
  "v_mov_b32 v52, 1.0\n\tv_mov_b32 v53, 1.0\n\tv_mov_b32 v54, 1.0\n\tv_mov_b32 v55, 1.0\n\t"                           \
  "s_nop 4\n\t"
#define EPILOGUE                                                                                                      \
  "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"                   \
  "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"                   \
  "v_max_f32 %0, v40, v41\n\tv_max_f32 %0, %0, v42\n\tv_max_f32 %0, %0, v43\n\t"                                      \
  "v_max_f32 %1, v44, v45\n\tv_max_f32 %1, %1, v46\n\tv_max_f32 %1, %1, v47\n\t"                                      \
  "v_max_f32 %2, v48, v49\n\tv_max_f32 %2, %2, v50\n\tv_max_f32 %2, %2, v51\n\t"                                      \
  "v_max_f32 %3, v52, v53\n\tv_max_f32 %3, %3, v54\n\tv_max_f32 %3, %3, v55"
#define REGS "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", \
             "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55"

// SHAPE 0: UNIT12, 1: RR12, 2: ONE v_mfma;  WHICH 0: clobber B, 1: clobber A
template <int SHAPE, int WHICH, int WS>
__device__ __forceinline__ void body(float& r0, float& r1, float& r2, float& r3) {
#define RUN(SEQ, CL)                                                                                      \
  if constexpr (WS == 0)                                                                                  \
    asm volatile(PROLOGUE SEQ CL EPILOGUE : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : : REGS);         \
  else                                                                                                    \
    asm volatile(PROLOGUE SEQ "s_nop %4\n\t" CL EPILOGUE : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)     \
                 : "n"(WS > 0 ? WS - 1 : 0) : REGS);
  if constexpr (SHAPE == 0 && WHICH == 0) { RUN(UNIT12, CLOBBER_B) }
  else if constexpr (SHAPE == 0 && WHICH == 1) { RUN(UNIT12, CLOBBER_A) }
  else if constexpr (SHAPE == 1 && WHICH == 0) { RUN(RR12, CLOBBER_B) }
  else if constexpr (SHAPE == 1 && WHICH == 1) { RUN(RR12, CLOBBER_A) }
  else if constexpr (SHAPE == 2 && WHICH == 0) { RUN(ONE, CLOBBER_B) }
  else { RUN(ONE, CLOBBER_A) }
#undef RUN
}

template <int SHAPE, int WHICH, int WS, int MODE>
__global__ void __launch_bounds__(512) k(unsigned* bad, int iters) {
  const int wave = threadIdx.x >> 6;
  unsigned wrong = 0;
  const float w012 = SHAPE == 2 ? 1.0f : 1.0f + 32.0f * 3, w3 = SHAPE == 2 ? 1.0f + 32.0f : 1.0f + 32.0f * 3;
  if (MODE == 1 && wave >= 4) {          // the second wave of every SIMD floods the matrix pipe
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)1.0f; b[i] = (_Float16)1.0f; }
    f32x4 x0 = {0, 0, 0, 0}, x1 = x0, x2 = x0, x3 = x0;
    for (int it = 0; it < iters * 5; ++it) {
      x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, x0, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, x1, 0, 0, 0);
      x2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, x2, 0, 0, 0);
      x3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, x3, 0, 0, 0);
    }
    if (x0[0] + x1[0] + x2[0] + x3[0] == -1.0f) atomicAdd(bad + 1, 1u);
    return;
  }
  for (int it = 0; it < iters; ++it) {
    const int dly = ((wave >> 2) * 5 + it * 3 + wave) & 15;
    for (int q = 0; q < dly; ++q) asm volatile("s_nop 3");
    float r0, r1, r2, r3;
    body<SHAPE, WHICH, WS>(r0, r1, r2, r3);
    wrong += (r0 != w012) | (r1 != w012) | (r2 != w012) | (r3 != w3);
  }
  if ((threadIdx.x & 63) == 0 && wrong) atomicAdd(bad, wrong);
}

template <int SHAPE, int WHICH, int WS, int MODE>
static void run(unsigned* bad) {
  (void)hipMemset(bad, 0, 8);
  k<SHAPE, WHICH, WS, MODE><<<256, 512>>>(bad, 4000);
  unsigned h = 0;
  (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf(" %8u", h);
}

template <int SHAPE, int WHICH, int MODE>
static void sweep(unsigned* bad) {
  static const char* shapes[] = {"3 x 4 acc (kernel's last unit)", "12 round robin", "one v_mfma"};
  printf("  %-30s clobber %c mode %d:", shapes[SHAPE], WHICH ? 'A' : 'B', MODE);
  run<SHAPE, WHICH, 0, MODE>(bad); run<SHAPE, WHICH, 1, MODE>(bad); run<SHAPE, WHICH, 2, MODE>(bad); run<SHAPE, WHICH, 3, MODE>(bad);
  run<SHAPE, WHICH, 4, MODE>(bad); run<SHAPE, WHICH, 6, MODE>(bad); run<SHAPE, WHICH, 8, MODE>(bad); run<SHAPE, WHICH, 10, MODE>(bad);
  run<SHAPE, WHICH, 12, MODE>(bad); run<SHAPE, WHICH, 16, MODE>(bad);
  printf("\n");
}

int main() {
  unsigned* bad;
  (void)hipMalloc(&bad, 8);
  printf("wrong results (lane 0 of each of 2048 / 1024 testing waves x 4000 runs) by wait states between the last v_mfma and the\n"
         "v_mov into its srcB / srcA registers:                          WS =  0 1 2 3 4 6 8 10 12 16\n");
  sweep<0, 0, 0>(bad); sweep<0, 1, 0>(bad); sweep<1, 0, 0>(bad); sweep<1, 1, 0>(bad); sweep<2, 0, 0>(bad); sweep<2, 1, 0>(bad);
  sweep<0, 0, 1>(bad); sweep<0, 1, 1>(bad); sweep<1, 0, 1>(bad); sweep<1, 1, 1>(bad); sweep<2, 0, 1>(bad); sweep<2, 1, 1>(bad);
  return 0;
}
