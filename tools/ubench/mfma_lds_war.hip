// Microbenchmark (gfx950), round 3: is an LDS read RETURNING INTO the source registers of a just-issued v_mfma safe?
//
// mfma_srcc_war.hip / mfma_srcab_war.hip showed that a VALU write to srcA / srcB / srcC of an in-flight v_mfma is
// interlocked by the hardware (0 wrong results at 0 wait states, busy pipe or not).  The other writer in the failing
// kernel tail is `ds_read_b128 v[50:53]` (the next pass's bias, into the dead srcC registers of the tail v_mfma, 9 wait
// states behind it): an LDS return is asynchronous, and nothing tracks that a queued v_mfma has yet to read the register.
//
//   12 x v_mfma (4 accumulators x 3 dependent, the kernel's last unit)  [+ untied tail out = A.B + acc3 for the srcC case]
//   s_nop ...                                   WS wait states in all
//   ds_read_b128 into the A / B / srcC registers (LDS holds f16 1e4 pairs)
//   -> every accumulator must hold its exact sum; a larger value read an operand after the LDS data landed.
//
//   hipcc --offload-arch=gfx950 -O2 -o mfma_lds_war mfma_lds_war.hip && ./mfma_lds_war
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
#define TAIL "v_mfma_f32_16x16x32_f16 v[56:59], v[32:35], v[36:39], v[52:55]\n\t"

// the clobber is an LDS read returning into the operand registers (address in v31; the LDS words hold f16 1e4 pairs = f32 2.6e30)
#define CLOBBER_B "ds_read_b128 v[36:39], v31\n\t"
#define CLOBBER_A "ds_read_b128 v[32:35], v31\n\t"
#define CLOBBER_C "ds_read_b128 v[52:55], v31\n\t"

#define PROLOGUE                                                                                                      \
  "v_mov_b32 v32, 0x3c003c00\n\tv_mov_b32 v33, 0x3c003c00\n\tv_mov_b32 v34, 0x3c003c00\n\tv_mov_b32 v35, 0x3c003c00\n\t" \
  "v_mov_b32 v36, 0x3c003c00\n\tv_mov_b32 v37, 0x3c003c00\n\tv_mov_b32 v38, 0x3c003c00\n\tv_mov_b32 v39, 0x3c003c00\n\t" \
  "v_mov_b32 v40, 1.0\n\tv_mov_b32 v41, 1.0\n\tv_mov_b32 v42, 1.0\n\tv_mov_b32 v43, 1.0\n\t"                           \
  "v_mov_b32 v44, 1.0\n\tv_mov_b32 v45, 1.0\n\tv_mov_b32 v46, 1.0\n\tv_mov_b32 v47, 1.0\n\t"                           \
  "v_mov_b32 v48, 1.0\n\tv_mov_b32 v49, 1.0\n\tv_mov_b32 v50, 1.0\n\tv_mov_b32 v51, 1.0\n\t"                           \
  "v_mov_b32 v52, 1.0\n\tv_mov_b32 v53, 1.0\n\tv_mov_b32 v54, 1.0\n\tv_mov_b32 v55, 1.0\n\t"                           \
  "v_mov_b32 v31, %4\n\tv_mov_b32 v56, 0\n\tv_mov_b32 v57, 0\n\tv_mov_b32 v58, 0\n\tv_mov_b32 v59, 0\n\ts_nop 4\n\t"
#define EPILOGUE                                                                                                      \
  "s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"     \
  "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"                   \
  "v_max_f32 %0, v40, v41\n\tv_max_f32 %0, %0, v42\n\tv_max_f32 %0, %0, v43\n\t"                                      \
  "v_max_f32 %1, v44, v45\n\tv_max_f32 %1, %1, v46\n\tv_max_f32 %1, %1, v47\n\t"                                      \
  "v_max_f32 %2, v48, v49\n\tv_max_f32 %2, %2, v50\n\tv_max_f32 %2, %2, v51\n\t"                                      \
  "v_max_f32 %3, v56, v57\n\tv_max_f32 %3, %3, v58\n\tv_max_f32 %3, %3, v59"
#define REGS "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", \
             "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v31", "memory"

// SHAPE 0: the kernel's last unit (3 x 4 accumulators) + untied tail, 1: round robin + tail, 2: the tail alone (idle pipe);
// WHICH 0: LDS data into srcB, 1: srcA, 2: srcC of the tail
template <int SHAPE, int WHICH, int WS>
__device__ __forceinline__ void body(unsigned lds_addr, float& r0, float& r1, float& r2, float& r3) {
#define RUN(SEQ, CL)                                                                                                 \
  if constexpr (WS == 0)                                                                                             \
    asm volatile(PROLOGUE SEQ TAIL CL EPILOGUE : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(lds_addr) : REGS);  \
  else                                                                                                               \
    asm volatile(PROLOGUE SEQ TAIL "s_nop %5\n\t" CL EPILOGUE : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)           \
                 : "v"(lds_addr), "n"(WS > 0 ? WS - 1 : 0) : REGS);
  if constexpr (SHAPE == 0 && WHICH == 0) { RUN(UNIT12, CLOBBER_B) }
  else if constexpr (SHAPE == 0 && WHICH == 1) { RUN(UNIT12, CLOBBER_A) }
  else if constexpr (SHAPE == 0 && WHICH == 2) { RUN(UNIT12, CLOBBER_C) }
  else if constexpr (SHAPE == 1 && WHICH == 0) { RUN(RR12, CLOBBER_B) }
  else if constexpr (SHAPE == 1 && WHICH == 1) { RUN(RR12, CLOBBER_A) }
  else if constexpr (SHAPE == 1 && WHICH == 2) { RUN(RR12, CLOBBER_C) }
  else if constexpr (SHAPE == 2 && WHICH == 0) { RUN("", CLOBBER_B) }
  else if constexpr (SHAPE == 2 && WHICH == 1) { RUN("", CLOBBER_A) }
  else { RUN("", CLOBBER_C) }
#undef RUN
}

template <int SHAPE, int WHICH, int WS, int MODE>
__global__ void __launch_bounds__(512) k(unsigned* bad, int iters) {
  __shared__ unsigned big[512 * 4];
  for (int q = 0; q < 4; ++q) big[threadIdx.x * 4 + q] = 0x70e270e2u;
  __syncthreads();
  const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&big[threadIdx.x * 4];
  const int wave = threadIdx.x >> 6;
  unsigned wrong = 0;
  // accumulators 0..2: 1 + 3 x 32 (or 1 when no sequence runs); the tail: acc3 + 32
  const float w012 = SHAPE == 2 ? 1.0f : 1.0f + 32.0f * 3, w3 = (SHAPE == 2 ? 1.0f : 1.0f + 32.0f * 3) + 32.0f;
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
    body<SHAPE, WHICH, WS>(lds_addr, r0, r1, r2, r3);
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
  static const char* shapes[] = {"3 x 4 acc + tail", "12 round robin + tail", "tail alone"};
  printf("  %-24s LDS data into src%c mode %d:", shapes[SHAPE], "BAC"[WHICH], MODE);
  run<SHAPE, WHICH, 0, MODE>(bad); run<SHAPE, WHICH, 1, MODE>(bad); run<SHAPE, WHICH, 2, MODE>(bad); run<SHAPE, WHICH, 3, MODE>(bad);
  run<SHAPE, WHICH, 4, MODE>(bad); run<SHAPE, WHICH, 6, MODE>(bad); run<SHAPE, WHICH, 8, MODE>(bad); run<SHAPE, WHICH, 10, MODE>(bad);
  run<SHAPE, WHICH, 12, MODE>(bad); run<SHAPE, WHICH, 16, MODE>(bad);
  printf("\n");
}

int main() {
  unsigned* bad;
  (void)hipMalloc(&bad, 8);
  printf("wrong results (lane 0 of each of 2048 / 1024 testing waves x 4000 runs) by wait states between the tail v_mfma and the\n"
         "ds_read_b128 into its srcB / srcA / srcC registers:              WS =  0 1 2 3 4 6 8 10 12 16\n");
  sweep<0, 0, 0>(bad); sweep<0, 1, 0>(bad); sweep<0, 2, 0>(bad); sweep<1, 2, 0>(bad); sweep<2, 0, 0>(bad); sweep<2, 1, 0>(bad); sweep<2, 2, 0>(bad);
  sweep<0, 0, 1>(bad); sweep<0, 1, 1>(bad); sweep<0, 2, 1>(bad); sweep<1, 2, 1>(bad); sweep<2, 0, 1>(bad); sweep<2, 1, 1>(bad); sweep<2, 2, 1>(bad);
  return 0;
}
