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
__global__ void __launch_bounds__(512) k(unsigned* bad, int iters, const unsigned* src) {
  __shared__ unsigned lds[8][2][256];
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)1.0f; b[i] = (_Float16)1.0f; }
  unsigned wrong = 0;
  for (int it = 0; it < iters; ++it) {
    f32x4 c0 = {1.0f, 1.0f, 1.0f, 1.0f};
    float r0, r1, r2, r3;
    unsigned y32 = 0; (void)y32;
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
    } else if constexpr (CHAIN == 14) {
      // the failing kernel's whole tail: M0, the address add into the dead registers, the staging DMA, an LDS read into
      // the same registers; the DMA must deliver this wave's 1 KiB of src to lds[wave][0]
      const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
      const unsigned long long sgpr = 0x0000000000a000ull;
      const unsigned* mine = src + ((size_t)(blockIdx.x * 8 + wave) * 2 + (it & 1)) * 256;
      unsigned long long x = (unsigned long long)(mine + lane * 4) - sgpr;
      const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)&lds[wave][0][0]);
      const unsigned other = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&lds[wave][1][lane * 4];
      lds[wave][0][lane * 4 + 0] = 0xdeadbeef; lds[wave][0][lane * 4 + 1] = 0xdeadbeef;
      lds[wave][0][lane * 4 + 2] = 0xdeadbeef; lds[wave][0][lane * 4 + 3] = 0xdeadbeef;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile(
          "s_barrier\n\t"
          "v_mfma_f32_16x16x32_f16 v[40:43], %7, %8, %6\n\t"
          "v_mfma_f32_16x16x32_f16 v[44:47], %7, %8, v[40:43]\n\t"
          "s_nop %9\n\t"
          "s_mov_b32 m0, %11\n\t"
          "s_nop 2\n\t"
          "v_lshl_add_u64 v[40:41], %5, 0, %10\n\t"
          "global_load_lds_dwordx4 v[40:41], off\n\t"
          "ds_read_b128 v[40:43], %12\n\t"
          "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
          "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
          "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v46\n\tv_mov_b32 %3, v47\n\t"
          "v_mov_b32 %4, v40"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(y32)
          : "v"(x), "v"(c0), "v"(a), "v"(b), "n"(NOPS), "s"(sgpr), "s"(dst), "v"(other)
          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");
      bool ok = true;
      for (int e = 0; e < 4; ++e) ok = ok && lds[wave][0][lane * 4 + e] == mine[lane * 4 + e];
      wrong += (r0 != 65.0f) | (r1 != 65.0f) | (r2 != 65.0f) | (r3 != 65.0f) | !ok;
    } else if constexpr (CHAIN >= 12) {
      // the VALU instruction of the failing kernel: a 64-bit address add into the dead registers (12) or fresh ones (13)
      unsigned long long x = 0x00007f0012345678ull + threadIdx.x * 16, y = 0;
      const unsigned long long sgpr = 0x00000000fffff000ull;      // carries into the high word
      asm volatile("" : "+v"(x));
      if constexpr (CHAIN == 12)
        asm volatile(
            "s_barrier\n\t"
            "v_mfma_f32_16x16x32_f16 v[40:43], %7, %8, %6\n\t"
            "v_mfma_f32_16x16x32_f16 v[44:47], %7, %8, v[40:43]\n\t"
            "s_nop %9\n\t"
            "v_lshl_add_u64 v[40:41], %5, 0, %10\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v46\n\tv_mov_b32 %3, v47\n\t"
            "v_lshl_add_u64 %4, v[40:41], 0, 0"
            : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(y)
            : "v"(x), "v"(c0), "v"(a), "v"(b), "n"(NOPS), "s"(sgpr)
            : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
      else
        asm volatile(
            "s_barrier\n\t"
            "v_mfma_f32_16x16x32_f16 v[40:43], %7, %8, %6\n\t"
            "v_mfma_f32_16x16x32_f16 v[44:47], %7, %8, v[40:43]\n\t"
            "s_nop %9\n\t"
            "v_lshl_add_u64 v[60:61], %5, 0, %10\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v46\n\tv_mov_b32 %3, v47\n\t"
            "v_lshl_add_u64 %4, v[60:61], 0, 0"
            : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(y)
            : "v"(x), "v"(c0), "v"(a), "v"(b), "n"(NOPS), "s"(sgpr)
            : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v60", "v61");
      wrong += (r0 != 65.0f) | (r1 != 65.0f) | (r2 != 65.0f) | (r3 != 65.0f) | (y != x + sgpr);
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
static void run(unsigned* bad, int threads, const unsigned* src) {
  hipMemset(bad, 0, 4);
  k<NOPS, CHAIN><<<256, threads>>>(bad, 2000, src);
  unsigned h = 0;
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("  chain %d  s_nop %2d  %d waves/SIMD: %u of %d waves saw a wrong accumulator\n", CHAIN, NOPS, threads / 256, h, 256 * threads / 64);
}

template <int CHAIN>
static void sweep(unsigned* bad, const unsigned* src) {
  for (int threads : {256, 512}) {
    run<0, CHAIN>(bad, threads, src);
    run<1, CHAIN>(bad, threads, src);
    run<2, CHAIN>(bad, threads, src);
    run<3, CHAIN>(bad, threads, src);
    run<4, CHAIN>(bad, threads, src);
    run<5, CHAIN>(bad, threads, src);
    run<6, CHAIN>(bad, threads, src);
    run<7, CHAIN>(bad, threads, src);
    run<9, CHAIN>(bad, threads, src);
    run<11, CHAIN>(bad, threads, src);
    run<15, CHAIN>(bad, threads, src);
  }
}

int main() {
  unsigned* bad;
  hipMalloc(&bad, 4);
  unsigned* src;
  const size_t words = (size_t)256 * 8 * 2 * 256;
  hipMalloc(&src, words * 4);
  {
    std::vector<unsigned> h(words);
    for (size_t i = 0; i < words; ++i) h[i] = (unsigned)(i * 2654435761u) | 1u;
    hipMemcpy(src, h.data(), words * 4, hipMemcpyHostToDevice);
  }
  printf("v_mov of the dead source-C register of the chain's last v_mfma, s_nop N in between (N+1 wait states):\n");
  sweep<2>(bad, src);
  sweep<3>(bad, src);
  printf("v_lshl_add_u64 (the address add in front of the staging DMA) into the dead registers (12) / fresh ones (13):\n");
  sweep<12>(bad, src);
  sweep<13>(bad, src);
  printf("the whole tail: m0, address add, staging DMA, LDS read into the same registers (14):\n");
  sweep<14>(bad, src);
  return 0;
}
