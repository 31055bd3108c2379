// Microbenchmark (gfx950): what does one wave's / one SIMD's instruction stream cost when f16 MFMAs are mixed with the
// other instruction kinds of the flow kernel?  One workgroup of 256 threads (one wave per SIMD) or 512 threads (two waves
// per SIMD, same program), s_memtime cycles of wave 0, per MFMA.
//
//   hipcc --offload-arch=gfx950 -O3 -o issue_model issue_model.hip && ./issue_model
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
#define FENCE() __builtin_amdgcn_sched_barrier(0)

enum { OP_NONE = 0, OP_FMA, OP_CVT, OP_EXP, OP_LDS, OP_DMA, OP_BARRIER, OP_FMAMIX, OP_PKFMA };

// BIG: 0 = v_mfma_f32_16x16x32_f16, 1 = v_mfma_f32_32x32x16_f16.  Every `PERIOD` MFMAs `NOPS` ops of kind OP are issued
// (spread: one after each of the first NOPS... MFMAs of the period when NOPS <= PERIOD, else NOPS/PERIOD after each).
template <int BIG, int OP, int NOPS, int PERIOD>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, const unsigned* gsrc, float a, float b, int seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 2654435761u;
  __syncthreads();
  f32x16 accB[4];
  f32x4 accS[4];
  for (int i = 0; i < 4; ++i) { for (int j = 0; j < 16; ++j) accB[i][j] = 0; accS[i] = f32x4{0, 0, 0, 0}; }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = a + i * 0.001f + threadIdx.x * 1e-4f;
  f16x8 av[2], bv[2];
  for (int i = 0; i < 2; ++i) {
    i32x4 t = i32x4{0x3c003c00 + seed + i, 0x38003800 + (int)threadIdx.x, 0x34003400 + i, 0x30003000};
    av[i] = __builtin_bit_cast(f16x8, t);
    t[0] += 17;
    bv[i] = __builtin_bit_cast(f16x8, t);
  }
  u32x4 ld[4] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
  const unsigned lds_addr = (unsigned)(lane * 16 + wave * 1024);
  const __attribute__((address_space(1))) char* gbase =
      (const __attribute__((address_space(1))) char*)gsrc + wave * 1024 + lane * 16;
  unsigned long long t0, t1;
  FENCE();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  FENCE();
  constexpr int ITERS = 32, M = 48;                 // 1536 MFMAs
  for (int it = 0; it < ITERS; ++it) {
    int vi = 0, li = 0, piece = 0;        // compile-time indices inside the unrolled body (no scratch)
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (BIG) accB[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[m & 1], bv[m & 1], accB[m & 3], 0, 0, 0);
      else accS[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[m & 1], bv[m & 1], accS[m & 3], 0, 0, 0);
      FENCE();
      const int ph = m % PERIOD;
      constexpr int PER = (NOPS + PERIOD - 1) / PERIOD;         // ops after one MFMA
      const int n_here = (NOPS >= PERIOD) ? PER : (ph < NOPS ? 1 : 0);
#pragma unroll
      for (int q = 0; q < n_here; ++q) {
        if (OP == OP_FMA) { v[vi & 15] = __builtin_fmaf(v[vi & 15], a, b); ++vi; }
        else if (OP == OP_EXP) { v[vi & 15] = __builtin_amdgcn_exp2f(v[vi & 15]); ++vi; }
        else if (OP == OP_CVT) {
          auto h = __builtin_amdgcn_cvt_pkrtz(v[vi & 15], v[(vi + 1) & 15]);
          v[(vi + 8) & 15] = __builtin_bit_cast(float, h); ++vi;
        } else if (OP == OP_FMAMIX) {
          float r;
          asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(v[vi & 15]), "v"(a), "v"(v[(vi + 1) & 15]));
          v[(vi + 8) & 15] = r; ++vi;
        } else if (OP == OP_PKFMA) {
          using f32x2 = __attribute__((ext_vector_type(2))) float;
          f32x2 x = {v[vi & 15], v[(vi + 1) & 15]}, y = {a, a}, z = {b, b}, r;
          asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
          v[vi & 15] = r[0]; v[(vi + 1) & 15] = r[1]; vi += 2;
        } else if (OP == OP_LDS) {
          asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(ld[li & 3]) : "v"(lds_addr + ((li & 7) << 12)));
          ++li;
        } else if (OP == OP_DMA) {
          __builtin_amdgcn_global_load_lds(gbase + ((piece & 15) << 12),
                                           (__attribute__((address_space(3))) void*)(lds + 4096 + wave * 256), 16, 0, 0);
          ++piece;
        } else if (OP == OP_BARRIER) {
          __builtin_amdgcn_s_barrier();
        }
        FENCE();
      }
    }
    if (OP == OP_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (OP == OP_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FENCE();
  }
  FENCE();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  FENCE();
  float s = 0;
  for (int i = 0; i < 4; ++i) { for (int j = 0; j < 16; ++j) s += accB[i][j]; s += accS[i][0] + accS[i][3]; }
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += __builtin_bit_cast(float, ld[i][0] ^ ld[i][3]);
  out[threadIdx.x] = s;
  if (lane == 0) cyc[wave] = t1 - t0;
}

static float* g_out; static unsigned long long* g_cyc; static unsigned* g_src;

template <int BIG, int OP, int NOPS, int PERIOD>
void run(const char* name) {
  for (int threads = 256; threads <= 512; threads *= 2) {
    for (int r = 0; r < 3; ++r)
      hipLaunchKernelGGL((k<BIG, OP, NOPS, PERIOD>), dim3(1), dim3(threads), 64 * 1024, 0, g_out, g_cyc, g_src, 1.0001f, 0.5f, 12345);
    hipDeviceSynchronize();
    unsigned long long c[8]; hipMemcpy(c, g_cyc, 64, hipMemcpyDeviceToHost);
    const double per = c[0] / 1536.0;
    if (threads == 256) printf("%-46s 1 wave/SIMD %6.1f cyc/MFMA", name, per);
    else printf("   2 waves/SIMD %6.1f cyc/MFMA/wave = %5.1f per MFMA on the SIMD\n", per, per / 2);
  }
}

int main() {
  hipMalloc(&g_out, 512 * 4); hipMalloc(&g_cyc, 64); hipMalloc(&g_src, 1 << 20); hipMemset(g_src, 1, 1 << 20);
  run<0, OP_NONE, 0, 1>("16x16x32 alone");
  run<0, OP_FMA, 1, 1>("16x16x32 + 1 fma / MFMA");
  run<0, OP_FMA, 2, 1>("16x16x32 + 2 fma / MFMA");
  run<0, OP_FMA, 3, 1>("16x16x32 + 3 fma / MFMA");
  run<0, OP_FMA, 4, 1>("16x16x32 + 4 fma / MFMA");
  run<0, OP_CVT, 2, 1>("16x16x32 + 2 cvt_pkrtz / MFMA");
  run<0, OP_FMAMIX, 2, 1>("16x16x32 + 2 fma_mix / MFMA");
  run<0, OP_PKFMA, 1, 1>("16x16x32 + 1 pk_fma / MFMA");
  run<0, OP_EXP, 1, 1>("16x16x32 + 1 exp / MFMA");
  run<0, OP_EXP, 2, 1>("16x16x32 + 2 exp / MFMA");
  run<0, OP_LDS, 1, 3>("16x16x32 + 1 ds_read_b128 / 3 MFMA");
  run<0, OP_LDS, 1, 1>("16x16x32 + 1 ds_read_b128 / MFMA");
  run<0, OP_DMA, 1, 12>("16x16x32 + 1 LDS-DMA piece / 12 MFMA");
  run<0, OP_DMA, 1, 6>("16x16x32 + 1 LDS-DMA piece / 6 MFMA");
  run<0, OP_BARRIER, 1, 48>("16x16x32 + 1 s_barrier / 48 MFMA");
  run<1, OP_NONE, 0, 1>("32x32x16 alone");
  run<1, OP_FMA, 2, 1>("32x32x16 + 2 fma / MFMA");
  run<1, OP_FMA, 4, 1>("32x32x16 + 4 fma / MFMA");
  run<1, OP_FMA, 6, 1>("32x32x16 + 6 fma / MFMA");
  run<1, OP_FMA, 8, 1>("32x32x16 + 8 fma / MFMA");
  run<1, OP_CVT, 4, 1>("32x32x16 + 4 cvt_pkrtz / MFMA");
  run<1, OP_EXP, 2, 1>("32x32x16 + 2 exp / MFMA");
  run<1, OP_EXP, 4, 1>("32x32x16 + 4 exp / MFMA");
  run<1, OP_LDS, 2, 3>("32x32x16 + 2 ds_read_b128 / 3 MFMA");
  run<1, OP_DMA, 1, 6>("32x32x16 + 1 LDS-DMA piece / 6 MFMA");
  run<1, OP_DMA, 1, 3>("32x32x16 + 1 LDS-DMA piece / 3 MFMA");
  run<1, OP_BARRIER, 1, 24>("32x32x16 + 1 s_barrier / 24 MFMA");
  return 0;
}
