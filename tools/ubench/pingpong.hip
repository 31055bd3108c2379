// Microbenchmark (gfx950): two waves per SIMD, the instruction mix of one hidden-layer pass of flow_kernel_hx3
// (42 v_mfma_f32_16x16x32_f16 in two dependent chains + the activation/split of 4 register pairs = 24 plain VALU +
// 16 transcendentals), scheduled two ways:
//   MIXED     both waves run the interleaved stream the kernel ships today (one pair's VALU between the MFMAs of a unit),
//             one s_barrier per pass;
//   PINGPONG  the halves of the workgroup alternate roles: waves 0-3 issue their 42 MFMAs back to back while waves 4-7 do
//             their VALU work, s_barrier, roles swap, s_barrier  (an in-order wave whose next instruction is an MFMA that
//             waits for the partner's MFMA cannot issue the VALU behind it: separating the streams avoids that)
// Prints cycles per PASS-PAIR (both waves of a SIMD have done one pass each); the pipe-bound value is 84 x 16.7 = 1403.
//   hipcc --offload-arch=gfx950 -O3 -o pingpong pingpong.hip && ./pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using i32x4 = __attribute__((ext_vector_type(4))) int;
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// one register pair: 2 exp, pk_add (as 2 adds here), 2 rcp, cvt_pk, 2 fma_mix-like fma, cvt_pk  = 6 VALU + 4 trans
__device__ __forceinline__ void pair_work(float& a, float& b, unsigned& o0, unsigned& o1) {
  float e0, e1, r0, r1, s0, s1;
  asm volatile("v_exp_f32 %0, %1" : "=v"(e0) : "v"(a));
  asm volatile("v_exp_f32 %0, %1" : "=v"(e1) : "v"(b));
  asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(e0) : "v"(e0));
  asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(e1) : "v"(e1));
  asm volatile("v_rcp_f32 %0, %1" : "=v"(r0) : "v"(e0));
  asm volatile("v_rcp_f32 %0, %1" : "=v"(r1) : "v"(e1));
  asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(o0) : "v"(r0), "v"(r1));
  asm volatile("v_fma_f32 %0, %1, -1.0, %2" : "=v"(s0) : "v"(r0), "v"(a));
  asm volatile("v_fma_f32 %0, %1, -1.0, %2" : "=v"(s1) : "v"(r1), "v"(b));
  asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(o1) : "v"(s0), "v"(s1));
  asm volatile("v_fma_f32 %0, %1, 0.5, 0.5" : "=v"(a) : "v"(s0));
  asm volatile("v_fma_f32 %0, %1, 0.5, 0.5" : "=v"(b) : "v"(s1));
}


struct PairState { float a, b, e0, e1, r0, r1, s0, s1; unsigned o0, o1; };
template <int STEP>
__device__ __forceinline__ void pair_step(PairState& p) {        // the same 14 instructions as pair_work, one at a time
  if constexpr (STEP == 0) asm volatile("v_exp_f32 %0, %1" : "=v"(p.e0) : "v"(p.a));
  if constexpr (STEP == 1) asm volatile("v_exp_f32 %0, %1" : "=v"(p.e1) : "v"(p.b));
  if constexpr (STEP == 2) asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(p.e0) : "v"(p.e0));
  if constexpr (STEP == 3) asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(p.e1) : "v"(p.e1));
  if constexpr (STEP == 4) asm volatile("v_rcp_f32 %0, %1" : "=v"(p.r0) : "v"(p.e0));
  if constexpr (STEP == 5) asm volatile("v_rcp_f32 %0, %1" : "=v"(p.r1) : "v"(p.e1));
  if constexpr (STEP == 6) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(p.o0) : "v"(p.r0), "v"(p.r1));
  if constexpr (STEP == 7) asm volatile("v_fma_f32 %0, %1, -1.0, %2" : "=v"(p.s0) : "v"(p.r0), "v"(p.a));
  if constexpr (STEP == 8) asm volatile("v_fma_f32 %0, %1, -1.0, %2" : "=v"(p.s1) : "v"(p.r1), "v"(p.b));
  if constexpr (STEP == 9) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(p.o1) : "v"(p.s0), "v"(p.s1));
  if constexpr (STEP == 10) asm volatile("v_fma_f32 %0, %1, 0.5, 0.5" : "=v"(p.a) : "v"(p.s0));
  if constexpr (STEP == 11) asm volatile("v_fma_f32 %0, %1, 0.5, 0.5" : "=v"(p.b) : "v"(p.s1));
}
// instruction K of the pass's VALU stream: pairs interleaved two by two (pair = 2*(K/24) + K%2, step = (K%24)/2): two independent chains in flight
template <int K>
__device__ __forceinline__ void stream_step(PairState (&ps)[4]) {
  constexpr int pair = 2 * (K / 24) + (K % 2), step = (K % 24) / 2;
  pair_step<step>(ps[pair]);
}
template <int K0, int K1>
__device__ __forceinline__ void stream_range(PairState (&ps)[4]) {
  if constexpr (K0 < K1) {
    stream_step<K0>(ps);
    __builtin_amdgcn_sched_barrier(0);
    stream_range<K0 + 1, K1>(ps);
  }
}
template <int M, int NM, int NV, typename F>
__device__ __forceinline__ void spread(PairState (&ps)[4], F&& mfma1) {       // MFMA m, then its share of the NV instructions
  if constexpr (M < NM) {
    mfma1(M);
    stream_range<(M * NV) / NM, ((M + 1) * NV) / NM>(ps);
    spread<M + 1, NM, NV>(ps, mfma1);
  }
}

template <int MODE, int CH>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, int seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[CH];
  for (int i = 0; i < CH; ++i) acc[i] = f32x4{0, 0, 0, 0};
  f16x8 av[2], bv[2];
  for (int i = 0; i < 2; ++i) {
    i32x4 t = i32x4{0x3c003c00 + seed + i, 0x38003800 + (int)threadIdx.x, 0x34003400 + i, 0x30003000};
    av[i] = __builtin_bit_cast(f16x8, t);
    t[0] += 17;
    bv[i] = __builtin_bit_cast(f16x8, t);
  }
  float va[8];
  for (int i = 0; i < 8; ++i) va[i] = 0.1f + 0.01f * i + lane * 1e-4f;
  unsigned sink = 0;
  __syncthreads();
  unsigned long long t0, t1;
  FENCE();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  FENCE();
  constexpr int PASSES = 64;
  auto mfma_unit = [&](int u) {          // one consumption unit: 6 MFMAs, two chains
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      acc[m % CH] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[(u + m) & 1], bv[m & 1], acc[m % CH], 0, 0, 0);
      FENCE();
    }
  };
  auto valu_pair = [&](int q) {
    unsigned o0, o1;
    pair_work(va[2 * q], va[2 * q + 1], o0, o1);
    sink ^= o0 ^ o1;
    FENCE();
  };
  if (MODE == 5 || MODE == 7) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
  if (MODE == 10) {
    // as MODE 9, with the VALU stream hand-spread: one instruction of two interleaved pair chains behind every MFMA
    extern __shared__ __attribute__((aligned(16))) unsigned frag_lds[];
    for (int w = threadIdx.x; w < 14 * 256; w += blockDim.x) frag_lds[w] = 0x3c003c00u + w;
    __syncthreads();
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    u32x4 A[3][2];
    auto load_unit = [&](int slot, int u) {
      A[slot][0] = *reinterpret_cast<const u32x4*>(frag_lds + (2 * u) * 256 + lane * 4);
      A[slot][1] = *reinterpret_cast<const u32x4*>(frag_lds + (2 * u + 1) * 256 + lane * 4);
    };
    PairState ps[4];
    for (int q = 0; q < 4; ++q) { ps[q].a = va[2 * q]; ps[q].b = va[2 * q + 1]; }
    auto mfma1 = [&](int m) {
      const int u = m / 6, j = m % 6;
      if (j == 0 && u + 2 < 7) load_unit((u + 2) % 3, u + 2);
      acc[j & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[u % 3][j < 2 ? 1 : 0]), bv[j & 1], acc[j & 1], 0, 0, 0);
      FENCE();
    };
    for (int p = 0; p < PASSES; ++p) {
      load_unit(0, 0); load_unit(1, 1);
      spread<0, 42, 48>(ps, mfma1);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    for (int q = 0; q < 4; ++q) { va[2 * q] = ps[q].a; va[2 * q + 1] = ps[q].b; sink ^= ps[q].o0 ^ ps[q].o1; }
  } else if (MODE == 9) {
    // the A operands come from LDS like in the kernel: 2 fragments (hi, mid) per unit, read two units ahead
    extern __shared__ __attribute__((aligned(16))) unsigned frag_lds[];
    for (int w = threadIdx.x; w < 14 * 256; w += blockDim.x) frag_lds[w] = 0x3c003c00u + w;
    __syncthreads();
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    u32x4 A[3][2];
    auto load_unit = [&](int slot, int u) {
      A[slot][0] = *reinterpret_cast<const u32x4*>(frag_lds + (2 * u) * 256 + lane * 4);
      A[slot][1] = *reinterpret_cast<const u32x4*>(frag_lds + (2 * u + 1) * 256 + lane * 4);
    };
    for (int p = 0; p < PASSES; ++p) {
      load_unit(0, 0); load_unit(1, 1);
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        if (u + 2 < 7) load_unit((u + 2) % 3, u + 2);
        if (u < 4) valu_pair(u);
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[u % 3][m < 2 ? 1 : 0]), bv[m & 1], acc[m & 1], 0, 0, 0);
          FENCE();
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  } else if (MODE == 6) {
    for (int p = 0; p < PASSES; ++p) {
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        if (u < 4) valu_pair(u);
        __builtin_amdgcn_s_setprio(1);
        mfma_unit(u);
        __builtin_amdgcn_s_setprio(0);
      }
      asm volatile("s_barrier" ::: "memory");
    }
  } else if (MODE == 8) {
    for (int p = 0; p < PASSES; ++p) {
      if (wave >= 4) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        if (u == 3) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        if (u < 4) valu_pair(u);
        mfma_unit(u);
      }
      asm volatile("s_barrier" ::: "memory");
    }
  } else if (MODE == 0 || MODE == 5) {
    for (int p = 0; p < PASSES; ++p) {
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        if (u < 4) valu_pair(u);
        mfma_unit(u);
      }
      asm volatile("s_barrier" ::: "memory");
    }
  } else if (MODE == 4 || MODE == 7) {      // mixed, the VALU stream spread evenly: 48 instructions behind 42 MFMAs
    PairState ps[4];
    for (int q = 0; q < 4; ++q) { ps[q].a = va[2 * q]; ps[q].b = va[2 * q + 1]; }
    auto mfma1 = [&](int m) {
      acc[m % CH] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[m & 1], bv[(m >> 1) & 1], acc[m % CH], 0, 0, 0);
      FENCE();
    };
    for (int p = 0; p < PASSES; ++p) {
      spread<0, 42, 48>(ps, mfma1);
      asm volatile("s_barrier" ::: "memory");
    }
    for (int q = 0; q < 4; ++q) { va[2 * q] = ps[q].a; va[2 * q + 1] = ps[q].b; sink ^= ps[q].o0 ^ ps[q].o1; }
  } else if (MODE == 2) {      // MFMAs only, both waves
    for (int p = 0; p < PASSES; ++p) {
#pragma unroll
      for (int u = 0; u < 7; ++u) mfma_unit(u);
      asm volatile("s_barrier" ::: "memory");
    }
  } else if (MODE == 3) {      // VALU only, both waves
    for (int p = 0; p < PASSES; ++p) {
#pragma unroll
      for (int q = 0; q < 4; ++q) valu_pair(q);
      asm volatile("s_barrier" ::: "memory");
    }
  } else {
    const bool first_half = wave < 4;
    for (int p = 0; p < PASSES; ++p) {
      if (first_half) {
#pragma unroll
        for (int u = 0; u < 7; ++u) mfma_unit(u);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) valu_pair(q);
      }
      asm volatile("s_barrier" ::: "memory");
      if (first_half) {
#pragma unroll
        for (int q = 0; q < 4; ++q) valu_pair(q);
      } else {
#pragma unroll
        for (int u = 0; u < 7; ++u) mfma_unit(u);
      }
      asm volatile("s_barrier" ::: "memory");
    }
  }
  FENCE();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  FENCE();
  float s = 0;
  for (int i = 0; i < CH; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += va[i];
  out[threadIdx.x] = s + (float)sink;
  if (lane == 0) cyc[wave] = t1 - t0;
}

template <int MODE, int CH>
static void run(const char* name, float* out, unsigned long long* cyc, int THREADS = 512) {
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<MODE, CH>), dim3(256), dim3(THREADS), 14 * 1024, 0, out, cyc, 12345);
  (void)hipDeviceSynchronize();
  unsigned long long c[8];
  (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
  unsigned long long mx = 0;
  for (int w = 0; w < THREADS / 64; ++w) mx = c[w] > mx ? c[w] : mx;
  printf("%-26s %7.1f cycles per pass of every wave   per wave:", name, mx / 64.0);
  for (int w = 0; w < THREADS / 64; ++w) printf(" %5.0f", c[w] / 64.0);
  printf("\n");
}

int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 512 * 4); (void)hipMalloc(&cyc, 64);
  run<0, 2>("mixed ch2", out, cyc);
  run<0, 3>("mixed ch3", out, cyc);
  run<0, 6>("mixed ch6", out, cyc);
  run<1, 2>("pingpong2", out, cyc);
  run<1, 6>("pingpong6", out, cyc);
  run<2, 2>("mfma-only 2", out, cyc);
  run<2, 6>("mfma-only 6", out, cyc);
  run<2, 2>("mfma-only 2, 1 wave/SIMD", out, cyc, 256);
  run<2, 6>("mfma-only 6, 1 wave/SIMD", out, cyc, 256);
  run<0, 2>("mixed ch2, 1 wave/SIMD", out, cyc, 256);
  run<0, 6>("mixed ch6, 1 wave/SIMD", out, cyc, 256);
  run<3, 2>("valu-only", out, cyc);
  run<4, 2>("mixed, VALU spread evenly", out, cyc);
  run<4, 2>("spread, 1 wave/SIMD", out, cyc, 256);
  run<5, 2>("mixed, waves 4-7 at prio 1", out, cyc);
  run<6, 2>("mixed, prio 1 around MFMAs", out, cyc);
  run<8, 2>("mixed, prio flips mid pass", out, cyc);
  run<7, 2>("spread, waves 4-7 at prio 1", out, cyc);
  run<9, 2>("mixed + LDS fragment reads", out, cyc);
  run<10, 2>("spread + LDS fragment reads", out, cyc);
  run<10, 2>("spread + LDS frags, 1 wave/SIMD", out, cyc, 256);
  run<9, 2>("mixed + LDS frags, 1 wave/SIMD", out, cyc, 256);
  return 0;
}
