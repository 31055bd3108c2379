// Microbenchmark (gfx950, round 6): balance.hip with the OTHER f16 MFMA shape, v_mfma_f32_32x32x16_f16 -- twice the FLOP per MFMA issue (an MFMA
// holds the vector issue port 8 cycles in either shape, MI355X_MICROARCH.md): a pass = one 32-unit tile of the hidden layer for a 32-sample wave
// = NM = 42 MFMAs (14 k = 16 chunks x 3 products, a fragment pair feeds 3 MFMAs) + NPAIR = 8 register-pair activations.  Compare per 32 hidden
// units: two passes of balance.hip's <42, 4> against one pass of <42, 8> here.  Also prints the clock held (s_memtime / s_memrealtime) after
// ~1 s of back-to-back launches.
// (round 4's text follows)  how much vector work per MFMA does a pass of flow_kernel_hx3 absorb?
// One "pass" = NM v_mfma_f32_16x16x32_f16 (two accumulation chains, A fragments from LDS two units ahead, as the kernel) with
// NPAIR register-pair activations (2 v_exp, v_pk_add, 2 v_rcp, v_cvt_pkrtz, 2 v_fma_mix, v_cvt_pkrtz = 52 issue cycles by the
// guide's prices) spread evenly behind the MFMAs, two chains interleaved; one s_barrier per pass.  Printed: cycles per pass for
// 8-wave workgroups (two waves per SIMD: the time in which BOTH do a pass; pipe bound 2 NM x 16.5) and 4-wave ones (a lone wave).
// The shipped hidden pass is NM = 42, NPAIR = 4; a stream that carried ALL of a flow step's vector work evenly would be ~6.9.
//   hipcc --offload-arch=gfx950 -O3 -o balance balance.hip && ./balance
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
#define FENCE() __builtin_amdgcn_sched_barrier(0)
static int WARM = 3;      // launches, the last one is read
static int PASSES_RT = 64; // passes per launch (argv[2]: 400000 = ~0.5 s per launch: the clock the chip HOLDS under the stream)

struct PairState { float a, b; f32x2 e; float r0, r1; unsigned hi, mid; };
template <int STEP>
__device__ __forceinline__ void pair_step(PairState& p) {
  if constexpr (STEP == 0) asm volatile("v_exp_f32 %0, %1" : "=v"(p.e[0]) : "v"(p.a));
  if constexpr (STEP == 1) asm volatile("v_exp_f32 %0, %1" : "=v"(p.e[1]) : "v"(p.b));
  if constexpr (STEP == 2) asm volatile("v_pk_add_f32 %0, %1, 1.0 op_sel_hi:[1,0]" : "=v"(p.e) : "v"(p.e));
  if constexpr (STEP == 3) asm volatile("v_rcp_f32 %0, %1" : "=v"(p.r0) : "v"(p.e[0]));
  if constexpr (STEP == 4) asm volatile("v_rcp_f32 %0, %1" : "=v"(p.r1) : "v"(p.e[1]));
  if constexpr (STEP == 5) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(p.hi) : "v"(p.r0), "v"(p.r1));
  if constexpr (STEP == 6) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(p.a) : "v"(p.hi), "v"(p.r0));
  if constexpr (STEP == 7) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(p.b) : "v"(p.hi), "v"(p.r1));
  if constexpr (STEP == 8) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(p.mid) : "v"(p.a), "v"(p.b));
}
// instruction K of the VALU stream of NP pairs: pairs interleaved two by two
template <int K, int NP>
__device__ __forceinline__ void stream_step(PairState (&ps)[NP > 0 ? NP : 1]) {
  constexpr int duo = K / 18, r = K % 18;
  constexpr int pair = (2 * duo + 1 < NP) ? 2 * duo + (r % 2) : 2 * duo;            // an odd last pair runs alone
  constexpr int step = (2 * duo + 1 < NP) ? r / 2 : K - 18 * duo;
  if constexpr (pair < NP && step < 9) pair_step<step>(ps[pair]);
}
template <int K0, int K1, int NP>
__device__ __forceinline__ void stream_range(PairState (&ps)[NP > 0 ? NP : 1]) {
  if constexpr (K0 < K1) {
    stream_step<K0, NP>(ps);
    FENCE();
    stream_range<K0 + 1, K1, NP>(ps);
  }
}
template <int M, int NM, int NV, int NP, typename F>
__device__ __forceinline__ void spread(PairState (&ps)[NP > 0 ? NP : 1], F&& mfma1) {
  if constexpr (M < NM) {
    mfma1(M);
    stream_range<(M * NV) / NM, ((M + 1) * NV) / NM, NP>(ps);
    spread<M + 1, NM, NV, NP>(ps, mfma1);
  }
}

#ifdef SHAPE16
#define DEF_MPU 6
#else
#define DEF_MPU 3
#endif
template <int NM, int NPAIR, int FRAGS, int MPU = DEF_MPU, int MAXT = 512>
__global__ void __launch_bounds__(MAXT) k(float* out, unsigned long long* cyc, int seed, int passes) {
  const int lane = threadIdx.x & 63;
  extern __shared__ __attribute__((aligned(16))) unsigned frag_lds[];
  constexpr int NU = NM / MPU;
  // operands with random mantissas and signs (f16 pairs in [0.5, 2)): the power an MFMA draws depends on its data (MI355X_MICROARCH.md, DVFS)
  auto rnd = [](unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; };
  auto f16pair = [&](unsigned x) { const unsigned r = rnd(x); return (r & 0x83ff83ffu) | 0x38003800u | ((r >> 3) & 0x04000400u); };
  for (int w = threadIdx.x; w < 2 * NU * 256; w += blockDim.x) frag_lds[w] = f16pair(w * 7u + seed + blockIdx.x * 977u);
#ifdef SHAPE16
  f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#else
  f32x16 acc[2];
  for (int q = 0; q < 16; ++q) { acc[0][q] = 0.f; acc[1][q] = 0.f; }
#endif
  f16x8 bv[2];
  for (int i = 0; i < 2; ++i) {
    u32x4 t = u32x4{f16pair(seed + i + threadIdx.x * 31u), f16pair(threadIdx.x * 131u + i), f16pair(threadIdx.x * 17u + 5u * i + 3u), f16pair(threadIdx.x + 1000u * i)};
    bv[i] = __builtin_bit_cast(f16x8, t);
  }
  PairState ps[NPAIR > 0 ? NPAIR : 1];
  for (int q = 0; q < (NPAIR > 0 ? NPAIR : 1); ++q) { ps[q].a = 0.1f + 0.01f * q + lane * 1e-4f; ps[q].b = 0.2f + 0.01f * q; ps[q].hi = ps[q].mid = 0; }
  __syncthreads();
  u32x4 A[3][2];
  auto load_unit = [&](int slot, int u) {
    if (FRAGS) {
      A[slot][0] = *reinterpret_cast<const u32x4*>(frag_lds + (2 * u) * 256 + lane * 4);
      A[slot][1] = *reinterpret_cast<const u32x4*>(frag_lds + (2 * u + 1) * 256 + lane * 4);
    }
  };
  if (!FRAGS) for (int s = 0; s < 3; ++s) for (int h = 0; h < 2; ++h) A[s][h] = *reinterpret_cast<const u32x4*>(frag_lds + (2 * s + h) * 256 + lane * 4);
  auto mfma1 = [&](int m) {
    const int u = m / MPU, j = m % MPU;
    if (j == 0 && u + 2 < NU) load_unit((u + 2) % 3, u + 2);
#ifdef SHAPE16
    acc[j & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[u % 3][j < MPU / 3 ? 1 : 0]), bv[j & 1], acc[j & 1], 0, 0, 0);
#else
    acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[u % 3][j == 0 ? 1 : 0]), bv[j == 2 ? 1 : 0], acc[u & 1], 0, 0, 0);
#endif
    FENCE();
  };
  unsigned long long t0, t1, r0, r1;
  FENCE();
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  FENCE();
  const int PASSES = passes;
  for (int p = 0; p < PASSES; ++p) {
    load_unit(0, 0); load_unit(1, 1);
    spread<0, NM, 9 * NPAIR, NPAIR>(ps, mfma1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  FENCE();
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  FENCE();
  unsigned sink = 0;
  for (int q = 0; q < (NPAIR > 0 ? NPAIR : 1); ++q) sink ^= ps[q].hi ^ ps[q].mid;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[0][3] + acc[1][2] + __builtin_bit_cast(float, sink & 0x3fffffffu);
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = (t1 - t0) / PASSES;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[256 * 16] = t1 - t0; cyc[256 * 16 + 1] = r1 - r0; }
}

template <int NM, int NPAIR, int FRAGS, int MPU = DEF_MPU, int MAXT = 512>
static void run(const char* what, std::initializer_list<int> wave_counts = {8, 4}) {
  for (int waves : wave_counts) {
    const int blocks = 256 * (waves == 4 ? 1 : 1);
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 1024 * 4); hipMalloc(&cyc, (blocks * 16 + 2) * 8);
    hipFuncSetAttribute((const void*)k<NM, NPAIR, FRAGS, MPU, MAXT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int rep = 0; rep < WARM; ++rep) hipLaunchKernelGGL((k<NM, NPAIR, FRAGS, MPU, MAXT>), dim3(blocks), dim3(64 * waves), 2 * (NM / MPU) * 1024, 0, out, cyc, rep, PASSES_RT);
    hipDeviceSynchronize();
    static unsigned long long h[256 * 16 + 2];
    hipMemcpy(h, cyc, blocks * waves * 8, hipMemcpyDeviceToHost);
    hipMemcpy(h + 256 * 16, cyc + 256 * 16, 16, hipMemcpyDeviceToHost);
    double s = 0; unsigned long long mx = 0;
    for (int i = 0; i < blocks * waves; ++i) { s += h[i]; mx = h[i] > mx ? h[i] : mx; }
    printf("%-34s NM %2d NPAIR %2d  %d waves/SIMD: %7.1f cycles per pass (max %llu)   per MFMA of the SIMD %5.2f   vector issue per MFMA %4.1f   clock %.2f GHz\n", what, NM, NPAIR,
           waves / 4, s / (blocks * waves), mx, s / (blocks * waves) / (NM * (waves / 4)), 52.0 * NPAIR / NM, 0.1 * (double)h[256 * 16] / (double)(h[256 * 16 + 1] ? h[256 * 16 + 1] : 1));
    hipFree(out); hipFree(cyc);
  }
}

int main(int argc, char** argv) {
  if (argc > 1) WARM = atoi(argv[1]);
  if (argc > 2) PASSES_RT = atoi(argv[2]);
#ifdef SHAPE16          // the shipped shape on the same harness (-DSHAPE16): per 32 hidden units TWO passes of <42, 4>
  run<42, 0, 1>("16x16x32: mfma + LDS frags");
  run<42, 4, 1>("16x16x32: spread (hidden pass, 16 units)");
  run<42, 7, 1>("16x16x32: spread (a whole step's mix)");
  run<12, 8, 1>("16x16x32: layer-0 mix (2 tiles)");
#else
  run<42, 0, 1>("32x32x16: mfma + LDS frags");
  run<42, 4, 1>("32x32x16: spread");
  run<42, 8, 1>("32x32x16: spread (hidden pass, 32 units)");
  run<42, 10, 1>("32x32x16: spread");
  run<42, 12, 1>("32x32x16: spread");
  run<42, 14, 1>("32x32x16: spread (a whole step's mix)");
  run<42, 8, 0>("32x32x16: spread, fragments in registers");
  run<6, 8, 1>("32x32x16: layer-0 mix (32 units)");
#endif
  return 0;
}
