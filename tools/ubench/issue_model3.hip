// Microbenchmark (gfx950), third pass: operand kinds (inline constants, SGPRs, literals) and mixes; WHICH forms co-issue with v_mfma_f32_16x16x32_f16?
// (issue_model.hip found v_fma_f32 fully additive at ~4 cycles each, v_cvt_pkrtz_f16_f32 / v_fma_mix_f32 / one
// transcendental per MFMA free.)  One workgroup: 256 threads = one wave per SIMD, 512 = two per SIMD (same program);
// cycles of the SLOWEST wave per MFMA-of-one-wave.  Every filler has VGPR-only operands.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using i32x4 = __attribute__((ext_vector_type(4))) int;
#define FENCE() __builtin_amdgcn_sched_barrier(0)

#define OPS(X) \
  X(NONE, "") \
  X(ADD_VV, "v_add_f32 %0, %1, %2") \
  X(ADD_INLINE, "v_add_f32 %0, 1.0, %1") \
  X(ADD_SGPR, "v_add_f32 %0, %4, %1") \
  X(FMA_VVV, "v_fma_f32 %0, %1, %2, %3") \
  X(FMA_INLINE, "v_fma_f32 %0, %1, -2.0, 1.0") \
  X(FMA_SGPR, "v_fma_f32 %0, %1, %4, %3") \
  X(FMAAK, "v_fmaak_f32 %0, %1, %2, 0x40490fdb") \
  X(MUL_LIT, "v_mul_f32 %0, 0x40490fdb, %1") \
  X(EXP_ADD, "v_exp_f32 %0, %1\n\tv_add_f32 %0, %0, %2") \
  X(EXP_2ADD, "v_exp_f32 %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %3") \
  X(EXP_RCP, "v_exp_f32 %0, %1\n\tv_rcp_f32 %0, %0") \
  X(TANHSEQ, "v_exp_f32 %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_rcp_f32 %0, %0\n\tv_fma_f32 %0, %0, %3, %2") \
  X(CMP, "v_cmp_lt_f32 vcc, %1, %2\n\tv_mov_b32 %0, %1") \
  X(ACCREAD, "v_accvgpr_write_b32 a0, %1\n\tv_accvgpr_read_b32 %0, a0")

enum Op {
#define X(n, s) OP_##n,
  OPS(X)
#undef X
  OP_COUNT
};

template <int OP>
__device__ __forceinline__ void filler(float& d, float a, float b, float c, float sc) {
#define X(n, s) \
  if constexpr (OP == OP_##n && OP != OP_NONE) asm volatile(s : "=v"(d) : "v"(a), "v"(b), "v"(c), "s"(sc) : "vcc");
  OPS(X)
#undef X
}

template <int OP, int NOPS>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, float a, float b, int seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = a + i * 0.001f + threadIdx.x * 1e-4f;
  float va = a + lane * 1e-6f, vb = b + lane * 1e-6f;     // VGPR copies of the scalars
  asm volatile("" : "+v"(va), "+v"(vb));
  f16x8 av[2], bv[2];
  for (int i = 0; i < 2; ++i) {
    i32x4 t = i32x4{0x3c003c00 + seed + i, 0x38003800 + (int)threadIdx.x, 0x34003400 + i, 0x30003000};
    av[i] = __builtin_bit_cast(f16x8, t);
    t[0] += 17;
    bv[i] = __builtin_bit_cast(f16x8, t);
  }
  __syncthreads();
  unsigned long long t0, t1;
  FENCE();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  FENCE();
  constexpr int ITERS = 32, M = 48;
  for (int it = 0; it < ITERS; ++it) {
    int vi = 0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[m & 1], bv[m & 1], acc[m & 3], 0, 0, 0);
      FENCE();
#pragma unroll
      for (int q = 0; q < NOPS; ++q) {
        filler<OP>(v[(vi + 8) & 15], v[vi & 15], va, vb, a);
        ++vi;
        FENCE();
      }
    }
  }
  FENCE();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  FENCE();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[threadIdx.x] = s;
  if (lane == 0) cyc[wave] = t1 - t0;
}

static float* g_out; static unsigned long long* g_cyc;

template <int OP, int NOPS>
void run(const char* name) {
  double r[2];
  for (int t = 0; t < 2; ++t) {
    const int threads = t ? 512 : 256;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<OP, NOPS>), dim3(1), dim3(threads), 0, 0, g_out, g_cyc, 1.0001f, 0.5f, 12345);
    (void)hipDeviceSynchronize();
    unsigned long long c[8]; (void)hipMemcpy(c, g_cyc, 64, hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int w = 0; w < threads / 64; ++w) mx = c[w] > mx ? c[w] : mx;
    r[t] = mx / 1536.0;
  }
  printf("%-12s x%d per MFMA:  1 wave/SIMD %6.1f cyc/MFMA   2 waves/SIMD (slowest) %6.1f = %5.1f per MFMA of the SIMD\n", name, NOPS, r[0], r[1], r[1] / 2);
}

int main() {
  (void)hipMalloc(&g_out, 512 * 4); (void)hipMalloc(&g_cyc, 64);
  run<OP_NONE, 0>("none");
#define X(n, s) \
  if (OP_##n != OP_NONE) { run<OP_##n, 1>(#n); run<OP_##n, 2>(#n); run<OP_##n, 4>(#n); }
  OPS(X)
#undef X
  return 0;
}
