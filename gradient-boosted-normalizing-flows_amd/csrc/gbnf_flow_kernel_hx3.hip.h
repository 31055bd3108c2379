// gbnf_flow_kernel_hx3.hip.h -- the fused flow kernel on the f16 / bf16 matrix pipe with split-f32 operands.
//
// Why: on gfx950 the f32-input MFMA (v_mfma_f32_16x16x4_f32) runs at the f32 VECTOR rate and blocks the
// vector ALU for its whole duration (tools/ubench/mfma_f32_issue.hip: 32 cycles per 2048 FLOP, VALU fully
// additive).  v_mfma_f32_16x16x32_{f16,bf16} delivers 16384 FLOP in ~17 cycles.  An f32 value x is split into
// pieces of the narrow type and a product is evaluated as a sum of narrow MFMAs with f32 accumulation (every
// narrow x narrow product is exact in f32):
//
//   PREC 0, "f16x3":  x ~= hi + mid, hi = f16(x), mid = f16(x - hi)   (22 of 24 significand bits, fp16 RANGE)
//                     a.b = a_mid.b_hi + a_hi.b_mid + a_hi.b_hi                             (3 MFMAs)
//   PREC 1, "bf16x6": x ~= p0 + p1 + p2, p_k = bf16_rne(x - p_0 - .. - p_{k-1})   (>= 24 bits, f32 RANGE)
//                     a.b = a2.b0 + a0.b2 + a1.b1 + a1.b0 + a0.b1 + a0.b0; the dropped terms are < 2^-25 |a.b|  (6 MFMAs)
//
// f16x3 is the fast path (end to end ~1e-7 relative in the log-likelihood on well-conditioned models); bf16x6 is
// f32-faithful for any finite input and any model the f32 reference itself evaluates accurately.  A f16x3 launch
// marks every sample whose operands left the fp16 range (NaN in its outputs); the library follows it with a bf16x6
// "repair" launch of the same grid whose workgroups exit at once unless they own a marked sample (gbnf_api.hip).
// For tanh networks 2*log2(e) is folded into the packed weights and biases of the layers that feed a tanh, and
// t = 1 - 2 r into the layer that consumes it: a tanh costs exp, add, rcp (tanh_hx3 below).
//
// Structure:
//   * a workgroup is WAVES (8, or 4 when a wave needs more than 256 registers) waves; each wave owns 16*NT samples and
//     its own LDS feature tile Z; all work on the SAME component, so the weight stream is fetched once per
//     workgroup: the packed fragments go L2 -> LDS by direct-to-LDS DMA (global_load_lds_dwordx4, 1 KiB per
//     wave-instruction) one stage ahead into the other of two staging slots (the blob is laid out in consumption order,
//     so "the next stage" is a running pointer with a compile-time size; a deeper ring was measured and bought nothing:
//     profiles/r2_*); every wave reads its A operands with conflict-free lane-linear ds_read_b128, two consumption units
//     ahead.  One s_barrier per stage.  Biases travel with a net's first stage into a double-buffered LDS area.
//   * D layout == B layout: two consecutive 16-unit accumulator tiles, after the activation and the split, ARE the
//     B operand (k = 32) of the next layer's chunk; nothing is shuffled or stored.
//   * the activation + split of tile u-1 (VALU) is issued between the MFMAs of tile u.
#pragma once

#include <type_traits>

#include "gbnf_flow_kernel.hip.h"

namespace gbnf {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int HX3_RING = 2;         // stage slots in LDS: the stage in use and the next one

constexpr int hx3_pieces(int prec) { return prec == 0 ? 2 : 3; }

// Packed layout of one coupling network for the hx3 kernel, in 32-bit words.  A "fragment" is the A operand of one
// v_mfma_f32_16x16x32_* for one 16-row tile: [64 lanes][8 narrow values] = 256 words; every weight tile is NP
// consecutive fragments (its pieces, largest first).  Stages are contiguous and in consumption order:
//   biases   : layer 0 [HT][16] | hidden layer j = 1..DEPTH [HT][16] each | output layer [OT][16] (f32), padded to whole
//              fragments (BIAS_FRAGS)
//   L0 stages: layer-0 tiles [0,TL0), [TL0,2 TL0) ...                                  TL0 = HC + OT tiles per stage
//   DEPTH >= 1, per hidden layer j:
//     PASS u : hidden row u, chunks c = 0..HC-1; then, in the LAST hidden layer, if u is even and u >= 2, the output-layer
//              chunk (u-2)/2: tiles o = 0..OT-1 (it is consumed in pass u)
//     DRAIN  : output-layer chunk HC-1
//   DEPTH == 0 (the output layer reads layer 0's activations): OUT stages of CG = TL0 / OT chunks, OT tiles each
struct Hx3Layout {
  static constexpr int MAXS = 144;
  int NP, HC, TL0, N_L0, NS, BIAS_FRAGS, BIAS_WORDS, NET_WORDS, STAGE_FRAGS, DEPTH, CG, N_OUT;
  int off[MAXS];   // word offset of stage s from the start of the net block
  int nf[MAXS];    // fragments in stage s
  constexpr Hx3Layout(int HT, int OT, int np, int depth = 1)
      : NP(np), HC((HT + 1) / 2), TL0((HT + 1) / 2 + OT), N_L0(0), NS(0), BIAS_FRAGS((((depth + 1) * HT + OT) * 16 + 255) / 256),
        BIAS_WORDS(0), NET_WORDS(0), STAGE_FRAGS(0), DEPTH(depth), CG(1), N_OUT(0), off{}, nf{} {
    BIAS_WORDS = BIAS_FRAGS * 256;
    N_L0 = (HT + TL0 - 1) / TL0;
    CG = TL0 / OT;
    int s = 0, w = BIAS_WORDS;
    for (int i = 0; i < N_L0; ++i) {
      const int t0 = i * TL0;
      const int cnt = (HT - t0 < TL0) ? HT - t0 : TL0;
      off[s] = w; nf[s] = NP * cnt; w += nf[s] * 256; ++s;
    }
    for (int j = 1; j <= depth; ++j)
      for (int u = 0; u < HT; ++u) {
        off[s] = w; nf[s] = NP * HC + ((j == depth && u % 2 == 0 && u >= 2) ? NP * OT : 0); w += nf[s] * 256; ++s;
      }
    if (depth >= 1) {
      off[s] = w; nf[s] = NP * OT; w += nf[s] * 256; ++s;
    } else {
      N_OUT = (HC + CG - 1) / CG;
      for (int k = 0; k < N_OUT; ++k) {
        const int cnt = (HC - k * CG < CG) ? HC - k * CG : CG;
        off[s] = w; nf[s] = NP * OT * cnt; w += nf[s] * 256; ++s;
      }
    }
    NS = s;
    NET_WORDS = w;
    for (int k = 0; k < s; ++k) STAGE_FRAGS = nf[k] > STAGE_FRAGS ? nf[k] : STAGE_FRAGS;
  }
};

template <int HT, int OT, int NP, int DEPTH>
struct Hx3LayoutOf {
  static constexpr Hx3Layout value = Hx3Layout(HT, OT, NP, DEPTH);
};

// Waves per workgroup: 8 (two per SIMD, 256 registers each) unless the register-resident hidden layer of a wave
// (HC chunks x NT tiles x NP pieces x 4 registers) needs the 512-register budget of one wave per SIMD.
constexpr int hx3_reg_estimate(int HT, int OT, int NT, int prec, int kind, int act_a, int act_b, int depth = 1) {
  const int np = hx3_pieces(prec);
  const int hc = (HT + 1) / 2;
  const int nn = kind == GBNF_KIND_REALNVP ? 2 : 1;
  const int relu = (act_a != GBNF_ACT_TANH || act_b != GBNF_ACT_TANH) ? 28 : 0;   // measured: ReLU / per-step variants keep more values live
  const int accs = np == 3 ? 3 : 1;                    // running sums per output tile (Products<NP>::NACC)
  // (a second hidden layer keeps the first one's activations AND its own, both as B operands)
  // a ResidualNet keeps layer 0's raw output tiles; two blocks (three fully unrolled middle layers): measured 135 registers over the
  // 256 of two waves per SIMD at 7 hidden tiles -- one wave per SIMD from the smallest geometry on
  const int res = act_a == 2 ? HT * NT * 4 + (depth == 4 ? 100 : 0) : 0;
  return (depth >= 2 ? 2 : 1) * hc * NT * np * 4 + (nn + accs - 1) * OT * NT * 4 + NT * 4 * (1 + accs) + 2 * NT * np * 4 + 3 * np * 4 + 36 + relu + res;
}
// TRAIN kernels (trace + operand saves, ~40 more registers per sample tile): 32-sample waves run one per SIMD with the
// 512-register budget, in 4-wave workgroups (a lone 32-sample wave does a pass in the time two co-resident 16-sample waves
// take for theirs together with half the LDS reads and DMA issue: the forward sweep of large batches)
constexpr bool hx3_train_wide(int HT, int OT, int NT, int prec, int kind, int act_a, int act_b, int depth, int train) {
  return train != 0 && NT == 2 && hx3_reg_estimate(HT, OT, 2, prec, kind, act_a, act_b, depth) + 80 <= 440;
}
constexpr int hx3_waves(int HT, int OT, int NT, int prec, int kind, int act_a, int act_b, int depth = 1, int train = 0) {
  if (hx3_train_wide(HT, OT, NT, prec, kind, act_a, act_b, depth, train)) return 4;
#ifdef GBNF_HX3_FORCE_WAVES       // experiment knob: 4 = two independent 4-wave workgroups per CU (where registers and LDS allow)
  return hx3_reg_estimate(HT, OT, NT, prec, kind, act_a, act_b, depth) <= 248 ? GBNF_HX3_FORCE_WAVES : 4;
#else
  return hx3_reg_estimate(HT, OT, NT, prec, kind, act_a, act_b, depth) <= 248 ? 8 : 4;
#endif
}
#ifndef GBNF_HX3_F16_CHAINS_NT1
#define GBNF_HX3_F16_CHAINS_NT1 1   // f16x3 running sums per hidden tile of a 16-sample wave: 3 = one per product (round 6, measured: 43.6 us
                                    // against 41.0 us for the single chain on one log_prob launch at 64-2048 rows: the extra zero-inits and
                                    // adds cost more than the dependent-MFMA latency they hide -- gpurun_out/latency_ablate.txt); 1 = shipped
#endif
#ifndef GBNF_HX3_F16_CHAINS_NT2
#define GBNF_HX3_F16_CHAINS_NT2 1   // ... of a 32-sample wave (two sample tiles: two chains already)
#endif
// f16x3 running sums per hidden tile (flow_kernel_hx3, `HCH`): three for 16-sample waves of the geometries with registers to spare
// (ResidualNets and the widest nets already spill: they keep the single sum)
constexpr int hx3_f16_hidden_chains(int HT, int OT, int NT, int kind, int act_a, int act_b, int depth) {
  if (NT != 1) return GBNF_HX3_F16_CHAINS_NT2;
  return (act_a != 2 && hx3_reg_estimate(HT, OT, 1, 0, kind, act_a, act_b, depth) <= 200) ? GBNF_HX3_F16_CHAINS_NT1 : 1;
}
// minimum waves per SIMD the kernel is compiled for (the register budget): 2 (256 registers) or 1 (512)
constexpr int hx3_waves_per_simd(int HT, int OT, int NT, int prec, int kind, int act_a, int act_b, int depth = 1, int train = 0) {
  if (hx3_train_wide(HT, OT, NT, prec, kind, act_a, act_b, depth, train)) return 1;
  return hx3_reg_estimate(HT, OT, NT, prec, kind, act_a, act_b, depth) <= 248 ? 2 : 1;
}
// Samples per wave actually compiled for a requested NT: 32-sample waves (NT = 2) of the widest geometries would
// spill even with one wave per SIMD; their NT = 2 entry runs the 16-sample kernel.
constexpr int hx3_eff_nt(int HT, int OT, int NT, int prec, int kind, int act_a, int act_b, int depth = 1, int train = 0) {
  if (train != 0 && NT == 2) return hx3_train_wide(HT, OT, NT, prec, kind, act_a, act_b, depth, train) ? 2 : 1;
  return (NT == 2 && hx3_reg_estimate(HT, OT, 2, prec, kind, act_a, act_b, depth) > 300) ? 1 : NT;
}

// round-4 knobs, each A/B'd on one box (tools/build_ab.sh + tools/ab_bench.py, profiles/r4_ab_headline_knobs.txt); the defaults are
// the shipped forms
#ifndef GBNF_HX3_MIXLO
#define GBNF_HX3_MIXLO 0          // 1: the f16 residual piece is converted and packed by v_fma_mixlo_f16 / v_fma_mixhi_f16 (measured: 1.2 % SLOWER)
#endif
#ifndef GBNF_HX3_DMA_RUNS
#define GBNF_HX3_DMA_RUNS 1       // 1: a wave stages a RUN of consecutive fragments per stage: one M0 / base per 4 pieces (immediate offsets; +0.2 %)
#endif
#ifndef GBNF_HX3_DMA_AT
#define GBNF_HX3_DMA_AT 0         // n > 0: a hidden pass issues the next stage's DMA in front of its unit n (0: at the top of the pass; measured: no change)
#endif
#ifndef GBNF_HX3_EDGE
#define GBNF_HX3_EDGE 1           // 1: the step boundary (coupling epilogue, input normalisation) as direction-specialised straight-line code (+1.3 %)
#endif


// ---- operand splitting ----------------------------------------------------------------------------------------
// f16: hi = f16(x) (toward zero), mid = f16(x - hi); 4 VALU ops per register pair: the residual x - hi is ONE
// v_fma_mix_f32 per value (the f16 half is widened inside the instruction; LLVM itself would emit v_cvt_f32_f16 + v_sub_f32)
__device__ __forceinline__ void split_pair_f16(float x0, float x1, unsigned (&p)[2]) {
  const auto h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
  const unsigned hw = __builtin_bit_cast(unsigned, h);
  p[0] = hw;
  float r0, r1;
#ifdef GBNF_HX3_NO_FMAMIX          // diagnostic: the residual as LLVM writes it (v_cvt_f32_f16 + v_sub_f32)
  r0 = x0 - (float)h[0];
  r1 = x1 - (float)h[1];
#else
  // IN PLACE (the residual overwrites x): hipcc's hazard recognizer does not see into inline asm, so an asm OUTPUT must never
  // be a fresh register -- the allocator may hand it a register that a v_mfma issued a few instructions earlier is still
  // going to write (its dead result), and nothing pads that write-after-write (tools/isa_hazard_lint.py found two such
  // places, 6 wait states behind the v_mfma where 7 are needed).  x's register was last written by a VALU instruction the
  // compiler did see.
  r0 = x0;
  r1 = x1;
#if GBNF_HX3_MIXLO
  // round 4: the residuals are converted to f16 and packed BY the fma (v_fma_mixlo_f16 / v_fma_mixhi_f16 write the low / high
  // half of the destination and keep the other half): two instructions instead of two v_fma_mix_f32 + v_cvt_pkrtz_f16_f32.
  // The residual x - hi is exact in f32; its conversion rounds to nearest here (toward zero before): hi + mid is the closer sum.
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "+v"(r0)
      : "v"(hw), "v"(r1));
  p[1] = __builtin_bit_cast(unsigned, r0);
  return;
#endif
  asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(r0) : "v"(hw));
  asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r1) : "v"(hw));
#endif
  const auto m = __builtin_amdgcn_cvt_pkrtz(r0, r1);
  p[1] = __builtin_bit_cast(unsigned, m);
}
// bf16: three round-to-nearest pieces (v_cvt_pk_bf16_f32); 11 VALU ops per register pair
__device__ __forceinline__ void split_pair_bf16(float x0, float x1, unsigned (&p)[3]) {
  const unsigned w0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
  p[0] = w0;
  const float r0 = x0 - __builtin_bit_cast(float, w0 << 16), r1 = x1 - __builtin_bit_cast(float, w0 & 0xffff0000u);
  const unsigned w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
  p[1] = w1;
  const float s0 = r0 - __builtin_bit_cast(float, w1 << 16), s1 = r1 - __builtin_bit_cast(float, w1 & 0xffff0000u);
  p[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}
template <int NP>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned (&p)[NP]) {
  if constexpr (NP == 2) split_pair_f16(x0, x1, p);
  else split_pair_bf16(x0, x1, p);
}

// tanh networks: the packer folds 2*log2(e) into every layer that feeds a tanh and the affine map t = 1 - 2 r into the
// layer that consumes it (W.t + b = (-2W).r + (b + W.1)), so the value handed on is r = 1/(2^y + 1), tanh(y') = 1 - 2 r:
// v_exp_f32, v_add_f32, v_rcp_f32
__device__ __forceinline__ float tanh_hx3(float y) {
  const float e = __builtin_amdgcn_exp2f(y);
  return __builtin_amdgcn_rcpf(e + 1.0f);
}

template <int PREC>
__device__ __forceinline__ f32x4 mfma_narrow(u32x4 a, u32x4 b, f32x4 c) {
#ifdef GBNF_ABLATE_MFMA           // diagnostic: one cheap VALU op instead of the MFMA (keeps every value live)
  c[0] += __builtin_bit_cast(float, a[0] ^ b[0]);
  return c;
#else
  if constexpr (PREC == 0)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}

// the products of one f32 product, (weight piece, activation piece), smallest terms first
template <int NP> struct Products;
// ACC: which running sum a product goes to.  bf16x6 keeps one sum per magnitude class (a0.b0 | 2^-9 terms | 2^-18 terms) and
// adds them once per output tile: an MFMA adds its 32 products to the accumulator at the accumulator's scale, which would
// cost the small classes most of their bits (measured: 1e-5 instead of 2e-6 on the ill-conditioned fixture g15).
template <> struct Products<2> { static constexpr int N = 3, NACC = 1; static constexpr int W[3] = {1, 0, 0}; static constexpr int X[3] = {0, 1, 0}; static constexpr int ACC[3] = {0, 0, 0}; };
template <> struct Products<3> { static constexpr int N = 6, NACC = 3; static constexpr int W[6] = {2, 0, 1, 1, 0, 0}; static constexpr int X[6] = {0, 2, 1, 0, 1, 0}; static constexpr int ACC[6] = {2, 2, 2, 1, 1, 0}; };

// the running sums of one 16x16 output tile
template <int NACC>
struct AccT {
  static constexpr int N = NACC;
  f32x4 s[NACC];
  __device__ __forceinline__ void init(f32x4 bias) {
    s[0] = bias;
#pragma unroll
    for (int k = 1; k < NACC; ++k) s[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  __device__ __forceinline__ f32x4 total() const {
    if constexpr (NACC == 1) return s[0];
    else return s[0] + (s[1] + s[2]);
  }
};

// s_waitcnt vmcnt(n) for a run-time n (an immediate in the instruction): wait until at most n of this wave's
// vector-memory operations are outstanding (in issue order), n clamped to [0, 8]
// One direct-to-LDS load: lane l moves 16 bytes from src + 16 l to lds_dst + 16 l.  Issued through inline asm on purpose: with
// __builtin_amdgcn_global_load_lds anywhere in a kernel, hipcc (ROCm 7.2) stops counting LDS reads -- EVERY wait for a
// ds_read becomes s_waitcnt lgkmcnt(0), i.e. each MFMA group waits for the fragment reads issued just before it instead of
// the ones it consumes (round 3, tools/ubench/lgkmcnt_dma.hip: counted waits come back when the compiler does not see the
// DMA).  The compiler then knows nothing of these loads: their completion is the stage-end `s_waitcnt vmcnt` + barrier, which
// the kernels did by hand already, and its own vmcnt counts only ever over-wait (hidden operations are extra younger ones).
// M0 = LDS address of the wave's 1 KiB piece; `saddr + 32-bit lane offset` form: no 64-bit vector address arithmetic.
#ifndef GBNF_DMA_ASM
#define GBNF_DMA_ASM 1
#endif
__device__ __forceinline__ void lds_dma16(const __attribute__((address_space(1))) uint32_t* src, uint32_t* lds_dst, unsigned lane_b16) {
#if GBNF_DMA_ASM
  const unsigned m0v = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_dst;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_b16), "s"(src), "s"(m0v) : "memory", "m0");
#else
  const __attribute__((address_space(1))) char* base = reinterpret_cast<const __attribute__((address_space(1))) char*>(src);
  __builtin_amdgcn_global_load_lds(base + lane_b16, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
#endif
}

// CNT (1..4) consecutive 1 KiB pieces with ONE M0 and ONE base: the instruction's immediate offset is added to the global AND
// to the LDS address (round 4: the per-piece s_mov m0 / s_nop / 64-bit base arithmetic was 4 of the 5 instructions a piece cost)
template <int CNT>
__device__ __forceinline__ void lds_dma16_run(const __attribute__((address_space(1))) uint32_t* src, uint32_t* lds_dst, unsigned lane_b16) {
  static_assert(CNT >= 1 && CNT <= 4, "immediate offsets 0 .. 3072");
  const unsigned m0v = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_dst;
  if constexpr (CNT == 1)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_b16), "s"(src), "s"(m0v) : "memory", "m0");
  else if constexpr (CNT == 2)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024"
                 ::"v"(lane_b16), "s"(src), "s"(m0v) : "memory", "m0");
  else if constexpr (CNT == 3)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:2048" ::"v"(lane_b16), "s"(src), "s"(m0v) : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072"
                 ::"v"(lane_b16), "s"(src), "s"(m0v) : "memory", "m0");
}

// a stage end that leaves the n youngest vector-memory operations of the wave in flight (n folds to a constant; even values,
// an odd or larger one is rounded down: waiting for more is always safe)
__device__ __forceinline__ void stage_wait_counted(int n) {
#define GBNF_STAGE_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
  if (n >= 24) GBNF_STAGE_WAIT(24);
  else if (n >= 20) GBNF_STAGE_WAIT(20);
  else if (n >= 16) GBNF_STAGE_WAIT(16);
  else if (n >= 12) GBNF_STAGE_WAIT(12);
  else if (n >= 8) GBNF_STAGE_WAIT(8);
  else if (n >= 6) GBNF_STAGE_WAIT(6);
  else if (n >= 4) GBNF_STAGE_WAIT(4);
  else if (n >= 2) GBNF_STAGE_WAIT(2);
  else GBNF_STAGE_WAIT(0);
#undef GBNF_STAGE_WAIT
}
__device__ __forceinline__ void wait_vmcnt(int n) {
  if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

template <int KIND, int HT, int OT, int NT, int ACTA, int ACTB, int PREC, int WV, int DEPTH, int TRAIN = 0>
__global__ void __launch_bounds__(64 * WV, hx3_waves_per_simd(HT, OT, NT, PREC, KIND, ACTA, ACTB, DEPTH, TRAIN))
flow_kernel_hx3(const FlowLaunch p) {
  static_assert((DEPTH >= 0 && DEPTH <= 2) || (DEPTH == 4 && ACTA == 2), "coupling_network_depth 0, 1 or 2; ResidualNets of one or two blocks");
  // ACT == 2 (GBNF_ACT_RESIDUAL_RELU): a ResidualNet of ONE block (models/layers.py:246-301) = layer 0 (no activation of its
  // own) -> [relu -> Linear -> relu -> Linear] + layer 0's output -> final layer: the two-hidden-layer pipeline with the
  // raw layer-0 tiles kept in registers, added to the second hidden layer's output and split WITHOUT an activation
  static_assert((ACTA == 2) == (ACTB == 2), "both nets of a step are ResidualNets or neither is");
  static_assert(ACTA != 2 || DEPTH == 2 || DEPTH == 4, "a ResidualNet has two hidden -> hidden layers per block");
  // Two blocks (round 5, evaluation only): DEPTH = 4.  The hidden -> hidden layers 1 .. DEPTH - 1 run as "middle" layers that ping-pong
  // between the two register sets of B operands; behind layer 2 (the first block's second Linear) the skip sum t = a2 + t0 replaces
  // the raw layer-0 tiles AND is what the second block starts from (relu(t)); the last layer's output + t goes to the final layer.
  // (TRAIN at DEPTH = 4: relu(a0), relu(a1), relu(t1), relu(a3) and t2 go to the operand rows of "hidden layers" 0 .. 4)
  static_assert(!TRAIN || PREC == 0, "the training forward runs on f16x3");
  constexpr int WAVES = WV;
  constexpr int NP = hx3_pieces(PREC);
  constexpr int NPROD = Products<NP>::N;
  using Acc = AccT<Products<NP>::NACC>;
  // Running sums of a HIDDEN tile (a tile lives for one pass: 3 HC dependent v_mfma per sample tile).  Round 6 experiment
  // (GBNF_HX3_F16_CHAINS_NT1 = 3): one sum per product, three independent chains for a 16-sample wave alone on its SIMD --
  // measured SLOWER than the single chain (see the knob): the latency form is not bound by the dependent-MFMA latency.
  constexpr int HCH = NP == 3 ? 3 : hx3_f16_hidden_chains(HT, OT, NT, KIND, ACTA, ACTB, DEPTH);
  using AccH = AccT<HCH>;
  auto acc_of = [](auto n_c, int pr) constexpr {
    constexpr int NA = decltype(n_c)::value;
    if constexpr (NA == Products<NP>::NACC) return Products<NP>::ACC[pr];
    else return pr % NA;
  };
  constexpr int ZS = 16 * NT + 1;
  constexpr int NNETS = (KIND == GBNF_KIND_REALNVP) ? 2 : 1;
  using LT = Hx3LayoutOf<HT, OT, NP, DEPTH>;          // LT::value: the layout (a static member: usable inside the lambdas below)
  constexpr int HC = LT::value.HC;
  constexpr int STEP_WORDS = SMALL_WORDS + NNETS * LT::value.NET_WORDS;
  constexpr int STAGE_WORDS = LT::value.STAGE_FRAGS * 256;
  constexpr bool WATCH = PREC == 0;        // the fp16 range can be left; bf16 pieces have the range of f32

  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int i = lane & 15;
  const int g = lane >> 4;

  // ---- a repair launch that has nothing to repair ends here: no launch from the one it follows on has raised the mark
  //      slot to its serial number (FlowLaunch::seq) and the handle's numerics guard has not tripped
  const bool redo_all = PREC == 1 && p.repair && p.guard != nullptr && p.guard[0] != 0u;
  if (p.repair && !redo_all) {
    if (p.repair == 2) return;
    if (p.sat != nullptr && p.sat[SAT_MARKS + p.seq % SAT_SLOTS] < p.seq) return;
  }

  // ---- work items: (component, batch, group of WAVES sample tiles).  A normal launch has one workgroup per item; a
  //      repair launch walks the items with a small grid and skips those without a marked sample.
  const int n_groups = (p.n_tiles + WAVES - 1) / WAVES;
#ifdef GBNF_CLOCK                 // diagnostic (tools/clock_probe.py): the clock this workgroup ran at = d s_memtime / d s_memrealtime x 100 MHz
  unsigned long long ck_c0 = 0, ck_r0 = 0;                                   // (MI355X_MICROARCH.md, "DVFS give-back" item 6)
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ck_c0), "=s"(ck_r0)::"memory");
#endif
  // (only the bf16x6 kernels serve repair launches: for the f16x3 instantiations the update is the constant -1, a single pass)
  for (int item = blockIdx.x; item >= 0;
       item = (PREC == 1 && p.repair && item + (int)gridDim.x < p.n_items) ? item + (int)gridDim.x : -1) {
  if (item != (int)blockIdx.x) __syncthreads();      // the previous item's LDS contents are dead
  int comp, grp, batch;
  {
    // XCD-aware: items are dealt round-robin over the 8 XCDs (as the hardware deals workgroups); each XCD gets a
    // contiguous run of the component-major work list
    const int total = p.n_items;
    const int b = item;
    const int xcd = b & 7, j = b >> 3;
    const int base = total >> 3, rem = total & 7;
    const int q = xcd * base + (xcd < rem ? xcd : rem) + j;
    const int per_comp = n_groups * p.n_batches;    // work list: component-major, then batch, then tile group
    comp = q / per_comp;
    const int r = q - comp * per_comp;
    batch = r / n_groups;
    grp = r - batch * n_groups;
  }
  const uint32_t* __restrict__ blob = p.blobs[p.c_begin + comp];
  const int d = p.d;
  // rows >= n are masked everywhere.  TRAIN: the last workgroup's spare waves own no rows; they shadow the last tile group
  // (same values stored twice) so that every wave issues the same vector-memory operations per stage (the counted stage waits)
  const int64_t row0_raw = ((int64_t)grp * WAVES + wave) * (16 * NT);
  const int64_t row0 = (TRAIN && row0_raw >= p.np) ? p.np - 16 * NT : row0_raw;
  const float* __restrict__ xin = p.xs[batch];
  const int64_t out_base = (int64_t)comp * p.out_stride + (int64_t)batch * p.n;
  // TRAIN: lane offsets into the trace ([slot][np]) and the operand regions ([16-sample tile][row][16]); 32-bit: one
  // (step, net) region holds net_rows * np < 2^31 floats for np up to 2 M rows
  [[maybe_unused]] constexpr bool tr_ok = true;
  [[maybe_unused]] const int tr_np = (int)p.np, tr_row = (int)row0 + (lane & 15);
  [[maybe_unused]] const int tr_ip = p.tr_ip, tr_hp = p.tr_hp, tr_ip16 = p.tr_ip * 16, tr_hp16 = p.tr_hp * 16, tr_op16 = p.tr_op * 16;
  [[maybe_unused]] const int tr_net_stride = p.net_rows * (int)p.np;
  [[maybe_unused]] const int tr_in_off = (int)(row0 >> 4) * tr_ip16 + 8 * (lane >> 4) * 16 + (lane & 15);      // net-input row 8 g + e
  [[maybe_unused]] const int tr_h_off = (int)(row0 >> 4) * tr_hp16 + 4 * (lane >> 4) * 16 + (lane & 15);       // hidden unit 16 t + 4 g + r
  [[maybe_unused]] const int tr_o_off = (int)(row0 >> 4) * tr_op16 + 4 * (lane >> 4) * 16 + (lane & 15);       // output row 16 o + 4 g + r

  // ---- repair launch: only items that own a sample marked by the f16x3 launch (NaN in its outputs) are evaluated
  if (p.repair && !redo_all) {
    bool need = false;
    if (lane < 16 * NT) {
      const int64_t n = row0 + lane;
      if (n < p.n) {
        float v;
        if (p.ll_out) v = p.ll_out[out_base + n];
        else if (p.ldj_out) v = p.ldj_out[out_base + n];
        else v = p.z_out[((int64_t)comp * p.n + n) * d];
        need = v != v;
      }
    }
    // workgroup-wide OR through the first LDS word (no static LDS: the dynamic allocation may be all 160 KB)
    if (threadIdx.x == 0) lds[0] = 0u;
    __syncthreads();
    if (__any(need) && lane == 0) lds[0] = 1u;
    __syncthreads();
    const bool any_need = lds[0] != 0u;
    __syncthreads();             // before the word is re-used
    if (!any_need) continue;
  }

  // ---- experiment knob (FlowLaunch::stagger, off by default): of two 4-wave workgroups on a CU the one in the SIMDs' odd
  //      wave slots starts late, so that its VALU-bound stretch at every step boundary (coupling epilogue, input
  //      preparation, layer 0: ~1/3 of a step at < 30 % matrix-pipe use, profiles/r2_timeline_f16x3_8wave.txt) could
  //      fall into the other one's MFMA-bound hidden passes.  Measured: no gain.
  if (p.stagger > 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) lds[0] = hw & 1u;             // WAVE_ID: the wave's slot on its SIMD
    __syncthreads();
    const bool late = lds[0] != 0u;
    __syncthreads();
    if (late)
      for (int k = 0; k < p.stagger; ++k) __builtin_amdgcn_s_sleep(32);       // 32 x 64 cycles each
  }

  // ---- LDS: per-step tables | 2 bias blocks | ring of stage slots | Z tiles
  const bool lds_tables = p.lds_tables != 0;
  constexpr int ring = 2;                                   // stage slots: the current stage and the next one
  uint32_t* SM = lds;
  uint32_t* BIAS = lds + (lds_tables ? p.n_steps * SMALL_WORDS : 0);
  uint32_t* STG = BIAS + 2 * LT::value.BIAS_WORDS;
  float* Z = reinterpret_cast<float*>(STG + ring * STAGE_WORDS) + wave * ((d + 1) * ZS);   // wave-private: d feature slots + a spare one (writes of dead table entries)

  // ---- weight staging, one stage ahead into the other of two slots.  The blob is laid out in consumption order, so
  //      "the next stage" is a running pointer; every call site knows the next stage's fragment count at compile time.
  //      Wave w moves fragments w, w + WAVES, ...; the LDS destination of a wave-instruction is base + lane*16 = the
  //      fragment.  A net's biases (BIAS_FRAGS fragments in front of its first stage) go to their own double-buffered area.
  using gwords = const __attribute__((address_space(1))) uint32_t*;
  using lptr = __attribute__((address_space(3))) void*;
  // z -> x (FlowLaunch::inverse, round 3; the exact-f32 kernel's scheme, gbnf_flow_kernel.hip.h): the steps are visited
  // K-1 .. 0 -- within a step the nets keep their order, so only the step-to-step hop of the running pointer changes --,
  // the in-half is read as the (already normalised) net input and un-normalised in place, the out-half gets coupling^-1
  // and norm^-1, log|det| accumulates with the opposite sign, z comes in through the FINAL slot map, x leaves through
  // the initial one (slot j = feature j)
  const bool inv = p.inverse != 0;
  // TRAIN: a step range [kb, ke) of the sweep (FlowLaunch::k_begin / k_end; the whole flow by default)
  const int kb = TRAIN ? p.k_begin : 0;
  const int ke = (TRAIN && p.k_end > 0) ? p.k_end : p.n_steps;
  const int first_step = inv ? p.n_steps - 1 : kb;
  gwords next_src = (gwords)blob + (size_t)first_step * STEP_WORDS + SMALL_WORDS;    // bias block of the first step's net 0
  int gs = 0;                                       // stage counter: slot = gs & 1
  // TRAIN: operand stores this wave has issued BEHIND the staging DMA of the stage in flight.  They may stay in flight across
  // the stage-end barrier (vmcnt counts in issue order: `s_waitcnt vmcnt(tr_later)` covers the DMA and everything older) --
  // waiting for them too made every stage as long as a store's round trip to HBM (16 us per flow step at N = 4096).  The
  // value is a compile-time constant at every stage end (straight-line code between issue() and stage_end()); only stores
  // that are issued unconditionally are counted (an undercount only waits longer); tools/isa_hazard_lint.py re-counts the
  // vector-memory instructions between every staging DMA and its counted wait in the shipped ISA.
  [[maybe_unused]] int tr_later = 0;
  int nets_issued = 0;                              // nets whose first stage has been issued (bias buffer = & 1)
  const unsigned lane_b16 = (unsigned)lane * 16u;
  auto dma = [&](gwords src, uint32_t* dst) {
#ifndef GBNF_ABLATE_DMA          // diagnostic: no weight staging at all (stale LDS contents, timing only)
    lds_dma16(src, dst, lane_b16);
#else
    (void)src; (void)dst;
#endif
  };
  // (GBNF_HX3_DMA_RUNS) wave w stages the run [w BASE + min(w, R), ...) of BASE (+1 if w < R) consecutive fragments, four per M0 / base
  auto dma_runs = [&](auto nf_c, gwords src, uint32_t* dst) {
    constexpr int NF = decltype(nf_c)::value;
    constexpr int BASE = NF / WAVES, R = NF % WAVES;
#ifndef GBNF_ABLATE_DMA
    const int start = wave * BASE + (wave < R ? wave : R);
    gwords s0 = src + start * 256;
    uint32_t* d0 = dst + start * 256;
    auto group = [&](auto g_c) {
      constexpr int G = decltype(g_c)::value;
      constexpr int lo = (BASE - 4 * G) < 0 ? 0 : ((BASE - 4 * G) > 4 ? 4 : (BASE - 4 * G));              // pieces of group G without ...
      constexpr int hi = (BASE + 1 - 4 * G) < 0 ? 0 : ((BASE + 1 - 4 * G) > 4 ? 4 : (BASE + 1 - 4 * G));  // ... and with the extra one
      if constexpr (R == 0 || lo == hi) {
        if constexpr (lo > 0) lds_dma16_run<lo>(s0 + G * 1024, d0 + G * 1024, lane_b16);
      } else {
        if (wave < R) lds_dma16_run<hi>(s0 + G * 1024, d0 + G * 1024, lane_b16);
        else if constexpr (lo > 0) lds_dma16_run<lo>(s0 + G * 1024, d0 + G * 1024, lane_b16);
      }
    };
    static_assert(BASE + 1 <= 16, "at most four groups of four pieces per wave and stage");
    group(std::integral_constant<int, 0>{});
    if constexpr (BASE + (R > 0) > 4) group(std::integral_constant<int, 1>{});
    if constexpr (BASE + (R > 0) > 8) group(std::integral_constant<int, 2>{});
    if constexpr (BASE + (R > 0) > 12) group(std::integral_constant<int, 3>{});
#else
    (void)src; (void)dst;
#endif
  };
  auto issue = [&](auto nf_c, int into) {           // the next stage: NF fragments at next_src
    constexpr int NF = decltype(nf_c)::value;
    uint32_t* dst = STG + (into & 1) * STAGE_WORDS;
#if GBNF_HX3_DMA_RUNS
    dma_runs(nf_c, next_src, dst);
#else
#pragma unroll
    for (int k = 0; k * WAVES < NF; ++k) {
      const int f = wave + k * WAVES;
      if ((k + 1) * WAVES <= NF || f < NF) dma(next_src + f * 256, dst + f * 256);
    }
#endif
    next_src += NF * 256;
    if constexpr (TRAIN) {
      __builtin_amdgcn_sched_barrier(0);           // nothing that is counted below moves in front of the DMA
      tr_later = 0;
    }
  };
  auto issue_net_start = [&](int into) {            // next_src points at a net's bias block: biases + first L0 stage
    uint32_t* bdst = BIAS + (nets_issued & 1) * LT::value.BIAS_WORDS;
#if GBNF_HX3_DMA_RUNS
    dma_runs(std::integral_constant<int, LT::value.BIAS_FRAGS>{}, next_src, bdst);
#else
#pragma unroll
    for (int k = 0; k * WAVES < LT::value.BIAS_FRAGS; ++k) {
      const int f = wave + k * WAVES;
      if ((k + 1) * WAVES <= LT::value.BIAS_FRAGS || f < LT::value.BIAS_FRAGS) dma(next_src + f * 256, bdst + f * 256);
    }
#endif
    next_src += LT::value.BIAS_WORDS;
    ++nets_issued;
    issue(std::integral_constant<int, LT::value.nf[0]>{}, into);
  };

  // ---- per-step tables -> LDS, x tile -> Z, the first stage in flight
  issue_net_start(0);
  if (lds_tables) {
    for (int s = 0; s < p.n_steps; ++s) {
      const uint32_t* src = blob + (size_t)s * STEP_WORDS;
      for (int w = (int)threadIdx.x * 4; w < SMALL_WORDS; w += 64 * WAVES * 4)
        *reinterpret_cast<i32x4*>(SM + s * SMALL_WORDS + w) = *reinterpret_cast<const i32x4*>(src + w);
    }
  }
  if (TRAIN && p.state_in != nullptr) {
    // the state parked by the launch of the previous step range: slot layout [d][np], 16 NT consecutive rows per slot
    constexpr int RW = 16 * NT, SPP = 64 / RW;       // rows of the wave's tile, slots per pass
    const int r = lane % RW, s0 = lane / RW;
    const float* sin = p.state_in + row0 + r;
    for (int slot = s0; slot < d; slot += SPP) Z[slot * ZS + r] = sin[(int64_t)slot * p.np];
  } else if (lane < d) {
    // every row's load in flight before the first LDS store (branch-free: rows past the batch re-read its last row and are
    // zeroed): eight at a time cost a global round trip per group at the head of every work item
    float xv[16 * NT];
    const int64_t last = p.n - 1;
#pragma unroll
    for (int r = 0; r < 16 * NT; ++r) {
      const int64_t n = row0 + r;
      xv[r] = xin[(n < p.n ? n : last) * d + lane];
    }
    const int slot0 = inv ? (int)blob[(size_t)p.n_steps * STEP_WORDS + lane] : lane;      // (the tail table: final slot of feature j)
#pragma unroll
    for (int r = 0; r < 16 * NT; ++r) Z[slot0 * ZS + r] = (row0 + r < p.n) ? xv[r] : 0.0f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the first stages have landed
  __syncthreads();               // tables + Z visible, every wave's pieces landed

  float ld[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) ld[nt] = 0.0f;
  float ld_const = 0.0f;
  [[maybe_unused]] float ld2[NT];  // (GBNF_HX3_EDGE) affine Glow coupling: sum of log2(1 + exp(-(raw + 2))); log|det| gets -ln 2 times it
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) ld2[nt] = 0.0f;
  bool sat[NT];                  // sample (i, nt) stored an operand beyond the fp16 range
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) sat[nt] = false;
  Stamps st;
  st.start();

  // ---- consumer side: slot gs & 1 holds the current stage
  int cnets = 0;                 // nets consumed so far (bias buffer = cnets & 1)
  const uint32_t* buf = STG;
  auto stage_end = [&]() {
#ifdef GBNF_TIMELINE              // diagnostic: absolute s_memtime stamps of workgroup 0 per wave and stage: [wave][stage][before wait | after barrier]
    unsigned long long tl0_ = 0, tl1_ = 0;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl0_)::"memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef GBNF_STAMPS
    const int phase_ = st.cur;
    st.mark(phase_);             // compute time of the stage so far -> its phase bucket
#endif
#ifndef GBNF_ABLATE_BARRIER       // diagnostic: no per-stage rendezvous (races on the staging buffers, timing only)
    // this wave's pieces of the next stage have landed, all waves are done with this slot
    if constexpr (TRAIN) stage_wait_counted(tr_later);
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
#ifdef GBNF_STAMPS
    st.mark(7);                  // bucket 7: time in the stage-end wait + barrier
    st.cur = phase_;
#endif
#ifdef GBNF_TIMELINE
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl1_)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    if (p.dbg != nullptr && blockIdx.x == 0 && lane == 0 && gs < 128) {
      p.dbg[(wave * 128 + gs) * 2 + 0] = tl0_;
      p.dbg[(wave * 128 + gs) * 2 + 1] = tl1_;
    }
#endif
    ++gs;
  };
#ifdef GBNF_ABLATE_FRAG            // diagnostic: one fragment read per kernel, reused for every MFMA (timing only)
  u32x4 frag_dummy = *reinterpret_cast<const u32x4*>(STG + lane * 4);
  asm volatile("" : "+v"(frag_dummy));
  auto frag = [&](int) -> u32x4 { return frag_dummy; };
#else
  auto frag = [&](int f) -> u32x4 { return *reinterpret_cast<const u32x4*>(buf + f * 256 + lane * 4); };
#endif

  // A consumption unit = one weight tile = NP fragments.  Units of a stage are read two ahead of their use; the first
  // two units of the NEXT stage are read right behind the stage-end barrier, in front of the last unit's MFMAs of the
  // stage that ends (N0, N1): all waves of a workgroup leave the barrier together, and without matrix work in hand every
  // SIMD would idle for the LDS latency at the top of every stage.
  struct Unit { u32x4 w[NP]; };
  auto load_unit = [&](Unit& a, int n) {
#pragma unroll
    for (int q = 0; q < NP; ++q) a.w[q] = frag(n * NP + q);
  };
  Unit N0, N1;
  auto preload = [&]() {
    buf = STG + (gs & 1) * STAGE_WORDS;
    load_unit(N0, 0);
    load_unit(N1, 1);
  };
  // MFMA-tail guard.  ROOT CAUSE (round 3, profiles/r3_mfma_hazard_root_cause.txt): reading a v_mfma result needs 7 wait
  // states behind the v_mfma (no hardware interlock; measured, tools/ubench/mfma_raw_latency.hip), and hipcc pads them only
  // along the FALL-THROUGH path of a conditional branch.  With a stage ending in front of its last unit, the last hidden
  // pass's last v_mfma (its result `pre` is the last hidden tile) is followed by the drain's
  // `if (another net or step follows) issue_net_start()`: on the fall-through side the staging DMA's address arithmetic
  // supplies the wait states, on the TAKEN side -- the last step of every component -- the drain's first v_exp_f32 read
  // `pre` 1-2 wait states behind the v_mfma and got the previous contents of the registers whenever the instruction fetch
  // at the branch target was fast: whole 16-sample tiles off by 5e-2, another set of waves on every launch (fixture
  // g5_glow_d63_h128_c2, 100 % of launches in tools/tail_repro.py).  Only the code layout of that one variant put the
  // reader right at the branch target, which is why nothing else failed.  The fix is the padding the compiler owes: eight
  // idle wait states behind the last unit of every stage that ends early (whatever follows, on whichever path, is then
  // >= 8 wait states behind the last v_mfma), no measurable cost -- and tools/isa_hazard_lint.py (tests/test_isa_lint.py)
  // walks the control-flow graph of EVERY shipped kernel and fails the build if any path reads a v_mfma result early.
  // (Round 2's reading -- a VALU / LDS write into the dead srcC registers -- was wrong: the hardware interlocks those,
  // tools/ubench/mfma_srcc_war.hip, mfma_srcab_war.hip, mfma_lds_war.hip: 0 wrong results at 0 wait states under load.)
#ifndef GBNF_HX3_TAIL_MODE
#define GBNF_HX3_TAIL_MODE 1        // experiment knob (tools/build_tail_experiments.sh): 0 = no guard, 1 = s_nop 7 (shipped), 2 = the last
#endif                              //   unit's A operands pinned live behind its v_mfma (no guard), 3 = both
  auto mfma_tail_guard = [&]() {
#if GBNF_HX3_TAIL_MODE == 1 || GBNF_HX3_TAIL_MODE == 3
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
  };
  auto guard_at = [&](int pos) {              // experiment: the guard's idle wait states at another place of the following pass
#if GBNF_HX3_TAIL_MODE >= 4
    if (pos == GBNF_HX3_TAIL_MODE) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 7" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
#else
    (void)pos;
#endif
  };
  auto pin_unit = [&](const Unit& a) {        // (TAIL_MODE 2 / 3) the unit's registers stay live up to here
#if GBNF_HX3_TAIL_MODE == 2 || GBNF_HX3_TAIL_MODE == 3
#pragma unroll
    for (int q = 0; q < NP; ++q) asm volatile("" ::"v"(a.w[q]));
#else
    (void)a;
#endif
  };
#ifndef GBNF_HX3_PIPE_MASK
#define GBNF_HX3_PIPE_MASK 7        // diagnostic: which stage kinds end early (1 layer-0 stages, 2 passes, 4 drain)
#endif
  // kind: 1 layer-0 stage, 2 pass, 4 drain.  `early`: in front of the stage's last unit (all of the stage's fragments
  // are in registers); `late`: behind it (the stage kinds that are not pipelined)
  auto stage_finish = [&](int kind, bool early) {
    const bool pipelined = (GBNF_HX3_PIPE_MASK & kind) != 0;
    if (pipelined != early) {
      if (pipelined && !early) mfma_tail_guard();
      return;
    }
    stage_end();
    preload();
  };
  preload();                       // the very first stage (landed behind the prologue's barrier)

  for (int sidx = kb; sidx < ke; ++sidx) {
    const int step = inv ? p.n_steps - 1 - sidx : sidx;
    const uint32_t* __restrict__ sp = blob + (size_t)step * STEP_WORDS;
    // TRAIN: this step's save regions as uniform bases + 32-bit lane offsets (no 64-bit vector arithmetic per store)
    [[maybe_unused]] float* tr_trace = nullptr;
    [[maybe_unused]] float* tr_acts = nullptr;
    if constexpr (TRAIN) {
      tr_trace = p.trace_out + (int64_t)step * d * p.np;
      tr_acts = p.acts_out + (int64_t)step * NNETS * p.net_rows * p.np;
    }
    // ---- normalise the coupling net's inputs in place; split them into the first layer's B operand:
    //      lane (i,g), element j  <->  input feature 8g + j of sample i
    u32x4 zp[NT][NP];
#if GBNF_HX3_EDGE
    // Round 4: direction-specialised straight-line code.  The direction (FlowLaunch::inverse) is tested ONCE per block, not per
    // value (the per-value form cost two scalar branches and three selects per value and broke the block into 16 pieces that the
    // scheduler could not overlap); ONE LDS address per entry serves the read and the write (dead entries use the spare slot d:
    // whatever they read is discarded by a select); slot * ZS is a shift-add (hipcc emitted the quarter-rate v_mul_lo_u32).
    // 32-bit LDS byte addresses (24-bit multiply-add: hipcc turned slot * ZS into the quarter-rate v_mul_lo_u32 / v_mad_u64_u32)
    using lf32 = __attribute__((address_space(3))) float;
    const unsigned zb_ = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)Z + 4u * (unsigned)i;
    auto zoff = [&](int slot) { return zb_ + (unsigned)__mul24(slot, 4 * ZS); };       // &Z[slot * ZS + i]
    auto zld = [&](unsigned a, int nt) { return *reinterpret_cast<lf32*>((uintptr_t)(a + 64u * (unsigned)nt)); };
    auto zst = [&](unsigned a, int nt, float v_) { *reinterpret_cast<lf32*>((uintptr_t)(a + 64u * (unsigned)nt)) = v_; };
    {
      LaneTable tin;
      if (lds_tables) {
        const float lc = as_f32(SM[step * SMALL_WORDS + 1]);
        ld_const += inv ? -lc : lc;
        tin.load(SM + step * SMALL_WORDS + SMALL_HDR + g * NENT);
      } else {
        const float lc = as_f32(sp[1]);
        ld_const += inv ? -lc : lc;
        tin.load(sp + SMALL_HDR + g * NENT);
      }
      unsigned za[NENT];
      bool live[NENT];
      float v[NT][NENT];
#pragma unroll
      for (int e = 0; e < NENT; ++e) {
        live[e] = tin.slot[e] >= 0;
        za[e] = zoff(live[e] ? tin.slot[e] : d);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < NENT; ++e) v[nt][e] = zld(za[e], nt);
      auto edge_in = [&](auto inv_c) {
        constexpr bool INV = decltype(inv_c)::value;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int e = 0; e < NENT; ++e) {
            // forward: normalise, keep; inverse: the stored value IS the normalised net input, the state gets norm^-1 of it
            float t = v[nt][e];
            if constexpr (!INV) {
              t = norm_fn<KIND>(t, tin.p0[e], tin.p1[e], tin.p2[e], tin.p3[e]);
              zst(za[e], nt, t);
            } else {
              zst(za[e], nt, invnorm_fn<KIND>(t, tin.p0[e], tin.p1[e], tin.p2[e], tin.p3[e]));
            }
            if constexpr (TRAIN && !INV) {
              if (live[e] && tr_ok) {       // the backward's inputs: the step's normalised state (slot layout) and the nets' input rows
                tr_trace[tin.slot[e] * tr_np + tr_row + 16 * nt] = t;
#pragma unroll
                for (int q = 0; q < NNETS; ++q) tr_acts[q * tr_net_stride + tr_in_off + nt * tr_ip16 + e * 16] = t;
              }
            }
            if constexpr (WATCH) {
              sat[nt] = sat[nt] || (live[e] && !(__builtin_fabsf(t) <= 65504.0f));
              v[nt][e] = live[e] ? __builtin_amdgcn_fmed3f(t, -65504.0f, 65504.0f) : 0.0f;
            } else {
              v[nt][e] = live[e] ? t : 0.0f;
            }
          }
      };
      if (!inv) edge_in(std::false_type{});
      else edge_in(std::true_type{});
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned pc[NP];
          split_pair<NP>(v[nt][2 * q], v[nt][2 * q + 1], pc);
#pragma unroll
          for (int k = 0; k < NP; ++k) zp[nt][k][q] = pc[k];
        }
    }
#else
    {
      LaneTable tin;
      if (lds_tables) {
        const float lc = as_f32(SM[step * SMALL_WORDS + 1]);
        ld_const += inv ? -lc : lc;
        tin.load(SM + step * SMALL_WORDS + SMALL_HDR + g * NENT);
      } else {
        const float lc = as_f32(sp[1]);
        ld_const += inv ? -lc : lc;
        tin.load(sp + SMALL_HDR + g * NENT);
      }
      float v[NT][NENT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)            // every load first: the slots of a step are distinct
#pragma unroll
        for (int e = 0; e < NENT; ++e) v[nt][e] = Z[(tin.slot[e] >= 0 ? tin.slot[e] : 0) * ZS + i + 16 * nt];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int e = 0; e < NENT; ++e) {
          const bool live = tin.slot[e] >= 0;
          // forward: normalise, keep; inverse: the stored value IS the normalised net input, the state gets norm^-1 of it
          const float t = inv ? v[nt][e] : norm_fn<KIND>(v[nt][e], tin.p0[e], tin.p1[e], tin.p2[e], tin.p3[e]);
          const float keep = inv ? invnorm_fn<KIND>(v[nt][e], tin.p0[e], tin.p1[e], tin.p2[e], tin.p3[e]) : t;
          Z[(live ? tin.slot[e] : d) * ZS + i + 16 * nt] = keep;      // dead entries go to the spare slot d: no exec masking
          if constexpr (TRAIN) {
            if (live && tr_ok) {       // the backward's inputs: the step's normalised state (slot layout) and the nets' input rows
              tr_trace[tin.slot[e] * tr_np + tr_row + 16 * nt] = t;
#pragma unroll
              for (int q = 0; q < NNETS; ++q) tr_acts[q * tr_net_stride + tr_in_off + nt * tr_ip16 + e * 16] = t;
            }
          }
          if constexpr (WATCH) {
            sat[nt] = sat[nt] || (live && !(__builtin_fabsf(t) <= 65504.0f));
            v[nt][e] = live ? __builtin_amdgcn_fmed3f(t, -65504.0f, 65504.0f) : 0.0f;
          } else {
            v[nt][e] = live ? t : 0.0f;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned pc[NP];
          split_pair<NP>(v[nt][2 * q], v[nt][2 * q + 1], pc);
#pragma unroll
          for (int k = 0; k < NP; ++k) zp[nt][k][q] = pc[k];
        }
      }
    }
#endif
    st.mark(0);
    st.set(1);

    f32x4 outA[OT][NT], outB[OT][NT];       // the nets' outputs (all products summed)
#pragma unroll
    for (int net = 0; net < NNETS; ++net) {
      f32x4 (&outF)[OT][NT] = (net == 0) ? outA : outB;
      Acc out[OT][NT];
      const int ACT = (net == 0) ? ACTA : ACTB;
      const uint32_t* bb = BIAS + (cnets & 1) * LT::value.BIAS_WORDS + g * 4;      // this net's biases (LDS)
#ifdef GBNF_ABLATE_BIAS            // diagnostic: no bias reads (timing only)
      auto ldb = [&](int) { return f32x4{0.01f, 0.02f, 0.03f, 0.04f}; };
#else
      auto ldb = [&](int tile) { return *reinterpret_cast<const f32x4*>(bb + tile * 16); };
#endif
      // ACT == 3 (GBNF_ACT_PER_STEP): the activation of this step's net comes from the step header (`--coupling_network random`): both are
      // computed and one is selected (the packer folded the tanh pre-scale into this net's layers only if it is a tanh net)
      const bool relu_rt = ACT == 3 && __builtin_amdgcn_readfirstlane(sp[2 + net]) != 0;
      float amax[NT];             // largest ReLU activation of sample (i, nt) in this net (fp16 range watch)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) amax[nt] = 0.0f;
      auto act1 = [&](float v, int nt) {
        if (ACT == GBNF_ACT_TANH) return tanh_hx3(v);
        float r = __builtin_fmaxf(v, 0.0f);
        if constexpr (WATCH) {
          if (ACT == GBNF_ACT_RELU || ACT == 2) amax[nt] = __builtin_fmaxf(amax[nt], r);
          else amax[nt] = __builtin_fmaxf(amax[nt], relu_rt ? r : 0.0f);
          r = __builtin_fminf(r, 65504.0f);
        }
        if (ACT == GBNF_ACT_RELU || ACT == 2) return r;
        const float t = tanh_hx3(v);
        return relu_rt ? r : t;
      };
      // activate + split one register pair (values 2hp, 2hp+1 of a raw accumulator tile)
      // (the empty asm pins the computation HERE: without it LLVM sinks the whole tanh + split into the later
      //  block that first consumes the operand, un-interleaving it from this region's MFMAs)
      // TRAIN: the activated pair (units 16 tile + 4 g + 2 hp + {0, 1} of hidden layer `layer`) also goes to the operand
      // workspace (the activation-side operand of the next layer's weight gradient, and act' for the backward chain)
      auto save_act = [&](float a0, float a1, bool r_form, int layer, int tile, int hp, int nt) {
        if constexpr (TRAIN) {
          if (r_form) { a0 = __builtin_fmaf(-2.0f, a0, 1.0f); a1 = __builtin_fmaf(-2.0f, a1, 1.0f); }   // tanh = 1 - 2 r
          // rows ip + layer * hp + 16 tile + 4 g + 2 hp + {0, 1} of this net's region; [tile of 16 samples][row][16]
          float* q = tr_acts + net * tr_net_stride + (tr_ip + layer * tr_hp) * tr_np + tr_h_off + nt * tr_hp16 + (16 * tile + 2 * hp) * 16;
          q[0] = a0;
          q[16] = a1;
          tr_later += 2;
        } else {
          (void)a0; (void)a1; (void)r_form; (void)layer; (void)tile; (void)hp; (void)nt;
        }
      };
      auto act_split = [&](const f32x4& raw, int hp, int nt, unsigned (&pc)[NP], int layer = 0, int tile = 0) {
#if defined(GBNF_ABLATE_ACT)      // diagnostic: no activation, no split (results wrong, timing only)
#pragma unroll
        for (int k = 0; k < NP; ++k) pc[k] = __builtin_bit_cast(unsigned, raw[(2 * hp + k) & 3]);
#elif defined(GBNF_ABLATE_SPLIT)  // diagnostic: activation but no split
#pragma unroll
        for (int k = 0; k < NP; ++k) pc[k] = __builtin_bit_cast(unsigned, act1(raw[(2 * hp + k) & 3], nt));
#else
        if (ACT == GBNF_ACT_TANH) {
          // the two "+ 1" of the pair as one v_pk_add_f32
          f32x2 e = {__builtin_amdgcn_exp2f(raw[2 * hp]), __builtin_amdgcn_exp2f(raw[2 * hp + 1])};
          e = e + f32x2{1.0f, 1.0f};
          const float a0 = __builtin_amdgcn_rcpf(e[0]), a1 = __builtin_amdgcn_rcpf(e[1]);
          save_act(a0, a1, true, layer, tile, hp, nt);
          split_pair<NP>(a0, a1, pc);
        } else {
          const float a0 = act1(raw[2 * hp], nt), a1 = act1(raw[2 * hp + 1], nt);
          save_act(a0, a1, ACT == 3 && !relu_rt, layer, tile, hp, nt);
          split_pair<NP>(a0, a1, pc);
        }
#endif
#pragma unroll
        for (int k = 0; k < NP; ++k) asm volatile("" : "+v"(pc[k]));
      };
      // ResidualNet: the block's output tile = second hidden layer's tile + the raw layer-0 tile, handed to the final layer
      // as it is (no activation): range-watched and clamped like a ReLU activation
      constexpr bool RES = (ACTA == 2);
      f32x4 t0r[RES ? HT : 1][NT];
      auto res_split = [&](const f32x4& raw, const f32x4& t0, int hp, int nt, unsigned (&pc)[NP], int tile) {
        float v0 = raw[2 * hp] + t0[2 * hp], v1 = raw[2 * hp + 1] + t0[2 * hp + 1];
        save_act(v0, v1, false, DEPTH, tile, hp, nt);        // TRAIN: the final layer's input t (operand rows of hidden "layer" DEPTH)
        if constexpr (WATCH) {
          amax[nt] = __builtin_fmaxf(amax[nt], __builtin_fmaxf(__builtin_fabsf(v0), __builtin_fabsf(v1)));
          v0 = __builtin_amdgcn_fmed3f(v0, -65504.0f, 65504.0f);
          v1 = __builtin_amdgcn_fmed3f(v1, -65504.0f, 65504.0f);
        }
        split_pair<NP>(v0, v1, pc);
#pragma unroll
        for (int k = 0; k < NP; ++k) asm volatile("" : "+v"(pc[k]));
      };
      // acc[nt] += W . X[nt] for one weight tile (NP pieces) and the B operands of the wave's NT sample tiles
      auto mac = [&](const Unit& a, const u32x4 (&x)[NT][NP], auto& acc) {
        using AT = std::remove_reference_t<decltype(acc[0])>;
#pragma unroll
        for (int pr = 0; pr < NPROD; ++pr) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            f32x4& t = acc[nt].s[acc_of(std::integral_constant<int, AT::N>{}, pr)];
            t = mfma_narrow<PREC>(a.w[Products<NP>::W[pr]], x[nt][Products<NP>::X[pr]], t);
            MFMA_ORDER_FENCE();
          }
        }
      };

      // layer-0 output = B operands of the first hidden layer (DEPTH == 0: of the output layer); with two hidden layers
      // the first one's output goes to the second set
      constexpr int NHB = DEPTH >= 2 ? 2 : 1;
      u32x4 hBs[NHB][HC][NT][NP];
      auto& hB = hBs[0];
#pragma unroll
      for (int b = 0; b < NHB; ++b)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)         // phantom half of an odd tile count stays zero
#pragma unroll
          for (int k = 0; k < NP; ++k) hBs[b][HC - 1][nt][k] = u32x4{0, 0, 0, 0};

      // ---- layer 0: tile t = W1[tile t] . z  (one k = 32 chunk); the activation + split of tile t-1 shares its region
      {
        f32x4 raw[NT];
        auto finish_tile = [&](int t) {       // tile t (held in raw) -> its half of chunk t/2
          const int c = t >> 1, hf = t & 1;
          if constexpr (RES) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) t0r[t][nt] = raw[nt];
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
              unsigned pc[NP];
              act_split(raw[nt], hp, nt, pc, 0, t);
#pragma unroll
              for (int k = 0; k < NP; ++k) hB[c][nt][k][2 * hf + hp] = pc[k];
            }
        };
        auto l0_stage = [&](auto sI_c) {
          constexpr int sI = decltype(sI_c)::value;
          issue(std::integral_constant<int, LT::value.nf[sI + 1]>{}, gs + 1);     // the next layer-0 stage or the first hidden pass
          constexpr int t0 = sI * LT::value.TL0;
          constexpr int cnt = (HT - t0 < LT::value.TL0) ? HT - t0 : LT::value.TL0;
          Unit A[3];
          A[0] = N0;
          A[1] = N1;
          f32x4 bias = ldb(t0);
          if (sI == 0) {
#pragma unroll
            for (int o = 0; o < OT; ++o) {
              const f32x4 b = ldb((DEPTH + 1) * HT + o);
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) out[o][nt].init(b);
            }
          }
#pragma unroll
          for (int tl = 0; tl < LT::value.TL0; ++tl) {
            const int t = t0 + tl;
            if (tl < cnt) {
              if (tl + 2 < cnt) load_unit(A[(tl + 2) % 3], tl + 2);
              const f32x4 bias_next = ldb(t + 1 < HT ? t + 1 : t);
              if (tl == cnt - 1) stage_finish(1, true);
              AccH cur[NT];
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) cur[nt].init(bias);
              // first product, then the previous tile's activation + split, then the rest
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                f32x4& t0_ = cur[nt].s[acc_of(std::integral_constant<int, HCH>{}, 0)];
                t0_ = mfma_narrow<PREC>(A[tl % 3].w[Products<NP>::W[0]], zp[nt][Products<NP>::X[0]], t0_);
                MFMA_ORDER_FENCE();
              }
              if (t > 0) finish_tile(t - 1);
#pragma unroll
              for (int pr = 1; pr < NPROD; ++pr)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                  f32x4& tp = cur[nt].s[acc_of(std::integral_constant<int, HCH>{}, pr)];
                  tp = mfma_narrow<PREC>(A[tl % 3].w[Products<NP>::W[pr]], zp[nt][Products<NP>::X[pr]], tp);
                  MFMA_ORDER_FENCE();
                }
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) raw[nt] = cur[nt].total();
              bias = bias_next;
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          stage_finish(1, false);
        };
        l0_stage(std::integral_constant<int, 0>{});
        if constexpr (LT::value.N_L0 > 1) l0_stage(std::integral_constant<int, 1>{});
        static_assert(LT::value.N_L0 <= 2, "layer 0 spans at most two stages");
        finish_tile(HT - 1);
      }
      st.mark(1);
      st.set(3);

      // ---- hidden layer, one 16-unit output tile per stage.  Tile u-1 is activated/split during pass u
      //      (one register pair per chunk region); an output-layer chunk (two hidden tiles = k 32) is
      //      consumed in the pass after its second tile.
      if constexpr (DEPTH == 0) {
        // ---- no hidden layer: the output layer contracts layer 0's activations, CG chunks (CG * OT tiles) per stage
        constexpr int CG = LT::value.CG, N_OUT = LT::value.N_OUT;
        auto out_stage = [&](auto k_c) {
          constexpr int k = decltype(k_c)::value;
          constexpr int c0 = k * CG;
          constexpr int cnt = (HC - c0 < CG) ? HC - c0 : CG;
          constexpr int NU = cnt * OT;
          if constexpr (k + 1 < N_OUT) {
            issue(std::integral_constant<int, LT::value.nf[LT::value.N_L0 + k + 1]>{}, gs + 1);
          } else {
            if ((net + 1 < NNETS) || (sidx + 1 < ke)) {
              if (net + 1 == NNETS)     // the next step's tables sit in front of its first net (inverse: the step before this one)
                next_src = inv ? (gwords)blob + (size_t)(step - 1) * STEP_WORDS + SMALL_WORDS : next_src + SMALL_WORDS;
              issue_net_start(gs + 1);
            } else if constexpr (TRAIN) {
              tr_later = 0;               // (no DMA to wait for: the same constant on both paths)
            }
          }
          Unit A[3];
          A[0] = N0;
          if (NU > 1) A[1] = N1;
#pragma unroll
          for (int n = 0; n < NU; ++n) {
            if (n + 2 < NU) load_unit(A[(n + 2) % 3], n + 2);
            if (n == NU - 1) stage_finish(4, true);
            mac(A[n % 3], hB[c0 + n / OT], out[n % OT]);
            __builtin_amdgcn_sched_barrier(0);
          }
          stage_finish(4, false);
        };
        static_assert(N_OUT <= 4, "output stages of a depth-0 net");
        out_stage(std::integral_constant<int, 0>{});
        if constexpr (N_OUT > 1) out_stage(std::integral_constant<int, 1>{});
        if constexpr (N_OUT > 2) out_stage(std::integral_constant<int, 2>{});
        if constexpr (N_OUT > 3) out_stage(std::integral_constant<int, 3>{});
#pragma unroll
        for (int o = 0; o < OT; ++o)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) outF[o][nt] = out[o][nt].total();
        st.mark(4);
      } else {
        f32x4 pre[NT];
        f32x4 bias;
        if constexpr (DEPTH >= 2) {
          // ---- the hidden layers in front of the last one (one for DEPTH = 2, three for a two-block ResidualNet): like the passes
          //      below, but tile u-1 (activated during pass u) becomes a B operand of the NEXT hidden layer instead of being
          //      consumed by the output layer -- layer j reads set (j - 1) & 1 and writes set j & 1.  Fully unrolled: the
          //      destination register of an activation must be a compile-time index.
          auto mid_layer = [&](auto j_c) {
            constexpr int J = decltype(j_c)::value;
            auto& hIn = hBs[(J - 1) & 1];
            auto& hOut = hBs[J & 1];
            if constexpr (J >= 3) {                 // (a set that is written a second time: the phantom half of an odd tile count stays zero)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int k = 0; k < NP; ++k) hOut[HC - 1][nt][k] = u32x4{0, 0, 0, 0};
            }
            bias = ldb(J * HT);
            auto mid_act = [&](auto t_c, int n) {           // register pairs n, n + HC, ... of tile t (held in pre)
              constexpr int t = decltype(t_c)::value;
#pragma unroll
              for (int q = n; q < 2 * NT; q += HC) {
                const int nt = q >> 1, hp = q & 1;
                unsigned pc[NP];
                if constexpr (RES && (J & 1) == 0) {
                  // the second Linear of a block that is not the last: t = its output + the skip source; t replaces the source
                  f32x4 tv = pre[nt];
                  tv[2 * hp] += t0r[t][nt][2 * hp];
                  tv[2 * hp + 1] += t0r[t][nt][2 * hp + 1];
                  t0r[t][nt][2 * hp] = tv[2 * hp];
                  t0r[t][nt][2 * hp + 1] = tv[2 * hp + 1];
                  act_split(tv, hp, nt, pc, J, t);
                } else {
                  act_split(pre[nt], hp, nt, pc, J, t);
                }
#pragma unroll
                for (int k = 0; k < NP; ++k) hOut[t >> 1][nt][k][2 * (t & 1) + hp] = pc[k];
              }
            };
            auto mid_pass = [&](auto u_c) {
              constexpr int u = decltype(u_c)::value;
              issue(std::integral_constant<int, NP * HC>{}, gs + 1);     // the next pass of this layer or pass 0 of the next one
              Unit A[3];
              A[0] = N0;
              A[1] = N1;
              const f32x4 bias_next = ldb(J * HT + (u + 1 < HT ? u + 1 : u));
              AccH acc[NT];
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) acc[nt].init(bias);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int n = 0; n < HC; ++n) {
                if (n + 2 < HC) load_unit(A[(n + 2) % 3], n + 2);
                if (n == HC - 1) stage_finish(2, true);
                if constexpr (u > 0) mid_act(std::integral_constant<int, (u > 0 ? u - 1 : 0)>{}, n);
                mac(A[n % 3], hIn[n], acc);
                __builtin_amdgcn_sched_barrier(0);
              }
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) pre[nt] = acc[nt].total();
              bias = bias_next;
              stage_finish(2, false);
            };
            auto mid_all = [&](auto self, auto u_c) -> void {
              constexpr int u = decltype(u_c)::value;
              if constexpr (u < HT) {
                mid_pass(u_c);
                self(self, std::integral_constant<int, u + 1>{});
              }
            };
            mid_all(mid_all, std::integral_constant<int, 0>{});
#pragma unroll
            for (int n = 0; n < HC; ++n) mid_act(std::integral_constant<int, HT - 1>{}, n);
          };
          mid_layer(std::integral_constant<int, 1>{});
          if constexpr (DEPTH == 4) {
            mid_layer(std::integral_constant<int, 2>{});
            mid_layer(std::integral_constant<int, 3>{});
          }
        }
        auto& hBin = hBs[(DEPTH - 1) & 1];         // the B operands of the last hidden layer (set 0: layer 0's output, DEPTH = 1)
        u32x4 hO[NT][NP];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int k = 0; k < NP; ++k) hO[nt][k] = u32x4{0, 0, 0, 0};
        bias = ldb(DEPTH * HT);
        // PREV: 0 = no previous tile (u = 0); 1 = tile u-1 is the FIRST half of its output-layer chunk
        // (u odd); 2 = it is the SECOND half and chunk (u-2)/2 is consumed at the end of this pass (u even >= 2)
        auto pass = [&](int u, auto prev_c, auto last_c) {
          constexpr int PREV = decltype(prev_c)::value;
          constexpr bool LAST = decltype(last_c)::value;
          constexpr int NU = HC + (PREV == 2 ? OT : 0);       // consumption units of this stage
          // the stage after this one: the drain, a pass with an output-layer chunk (after an odd pass), or a plain pass
          constexpr int NF_NEXT = LAST ? NP * OT : (PREV == 1 ? NP * (HC + OT) : NP * HC);
          guard_at(6);
          // (GBNF_HX3_DMA_AT > 0: the next stage's DMA is issued in front of unit DMA_AT instead of at the top of the pass, where all
          //  waves of the workgroup leave the barrier together and queue on the vector-memory port; not for TRAIN: its counted
          //  stage waits count the stores BEHIND the DMA)
          constexpr int DMA_AT = TRAIN ? 0 : GBNF_HX3_DMA_AT;
          if constexpr (DMA_AT == 0) issue(std::integral_constant<int, NF_NEXT>{}, gs + 1);
          guard_at(4);
          Unit A[3];
          A[0] = N0;
          A[1] = N1;
          const f32x4 bias_next = ldb(DEPTH * HT + (u + 1 < HT ? u + 1 : u));
          guard_at(5);
          AccH acc[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt].init(bias);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int n = 0; n < NU; ++n) {
            if (n + 2 < NU) load_unit(A[(n + 2) % 3], n + 2);
            if (DMA_AT > 0 && n == (DMA_AT < NU - 1 ? DMA_AT : NU - 2)) issue(std::integral_constant<int, NF_NEXT>{}, gs + 1);
            if (n == NU - 1) stage_finish(2, true);
            if (n < HC) {
              if (PREV != 0) {
                // register pairs q = n, n + HC, ... of the previous tile (2*NT pairs in all), ahead of the region's MFMAs
#pragma unroll
                for (int q = n; q < 2 * NT; q += HC) {
                  const int nt = q >> 1, hp = q & 1;
                  unsigned pc[NP];
                  if constexpr (RES) res_split(pre[nt], t0r[u - 1][nt], hp, nt, pc, u - 1);
                  else act_split(pre[nt], hp, nt, pc, DEPTH, u - 1);
#pragma unroll
                  for (int k = 0; k < NP; ++k) hO[nt][k][(PREV == 2 ? 2 : 0) + hp] = pc[k];
                }
              }
              mac(A[n % 3], hBin[n], acc);
            } else {
              // output-layer chunk (u-2)/2 = hidden tiles (u-2, u-1): its tiles follow the hidden row
              mac(A[n % 3], hO, out[n - HC]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          pin_unit(A[(NU - 1) % 3]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) pre[nt] = acc[nt].total();
          bias = bias_next;
          stage_finish(2, false);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using BF = std::false_type;
        using BT = std::true_type;
        if constexpr (HT == 1) {
          pass(0, I0{}, BT{});
        } else {
          pass(0, I0{}, BF{});
          int u = 1;
          if constexpr (RES) {       // (the raw layer-0 tile of pass u is a register array indexed by u: compile-time passes)
#pragma unroll
            for (int uu = 1; uu + 2 < HT; uu += 2) {
              pass(uu, I1{}, BF{});
              pass(uu + 1, I2{}, BF{});
            }
            u = 1 + 2 * ((HT - 2) / 2);
          } else {
#pragma unroll 1
            for (; u + 2 < HT; u += 2) {
              pass(u, I1{}, BF{});
              pass(u + 1, I2{}, BF{});
            }
          }
          if constexpr (HT % 2 == 1) {       // two passes left: HT-2 (odd), HT-1 (even, last)
            pass(u, I1{}, BF{});
            pass(u + 1, I2{}, BT{});
          } else {                           // one pass left: HT-1 (odd, last)
            pass(u, I1{}, BT{});
          }
        }
        st.mark(3);
        st.set(4);
        // ---- drain: last tile, last output-layer chunk (HC-1); the next net's / step's first stage goes in flight
        {
          guard_at(7);
          if ((net + 1 < NNETS) || (sidx + 1 < ke)) {
            if (net + 1 == NNETS)       // the next step's tables sit in front of its first net (inverse: the step before this one)
              next_src = inv ? (gwords)blob + (size_t)(step - 1) * STEP_WORDS + SMALL_WORDS : next_src + SMALL_WORDS;
            issue_net_start(gs + 1);
          } else if constexpr (TRAIN) {
            tr_later = 0;                 // (no DMA to wait for: the same constant on both paths)
          }
          Unit A[OT];
          A[0] = N0;
          if (OT > 1) A[1] = N1;
#pragma unroll
          for (int o = 2; o < OT; ++o) load_unit(A[o], o);
          constexpr bool odd_last = ((HT - 1) & 1) != 0;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
              unsigned pc[NP];
              if constexpr (RES) res_split(pre[nt], t0r[HT - 1][nt], hp, nt, pc, HT - 1);
              else act_split(pre[nt], hp, nt, pc, DEPTH, HT - 1);
#pragma unroll
              for (int k = 0; k < NP; ++k) hO[nt][k][(odd_last ? 2 : 0) + hp] = pc[k];
            }
            if (!odd_last) {
#pragma unroll
              for (int k = 0; k < NP; ++k) { hO[nt][k][2] = 0; hO[nt][k][3] = 0; }
            }
          }
#pragma unroll
          for (int o = 0; o < OT; ++o) {
            if (o == OT - 1) stage_finish(4, true);
            mac(A[o], hO, out[o]);
          }
#pragma unroll
          for (int o = 0; o < OT; ++o)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) outF[o][nt] = out[o][nt].total();
          stage_finish(4, false);
        }
        st.mark(4);
      }
      if constexpr (TRAIN) if (tr_ok) {     // the net's output rows (the backward needs shift / scale)
        // behind the gradient-side rows: input | DEPTH + 1 activation regions | DEPTH + 1 gradient regions | output gradient | output
        float* q0 = tr_acts + net * tr_net_stride + (tr_ip + 2 * (DEPTH + 1) * tr_hp + p.tr_op) * tr_np + tr_o_off;
#pragma unroll
        for (int o = 0; o < OT; ++o)
          if (16 * o < p.tr_op)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int r = 0; r < 4; ++r) q0[nt * tr_op16 + (16 * o + r) * 16] = outF[o][nt][r];
      }
      if constexpr (WATCH) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) sat[nt] = sat[nt] || !(amax[nt] <= 65504.0f);
      }
      ++cnets;
    }

    // ---- coupling transform of the other half, in place, + per-lane log-det partials
#if GBNF_HX3_EDGE
    {
      LaneTable tout;
      if (lds_tables) tout.load(SM + step * SMALL_WORDS + SMALL_HDR + 160 + g * NENT);
      else tout.load(sp + SMALL_HDR + 160 + g * NENT);
      if (KIND == GBNF_KIND_GLOW && !p.additive) {
        // affine coupling, models/glow.py:331-338 / 352-355: scale = sigmoid(raw + 2) = 1 / s1, s1 = 1 + exp(-(raw + 2));
        // log scale = -ln 2 . log2(s1): the log2 terms are summed as they are and scaled once at the end of the kernel (ld2)
        constexpr int NE = (2 * OT < NENT) ? 2 * OT : NENT;
        unsigned za[NE];
        bool live[NE];
        float v[NT][NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          live[e] = tout.slot[e] >= 0;
          za[e] = zoff(live[e] ? tout.slot[e] : d);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int e = 0; e < NE; ++e) v[nt][e] = zld(za[e], nt);
        auto edge_out = [&](auto inv_c) {
          constexpr bool INV = decltype(inv_c)::value;
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            const int o = e >> 1, pp = e & 1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const float shift = outA[o][nt][2 * pp], raw = outA[o][nt][2 * pp + 1];
              const float s1 = 1.0f + __builtin_amdgcn_exp2f((raw + 2.0f) * -1.4426950408889634f);
              const float l2 = __builtin_amdgcn_logf(s1);
              float t;
              if constexpr (!INV) {
                t = norm_fn<KIND>(v[nt][e], tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
                if constexpr (TRAIN) {
                  if (live[e] && tr_ok) tr_trace[tout.slot[e] * tr_np + tr_row + 16 * nt] = t;
                }
                t = (t + shift) * __builtin_amdgcn_rcpf(s1);
                ld2[nt] += live[e] ? l2 : 0.0f;
              } else {                                                   // FlowStep.decode, models/glow.py:352-355
                t = v[nt][e] * s1 - shift;                               // z2 / scale - shift
                t = invnorm_fn<KIND>(t, tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
                ld2[nt] -= live[e] ? l2 : 0.0f;
              }
              zst(za[e], nt, t);
            }
          }
        };
        if (!inv) edge_out(std::false_type{});
        else edge_out(std::true_type{});
      } else {
        constexpr int NE = (4 * OT < NENT) ? 4 * OT : NENT;
        unsigned za[NE];
        bool live[NE];
        float v[NT][NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          live[e] = tout.slot[e] >= 0;
          za[e] = zoff(live[e] ? tout.slot[e] : d);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int e = 0; e < NE; ++e) v[nt][e] = zld(za[e], nt);
        auto edge_out = [&](auto inv_c) {
          constexpr bool INV = decltype(inv_c)::value;
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            const int o = e >> 2, r = e & 3;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              float t = v[nt][e];
              if constexpr (!INV) {
                t = norm_fn<KIND>(t, tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
                if constexpr (TRAIN) {
                  if (live[e] && tr_ok) tr_trace[tout.slot[e] * tr_np + tr_row + 16 * nt] = t;
                }
              }
              if constexpr (KIND == GBNF_KIND_GLOW) {                     // additive, models/glow.py:328-329 / 349-350
                if constexpr (!INV) t = t + outA[o][nt][r];
                else t = invnorm_fn<KIND>(t - outA[o][nt][r], tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
              } else {
                const float shift = outA[o][nt][r], scale = outB[o][nt][r];
                if constexpr (!INV) {
                  t = shift + t * exp_fast(scale);                       // models/transformations.py:575
                  ld[nt] += live[e] ? scale : 0.0f;
                } else {                                                 // its true inverse (the reference's own .inverse is not: SURVEY S3)
                  t = (t - shift) * exp_fast(-scale);
                  t = invnorm_fn<KIND>(t, tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
                  ld[nt] -= live[e] ? scale : 0.0f;
                }
              }
              zst(za[e], nt, t);
            }
          }
        };
        if (!inv) edge_out(std::false_type{});
        else edge_out(std::true_type{});
      }
    }
#else
    {
      LaneTable tout;
      if (lds_tables) tout.load(SM + step * SMALL_WORDS + SMALL_HDR + 160 + g * NENT);
      else tout.load(sp + SMALL_HDR + 160 + g * NENT);
      if (KIND == GBNF_KIND_GLOW && !p.additive) {
        constexpr int NE = (2 * OT < NENT) ? 2 * OT : NENT;
        float v[NT][NE];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int e = 0; e < NE; ++e) v[nt][e] = Z[(tout.slot[e] >= 0 ? tout.slot[e] : 0) * ZS + i + 16 * nt];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int o = e >> 1, pp = e & 1;
          const bool live = tout.slot[e] >= 0;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float shift = outA[o][nt][2 * pp], raw = outA[o][nt][2 * pp + 1];
            float sc, lsc, t;
            sigmoid_logsigmoid(raw + 2.0f, sc, lsc);
            if (!inv) {
              t = norm_fn<KIND>(v[nt][e], tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
              if constexpr (TRAIN) {
                if (live && tr_ok) tr_trace[tout.slot[e] * tr_np + tr_row + 16 * nt] = t;
              }
              t = (t + shift) * sc;
            } else {                                                   // FlowStep.decode, models/glow.py:352-355
              t = v[nt][e] / sc - shift;
              t = invnorm_fn<KIND>(t, tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
              lsc = -lsc;
            }
            Z[(live ? tout.slot[e] : d) * ZS + i + 16 * nt] = t;
            ld[nt] += live ? lsc : 0.0f;
          }
        }
      } else {
        constexpr int NE = (4 * OT < NENT) ? 4 * OT : NENT;
        float v[NT][NE];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int e = 0; e < NE; ++e) v[nt][e] = Z[(tout.slot[e] >= 0 ? tout.slot[e] : 0) * ZS + i + 16 * nt];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int o = e >> 2, r = e & 3;
          const bool live = tout.slot[e] >= 0;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            float t = inv ? v[nt][e] : norm_fn<KIND>(v[nt][e], tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
            if constexpr (TRAIN) {
              if (live && tr_ok) tr_trace[tout.slot[e] * tr_np + tr_row + 16 * nt] = t;
            }
            if constexpr (KIND == GBNF_KIND_GLOW) {
              t = inv ? t - outA[o][nt][r] : t + outA[o][nt][r];      // additive, models/glow.py:328-329 / 349-350
              if (inv) t = invnorm_fn<KIND>(t, tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
              Z[(live ? tout.slot[e] : d) * ZS + i + 16 * nt] = t;
            } else {
              const float shift = outA[o][nt][r], scale = outB[o][nt][r];
              if (!inv) {
                t = shift + t * exp_fast(scale);                       // models/transformations.py:575
              } else {                                                 // its true inverse (the reference's own .inverse is not: SURVEY S3)
                t = (t - shift) * exp_fast(-scale);
                t = invnorm_fn<KIND>(t, tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
              }
              Z[(live ? tout.slot[e] : d) * ZS + i + 16 * nt] = t;
              ld[nt] += live ? (inv ? -scale : scale) : 0.0f;
            }
          }
        }
      }
    }
#endif
    // Z is wave-private: the wave's own LDS writes are ordered before its next reads (in-order LDS queue);
    // make that explicit for the compiler
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    st.mark(5);
  }

  // ---- base log-density + log|det J|, folded over the 4 lane groups
  const uint32_t* tail = blob + (size_t)p.n_steps * STEP_WORDS;
  float quad[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) quad[nt] = 0.0f;
  if (p.base_mean == nullptr) {
    // standard-normal base: the sum of squares runs over the slots themselves (no feature -> slot look-up, which was a
    // dependent global load per feature at the end of every work item)
    for (int j = g; j < d; j += 4) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float v = Z[j * ZS + 16 * nt + i];
        quad[nt] += -0.5f * v * v;
      }
    }
  } else {
    for (int j = g; j < d; j += 4) {
      const int slot = (int)tail[j];
      const float mu = p.base_mean[j], sd = p.base_std[j];
      const float inv_sd = 1.0f / sd, lsd = logf(sd);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float v = (Z[slot * ZS + 16 * nt + i] - mu) * inv_sd;
        quad[nt] += -0.5f * v * v - lsd;
      }
    }
  }
  bool any_sat = false;
  unsigned long long bad_rows = 0;      // bit r: row r of the wave's tile left the fp16 range (its outputs are marked)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    float q = quad[nt], l = ld[nt];
#if GBNF_HX3_EDGE
    l = __builtin_fmaf(-0.69314718055994531f, ld2[nt], l);
#endif
    q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
    l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
    bool bad = false;
    if constexpr (WATCH) {
      const unsigned long long m = __ballot(sat[nt]);          // bit (16 g + i): fold the 4 lane groups per sample
      const unsigned rows = (unsigned)((m | (m >> 16) | (m >> 32) | (m >> 48)) & 0xffffull);
      bad_rows |= (unsigned long long)rows << (16 * nt);
      // TRAIN: nothing repairs a marked row behind the training forward (the bf16x6 repair pass exists for the evaluation
      // kernels only), and one NaN row would make the loss and every gradient NaN: the operands were CLAMPED above, the
      // outputs stay finite (what the round-1 train_kernel does) and the launch is COUNTED (gbnf_saturation_count)
      bad = !TRAIN && ((rows >> i) & 1u);
      any_sat = any_sat || rows != 0;
    }
    const int64_t n = row0 + 16 * nt + i;
    if (g == 0 && n < p.n) {
      float ldj = l + ld_const;
      if constexpr (TRAIN) {
        if (p.ldj_accumulate && p.ldj_out) ldj += p.ldj_out[out_base + n];      // the ranges before this one
      }
      const float nanv = __builtin_nanf("");
      if (p.ldj_out) p.ldj_out[out_base + n] = bad ? nanv : ldj;
      if (p.ll_out) p.ll_out[out_base + n] = bad ? nanv : (q - 0.91893853320467274f * (float)d) + ldj;
    }
  }
  if (TRAIN && p.state_out != nullptr) {        // the range does not end the flow: park the state (slot layout), no z
    constexpr int RW = 16 * NT, SPP = 64 / RW;
    const int r = lane % RW, s0 = lane / RW;
    float* sout = p.state_out + row0 + r;
    for (int slot = s0; slot < d; slot += SPP) sout[(int64_t)slot * p.np] = Z[slot * ZS + r];
  }
  if (WATCH && p.sat != nullptr && any_sat && lane == 0) {
    atomicAdd(p.sat, 1ull);
    if constexpr (!TRAIN) atomicMax(p.sat + SAT_MARKS + p.seq % SAT_SLOTS, p.seq);      // tells the repair launch behind this one that it has work
  }
  if (p.z_out != nullptr && lane < d && !(TRAIN && p.state_out != nullptr)) {
    const int slot = inv ? lane : (int)tail[lane];
    float* zo = p.z_out + (int64_t)comp * p.n * d;
#pragma unroll 8
    for (int r = 0; r < 16 * NT; ++r) {
      const int64_t n = row0 + r;
      float v = Z[slot * ZS + r];
      if (WATCH && !TRAIN && ((bad_rows >> r) & 1ull)) v = __builtin_nanf("");
      if (n < p.n) zo[n * d + lane] = v;
    }
  }
#ifdef GBNF_STAMPS
  st.mark(6);
  if (p.dbg != nullptr && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) p.dbg[((size_t)blockIdx.x * WAVES + wave) * 8 + k] = st.acc[k];
  }
#endif
  }    // work items
#ifdef GBNF_CLOCK
  {
    unsigned long long ck_c1 = 0, ck_r1 = 0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ck_c1), "=s"(ck_r1)::"memory");
    if (p.dbg != nullptr && threadIdx.x == 0) {       // a buffer of their own: no output depends on the stamps
      p.dbg[(size_t)blockIdx.x * 2 + 0] = ck_c1 - ck_c0;
      p.dbg[(size_t)blockIdx.x * 2 + 1] = ck_r1 - ck_r0;
    }
  }
#endif
}

// LDS bytes of a launch with `ring` stage slots.
inline size_t flow_hx3_lds_bytes(int n_steps, int nt, int waves, int stage_frags, int bias_frags, int d, int ring,
                                 bool lds_tables) {
  const size_t tables = lds_tables ? (size_t)n_steps * SMALL_WORDS : 0;
  return (tables + 2 * (size_t)bias_frags * 256 + (size_t)ring * stage_frags * 256 + (size_t)waves * (d + 1) * (16 * nt + 1)) * 4;
}
// hx3 variants are keyed like the f32 ones with the lmid-independent fields fixed:
//   VariantKey{kind, ht, /*ksl*/ -3 (f16x3) | -6 (bf16x6), /*ks1*/ 0 (evaluation) | 1 (training forward), ot, nt, /*lmid*/ depth (0, 1, 2), act_a, act_b}
// The launcher sizes its own grid (it knows its waves per workgroup) and ring; the `grid` argument is ignored.
// An 8-wave kernel whose register budget is 256 also exists as a 4-wave workgroup: when two of those fit a CU's LDS
// (small d / K / hidden width) they run instead, staggered by about half a flow step (FlowLaunch::stagger).
template <int KIND, int HT, int OT, int ENT, int ACTA, int ACTB, int PREC, int WV, int DEPTH, int TRAIN = 0>
static hipError_t hx3_launch_wv(FlowLaunch p, bool staggered, hipStream_t s) {
  constexpr Hx3Layout L(HT, OT, hx3_pieces(PREC), DEPTH);
  p.n_tiles = (int32_t)((p.n + 16 * ENT - 1) / (16 * ENT));
  if (TRAIN) p.n_tiles = (int32_t)(p.np / (16 * ENT));      /* every padded row: the operand workspace is summed over all np samples */
  /* per-step tables in LDS when they fit beside the staging slots and the Z tiles, else read from the blob */
  const size_t budget = staggered ? 80 * 1024 : 160 * 1024;
  p.lds_tables = p.n_steps <= LDS_TABLE_STEPS &&
      flow_hx3_lds_bytes(p.n_steps, ENT, WV, L.STAGE_FRAGS, L.BIAS_FRAGS, p.d, HX3_RING, true) <= budget;
  p.ring = HX3_RING;
  const size_t lds = flow_hx3_lds_bytes(p.n_steps, ENT, WV, L.STAGE_FRAGS, L.BIAS_FRAGS, p.d, p.ring, p.lds_tables != 0);
  if (lds > budget) return hipErrorInvalidValue;
  long long grid = (long long)((p.n_tiles + WV - 1) / WV) * p.n_comp * p.n_batches;
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
  p.n_items = (int32_t)grid;
  if (p.repair && grid > 256) grid = 256;           // a repair launch walks the items with one workgroup per CU (it almost always returns at once:
                                                    // 512 workgroups of 160 KB of LDS took 7-10 us to do that, profiles/r5_emulated_rank_steps20_*)
  if (staggered) {
    // A start offset between the two co-resident workgroups was measured and bought nothing
    // (profiles/r2_wg_pairs_stagger_hx32.txt): off by default; GBNF_STAGGER = sleeps of 2048 cycles for experiments
    static const int forced = [] { const char* e = getenv("GBNF_STAGGER"); return e ? atoi(e) : 0; }();
    p.stagger = forced;
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)flow_kernel_hx3<KIND, HT, OT, ENT, ACTA, ACTB, PREC, WV, DEPTH, TRAIN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((flow_kernel_hx3<KIND, HT, OT, ENT, ACTA, ACTB, PREC, WV, DEPTH, TRAIN>), dim3((unsigned)grid), dim3(64 * WV), lds, s, p);
  return hipGetLastError();
}

// TRAIN = 1: the training path's forward sweep (trace + operand saves; registry key field ks1 = 1)
#define GBNF_INSTANTIATE_HX3_T(KIND, HT, OT, NT, ACTA, ACTB, PREC, DEPTH, TRAIN)                            \
  namespace gbnf {                                                                                          \
  static hipError_t launch_hx3_##KIND##_##HT##_##OT##_##NT##_##ACTA##_##ACTB##_##PREC##_##DEPTH##_##TRAIN(const FlowLaunch& p0, \
                                                                                     unsigned, hipStream_t s) { \
    constexpr Hx3Layout L(HT, OT, hx3_pieces(PREC), DEPTH);                                                 \
    constexpr int ENT = hx3_eff_nt(HT, OT, NT, PREC, KIND, ACTA, ACTB, DEPTH, TRAIN);                           \
    constexpr int WAVES = hx3_waves(HT, OT, ENT, PREC, KIND, ACTA, ACTB, DEPTH, TRAIN);                         \
    if constexpr (WAVES == 8) {                                                                             \
      const int pairs = tuning_wg_pairs();       /* -1 automatic, 0 never, 1 whenever they fit (gbnf_tuning_set) */        \
      /* two 4-wave workgroups per CU where they fit: 80 KB each, tables included */                        \
      const bool fits4 = flow_hx3_lds_bytes(p0.n_steps, ENT, 4, L.STAGE_FRAGS, L.BIAS_FRAGS, p0.d, HX3_RING,  \
                                            p0.n_steps <= 12) <= 80 * 1024;  /* (tables of 13 .. LDS_TABLE_STEPS steps: in LDS only if they fit beside the pair form) */                    \
      /* a lone wave per SIMD runs a stage in half the time of two sharing it (profiles/r2_ubench_pingpong.txt): 4-wave        \
         workgroups spread the same waves over twice the CUs.  Round 5: ALWAYS where they fit, not only from 1024 waves on --  \
         at the reference's own batch sizes (density_experiment.py:80-81: 512 rows to train on, 1024 to evaluate) one           \
         log_prob call of the MINIBOONE C = 8 model takes 47.6 us instead of 63.5 (19.9 vs 15.3 M samples/s at 1024 rows,       \
         10.0 vs 7.7 M at 512; tools/bench_latency.py, profiles/r5_latency_small_batches.txt) */                              \
      /* Round 6 (tools/ab_wg8.sh, profiles/r6_ab_workgroup_forms.txt): a LONG launch of a geometry with heavy stages (>= 16 KiB of     \
         weight fragments per stage: MINIBOONE, 20) runs 1.1-1.3 % faster as ONE 8-wave workgroup per CU -- half the L2 -> LDS    \
         weight DMA per CU -- than as two 4-wave ones: 78.7 vs 77.8 M samples/s on the headline, 75.1 vs 74.1 M at the driver's    \
         --steps 20, 154.1 vs 152.3 M at C = 4; short launches (the 1.25-round launch of one rank of eight: 262 vs 226 us) and    \
         light stages (HEPMASS RealNVP, 10 KiB: 111.3 vs 114.0 M) keep the pairs */                                                \
      const long long items4 = (long long)((((p0.n + 16 * ENT - 1) / (16 * ENT)) + 3) / 4) * p0.n_comp * p0.n_batches;             \
      const bool long_heavy = pairs == -1 && !TRAIN && L.STAGE_FRAGS >= 16 && items4 >= 4096;                                      \
      if (fits4 && !p0.repair && pairs != 0 && !long_heavy)                                                 \
        return hx3_launch_wv<KIND, HT, OT, ENT, ACTA, ACTB, PREC, 4, DEPTH, TRAIN>(p0, true, s);            \
    }                                                                                                       \
    return hx3_launch_wv<KIND, HT, OT, ENT, ACTA, ACTB, PREC, WAVES, DEPTH, TRAIN>(p0, false, s);           \
  }                                                                                                         \
  static const int reg_hx3_##KIND##_##HT##_##OT##_##NT##_##ACTA##_##ACTB##_##PREC##_##DEPTH##_##TRAIN =     \
      (register_variant(VariantKey{KIND, HT, (PREC) == 0 ? -3 : -6, TRAIN, OT, NT, DEPTH, ACTA, ACTB},      \
                        launch_hx3_##KIND##_##HT##_##OT##_##NT##_##ACTA##_##ACTB##_##PREC##_##DEPTH##_##TRAIN, \
                        "flow_kernel_hx3<" #KIND "," #HT "," #OT "," #NT "," #ACTA "," #ACTB "," #PREC "," #DEPTH "," #TRAIN ">"),\
       0);                                                                                                  \
  }
#define GBNF_INSTANTIATE_HX3(KIND, HT, OT, NT, ACTA, ACTB, PREC, DEPTH) GBNF_INSTANTIATE_HX3_T(KIND, HT, OT, NT, ACTA, ACTB, PREC, DEPTH, 0)

}  // namespace gbnf
