// gbnf_flow_kernel_hx3.hip.h -- the fused flow kernel on the f16 matrix pipe with split-f32 operands.
//
// Why: on gfx950 the f32-input MFMA (v_mfma_f32_16x16x4_f32) runs at the f32 VECTOR rate and blocks the
// vector ALU for its whole duration (tools/ubench/mfma_f32_issue.hip: 32 cycles per 2048 FLOP, VALU fully
// additive).  v_mfma_f32_16x16x32_f16 delivers 16384 FLOP in ~17 cycles and ~8 of those cycles accept
// VALU work (tools/ubench/mfma_bf16_issue.hip; the f16 forms take the same cycles).  An f32 value x is
// split into two fp16 pieces
//      x ~= hi + mid,   hi = f16(x), mid = f16(x - hi)            (x - hi is exact in f32)
// which carry 22 of its 24 significand bits (absolute floor 2^-25 from fp16 subnormals), and a product is
// evaluated as  a_mid.b_hi + a_hi.b_mid + a_hi.b_hi : three f16 MFMAs with f32 accumulation, every
// f16 x f16 product being exact in f32.  The dropped a_mid.b_mid term is 2^-22 relative, i.e. the result
// is within a few ulp of the f32 dot product; end to end the log-likelihood agrees with float64 to ~1e-7
// relative, the same as the torch-f32 reference itself (tests/test_hip_parity.py runs every fixture in
// both math modes).  Range: tanh outputs are in [-1,1]; ReLU outputs and the network inputs are clamped to
// the fp16 range (|v| <= 65504) with one v_med3 -- such values do not occur in a normalising flow.
// For tanh networks 2*log2(e) is folded into the packed weights and biases of the layers that feed a tanh
// (the split is as accurate for c*w as for w), so tanh(y) = 1 - 2/(2^y + 1) costs exp, add, rcp, fma.
//
// Structure (differences to gbnf_flow_kernel.hip.h, whose register-resident chain idea is kept):
//   * a workgroup is 4 waves; each wave owns 16*NT samples and its own LDS feature tile Z, all four work
//     on the SAME component, so the weight stream is fetched once per workgroup: the packed f16 fragments
//     go L2 -> LDS by direct-to-LDS DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, issued a
//     full stage ahead into the other of two staging buffers) and every wave reads its A operands with
//     conflict-free lane-linear ds_read_b128.  One s_barrier per stage (= per 16-unit output tile).
//   * D layout == B layout still holds: two consecutive 16-unit accumulator tiles, after tanh and the
//     hi/mid split, ARE the B operand (k = 32) of the next layer's chunk; nothing is shuffled or stored.
//   * the tanh + split of tile u-1 (VALU) is issued between the MFMAs of tile u.
#pragma once

#include <type_traits>

#include "gbnf_flow_kernel.hip.h"

namespace gbnf {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int HX3_WAVES = 4;
constexpr int HX3_L0_TILES = 10;   // layer-0 tiles per staging stage

// Packed layout of one coupling network for the hx3 kernel, in 32-bit words.  A "fragment" is the A
// operand of one v_mfma_f32_16x16x32_f16 for one 16-row tile: [64 lanes][8 f16] = 256 words; every
// weight tile is a (hi, mid) fragment pair.  Stages are contiguous so one stage = one DMA burst:
//   biases   : B1 [HT][16] | B2 [HT][16] | B3 [OT][16]                                   (f32)
//   L0 stages: tiles [0,10), [10,20)...  each tile (hi, mid)
//   PASS u   : hidden row u, chunks c = 0..HC-1 each (hi, mid); then, if u is even and u >= 2, the
//              output-layer chunk (u-2)/2: tiles o = 0..OT-1 each (hi, mid)   (it is consumed in pass u)
//   DRAIN    : output-layer chunk HC-1
struct Hx3Layout {
  static constexpr int MAXS = 40;
  int HC, N_L0, NS, BIAS_WORDS, NET_WORDS, STAGE_FRAGS;
  int off[MAXS];   // word offset of stage s from the start of the net block
  int nf[MAXS];    // fragments in stage s
  constexpr Hx3Layout(int HT, int OT)
      : HC((HT + 1) / 2), N_L0((HT + HX3_L0_TILES - 1) / HX3_L0_TILES), NS(0), BIAS_WORDS((2 * HT + OT) * 16),
        NET_WORDS(0), STAGE_FRAGS(0), off{}, nf{} {
    int s = 0, w = BIAS_WORDS;
    for (int i = 0; i < N_L0; ++i) {
      const int t0 = i * HX3_L0_TILES;
      const int cnt = (HT - t0 < HX3_L0_TILES) ? HT - t0 : HX3_L0_TILES;
      off[s] = w; nf[s] = 2 * cnt; w += nf[s] * 256; ++s;
    }
    for (int u = 0; u < HT; ++u) {
      off[s] = w; nf[s] = 2 * HC + ((u % 2 == 0 && u >= 2) ? 2 * OT : 0); w += nf[s] * 256; ++s;
    }
    off[s] = w; nf[s] = 2 * OT; w += nf[s] * 256; ++s;
    NS = s;
    NET_WORDS = w;
    for (int k = 0; k < s; ++k) STAGE_FRAGS = nf[k] > STAGE_FRAGS ? nf[k] : STAGE_FRAGS;
  }
};

template <int HT, int OT>
struct Hx3LayoutOf {
  static constexpr Hx3Layout value = Hx3Layout(HT, OT);
};

// 2-piece fp16 split of a pair of f32 values: hi = f16(x) (toward zero), mid = f16(x - hi); 6 VALU ops
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& mid) {
  const auto h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
  hi = __builtin_bit_cast(unsigned, h);
  const float h0 = (float)h[0], h1 = (float)h[1];
  const auto m = __builtin_amdgcn_cvt_pkrtz(x0 - h0, x1 - h1);
  mid = __builtin_bit_cast(unsigned, m);
}

// activations on PRE-SCALED sums: for tanh the packer folded 2*log2(e) into the layer, so
// tanh = 1 - 2/(2^y + 1); ReLU is clamped to the fp16 range in the same instruction
template <int ACT>
__device__ __forceinline__ float act_hx3(float y) {
  if constexpr (ACT == GBNF_ACT_TANH) {
    const float e = __builtin_amdgcn_exp2f(y);
    const float r = __builtin_amdgcn_rcpf(e + 1.0f);
    return __builtin_fmaf(-2.0f, r, 1.0f);
  } else {
    return __builtin_amdgcn_fmed3f(y, 0.0f, 65504.0f);
  }
}

__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
#ifdef GBNF_ABLATE_MFMA           // diagnostic: one cheap VALU op instead of the MFMA (keeps every value live)
  c[0] += __builtin_bit_cast(float, a[0] ^ b[0]);
  return c;
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#endif
}

// acc += W.x with W = (w_hi, w_mid), x = (x_hi, x_mid): small terms first
__device__ __forceinline__ f32x4 mfma_x3(u32x4 w_hi, u32x4 w_mid, u32x4 x_hi, u32x4 x_mid, f32x4 acc) {
  acc = mfma_f16(w_mid, x_hi, acc);
  acc = mfma_f16(w_hi, x_mid, acc);
  acc = mfma_f16(w_hi, x_hi, acc);
  return acc;
}

template <int KIND, int HT, int OT, int NT, int ACTA, int ACTB>
__global__ void __launch_bounds__(64 * HX3_WAVES, (NT == 1) ? 2 : 1) flow_kernel_hx3(const FlowLaunch p) {
  constexpr int ZS = 16 * NT + 1;
  constexpr int NNETS = (KIND == GBNF_KIND_REALNVP) ? 2 : 1;
  constexpr const Hx3Layout& L = Hx3LayoutOf<HT, OT>::value;
  constexpr int HC = L.HC;
  constexpr int STEP_WORDS = SMALL_WORDS + NNETS * L.NET_WORDS;
  constexpr int STAGE_WORDS = L.STAGE_FRAGS * 256;

  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const bool lds_tables = p.n_steps <= LDS_TABLE_STEPS;
  uint32_t* SM = lds;
  uint32_t* STG = lds + (lds_tables ? p.n_steps * SMALL_WORDS : 0);       // 2 staging buffers
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* Z = reinterpret_cast<float*>(STG + 2 * STAGE_WORDS) + wave * (p.d * ZS);   // wave-private, d slots
  const int i = lane & 15;
  const int g = lane >> 4;

  // ---- XCD-aware block -> (component, group of 4 sample tiles)
  const int n_groups = (p.n_tiles + HX3_WAVES - 1) / HX3_WAVES;
  int comp, grp, batch;
  {
    const int total = gridDim.x;
    const int b = blockIdx.x;
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
  const int64_t row0 = ((int64_t)grp * HX3_WAVES + wave) * (16 * NT);    // rows >= n are masked everywhere
  const float* __restrict__ xin = p.xs[batch];

  // ---- weight staging.  The blob is laid out in consumption order, so "the next stage" is a running
  //      pointer; every call site knows the next stage's fragment count at compile time.  Wave w moves
  //      fragments w, w+4, ...; the LDS destination of a wave-instruction is base + lane*16 = the fragment.
  using gwords = const __attribute__((address_space(1))) uint32_t*;
  gwords next_src = (gwords)blob + SMALL_WORDS + L.BIAS_WORDS;   // first stage of step 0, net 0
  int gs = 0;                                                      // stage counter: buffer = gs & 1
  const unsigned lane_b16 = (unsigned)lane * 16u;
  auto issue = [&](auto nf_c, int into) {
    constexpr int NF = decltype(nf_c)::value;
    uint32_t* dst = STG + (into & 1) * STAGE_WORDS;
#pragma unroll
    for (int k = 0; 4 * k < NF; ++k) {
      const int f = wave + 4 * k;
      if (4 * k + 3 < NF || f < NF) {
        // uniform base + 32-bit per-lane offset -> saddr form, no 64-bit VALU address arithmetic
        const __attribute__((address_space(1))) char* base =
            reinterpret_cast<const __attribute__((address_space(1))) char*>(next_src + f * 256);
#ifndef GBNF_ABLATE_DMA          // diagnostic: no weight staging at all (stale LDS contents, timing only)
        __builtin_amdgcn_global_load_lds(base + lane_b16, (__attribute__((address_space(3))) void*)(dst + f * 256),
                                         16, 0, 0);
#else
        (void)base; (void)dst;
#endif
      }
    }
    next_src += NF * 256;
  };
  constexpr int NF_L0_FIRST = 2 * (HT < HX3_L0_TILES ? HT : HX3_L0_TILES);

  // ---- per-step tables -> LDS, x tile -> Z, first weight stage in flight
  issue(std::integral_constant<int, NF_L0_FIRST>{}, 0);
  if (lds_tables) {
    for (int s = 0; s < p.n_steps; ++s) {
      const uint32_t* src = blob + (size_t)s * STEP_WORDS;
      for (int w = (int)threadIdx.x * 4; w < SMALL_WORDS; w += 256 * 4)
        *reinterpret_cast<i32x4*>(SM + s * SMALL_WORDS + w) = *reinterpret_cast<const i32x4*>(src + w);
    }
  }
  if (lane < d) {
#pragma unroll 8
    for (int r = 0; r < 16 * NT; ++r) {
      const int64_t n = row0 + r;
      float v = 0.0f;
      if (n < p.n) v = xin[n * d + lane];
      Z[lane * ZS + r] = v;
    }
  }
  __syncthreads();               // tables + Z visible, stage 0 landed (the barrier drains the DMA)

  float ld[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) ld[nt] = 0.0f;
  float ld_const = 0.0f;
  bool sat = false;
  Stamps st;
  st.start();

#ifdef GBNF_ABLATE_FRAG            // diagnostic: one fragment read per kernel, reused for every MFMA (timing only)
  u32x4 frag_dummy = *reinterpret_cast<const u32x4*>(STG + lane * 4);
  asm volatile("" : "+v"(frag_dummy));
  auto frag = [&](const uint32_t*, int) -> u32x4 { return frag_dummy; };
#else
  auto frag = [&](const uint32_t* buf, int f) -> u32x4 {
    return *reinterpret_cast<const u32x4*>(buf + f * 256 + lane * 4);
  };
#endif
  auto stage_end = [&]() {
#ifndef GBNF_ABLATE_BARRIER       // diagnostic: no per-stage rendezvous (races on the staging buffers, timing only)
    __syncthreads();             // all waves done with this buffer; the next stage's DMA has landed
#endif
    ++gs;
  };

  for (int step = 0; step < p.n_steps; ++step) {
    const uint32_t* __restrict__ sp = blob + (size_t)step * STEP_WORDS;
    // ---- normalise the coupling net's inputs in place; split them into the first layer's B operand:
    //      lane (i,g), element j  <->  input feature 8g + j of sample i
    u32x4 zhi[NT], zmid[NT];
    {
      LaneTable tin;
      if (lds_tables) {
        ld_const += as_f32(SM[step * SMALL_WORDS + 1]);
        tin.load(SM + step * SMALL_WORDS + SMALL_HDR + g * NENT);
      } else {
        ld_const += as_f32(sp[1]);
        tin.load(sp + SMALL_HDR + g * NENT);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float v[NENT];
#pragma unroll
        for (int e = 0; e < NENT; ++e) {
          const bool live = tin.slot[e] >= 0;
          const int zoff = (live ? tin.slot[e] : 0) * ZS + i + 16 * nt;
          float t = Z[zoff];
          t = norm_fn<KIND>(t, tin.p0[e], tin.p1[e], tin.p2[e], tin.p3[e]);
          if (live) Z[zoff] = t;
          sat = sat || (live && !(__builtin_fabsf(t) <= 65504.0f));       // beyond the fp16 range the operand saturates
          v[e] = live ? __builtin_amdgcn_fmed3f(t, -65504.0f, 65504.0f) : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned hq, mq;
          split_pair(v[2 * q], v[2 * q + 1], hq, mq);
          zhi[nt][q] = hq;
          zmid[nt][q] = mq;
        }
      }
    }
    st.mark(0);

    f32x4 outA[OT][NT], outB[OT][NT];
#pragma unroll
    for (int net = 0; net < NNETS; ++net) {
      f32x4 (&out)[OT][NT] = (net == 0) ? outA : outB;
      const int ACT = (net == 0) ? ACTA : ACTB;
      using gf4 = const __attribute__((address_space(1))) f32x4*;
      gwords nb = (gwords)blob + (size_t)step * STEP_WORDS + SMALL_WORDS + net * L.NET_WORDS;
      gf4 b1 = (gf4)nb + g;
      gf4 b2 = (gf4)(nb + HT * 16) + g;
      gf4 b3 = (gf4)(nb + 2 * HT * 16) + g;
#ifdef GBNF_ABLATE_BIAS            // diagnostic: no bias loads from global memory (timing only)
      auto ldb = [&](gf4, int) { return f32x4{0.01f, 0.02f, 0.03f, 0.04f}; };
#else
      auto ldb = [&](gf4 b, int idx) { return b[idx]; };
#endif
      // ACT == 3 (GBNF_ACT_PER_STEP): the activation of this step's net comes from the step header (`--coupling_network random`): both are
      // computed and one is selected (the packer folded the tanh pre-scale into this net's layers only if it is a tanh net)
      const bool relu_rt = ACT == 3 && __builtin_amdgcn_readfirstlane(sp[2 + net]) != 0;
      auto act = [&](float v) {
        if (ACT == GBNF_ACT_TANH) return act_hx3<GBNF_ACT_TANH>(v);
        if (ACT == GBNF_ACT_RELU) return act_hx3<GBNF_ACT_RELU>(v);
        const float t = act_hx3<GBNF_ACT_TANH>(v), r = act_hx3<GBNF_ACT_RELU>(v);
        return relu_rt ? r : t;
      };
      // activate + split one register pair (values 2hp, 2hp+1 of a raw accumulator tile)
      // (the empty asm pins the computation HERE: without it LLVM sinks the whole tanh + split into the later
      //  block that first consumes the operand, un-interleaving it from this region's MFMAs)
      auto act_split = [&](const f32x4& raw, int hp, unsigned& h, unsigned& m) {
#if defined(GBNF_ABLATE_ACT)      // diagnostic: no tanh, no split (results wrong, timing only)
        h = __builtin_bit_cast(unsigned, raw[2 * hp]);
        m = __builtin_bit_cast(unsigned, raw[2 * hp + 1]);
#elif defined(GBNF_ABLATE_SPLIT)  // diagnostic: tanh but no split
        h = __builtin_bit_cast(unsigned, act(raw[2 * hp]));
        m = __builtin_bit_cast(unsigned, act(raw[2 * hp + 1]));
#else
        split_pair(act(raw[2 * hp]), act(raw[2 * hp + 1]), h, m);
#endif
        asm volatile("" : "+v"(h), "+v"(m));
      };

      u32x4 hBhi[HC][NT], hBmid[HC][NT];     // layer-0 output = B operands of the hidden layer
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {       // phantom half of an odd tile count stays zero
        hBhi[HC - 1][nt] = u32x4{0, 0, 0, 0};
        hBmid[HC - 1][nt] = u32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        const f32x4 b = ldb(b3, o * 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) out[o][nt] = b;
      }

      // ---- layer 0: tile t = W1[tile t] . z  (one k = 32 chunk); the tanh + split of tile t-1 shares its region
      {
        f32x4 raw[NT];
        f32x4 bias = ldb(b1, 0);
#pragma unroll
        for (int sI = 0; sI < L.N_L0; ++sI) {
          const uint32_t* buf = STG + (gs & 1) * STAGE_WORDS;
          constexpr int NF_PASS0 = 2 * HC;
          const int t0 = sI * HX3_L0_TILES;
          // next stage: another layer-0 stage or the first hidden pass
          if (sI + 1 < L.N_L0) {
            if (HT - (sI + 1) * HX3_L0_TILES >= HX3_L0_TILES) issue(std::integral_constant<int, 2 * HX3_L0_TILES>{}, gs + 1);
            else issue(std::integral_constant<int, 2 * (HT % HX3_L0_TILES == 0 ? HX3_L0_TILES : HT % HX3_L0_TILES)>{}, gs + 1);
          } else {
            issue(std::integral_constant<int, NF_PASS0>{}, gs + 1);
          }
          u32x4 AL[2 * HX3_L0_TILES];
#pragma unroll
          for (int tl = 0; tl < HX3_L0_TILES; ++tl)
            if (t0 + tl < HT) {
              AL[2 * tl] = frag(buf, 2 * tl);
              AL[2 * tl + 1] = frag(buf, 2 * tl + 1);
            }
#pragma unroll
          for (int tl = 0; tl < HX3_L0_TILES; ++tl) {
            const int t = t0 + tl;
            if (t < HT) {
              const u32x4 c_hi = AL[2 * tl], c_mid = AL[2 * tl + 1];
              const f32x4 bias_next = ldb(b1, (t + 1 < HT ? t + 1 : t) * 4);
              f32x4 cur[NT];
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                cur[nt] = mfma_f16(c_mid, zhi[nt], bias);
                MFMA_ORDER_FENCE();
              }
              if (t > 0) {
                const int c = (t - 1) >> 1, hf = (t - 1) & 1;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                  for (int hp = 0; hp < 2; ++hp) {
                    unsigned h, m;
                    act_split(raw[nt], hp, h, m);
                    hBhi[c][nt][2 * hf + hp] = h;
                    hBmid[c][nt][2 * hf + hp] = m;
                  }
              }
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                cur[nt] = mfma_f16(c_hi, zmid[nt], cur[nt]);
                MFMA_ORDER_FENCE();
              }
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                cur[nt] = mfma_f16(c_hi, zhi[nt], cur[nt]);
                MFMA_ORDER_FENCE();
              }
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) raw[nt] = cur[nt];
              bias = bias_next;
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          stage_end();
        }
        {
          const int c = (HT - 1) >> 1, hf = (HT - 1) & 1;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
              unsigned h, m;
              act_split(raw[nt], hp, h, m);
              hBhi[c][nt][2 * hf + hp] = h;
              hBmid[c][nt][2 * hf + hp] = m;
            }
        }
      }
      st.mark(1);

      // ---- hidden layer, one 16-unit output tile per stage.  Tile u-1 is activated/split during pass u
      //      (one register pair per chunk region); an output-layer chunk (two hidden tiles = k 32) is
      //      consumed in the pass after its second tile.
      {
        f32x4 pre[NT];
        u32x4 hOhi[NT], hOmid[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          hOhi[nt] = u32x4{0, 0, 0, 0};
          hOmid[nt] = u32x4{0, 0, 0, 0};
        }
        f32x4 bias = ldb(b2, 0);
        // PREV: 0 = no previous tile (u = 0); 1 = tile u-1 is the FIRST half of its output-layer chunk
        // (u odd); 2 = it is the SECOND half and chunk (u-2)/2 is consumed at the end of this pass (u even >= 2)
        auto pass = [&](int u, auto prev_c, auto last_c) {
          constexpr int PREV = decltype(prev_c)::value;
          constexpr bool LAST = decltype(last_c)::value;
          constexpr int NF_CUR = 2 * HC + (PREV == 2 ? 2 * OT : 0);
          constexpr int NF_NEXT = LAST ? 2 * OT : (PREV == 1 ? 2 * HC + 2 * OT : 2 * HC);
          const uint32_t* buf = STG + (gs & 1) * STAGE_WORDS;
          issue(std::integral_constant<int, NF_NEXT>{}, gs + 1);
          u32x4 a_hi = frag(buf, 0), a_mid = frag(buf, 1);
          const f32x4 bias_next = ldb(b2, (u + 1 < HT ? u + 1 : u) * 4);
          f32x4 acc[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt] = bias;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < HC; ++c) {
            const u32x4 c_hi = a_hi, c_mid = a_mid;
            if (c + 1 < HC || PREV == 2) {      // next chunk's fragments (or the first output-layer tile's)
              a_hi = frag(buf, 2 * c + 2);
              a_mid = frag(buf, 2 * c + 3);
            }
            if (PREV != 0) {
              // register pairs q = c, c + HC, ... of the previous tile (2*NT pairs in all); issued ahead of the
              // region's MFMAs so that the first region covers the LDS latency of the fragment reads
#pragma unroll
              for (int q = c; q < 2 * NT; q += HC) {
                const int nt = q >> 1, hp = q & 1;
                unsigned h, m;
                act_split(pre[nt], hp, h, m);
                hOhi[nt][(PREV == 2 ? 2 : 0) + hp] = h;
                hOmid[nt][(PREV == 2 ? 2 : 0) + hp] = m;
              }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              acc[nt] = mfma_f16(c_mid, hBhi[c][nt], acc[nt]);
              MFMA_ORDER_FENCE();
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              acc[nt] = mfma_f16(c_hi, hBmid[c][nt], acc[nt]);
              MFMA_ORDER_FENCE();
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              acc[nt] = mfma_f16(c_hi, hBhi[c][nt], acc[nt]);
              MFMA_ORDER_FENCE();
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          if (PREV == 2) {
            // output-layer chunk (u-2)/2 = hidden tiles (u-2, u-1): its fragments follow the hidden row
#pragma unroll
            for (int o = 0; o < OT; ++o) {
              const u32x4 c_hi = a_hi, c_mid = a_mid;
              if (o + 1 < OT) {
                a_hi = frag(buf, 2 * HC + 2 * o + 2);
                a_mid = frag(buf, 2 * HC + 2 * o + 3);
              }
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                out[o][nt] = mfma_f16(c_mid, hOhi[nt], out[o][nt]);
                MFMA_ORDER_FENCE();
              }
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                out[o][nt] = mfma_f16(c_hi, hOmid[nt], out[o][nt]);
                MFMA_ORDER_FENCE();
              }
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                out[o][nt] = mfma_f16(c_hi, hOhi[nt], out[o][nt]);
                MFMA_ORDER_FENCE();
              }
            }
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) pre[nt] = acc[nt];
          bias = bias_next;
          stage_end();
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
#pragma unroll 1
          for (; u + 2 < HT; u += 2) {
            pass(u, I1{}, BF{});
            pass(u + 1, I2{}, BF{});
          }
          if constexpr (HT % 2 == 1) {       // two passes left: HT-2 (odd), HT-1 (even, last)
            pass(u, I1{}, BF{});
            pass(u + 1, I2{}, BT{});
          } else {                           // one pass left: HT-1 (odd, last)
            pass(u, I1{}, BT{});
          }
        }
        st.mark(3);
        // ---- drain: last tile, last output-layer chunk (HC-1); put the next net's / step's first stage in flight
        {
          const uint32_t* buf = STG + (gs & 1) * STAGE_WORDS;
          const bool more = (net + 1 < NNETS) || (step + 1 < p.n_steps);
          if (more) {
            next_src += (net + 1 < NNETS) ? L.BIAS_WORDS : SMALL_WORDS + L.BIAS_WORDS;
            issue(std::integral_constant<int, NF_L0_FIRST>{}, gs + 1);
          }
          u32x4 A[2 * OT];
#pragma unroll
          for (int f = 0; f < 2 * OT; ++f) A[f] = frag(buf, f);
          constexpr bool odd_last = ((HT - 1) & 1) != 0;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
              unsigned h, m;
              act_split(pre[nt], hp, h, m);
              hOhi[nt][(odd_last ? 2 : 0) + hp] = h;
              hOmid[nt][(odd_last ? 2 : 0) + hp] = m;
            }
            if (!odd_last) {
              hOhi[nt][2] = 0; hOhi[nt][3] = 0; hOmid[nt][2] = 0; hOmid[nt][3] = 0;
            }
          }
#pragma unroll
          for (int o = 0; o < OT; ++o) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) out[o][nt] = mfma_x3(A[2 * o], A[2 * o + 1], hOhi[nt], hOmid[nt], out[o][nt]);
          }
          stage_end();
        }
        st.mark(4);
      }
    }

    // ---- coupling transform of the other half, in place, + per-lane log-det partials
    {
      LaneTable tout;
      if (lds_tables) tout.load(SM + step * SMALL_WORDS + SMALL_HDR + 160 + g * NENT);
      else tout.load(sp + SMALL_HDR + 160 + g * NENT);
      if (KIND == GBNF_KIND_GLOW && !p.additive) {
#pragma unroll
        for (int e = 0; e < 2 * OT && e < NENT; ++e) {
          const int o = e >> 1, pp = e & 1;
          const bool live = tout.slot[e] >= 0;
          const int zoff = (live ? tout.slot[e] : 0) * ZS + i;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            float v = Z[zoff + 16 * nt];
            v = norm_fn<KIND>(v, tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
            const float shift = outA[o][nt][2 * pp], raw = outA[o][nt][2 * pp + 1];
            float sc, lsc;
            sigmoid_logsigmoid(raw + 2.0f, sc, lsc);
            v = (v + shift) * sc;
            if (live) {
              Z[zoff + 16 * nt] = v;
              ld[nt] += lsc;
            }
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4 * OT && e < NENT; ++e) {
          const int o = e >> 2, r = e & 3;
          const bool live = tout.slot[e] >= 0;
          const int zoff = (live ? tout.slot[e] : 0) * ZS + i;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            float v = Z[zoff + 16 * nt];
            v = norm_fn<KIND>(v, tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
            if constexpr (KIND == GBNF_KIND_GLOW) {
              v = v + outA[o][nt][r];
              if (live) Z[zoff + 16 * nt] = v;
            } else {
              const float shift = outA[o][nt][r], scale = outB[o][nt][r];
              v = shift + v * exp_fast(scale);
              if (live) {
                Z[zoff + 16 * nt] = v;
                ld[nt] += scale;
              }
            }
          }
        }
      }
    }
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
  for (int j = g; j < d; j += 4) {
    const int slot = (int)tail[j];
    float mu = 0.0f, inv_sd = 1.0f, lsd = 0.0f;
    if (p.base_mean != nullptr) {
      mu = p.base_mean[j];
      const float sd = p.base_std[j];
      inv_sd = 1.0f / sd;
      lsd = logf(sd);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float v = (Z[slot * ZS + 16 * nt + i] - mu) * inv_sd;
      quad[nt] += -0.5f * v * v - lsd;
    }
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    float q = quad[nt], l = ld[nt];
    q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
    l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
    const int64_t n = row0 + 16 * nt + i;
    if (g == 0 && n < p.n) {
      const float ldj = l + ld_const;
      const int64_t o = (int64_t)comp * p.out_stride + (int64_t)batch * p.n + n;
      if (p.ldj_out) p.ldj_out[o] = ldj;
      if (p.ll_out) p.ll_out[o] = (q - 0.91893853320467274f * (float)d) + ldj;
    }
  }
  if (p.sat != nullptr && __any(sat) && lane == 0) atomicAdd(p.sat, 1u);
  if (p.z_out != nullptr && lane < d) {
    const int slot = (int)tail[lane];
    float* zo = p.z_out + (int64_t)comp * p.n * d;
#pragma unroll 8
    for (int r = 0; r < 16 * NT; ++r) {
      const int64_t n = row0 + r;
      if (n < p.n) zo[n * d + lane] = Z[slot * ZS + r];
    }
  }
#ifdef GBNF_STAMPS
  st.mark(6);
  if (p.dbg != nullptr && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) p.dbg[((size_t)blockIdx.x * HX3_WAVES + wave) * 8 + k] = st.acc[k];
  }
#endif
}

inline size_t flow_hx3_lds_bytes(int n_steps, int nt, int stage_frags, int d) {
  const size_t tables = n_steps <= LDS_TABLE_STEPS ? (size_t)n_steps * SMALL_WORDS : 0;
  return (tables + 2 * (size_t)stage_frags * 256 + (size_t)HX3_WAVES * d * (16 * nt + 1)) * 4;
}

// hx3 variants are keyed like the f32 ones with ksl = ks1 = lmid-independent fields fixed:
//   VariantKey{kind, ht, /*ksl*/ -3, /*ks1*/ 0, ot, nt, /*lmid*/ 1, act_a, act_b}
#define GBNF_INSTANTIATE_HX3(KIND, HT, OT, NT, ACTA, ACTB)                                                  \
  namespace gbnf {                                                                                          \
  static hipError_t launch_hx3_##KIND##_##HT##_##OT##_##NT##_##ACTA##_##ACTB(const FlowLaunch& p,           \
                                                                             unsigned grid, hipStream_t s) { \
    constexpr Hx3Layout L(HT, OT);                                                                          \
    const size_t lds = flow_hx3_lds_bytes(p.n_steps, NT, L.STAGE_FRAGS, p.d);                                    \
    static bool attr_set = false;                                                                           \
    if (!attr_set) {                                                                                        \
      hipError_t e = hipFuncSetAttribute((const void*)flow_kernel_hx3<KIND, HT, OT, NT, ACTA, ACTB>,        \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);           \
      if (e != hipSuccess) return e;                                                                        \
      attr_set = true;                                                                                      \
    }                                                                                                       \
    hipLaunchKernelGGL((flow_kernel_hx3<KIND, HT, OT, NT, ACTA, ACTB>), dim3(grid), dim3(64 * HX3_WAVES),   \
                       lds, s, p);                                                                          \
    return hipGetLastError();                                                                               \
  }                                                                                                         \
  static const int reg_hx3_##KIND##_##HT##_##OT##_##NT##_##ACTA##_##ACTB =                                  \
      (register_variant(VariantKey{KIND, HT, -3, 0, OT, NT, 1, ACTA, ACTB},                                 \
                        launch_hx3_##KIND##_##HT##_##OT##_##NT##_##ACTA##_##ACTB,                           \
                        "flow_kernel_hx3<" #KIND "," #HT "," #OT "," #NT "," #ACTA "," #ACTB ">"),          \
       0);                                                                                                  \
  }

}  // namespace gbnf
