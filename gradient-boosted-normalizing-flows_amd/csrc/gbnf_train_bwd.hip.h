// gbnf_train_bwd.hip.h -- the backward sweep of the training path, register-chained (round 3).
//
// What loss.backward() of density_experiment.py:366-374 does to one component, given what the forward sweep
// (flow_kernel_hx3<..., TRAIN = 1>) saved: every step's normalised state (trace), the coupling nets' inputs, hidden
// activations and outputs (operand workspace).  Nothing is recomputed: per step, last to first,
//
//   coupling backward   g_z2 -> g_y2, g_shift, g_raw / g_scale   (element-wise, from the saved net outputs and the trace)
//   dgrad chain         g_o -> W3^T -> (.) act'(h2) -> W2^T -> (.) act'(h1) -> W1^T -> g_y1      (split-f16 MFMA; written for depth 1:
//                       depth 0 has no W2^T link, depth 2 one more, a one-block ResidualNet is depth 2 with its skip -- template DEPTH)
//   normalisation bwd   g_y -> g_x, ActNorm / BatchNorm parameter gradients (16-sample sums + atomics)
//
// and the gradient-side operands of every weight gradient (g_o, g_a2, g_a1) go to the operand workspace for wgrad_kernel.
// The chain has the shape of the forward chain -- a layer with K0 = ceil(OT / 2) input chunks, one with HC chunks, an output
// layer of two tiles -- and is written like it (gbnf_flow_kernel_hx3.hip.h): a wave owns its 16 samples through all three
// layers (D layout = B layout: the accumulator tile of one layer, times act' and split, IS the next layer's B operand),
// the transposed weights are staged once per workgroup by LDS DMA, one stage ahead, in consumption order.
// Reference semantics: FlowStep.encode models/glow.py:317-342, RealNVP.forward models/transformations.py:560-579,
// _ActNorm models/layers.py:488-533, BatchNorm (running statistics) models/layers.py:337-358, TanhNet / ReLUNet :208-243.
#pragma once

#include "gbnf_flow_kernel_hx3.hip.h"

#ifndef GBNF_BWD_PREFETCH
#define GBNF_BWD_PREFETCH 0        // 1: L2-warming look-ahead of the saved operands (see bwd_warm_lines)
#endif
#ifndef GBNF_BWD_ABLATE
#define GBNF_BWD_ABLATE 0          // diagnostic builds (tools/build_bwd_ablations.sh, timing only): 1 no parameter sums, 2 no operand stores, 4 no activation loads
#endif

namespace gbnf {

// Packed layout of one coupling net's TRANSPOSED weights (f16x3: NP = 2 fragments per tile), in consumption order.  The net has
// layer 0 (input -> hidden), DEPTH hidden -> hidden layers 1 .. DEPTH and the output layer DEPTH + 1 (coupling_network_depth 0, 1, 2):
//   L0' stages : W(DEPTH+1)^T tile rows t (hidden tile t = 16 output units of this layer), K0 tiles (k-chunks of 32 net outputs)
//                each, ROWS0 tile rows per stage
//   MID u      : (DEPTH = 2) W2^T row u, chunks c = 0..HC-1
//   PASS u     : (DEPTH >= 1) W1^T row u, chunks c = 0..HC-1; then, if u is even and u >= 2, the W0^T chunk (u-2)/2: IT = 2 tiles
//   DRAIN      : (DEPTH >= 1) W0^T chunk HC-1
//   IN k       : (DEPTH = 0) W0^T chunks k CGI .. : IT tiles each
struct BwdLayout {
  static constexpr int MAXS = 96;
  int HC, K0, ROWS0, N_L0, NS, NET_WORDS, STAGE_FRAGS, DEPTH, CGI, N_IN;
  int off[MAXS], nf[MAXS];
  constexpr BwdLayout(int HT, int OT, int depth = 1)
      : HC((HT + 1) / 2), K0((OT + 1) / 2), ROWS0(1), N_L0(0), NS(0), NET_WORDS(0), STAGE_FRAGS(0), DEPTH(depth), CGI(1), N_IN(0),
        off{}, nf{} {
    ROWS0 = (HC + 2) / K0 > 0 ? (HC + 2) / K0 : 1;
    N_L0 = (HT + ROWS0 - 1) / ROWS0;
    int s = 0, w = 0;
    for (int i = 0; i < N_L0; ++i) {
      const int cnt = (HT - i * ROWS0 < ROWS0) ? HT - i * ROWS0 : ROWS0;
      off[s] = w; nf[s] = 2 * cnt * K0; w += nf[s] * 256; ++s;
    }
    if (depth >= 1) {
      for (int j = depth; j >= 2; --j)
        for (int u = 0; u < HT; ++u) {
          off[s] = w; nf[s] = 2 * HC; w += nf[s] * 256; ++s;
        }
      for (int u = 0; u < HT; ++u) {
        off[s] = w; nf[s] = 2 * HC + ((u % 2 == 0 && u >= 2) ? 2 * 2 : 0); w += nf[s] * 256; ++s;
      }
      off[s] = w; nf[s] = 2 * 2; w += nf[s] * 256; ++s;
    } else {
      CGI = (HC + 2) / 2;                                 // chunks per stage: about the fragments of a pass
      N_IN = (HC + CGI - 1) / CGI;
      for (int k = 0; k < N_IN; ++k) {
        const int cnt = (HC - k * CGI < CGI) ? HC - k * CGI : CGI;
        off[s] = w; nf[s] = 2 * 2 * cnt; w += nf[s] * 256; ++s;
      }
    }
    NS = s;
    NET_WORDS = w;
    for (int k = 0; k < s; ++k) STAGE_FRAGS = nf[k] > STAGE_FRAGS ? nf[k] : STAGE_FRAGS;
  }
};
template <int HT, int OT, int DEPTH>
struct BwdLayoutOf {
  static constexpr BwdLayout value = BwdLayout(HT, OT, DEPTH);
};

// the device side of tr_grad_scale (gbnf_train.hip): alpha = the power of two that puts the largest upstream entry at [4, 8)
__device__ __forceinline__ void bwd_grad_scale(unsigned max_bits, float& alpha, float& inv_alpha) {
  const int e = (int)((max_bits >> 23) & 255u);
  if (max_bits == 0u || e == 255) { alpha = 1.0f; inv_alpha = 1.0f; return; }
  int k = 2 - (e - 127);
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  alpha = __builtin_bit_cast(float, (unsigned)(k + 127) << 23);
  inv_alpha = __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
}

// sum over the 16 lanes of a lane group (the wave's 16 samples): a lane group is one DPP row, so four rotate-and-add steps on
// the VALU leave the sum in every lane (__shfl_xor compiles to ds_bpermute: 128 LDS round trips per step, a third of the kernel)
__device__ __forceinline__ float bwd_sum16(float v) {
  auto ror = [](float a, auto n_c) {
    constexpr int CTRL = 0x120 + decltype(n_c)::value;      // row_ror:n
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), CTRL, 0xf, 0xf, false));
  };
  v += ror(v, std::integral_constant<int, 8>{});
  v += ror(v, std::integral_constant<int, 4>{});
  v += ror(v, std::integral_constant<int, 2>{});
  v += ror(v, std::integral_constant<int, 1>{});
  return v;
}

// L2-warming look-ahead (GBNF_BWD_PREFETCH): one LDS-DMA dword per lane from 64 DIFFERENT cache lines (lane l reads the first word of
// line l of a block) into a dead 256-byte LDS cell -- no register is written, so nothing can be clobbered when the data lands, and the
// kernel's real loads of those lines (the saved activations a pass needs two tiles ahead, the next step's boundary operands) then
// find them in the L2 / Infinity Cache instead of waiting out an HBM round trip with nothing on the SIMD to switch to.  Counts as ONE
// vector-memory operation (vmcnt) like every LDS-DMA; the dword form is what tools/isa_hazard_lint.py tells from a staging DMA (x4).
__device__ __forceinline__ void bwd_warm_lines(const float* base, unsigned lane_off, uint32_t* dead_cell) {
  const unsigned m0v = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)dead_cell;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(lane_off), "s"(base), "s"(m0v) : "memory", "m0");
}

// Waves per SIMD the 4-wave form is compiled for: 2 (256 registers, two workgroups share a CU and cover each other's memory
// latency: 498 -> 3xx us at N = 65536) where the kernel fits -- its register count is 8 per hidden tile (the saved second-layer
// activations and the split g_a2 operands) + ~125 (measured: HT = 14 -> 235) -- else 1 (512 registers).
constexpr int bwd_hx3_occupancy(int KIND, int HT, int OT, int DEPTH = 1) {
#ifdef GBNF_BWD_OCC
  return GBNF_BWD_OCC;
#else
  // (two hidden -> hidden layers: a second set of split gradient operands, 4 registers per hidden tile)
  // (a one-block ResidualNet at 7 hidden tiles spills 65 registers with two waves per SIMD and is still 12 % faster than with one:
  //  measured 0.98 against 1.10 ms of backward + wgrad at N = 65536 -- the skip tiles are NOT counted here)
  return (DEPTH >= 2 ? 12 : 8) * HT + (DEPTH == 4 ? 60 : 0) + 125 + (KIND == GBNF_KIND_REALNVP ? 8 * OT + 8 : 0) <= 250 ? 2 : 1;
#endif
}
template <int KIND, int HT, int OT, int ACTA, int ACTB, int WV, int DEPTH = 1>
__global__ void __launch_bounds__(64 * WV, WV == 8 ? 2 : bwd_hx3_occupancy(KIND, HT, OT, DEPTH)) bwd_kernel_hx3(const FlowLaunch p) {
  static_assert((DEPTH >= 0 && DEPTH <= 2) || (DEPTH == 4 && ACTA == 2), "coupling_network_depth 0, 1 or 2; ResidualNets of one or two blocks");
  // ACT == 2 (GBNF_ACT_RESIDUAL_RELU): a ResidualNet of ONE block (models/layers.py:246-301) = layer 0 -> [relu -> Linear -> relu ->
  // Linear] + layer 0's output -> final layer.  Backward: the final layer's input gradient g_t passes the block's exit unchanged
  // (no activation in front of the final layer), runs back through the two inner layers with relu', and is ADDED to the block's
  // input gradient (the skip connection): DEPTH = 2 with the raw g_t tiles kept in registers.
  constexpr bool RES = ACTA == 2;
  static_assert((ACTA == 2) == (ACTB == 2), "both nets of a step are ResidualNets or neither is");
  static_assert(!RES || DEPTH == 2 || DEPTH == 4, "a ResidualNet has two hidden -> hidden layers per block");
  // Two blocks (DEPTH = 4): three middle layers J = 4, 3, 2 ping-pong between the operand sets; behind layer 3 (the second block's
  // first Linear) the gradient meets the second block's skip path -- g_t1 = relu'(t1) (W3^T g_a3) + g_t2 -- and g_t1 REPLACES the kept
  // skip gradient (what the first block's skip hands to layer 0's output).
  constexpr int WAVES = WV, NP = 2, NT = 1, ZS = 17, IT = 2;
  constexpr int NNETS = (KIND == GBNF_KIND_REALNVP) ? 2 : 1;
  constexpr int NH = DEPTH + 1;                          // hidden activations per net: operand rows in | NH x act | NH x grad | out grad | out
  using FL = Hx3LayoutOf<HT, OT, NP, DEPTH>;             // the forward blob: only its per-step tables are read here
  using BL = BwdLayoutOf<HT, OT, DEPTH>;
  constexpr int HC = BL::value.HC, K0 = BL::value.K0, ROWS0 = BL::value.ROWS0;
  constexpr int STEP_WORDS_F = SMALL_WORDS + NNETS * FL::value.NET_WORDS;
  constexpr int STEP_WORDS_B = NNETS * BL::value.NET_WORDS;
  constexpr int STAGE_WORDS = BL::value.STAGE_FRAGS * 256;
  using Acc = AccT<1>;

  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int i = lane & 15, g = lane >> 4;
  const int d = p.d, K = p.n_steps;
  const int kb = p.k_begin, ke = p.k_end > 0 ? p.k_end : K;       // this launch's step range (FlowLaunch::k_begin / k_end)
  const uint32_t* __restrict__ blobF = p.blobs[0];
  const uint32_t* __restrict__ blobB = p.blobs_bwd[0];
  // rows >= n are padding: their upstream gradients are zero.  The last workgroup's spare waves own no rows: they shadow the
  // last tile (same loads, same values stored twice: the vector-memory count per stage is the same for every wave) and only
  // leave out the atomic parameter-gradient sums.
  const int64_t row0_raw = ((int64_t)blockIdx.x * WAVES + wave) * 16;
  const bool wave_ok = row0_raw < p.np;
  const int64_t row0 = wave_ok ? row0_raw : p.np - 16;
  const int np = (int)p.np;
  const int row = (int)row0 + i;

  float alpha = 1.0f, inv_alpha = 1.0f;
  if (p.gmax != nullptr) bwd_grad_scale(__builtin_amdgcn_readfirstlane(*p.gmax), alpha, inv_alpha);

  // ---- LDS: per-step tables (of the forward blob) | 2 stage slots | G tiles | scatter scratch
  uint32_t* SM = lds;
  uint32_t* STG = lds + K * SMALL_WORDS;
  float* G = reinterpret_cast<float*>(STG + 2 * STAGE_WORDS) + wave * ((d + 1) * ZS);     // gradient state, slot layout (+ a spare slot)
  float* SC = reinterpret_cast<float*>(STG + 2 * STAGE_WORDS) + WAVES * ((d + 1) * ZS) + wave * (32 * ZS);
  // every wave's 16-sample sums of the ActNorm / BatchNorm parameter gradients: [wave][step][2][64].  (Atomic adds into the
  // gradient buffer -- 4096 waves x 86 parameters x K steps on 430 addresses -- took 80 % of this kernel's time.)
  float* PG = reinterpret_cast<float*>(STG + 2 * STAGE_WORDS) + WAVES * ((d + 1) * ZS) + WAVES * (32 * ZS);
  for (int e = (int)threadIdx.x; e < WAVES * K * 128; e += 64 * WAVES) PG[e] = 0.0f;
  [[maybe_unused]] uint32_t* WARM = reinterpret_cast<uint32_t*>(PG + WAVES * K * 128);      // (GBNF_BWD_PREFETCH) 64 dead words

  // ---- weight staging (as in flow_kernel_hx3): the transposed blob is in consumption order, steps last to first
  using gwords = const __attribute__((address_space(1))) uint32_t*;
  using lptr = __attribute__((address_space(3))) void*;
  gwords next_src = (gwords)blobB + (size_t)(ke - 1) * STEP_WORDS_B;
  int gs = 0;
  const unsigned lane_b16 = (unsigned)lane * 16u;
  auto dma = [&](gwords src, uint32_t* dst) {
    lds_dma16(src, dst, lane_b16);
  };
  int later = 0;             // vector-memory operations issued behind the staging DMA in flight (stage_end)
  auto issue = [&](auto nf_c, int into) {
    constexpr int NF = decltype(nf_c)::value;
    uint32_t* dst = STG + (into & 1) * STAGE_WORDS;
#pragma unroll
    for (int k = 0; k * WAVES < NF; ++k) {
      const int f = wave + k * WAVES;
      if ((k + 1) * WAVES <= NF || f < NF) dma(next_src + f * 256, dst + f * 256);
    }
    next_src += NF * 256;
    __builtin_amdgcn_sched_barrier(0);       // nothing that is counted below moves in front of the DMA
    later = 0;
  };
  issue(std::integral_constant<int, BL::value.nf[0]>{}, 0);
  for (int s = 0; s < K; ++s) {
    const uint32_t* src = blobF + (size_t)s * STEP_WORDS_F;
    for (int w = (int)threadIdx.x * 4; w < SMALL_WORDS; w += 64 * WAVES * 4)
      *reinterpret_cast<i32x4*>(SM + s * SMALL_WORDS + w) = *reinterpret_cast<const i32x4*>(src + w);
  }
  // ---- upstream gradients -> G (through the final slot map), scaled -- or the gradient state the launch of the following step
  //      range parked (slot layout [d][np], already scaled)
  if (p.state_in != nullptr) {
    const int r = lane & 15, s0 = lane >> 4;
    const float* gin = p.state_in + row0 + r;
    for (int slot = s0; slot < d; slot += 4) G[slot * ZS + r] = gin[(int64_t)slot * p.np];
  } else {
    const uint32_t* tail = blobF + (size_t)K * STEP_WORDS_F;
    if (lane < d) {
      const int slot = (int)tail[lane];
      float gv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t n = row0 + r;
        gv[r] = (p.g_z != nullptr && n < p.n) ? p.g_z[n * d + lane] * alpha : 0.0f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) G[slot * ZS + r] = gv[r];
    }
  }
  const float gl = (p.g_ldj != nullptr && row0 + i < p.n) ? p.g_ldj[row0 + i] * alpha : 0.0f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  bool sat = false;
  const uint32_t* buf = STG;
  // `later`: vector-memory operations this wave has issued BEHIND the staging DMA of the stage in flight (operand stores,
  // activation prefetches): they may stay in flight across the barrier -- waiting for all of them (vmcnt(0)) makes every
  // stage as long as a store's round trip to HBM.  vmcnt counts in issue order and an undercount is safe; the value is a
  // compile-time constant at every stage end (reset by issue(), straight-line code up to the wait), and
  // tools/isa_hazard_lint.py re-counts the instructions between every staging DMA and its counted wait in the shipped ISA.
  Stamps st;                  // diagnostic builds (-DGBNF_STAMPS, tools/build_train_stamps2.sh): cycles per phase and wave
  auto stage_end = [&]() {
#ifdef GBNF_STAMPS
    const int phase_ = st.cur;
    st.mark(phase_);
#endif
    stage_wait_counted(later);
#ifdef GBNF_STAMPS
    st.mark(7);                // bucket 7: stage-end wait + barrier
    st.cur = phase_;
#endif
    ++gs;
  };
  auto frag = [&](int f) -> u32x4 { return *reinterpret_cast<const u32x4*>(buf + f * 256 + lane * 4); };
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
  // the padding hipcc omits on the taken side of a branch between a v_mfma and the first use of its result
  // (gbnf_flow_kernel_hx3.hip.h, mfma_tail_guard; tools/isa_hazard_lint.py checks every path of this kernel too)
  auto mfma_tail_guard = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto stage_finish = [&](bool early) {        // every stage ends in front of its last unit
    if (!early) { mfma_tail_guard(); return; }
    stage_end();
    preload();
  };
  preload();
  auto mac = [&](const Unit& a, const u32x4 (&x)[NP], Acc& acc) {
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) {
      acc.s[0] = mfma_narrow<0>(a.w[Products<2>::W[pr]], x[Products<2>::X[pr]], acc.s[0]);
      MFMA_ORDER_FENCE();
    }
  };
  auto split4 = [&](const f32x4& v, unsigned (&lo)[NP], unsigned (&hi)[NP]) {       // a tile's 4 values -> two register pairs of pieces
    f32x4 c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sat = sat || !(__builtin_fabsf(v[r]) <= 65504.0f);
      c[r] = __builtin_amdgcn_fmed3f(v[r], -65504.0f, 65504.0f);
    }
    split_pair<NP>(c[0], c[1], lo);
    split_pair<NP>(c[2], c[3], hi);
  };

  const int tile0 = (int)(row0 >> 4);
  const int hp16 = p.tr_hp * 16, op16 = p.tr_op * 16;
  const int h_off = tile0 * hp16 + 4 * g * 16 + i;         // unit 16 t + 4 g + r of this lane's sample: + (16 t + r) * 16
  const int o_off = tile0 * op16 + 4 * g * 16 + i;

  st.start();
  for (int step = ke - 1; step >= kb; --step) {
    st.set(0);
    const uint32_t* smt = SM + step * SMALL_WORDS;
    const float* trace = p.trace_in + (int64_t)step * d * p.np;
    float* acts = p.acts_out + (int64_t)step * NNETS * p.net_rows * p.np;
    const int32_t* ptab = p.bwd_tab + step * (2 * 4 * NENT);
    const int64_t g_na = p.bwd_goff[2 * step], g_nb = p.bwd_goff[2 * step + 1];

    // ---- (a) coupling backward: gradient of the net output(s) in the D layout, G[out slots] <- gradient w.r.t. the normalised y2
    LaneTable tout;
    tout.load(smt + SMALL_HDR + 160 + g * NENT);
    f32x4 gA[OT], gBo[OT];                       // net 0 / net 1 output gradients (rows 16 o + 4 g + r)
#pragma unroll
    for (int o = 0; o < OT; ++o) { gA[o] = f32x4{0.f, 0.f, 0.f, 0.f}; gBo[o] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // (c)'s inputs -- the step's normalised in-half state -- are requested here: behind the chain they would be a global round
    // trip per step with nothing to hide it
    LaneTable tin;
    tin.load(smt + SMALL_HDR + g * NENT);
    float yin[NENT];
#pragma unroll
    for (int e = 0; e < NENT; ++e) yin[e] = trace[(tin.slot[e] >= 0 ? tin.slot[e] : 0) * np + row];
    // ... and so are net 0's last-hidden-layer activations (56 loads for h = 215): one exposed round trip per step, not two
    f32x4 h2first[HT];
    {
      const float* h2p0 = acts + (int64_t)(p.tr_ip + DEPTH * p.tr_hp) * np + h_off;
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h2first[t][r] = ((GBNF_BWD_ABLATE & 4) || RES) ? 0.5f : h2p0[(16 * t + r) * 16];
    }
#if GBNF_BWD_PREFETCH
    {
      // this step's layer-0 activations (the W1^T passes ask for them two tiles ahead: less than an HBM round trip), and what the
      // step BEFORE this one (the next to be processed) reads at its top: its last-hidden-layer activations, net outputs, trace
      const unsigned l128 = (unsigned)lane * 128u;
      const int hp_lines = (p.tr_hp * 64 + 127) / 128, op_lines = (p.tr_op * 64 + 127) / 128;
      const bool prev = step > kb;
      const float* actp = prev ? acts - (int64_t)NNETS * p.net_rows * np : acts;          // (the first step: its own rows again)
#pragma unroll
      for (int net = 0; net < NNETS; ++net) {
        const float* a0b = acts + (int64_t)net * p.net_rows * np + (int64_t)p.tr_ip * np + tile0 * hp16;
        const float* a1b = actp + (int64_t)net * p.net_rows * np + (int64_t)(p.tr_ip + DEPTH * p.tr_hp) * np + tile0 * hp16;
        const float* ob = actp + (int64_t)net * p.net_rows * np + (int64_t)(p.tr_ip + 2 * NH * p.tr_hp + p.tr_op) * np + tile0 * op16;
        for (int l0 = 0; l0 < hp_lines; l0 += 64) {
          const unsigned lo = (unsigned)(l0 + lane < hp_lines ? l0 + lane : hp_lines - 1) * 128u;
          bwd_warm_lines(a0b, lo, WARM);
          bwd_warm_lines(a1b, lo, WARM);
        }
        bwd_warm_lines(ob, (unsigned)(lane < op_lines ? lane : op_lines - 1) * 128u, WARM);
      }
      (void)l128;
      const float* trp = prev ? trace - (int64_t)d * p.np : trace;
      bwd_warm_lines(trp + row0, (unsigned)(lane < d ? lane : d - 1) * (unsigned)np * 4u, WARM);
    }
#endif
    float y2v[NENT];
    {
      const float* oA = acts + (int64_t)(p.tr_ip + 2 * NH * p.tr_hp + p.tr_op) * np + o_off;     // the forward sweep's saved net outputs
      const float* oB = oA + (int64_t)p.net_rows * np;
      if (KIND == GBNF_KIND_GLOW && !p.additive) {
        constexpr int NE = (2 * OT < NENT) ? 2 * OT : NENT;
        float sh[NE], rw[NE], g2[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int o = e >> 1, pp = e & 1;
          const int sl = tout.slot[e] >= 0 ? tout.slot[e] : 0;
          sh[e] = oA[(16 * o + 2 * pp) * 16];
          rw[e] = oA[(16 * o + 2 * pp + 1) * 16];
          y2v[e] = trace[sl * np + row];
          g2[e] = G[sl * ZS + i];
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int o = e >> 1, pp = e & 1;
          const bool live = tout.slot[e] >= 0;
          const float ex = __expf(-(rw[e] + 2.0f));
          const float sc = 1.0f / (1.0f + ex);
          const float omsc = ex < 1e30f ? ex * sc : 1.0f;                   // 1 - scale
          const float gy = g2[e] * sc;
          gA[o][2 * pp] = live ? gy : 0.0f;                                  // d/d shift
          gA[o][2 * pp + 1] = live ? (g2[e] * (y2v[e] + sh[e]) * sc + gl) * omsc : 0.0f;   // d/d raw: z2 = (y2 + shift) s, ld += log s
          G[(live ? tout.slot[e] : d) * ZS + i] = gy;
        }
      } else {
        constexpr int NE = (4 * OT < NENT) ? 4 * OT : NENT;
        float sv[NE], g2[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int o = e >> 2, r = e & 3;
          const int sl = tout.slot[e] >= 0 ? tout.slot[e] : 0;
          sv[e] = (KIND == GBNF_KIND_REALNVP) ? oB[(16 * o + r) * 16] : 0.0f;     // the log-scale net's output
          y2v[e] = trace[sl * np + row];
          g2[e] = G[sl * ZS + i];
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int o = e >> 2, r = e & 3;
          const bool live = tout.slot[e] >= 0;
          if constexpr (KIND == GBNF_KIND_GLOW) {                            // additive: z2 = y2 + h
            gA[o][r] = live ? g2[e] : 0.0f;
            G[(live ? tout.slot[e] : d) * ZS + i] = g2[e];
          } else {                                                           // z2 = shift + y2 e^scale, ld += scale
            const float es = __expf(sv[e]);
            gA[o][r] = live ? g2[e] : 0.0f;                                  // d/d shift
            gBo[o][r] = live ? g2[e] * y2v[e] * es + gl : 0.0f;              // d/d scale
            G[(live ? tout.slot[e] : d) * ZS + i] = g2[e] * es;
          }
        }
      }
    }

    st.mark(0);                 // bucket 0: coupling backward (with the wait for its loads)
    // ---- (b) the dgrad chain of every net; its output (rows k = 16 o + 4 g + r of d loss / d net input) is summed in SC
#pragma unroll
    for (int net = 0; net < NNETS; ++net) {
      const f32x4 (&gOut)[OT] = (net == 0) ? gA : gBo;
      const int ACT = (net == 0) ? ACTA : ACTB;
      const bool relu_rt = ACT == 3 && __builtin_amdgcn_readfirstlane(smt[2 + net]) != 0;
      float* an = acts + (int64_t)net * p.net_rows * np;
      // saved activations (forward sweep) of hidden layer l: rows ip + l hp; gradient-side operands for wgrad_kernel: ip + (NH + l) hp
      const float* h1p = an + (int64_t)p.tr_ip * np + h_off;                          // layer 0 (the last pass layer's act')
      const float* h2p = an + (int64_t)(p.tr_ip + DEPTH * p.tr_hp) * np + h_off;      // the last hidden layer
      [[maybe_unused]] const float* hmp = an + (int64_t)(p.tr_ip + p.tr_hp) * np + h_off;      // (DEPTH = 2) the middle one
      float* d1p = an + (int64_t)(p.tr_ip + NH * p.tr_hp) * np + h_off;
      float* d2p = an + (int64_t)(p.tr_ip + (NH + DEPTH) * p.tr_hp) * np + h_off;
      [[maybe_unused]] float* dmp = an + (int64_t)(p.tr_ip + (NH + 1) * p.tr_hp) * np + h_off;
      float* dop = an + (int64_t)(p.tr_ip + 2 * NH * p.tr_hp) * np + o_off;
      auto dact = [&](float gv, float hv) {       // gv * act'(pre-activation), through the saved activation hv
        const float t = gv * __builtin_fmaf(-hv, hv, 1.0f), r = hv > 0.0f ? gv : 0.0f;
        if (ACT == GBNF_ACT_TANH) return t;
        if (ACT == GBNF_ACT_RELU || ACT == 2) return r;
        return relu_rt ? r : t;
      };
      // the chain's input: the output gradient, emitted and split two tiles per k-chunk
      u32x4 gO[K0][NP];
#pragma unroll
      for (int c = 0; c < K0; ++c)
#pragma unroll
        for (int k = 0; k < NP; ++k) gO[c][k] = u32x4{0, 0, 0, 0};
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        if (16 * o < p.tr_op) {
#pragma unroll
          for (int r = 0; r < 4; ++r) dop[(16 * o + r) * 16] = gOut[o][r];
        }
        unsigned lo[NP], hi[NP];
        split4(gOut[o], lo, hi);
#pragma unroll
        for (int k = 0; k < NP; ++k) { gO[o >> 1][k][2 * (o & 1)] = lo[k]; gO[o >> 1][k][2 * (o & 1) + 1] = hi[k]; }
      }
      // all of the second hidden layer's saved activations are requested up front (layer W3^T is short)
      f32x4 h2v[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        if constexpr (RES) {
          h2v[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else if (net == 0) {
          h2v[t] = h2first[t];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) h2v[t][r] = (GBNF_BWD_ABLATE & 4) ? 0.5f : h2p[(16 * t + r) * 16];
        }
      }

      f32x4 gskip[RES ? HT : 1];                  // (ResidualNet) g_t, tile by tile: added to the block's input gradient
      u32x4 gB[HC][NP];                           // g_a2 = (W3^T g_o) * act'(h2), split: the B operands of the W2^T layer
#pragma unroll
      for (int k = 0; k < NP; ++k) gB[HC - 1][k] = u32x4{0, 0, 0, 0};
      {
        f32x4 rawp = f32x4{0.f, 0.f, 0.f, 0.f};
        auto finish_tile = [&](int t, const f32x4& raw) {
          f32x4 ga;
          if constexpr (RES) {
            ga = raw;                              // d/dt: the final layer reads t itself
            gskip[t] = raw;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) ga[r] = dact(raw[r], h2v[t][r]);
          }
          if (!(GBNF_BWD_ABLATE & 2)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) d2p[(16 * t + r) * 16] = ga[r];
            later += 4;
          }
          unsigned lo[NP], hi[NP];
          split4(ga, lo, hi);
#pragma unroll
          for (int k = 0; k < NP; ++k) { gB[t >> 1][k][2 * (t & 1)] = lo[k]; gB[t >> 1][k][2 * (t & 1) + 1] = hi[k]; }
        };
        auto l0_stage = [&](auto sI_c) {
          constexpr int sI = decltype(sI_c)::value;
          issue(std::integral_constant<int, BL::value.nf[sI + 1]>{}, gs + 1);
          constexpr int t0 = sI * ROWS0;
          constexpr int cnt = (HT - t0 < ROWS0) ? HT - t0 : ROWS0;
          constexpr int NU = cnt * K0;
          Unit A[3];
          A[0] = N0;
          if (NU > 1) A[1] = N1;
          Acc cur;
#pragma unroll
          for (int n = 0; n < NU; ++n) {
            const int t = t0 + n / K0, c = n % K0;
            if (n + 2 < NU) load_unit(A[(n + 2) % 3], n + 2);
            if (n == NU - 1) stage_finish(true);
            if (c == 0) cur.init(f32x4{0.f, 0.f, 0.f, 0.f});
            mac(A[n % 3], gO[c], cur);
            if (c == K0 - 1) {
              if (t > 0) finish_tile(t - 1, rawp);
              rawp = cur.total();
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          stage_finish(false);
        };
        auto l0_all = [&](auto self, auto s_c) -> void {
          constexpr int sI = decltype(s_c)::value;
          if constexpr (sI < BL::value.N_L0) {
            l0_stage(s_c);
            self(self, std::integral_constant<int, sI + 1>{});
          }
        };
        st.mark(5);             // bucket 5: net start (output-gradient stores and split, activation requests)
        st.set(1);
        l0_all(l0_all, std::integral_constant<int, 0>{});
        finish_tile(HT - 1, rawp);
        st.mark(1);             // bucket 1: W3^T stages
        st.set(2);
      }

      Acc outG[IT];
#pragma unroll
      for (int o = 0; o < IT; ++o) outG[o].init(f32x4{0.f, 0.f, 0.f, 0.f});
      // ---- (DEPTH >= 2) the middle layers J = DEPTH .. 2: WJ^T, one output tile per stage; tile u-1 times act'(saved activation of
      //      layer J - 1) is emitted / split during pass u into the OTHER operand set -- gB -> gB2 -> gB -> gB2: the B operands of the
      //      next layer down.  Fully unrolled: the destination register of a finished tile is a compile-time index.
      u32x4 gB2[DEPTH >= 2 ? HC : 1][NP];
      if constexpr (DEPTH == 4) {
        auto mid_layer = [&](auto j_c, auto& gIn, auto& gOut) {
          constexpr int J = decltype(j_c)::value;
          const float* hjp = an + (int64_t)(p.tr_ip + (J - 1) * p.tr_hp) * np + h_off;           // saved activations of layer J - 1
          float* djp = an + (int64_t)(p.tr_ip + (NH + J - 1) * p.tr_hp) * np + h_off;         // ... and its gradient-side operand rows
#pragma unroll
          for (int k = 0; k < NP; ++k) gOut[HC - 1][k] = u32x4{0, 0, 0, 0};
          f32x4 prem = f32x4{0.f, 0.f, 0.f, 0.f};
          auto load_hm = [&](int t) {
            f32x4 v;
            const int tt = t < HT ? t : HT - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (GBNF_BWD_ABLATE & 4) ? 0.5f : hjp[(16 * tt + r) * 16];
            return v;
          };
          f32x4 hmv[2] = {load_hm(0), load_hm(1)};       // saved activations of the even / odd tile that is finished next
          auto finish_mid = [&](auto t_c) {              // tile t of this layer's input gradient -> operand workspace + gOut
            constexpr int t = decltype(t_c)::value;
            f32x4 ga;
#pragma unroll
            for (int r = 0; r < 4; ++r) ga[r] = dact(prem[r], hmv[t & 1][r]);
            if constexpr (RES && (J & 1) == 1 && J > 1) {       // the entry of a block that is not the first: + the skip path, and on
              ga += gskip[t];
              gskip[t] = ga;
            }
            if (!(GBNF_BWD_ABLATE & 2)) {
#pragma unroll
              for (int r = 0; r < 4; ++r) djp[(16 * t + r) * 16] = ga[r];
              later += 4;
            }
            unsigned lo[NP], hi[NP];
            split4(ga, lo, hi);
#pragma unroll
            for (int k = 0; k < NP; ++k) { gOut[t >> 1][k][2 * (t & 1)] = lo[k]; gOut[t >> 1][k][2 * (t & 1) + 1] = hi[k]; }
          };
          auto mid_pass = [&](auto u_c) {
            constexpr int u = decltype(u_c)::value;
            issue(std::integral_constant<int, NP * HC>{}, gs + 1);      // the next pass of this layer or pass 0 of the next layer down
            __builtin_amdgcn_sched_barrier(0);       // (the stores + loads below stay BEHIND the staging DMA: stage_end counts on it)
            Unit A[3];
            A[0] = N0;
            A[1] = N1;
            Acc acc;
            acc.init(f32x4{0.f, 0.f, 0.f, 0.f});
            if constexpr (u > 0) {
              finish_mid(std::integral_constant<int, (u > 0 ? u - 1 : 0)>{});
              // (the last pass has no tile u + 1 to request: a load whose value is never used would be dropped by the compiler and the
              //  counted wait below would then let the next stage's staging DMA slip)
              if constexpr (u + 1 < HT) {
                hmv[(u - 1) & 1] = load_hm(u + 1);
                later += 4;
              }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < HC; ++n) {
              if (n + 2 < HC) load_unit(A[(n + 2) % 3], n + 2);
              if (n == HC - 1) stage_finish(true);
              mac(A[n % 3], gIn[n], acc);
              __builtin_amdgcn_sched_barrier(0);
            }
            prem = acc.total();
            stage_finish(false);
          };
          auto mid_all = [&](auto self, auto u_c) -> void {
            constexpr int u = decltype(u_c)::value;
            if constexpr (u < HT) {
              mid_pass(u_c);
              self(self, std::integral_constant<int, u + 1>{});
            }
          };
          mid_all(mid_all, std::integral_constant<int, 0>{});
          finish_mid(std::integral_constant<int, HT - 1>{});
        };
        mid_layer(std::integral_constant<int, 4>{}, gB, gB2);
        mid_layer(std::integral_constant<int, 3>{}, gB2, gB);
        mid_layer(std::integral_constant<int, 2>{}, gB, gB2);
      }
      // (DEPTH = 2 keeps its own hand-written block below: the generic form compiled, for the one-block ResidualNet at 16 hidden tiles,
      //  to a kernel that faulted in workgroups with spare waves -- 101 spilled registers, cause not found, HISTORY round 5 -- while this
      //  form of the same arithmetic is the one every depth-2 test and stress run of the round has passed on)
      // ---- (DEPTH = 2) W2^T: one output tile per stage; tile u-1 times act'(h of the middle layer) is emitted / split during pass u
      //      into the second operand set gB2 -- the B operands of the W1^T passes.  Fully unrolled: the destination register of
      //      a finished tile is a compile-time index.
      if constexpr (DEPTH == 2) {
#pragma unroll
        for (int k = 0; k < NP; ++k) gB2[HC - 1][k] = u32x4{0, 0, 0, 0};
        f32x4 prem = f32x4{0.f, 0.f, 0.f, 0.f};
        auto load_hm = [&](int t) {
          f32x4 v;
          const int tt = t < HT ? t : HT - 1;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (GBNF_BWD_ABLATE & 4) ? 0.5f : hmp[(16 * tt + r) * 16];
          return v;
        };
        f32x4 hmv[2] = {load_hm(0), load_hm(1)};       // saved activations of the even / odd tile that is finished next
        auto finish_mid = [&](auto t_c) {              // tile t of the middle layer's gradient -> operand workspace + gB2
          constexpr int t = decltype(t_c)::value;
          f32x4 ga;
#pragma unroll
          for (int r = 0; r < 4; ++r) ga[r] = dact(prem[r], hmv[t & 1][r]);
          if (!(GBNF_BWD_ABLATE & 2)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dmp[(16 * t + r) * 16] = ga[r];
            later += 4;
          }
          unsigned lo[NP], hi[NP];
          split4(ga, lo, hi);
#pragma unroll
          for (int k = 0; k < NP; ++k) { gB2[t >> 1][k][2 * (t & 1)] = lo[k]; gB2[t >> 1][k][2 * (t & 1) + 1] = hi[k]; }
        };
        auto mid_pass = [&](auto u_c) {
          constexpr int u = decltype(u_c)::value;
          issue(std::integral_constant<int, NP * HC>{}, gs + 1);      // the next pass of this layer or pass 0 of the W1^T layer
          __builtin_amdgcn_sched_barrier(0);       // (the stores + loads below stay BEHIND the staging DMA: stage_end counts on it)
          Unit A[3];
          A[0] = N0;
          A[1] = N1;
          Acc acc;
          acc.init(f32x4{0.f, 0.f, 0.f, 0.f});
          if constexpr (u > 0) {
            finish_mid(std::integral_constant<int, (u > 0 ? u - 1 : 0)>{});
            // (the last pass has no tile u + 1 to request: a load whose value is never used would be dropped by the compiler and the
            //  counted wait below would then let the next stage's staging DMA slip)
            if constexpr (u + 1 < HT) {
              hmv[(u - 1) & 1] = load_hm(u + 1);
              later += 4;
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int n = 0; n < HC; ++n) {
            if (n + 2 < HC) load_unit(A[(n + 2) % 3], n + 2);
            if (n == HC - 1) stage_finish(true);
            mac(A[n % 3], gB[n], acc);
            __builtin_amdgcn_sched_barrier(0);
          }
          prem = acc.total();
          stage_finish(false);
        };
        auto mid_all = [&](auto self, auto u_c) -> void {
          constexpr int u = decltype(u_c)::value;
          if constexpr (u < HT) {
            mid_pass(u_c);
            self(self, std::integral_constant<int, u + 1>{});
          }
        };
        mid_all(mid_all, std::integral_constant<int, 0>{});
        finish_mid(std::integral_constant<int, HT - 1>{});
      }
      auto& gBin = [&]() -> auto& {                 // the B operands of the W1^T passes
        if constexpr (DEPTH >= 2) return gB2;
        else return gB;
      }();
      if constexpr (DEPTH == 0) {
        // ---- no hidden -> hidden layer: W0^T contracts the layer-0 gradient gB, CGI chunks (IT tiles each) per stage
        constexpr int CGI = BL::value.CGI, N_IN = BL::value.N_IN;
        auto in_stage = [&](auto k_c) {
          constexpr int k = decltype(k_c)::value;
          constexpr int c0 = k * CGI;
          constexpr int cnt = (HC - c0 < CGI) ? HC - c0 : CGI;
          constexpr int NU = cnt * IT;
          if constexpr (k + 1 < N_IN) {
            issue(std::integral_constant<int, BL::value.nf[BL::value.N_L0 + k + 1]>{}, gs + 1);
          } else {
            if (net + 1 < NNETS || step > kb) {
              if (net + 1 == NNETS) next_src = (gwords)blobB + (size_t)(step - 1) * STEP_WORDS_B;
              issue(std::integral_constant<int, BL::value.nf[0]>{}, gs + 1);
            } else {
              later = 0;               // (no DMA to wait for: the same constant on both paths)
            }
          }
          Unit A[3];
          A[0] = N0;
          if (NU > 1) A[1] = N1;
#pragma unroll
          for (int n = 0; n < NU; ++n) {
            if (n + 2 < NU) load_unit(A[(n + 2) % 3], n + 2);
            if (n == NU - 1) stage_finish(true);
            mac(A[n % 3], gB[c0 + n / IT], outG[n % IT]);
            __builtin_amdgcn_sched_barrier(0);
          }
          stage_finish(false);
        };
        static_assert(N_IN <= 4, "input stages of a depth-0 net");
        st.set(3);
        in_stage(std::integral_constant<int, 0>{});
        if constexpr (N_IN > 1) in_stage(std::integral_constant<int, 1>{});
        if constexpr (N_IN > 2) in_stage(std::integral_constant<int, 2>{});
        if constexpr (N_IN > 3) in_stage(std::integral_constant<int, 3>{});
      } else {
      // ---- W2^T: one output tile per stage; tile u-1 times act'(h1) is emitted / split during pass u and consumed, two
      //      tiles per chunk, by the W1^T tiles (two output tiles: the net input's <= 32 rows)
      u32x4 hO[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) hO[k] = u32x4{0, 0, 0, 0};
      f32x4 pre = f32x4{0.f, 0.f, 0.f, 0.f};
      auto load_h1 = [&](int t) {
        f32x4 v;
        const int tt = t < HT ? t : HT - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (GBNF_BWD_ABLATE & 4) ? 0.5f : h1p[(16 * tt + r) * 16];
        return v;
      };
      f32x4 hE = load_h1(0), hOd = load_h1(1);       // saved activations of the even / odd tile that is finished next
      auto finish_h1 = [&](int t, const f32x4& hv, int half) {        // tile t of g_a1 -> operand workspace + half `half` of hO
        f32x4 ga;
#pragma unroll
        for (int r = 0; r < 4; ++r) ga[r] = dact(pre[r], hv[r]);
        if constexpr (RES) ga += gskip[t];          // (t is a compile-time index here: the ResidualNet passes are fully unrolled)
        if (!(GBNF_BWD_ABLATE & 2)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) d1p[(16 * t + r) * 16] = ga[r];
          later += 4;
        }
        unsigned lo[NP], hi[NP];
        split4(ga, lo, hi);
#pragma unroll
        for (int k = 0; k < NP; ++k) { hO[k][2 * half] = lo[k]; hO[k][2 * half + 1] = hi[k]; }
      };
      // PREV: 0 = no previous tile (u = 0); 1 = tile u-1 is even (first half of its chunk); 2 = it is odd (second half):
      // chunk (u-2)/2 is consumed at the end of this pass
      auto pass = [&](int u, auto prev_c, auto last_c) {
        constexpr int PREV = decltype(prev_c)::value;
        constexpr bool LAST = decltype(last_c)::value;
        constexpr int NU = HC + (PREV == 2 ? IT : 0);
        constexpr int NF_NEXT = LAST ? NP * IT : (PREV == 1 ? NP * (HC + IT) : NP * HC);
        issue(std::integral_constant<int, NF_NEXT>{}, gs + 1);
        __builtin_amdgcn_sched_barrier(0);         // (the 4 stores + 4 loads below stay BEHIND the staging DMA: stage_end counts on it)
        Unit A[3];
        A[0] = N0;
        A[1] = N1;
        Acc acc;
        acc.init(f32x4{0.f, 0.f, 0.f, 0.f});
        // (the last pass has no tile u + 1 to request: a load whose value is never used would be dropped by the compiler and the
        // counted wait below would then let the drain's staging DMA slip)
        if (PREV == 1) { finish_h1(u - 1, hE, 0); if constexpr (!LAST) { hE = load_h1(u + 1); later += 4; } }
        if (PREV == 2) { finish_h1(u - 1, hOd, 1); if constexpr (!LAST) { hOd = load_h1(u + 1); later += 4; } }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NU; ++n) {
          if (n + 2 < NU) load_unit(A[(n + 2) % 3], n + 2);
          if (n == NU - 1) {
            stage_finish(true);
          }
          if (n < HC) mac(A[n % 3], gBin[n], acc);
          else mac(A[n % 3], hO, outG[n - HC]);
          __builtin_amdgcn_sched_barrier(0);
        }
        pre = acc.total();
        stage_finish(false);
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
        if constexpr (RES) {       // (the skip tile of pass u is a register array indexed by u: compile-time passes)
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
        if constexpr (HT % 2 == 1) {
          pass(u, I1{}, BF{});
          pass(u + 1, I2{}, BT{});
        } else {
          pass(u, I1{}, BT{});
        }
      }
      st.mark(2);               // bucket 2: W2^T passes
      st.set(3);
      // ---- drain: last tile of g_a1, last W1^T chunk; the next net's / step's first stage goes in flight
      {
        if (net + 1 < NNETS || step > kb) {
          if (net + 1 == NNETS) next_src = (gwords)blobB + (size_t)(step - 1) * STEP_WORDS_B;
          issue(std::integral_constant<int, BL::value.nf[0]>{}, gs + 1);
        } else {
          later = 0;               // (no DMA to wait for: the same constant on both paths)
        }
        Unit A[IT];
        A[0] = N0;
        A[1] = N1;
        constexpr bool odd_last = ((HT - 1) & 1) != 0;
        finish_h1(HT - 1, odd_last ? hOd : hE, odd_last ? 1 : 0);
        if (!odd_last) {
#pragma unroll
          for (int k = 0; k < NP; ++k) { hO[k][2] = 0; hO[k][3] = 0; }
        }
#pragma unroll
        for (int o = 0; o < IT; ++o) {
          if (o == IT - 1) stage_finish(true);
          mac(A[o], hO, outG[o]);
        }
        stage_finish(false);
      }
      }
      // the net's contribution to d loss / d net input: rows k = 16 o + 4 g + r of this lane's sample
#pragma unroll
      for (int o = 0; o < IT; ++o) {
        const f32x4 v = outG[o].total();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* q = SC + (16 * o + 4 * g + r) * ZS + i;
          *q = (net == 0) ? v[r] : *q + v[r];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    st.mark(3);                 // bucket 3: drain, net-input gradient to LDS

    // ---- (c) + (d): normalisation backward of every slot (in-half: pass-through gradient + the nets' contribution), its
    //      parameter gradients summed over the wave's 16 samples
    auto norm_bwd = [&](const LaneTable& tb, int e, float gy, float y, int m) {
      float gx, ga, gb;
      if constexpr (KIND == GBNF_KIND_GLOW) {
        gx = gy * tb.p1[e];                       // y = (x + bias) e^logs
        ga = gx;                                  // d/d bias
        gb = gy * y + gl;                         // d/d logs (and logdet += logs for every sample)
      } else {
        gx = gy * (tb.p2[e] / tb.p1[e]);          // y = (x - mean) / sqrt(var + eps) * e^log_gamma + beta
        ga = gy * (y - tb.p3[e]) + gl;            // d/d log_gamma
        gb = gy;                                  // d/d beta
      }
      if (!(GBNF_BWD_ABLATE & 1)) {
        // NO divergent block here (round 5).  Every lane of a 16-lane group holds the group's sum and stores it -- the same value to
        // the same address; a group without a parameter (m < 0: uniform per lane group, a RealNVP step without BatchNorm has none) and
        // a spare wave store into the wave's scatter scratch, which is dead at this point (gyi has been read).  Why: an unshipped
        // variant of this kernel (one-block ResidualNet, 16 hidden tiles, 101 spilled registers) reloaded an ADDRESS register from
        // its spill slot with valid values in the lanes i == 0 only -- the lanes of the `if (i == 0)` block that used to stand here
        // -- and float data in all others (rocgdb register dump, HISTORY): under that much register pressure hipcc let a value that
        // is live across the divergent block be (re)defined inside it.  No divergent region, nothing to get wrong.
        ga = bwd_sum16(ga);
        gb = bwd_sum16(gb);
        const bool keep = m >= 0 && wave_ok;
        float* q = keep ? PG + ((wave * K + step) * 2) * 64 + m : SC + 2 * g;
        q[0] = ga * inv_alpha;
        q[keep ? 64 : 1] = gb * inv_alpha;
      }
      return gx;
    };
    // (every LDS read of the section in front of its first write: the slots of a step are distinct, which the compiler cannot
    //  know -- interleaved, each entry waited for its own LDS round trip behind the previous entry's write.  The two tables
    //  are read again here rather than kept in 32-64 registers across the chain: the kernel fits 256 registers with them gone)
    int mi[NENT], mo[NENT];             // parameter index of every table entry (bwd_tab), -1: none
#pragma unroll
    for (int e = 0; e < NENT; ++e) {
      mi[e] = ptab[g * NENT + e];
      mo[e] = ptab[4 * NENT + g * NENT + e];
    }
    tin.load(smt + SMALL_HDR + g * NENT);
    tout.load(smt + SMALL_HDR + 160 + g * NENT);
    float gyi[NENT];
#pragma unroll
    for (int e = 0; e < NENT; ++e) {
      const bool live = tin.slot[e] >= 0;
      const int sl = live ? tin.slot[e] : 0;
      gyi[e] = G[sl * ZS + i] + SC[(8 * g + e) * ZS + i];
      mi[e] = live ? mi[e] : -1;
    }
    constexpr int NE_O = (KIND == GBNF_KIND_GLOW) ? ((2 * OT < NENT) ? 2 * OT : NENT) : ((4 * OT < NENT) ? 4 * OT : NENT);
    const int ne_o = (KIND == GBNF_KIND_GLOW && p.additive) ? ((4 * OT < NENT) ? 4 * OT : NENT) : NE_O;
    float gyo[NENT];
#pragma unroll
    for (int e = 0; e < NENT; ++e) {
      const bool live = tout.slot[e] >= 0;
      gyo[e] = G[(live ? tout.slot[e] : 0) * ZS + i];
      mo[e] = live ? mo[e] : -1;
    }
#pragma unroll
    for (int e = 0; e < NENT; ++e) {
      const bool live = tin.slot[e] >= 0;
      const float gx = norm_bwd(tin, e, gyi[e], yin[e], mi[e]);
      G[(live ? tin.slot[e] : d) * ZS + i] = gx;
    }
#pragma unroll
    for (int e = 0; e < NENT; ++e) {
      if (e < ne_o) {
        const bool live = tout.slot[e] >= 0;
        const float gx = norm_bwd(tout, e, gyo[e], y2v[e], mo[e]);
        G[(live ? tout.slot[e] : d) * ZS + i] = gx;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    st.mark(4);                 // bucket 4: normalisation backward, parameter sums
  }
#ifdef GBNF_STAMPS
  if (p.dbg != nullptr && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) p.dbg[((size_t)blockIdx.x * WAVES + wave) * 8 + k] = st.acc[k];
  }
#endif

  // ---- the gradient state in front of step kb: parked for the launch of the preceding range (slot layout, still scaled) ...
  if (p.state_out != nullptr) {
    const int r = lane & 15, s0 = lane >> 4;
    float* gout = p.state_out + row0 + r;
    if (wave_ok)
      for (int slot = s0; slot < d; slot += 4) gout[(int64_t)slot * p.np] = G[slot * ZS + r];
  }
  // ---- ... or d loss / d x: slot j = feature j at the input of step 0
  if (p.state_out == nullptr && p.g_x != nullptr && lane < d) {
#pragma unroll 8
    for (int r = 0; r < 16; ++r) {
      const int64_t n = row0 + r;
      if (n < p.n) p.g_x[n * d + lane] = G[lane * ZS + r] * inv_alpha;
    }
  }
  if (p.sat != nullptr && __any(sat) && lane == 0) atomicAdd(p.sat, 1ull);
  // ---- this workgroup's parameter-gradient sums, waves added in a fixed order
  __syncthreads();
  for (int e = kb * 128 + (int)threadIdx.x; e < ke * 128; e += 64 * WAVES) {        // (this range's steps only)
    float v = 0.0f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) v += PG[w * K * 128 + e];
    p.partials[(int64_t)blockIdx.x * K * 128 + e] = v;
  }
}

inline size_t bwd_hx3_lds_bytes(int n_steps, int waves, int stage_frags, int d) {
  return ((size_t)n_steps * SMALL_WORDS + 2 * (size_t)stage_frags * 256 + (size_t)waves * (d + 1) * 17 + (size_t)waves * 32 * 17 +
          (size_t)waves * n_steps * 128 + (GBNF_BWD_PREFETCH ? 64 : 0)) * 4;
}

// 4-wave workgroups, one wave per SIMD (the kernel keeps a whole layer of saved activations in registers: 290-330 of a lone
// wave's 512); the 8-wave form exists for geometries whose 4-wave LDS does not fit at all
inline int bwd_hx3_waves(int n_steps, int stage_frags, int d) {
  return bwd_hx3_lds_bytes(n_steps, 4, stage_frags, d) <= 160 * 1024 ? 4 : 8;
}

template <int KIND, int HT, int OT, int ACTA, int ACTB, int WV, int DEPTH>
static hipError_t bwd_launch_wv(FlowLaunch p, hipStream_t s) {
  constexpr BwdLayout L(HT, OT, DEPTH);
  const size_t lds = bwd_hx3_lds_bytes(p.n_steps, WV, L.STAGE_FRAGS, p.d);
  if (lds > 160 * 1024 || p.n_steps > LDS_TABLE_STEPS) return hipErrorInvalidValue;
  const long long tiles = p.np / 16;               // every padded row (np is a multiple of 32): wgrad_kernel sums over all of them
  const long long grid = (tiles + WV - 1) / WV;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)bwd_kernel_hx3<KIND, HT, OT, ACTA, ACTB, WV, DEPTH>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((bwd_kernel_hx3<KIND, HT, OT, ACTA, ACTB, WV, DEPTH>), dim3((unsigned)grid), dim3(64 * WV), lds, s, p);
  return hipGetLastError();
}

// registry key: VariantKey{kind, ht, -3, /*ks1*/ 2 (backward), ot, /*nt*/ 1, depth, act_a, act_b}
#define GBNF_INSTANTIATE_HX3_BWD(KIND, HT, OT, ACTA, ACTB, DEPTH)                                           \
  namespace gbnf {                                                                                          \
  static hipError_t launch_hx3b_##KIND##_##HT##_##OT##_##ACTA##_##ACTB##_##DEPTH(const FlowLaunch& p0, unsigned, hipStream_t s) { \
    constexpr BwdLayout L(HT, OT, DEPTH);                                                                   \
    if (bwd_hx3_waves(p0.n_steps, L.STAGE_FRAGS, p0.d) == 4) return bwd_launch_wv<KIND, HT, OT, ACTA, ACTB, 4, DEPTH>(p0, s); \
    return bwd_launch_wv<KIND, HT, OT, ACTA, ACTB, 8, DEPTH>(p0, s);                                        \
  }                                                                                                         \
  static const int reg_hx3b_##KIND##_##HT##_##OT##_##ACTA##_##ACTB##_##DEPTH =                              \
      (register_variant(VariantKey{KIND, HT, -3, 2, OT, 1, DEPTH, ACTA, ACTB}, launch_hx3b_##KIND##_##HT##_##OT##_##ACTA##_##ACTB##_##DEPTH, \
                        "bwd_kernel_hx3<" #KIND "," #HT "," #OT "," #ACTA "," #ACTB "," #DEPTH ">"),        \
       0);                                                                                                  \
  }

}  // namespace gbnf
