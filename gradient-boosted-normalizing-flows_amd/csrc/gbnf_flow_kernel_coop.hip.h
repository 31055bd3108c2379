// gbnf_flow_kernel_coop.hip.h -- the LATENCY form of the fused f16x3 flow kernel (round 6).
//
// Why.  One `model.log_prob(x)` call at the reference's own batch sizes (density_experiment.py:80-81: 512 rows to train on,
// 1024 to evaluate) gives flow_kernel_hx3 at most one 16-sample wave per SIMD, and that wave walks 5 steps x 17 stages ALONE:
// 41 us whatever the batch, 64 rows or 2048 (profiles/r6_latency_ablations.txt: 15 us of it waiting for the stage's weights /
// barrier / fragments, 7 us activations, the rest the wave's own serial MFMA + VALU issue; more accumulator chains bought
// nothing).  Here the WV (4 or 8) waves of a workgroup share ONE tile of 16 NT samples of one component:
//
//   * wave w owns the hidden chunks [w CPW, (w+1) CPW) (a chunk = two 16-unit tiles = one k = 32 B operand) of every layer: a
//     quarter (an eighth) of the MFMAs, of the tanh / split work and of the weight fragments per wave;
//   * layer 0's activations meet in LDS (already split, already in B-operand order: D layout == B layout, so a wave stores its
//     two tiles of a chunk as the 16 bytes per lane the consumers read back with one lane-linear ds_read_b128);
//   * the output layer is split along K: a wave contracts the chunks it produced itself, straight from its registers, and the
//     WV partial sums of every output tile meet in LDS; the coupling epilogue of output tile (o, nt) runs on wave (o NT + nt) % WV;
//   * weights: every fragment is used by exactly one wave, so nothing is staged -- each wave streams its own fragments
//     L2 -> registers (buffer_load_dwordx4 with a scalar offset, 1 KiB per wave-instruction, the hx3 blob as gbnf_flow_create
//     packed it) through a ring of GBNF_COOP_RING fragments that runs ahead across layers, nets and steps (weights do not depend on data);
//   * 3 workgroup barriers per net (+1 per step) instead of 17 stage barriers.
//
// Same blob, same arithmetic per product (f16x3: 3 x v_mfma_f32_16x16x32_f16 on (hi, mid) pieces, f32 accumulate), same
// range protocol as flow_kernel_hx3 (out-of-range rows are marked NaN and the bf16x6 repair launch behind the call
// re-evaluates them).  Forward direction, depth-1 TanhNet / ReLUNet (also with the activation drawn per step).  Forms (the registry's
// `nt` field): 1 = 16-sample tiles on 4 waves, 2 = 32-sample tiles on 4 waves, 3 = 32-sample tiles on 8 waves (4 = 16 on 8: measured,
// not built).  The launcher picks this kernel when a call has so few sample tiles that every workgroup gets a CU to itself
// (gbnf_api.hip, pick_coop); csrc/variants.list says which forms are built for which geometry (measured: profiles/r6_coop_geometries.txt).
// What bounds it: profiles/r6_latency_ablations.txt (3) -- the L2 -> CU weight stream and, right under it, the wave's dependent issue.
//
// Reference semantics: the same as gbnf_flow_kernel_hx3.hip.h (models/glow.py:317-342, models/transformations.py:560-579).
#pragma once

#include "gbnf_flow_kernel_hx3.hip.h"

namespace gbnf {

#ifdef GBNF_COOP_ABLATE_BARRIER    // diagnostic: no workgroup barriers inside the step loop (races, timing only)
#define GBNF_COOP_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define GBNF_COOP_SYNC() __syncthreads()
#endif
#ifndef GBNF_COOP_L1_JOINT
#define GBNF_COOP_L1_JOINT 0       // 1: the hidden layer accumulates all of a wave's tiles at once (a chunk's B operands read once); measured slower
#endif
#ifndef GBNF_COOP_PR_OUTER
#define GBNF_COOP_PR_OUTER 1       // 1: the products of the tiles of a group are interleaved (a dependent MFMA is G MFMAs behind its predecessor)
#endif
#ifndef GBNF_COOP_HOIST_TABLES
#define GBNF_COOP_HOIST_TABLES 1   // 1: a step's lane tables are loaded ahead of the barrier in front of their use
#endif
#ifndef GBNF_COOP_XB_ALL
#define GBNF_COOP_XB_ALL 0         // 1: four-wave forms read every chunk of layer 0's activations once per layer, all reads in flight at its top (measured: 21.9 vs 21.1 us, no gain)
#endif
#ifndef GBNF_COOP_RING
#define GBNF_COOP_RING 16          // weight fragments (1 KiB each) a wave keeps in flight
#endif

// LDS bytes of a launch
inline size_t flow_coop_lds_bytes(int n_steps, int nt, int ht, int ot, int nnets, int d, bool lds_tables, int waves) {
  const Hx3Layout L(ht, ot, 2, 1);
  const int cpw = (L.HC + waves - 1) / waves;
  const size_t tables = lds_tables ? (size_t)n_steps * SMALL_WORDS : 0;
  const size_t z = (size_t)(d + 1) * (16 * nt + 1);
  const size_t act = (size_t)(nnets > 1 ? 2 : 1) * waves * cpw * nt * 2 * 256;
  const size_t red = (size_t)nnets * waves * ot * nt * 256;
  const size_t bias = 2 * (size_t)nnets * L.BIAS_WORDS;
  const size_t tail = (size_t)ot * nt * 64 + 16;
  return (tables + z + act + red + bias + tail) * 4 + 64;
}

// WV = 4: one wave per SIMD; WV = 8: two per SIMD -- the same weight bytes per workgroup spread over twice the waves (the form for
// 32-sample tiles: the L2 -> CU weight stream of a workgroup, ~35 B/clk, is what a tile costs, and eight waves keep their MFMA + VALU
// work under it)
template <int KIND, int HT, int OT, int NT, int ACTA, int ACTB, int WV>
__global__ void __launch_bounds__(64 * WV, 1) flow_kernel_coop(const FlowLaunch p) {
  // ACT 3 (GBNF_ACT_PER_STEP): the activation of a step's net comes from the step header (`--coupling_network random`, and every uniform
  // net of an activation pair without a variant of its own): both are computed and one is selected, as flow_kernel_hx3 does
  static_assert(ACTA == GBNF_ACT_TANH || ACTA == GBNF_ACT_RELU || ACTA == 3, "TanhNet / ReLUNet");
  static_assert(ACTB == GBNF_ACT_TANH || ACTB == GBNF_ACT_RELU || ACTB == 3, "TanhNet / ReLUNet");
  constexpr int NP = 2, WAVES = WV;
  constexpr int NNETS = (KIND == GBNF_KIND_REALNVP) ? 2 : 1;
  using LT = Hx3LayoutOf<HT, OT, NP, 1>;
  constexpr int HC = LT::value.HC, N_L0 = LT::value.N_L0, TL0 = LT::value.TL0;
  constexpr int NET_WORDS = LT::value.NET_WORDS, BIAS_WORDS = LT::value.BIAS_WORDS;
  constexpr int STEP_WORDS = SMALL_WORDS + NNETS * NET_WORDS;
  constexpr int CPW = (HC + WAVES - 1) / WAVES, TPW = 2 * CPW;        // chunks / hidden tiles per wave
  constexpr int ZS = 16 * NT + 1;
  // the wave's weight-fragment sequence of one (step, net), in consumption order:
  //   layer 0: local tile j, piece q | hidden layer: contraction chunk c, local tile j, piece | output layer: own chunk, output tile, piece
  constexpr int F0 = TPW * NP, F1 = CPW * HC * 2 * NP, F2 = CPW * OT * NP, F = F0 + F1 + F2;
  constexpr int R = F >= GBNF_COOP_RING ? GBNF_COOP_RING : (F >= 8 ? 8 : 4);       // ring depth (fragments in flight per wave)
  constexpr int FP = (F + R - 1) / R * R;                  // positions per (step, net), padded: the ring slot of a position is the same in every net
  static_assert(N_L0 <= 2, "layer 0 spans at most two stages of the blob");
  // product-outer MFMA order (see layer 0): measured +4 % for 32-sample tiles, -5 % for 16-sample ones (profiles/r6_latency_ablations.txt)
  constexpr bool PRO = GBNF_COOP_PR_OUTER != 0 && NT == 2;

  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int i = lane & 15;
  const int g = lane >> 4;

  // ---- work item = (component, batch, tile of 16 NT rows); XCD-aware as flow_kernel_hx3 (one component's weights per L2)
  int comp, tile, batch;
  {
    const int total = p.n_items;
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int base = total >> 3, rem = total & 7;
    const int q = xcd * base + (xcd < rem ? xcd : rem) + j;
    const int per_comp = p.n_tiles * p.n_batches;
    comp = q / per_comp;
    const int r = q - comp * per_comp;
    batch = r / p.n_tiles;
    tile = r - batch * p.n_tiles;
  }
  using gwords = const __attribute__((address_space(1))) uint32_t*;      // global memory, known to the compiler as such: global_load, not flat_load
  using gbytes = const __attribute__((address_space(1))) char*;
  using gu32x4 = const __attribute__((address_space(1))) u32x4*;
  const gwords blob = (gwords)p.blobs[p.c_begin + comp];
  const int d = p.d;
  const int64_t row0 = (int64_t)tile * (16 * NT);
  const float* __restrict__ xin = p.xs[batch];
  const int64_t out_base = (int64_t)comp * p.out_stride + (int64_t)batch * p.n;

  // ---- LDS: per-step tables | Z | ACT (hidden activations, split, B-operand order) | RED (output-layer partials) | biases x 2 | tail
  const bool lds_tables = p.lds_tables != 0;
  uint32_t* SM = lds;
  float* Z = reinterpret_cast<float*>(lds + (lds_tables ? p.n_steps * SMALL_WORDS : 0));     // [d + 1][ZS], shared by the workgroup's waves
  uint32_t* ACT = reinterpret_cast<uint32_t*>(Z) + (((d + 1) * ZS + 3) & ~3);                // (16-byte aligned: SMALL_WORDS is a multiple of 4)
  constexpr int ACT_WORDS = WAVES * CPW * NT * NP * 256;                                         // one net's activations: [chunk][nt][piece][lane][4]
  uint32_t* RED = ACT + (NNETS > 1 ? 2 : 1) * ACT_WORDS;                                     // [net][wave][o][nt][lane][4] f32
  uint32_t* BIAS = RED + NNETS * WAVES * OT * NT * 256;                                      // [2][net][BIAS_WORDS]
  float* LDP = reinterpret_cast<float*>(BIAS + 2 * NNETS * BIAS_WORDS);                      // [o][nt][64] log-det partials
  uint32_t* SATW = reinterpret_cast<uint32_t*>(LDP + OT * NT * 64);                          // [nt] rows that left the fp16 range

  // ---- this wave's fragment offsets inside a net block (words), uniform
  const int t_first = TPW * wave;
  int l0off[TPW], l1off[TPW], outoff[CPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    int t = t_first + j;
    t = t < HT ? t : HT - 1;                         // phantom tiles read the last tile's weights (their results are discarded)
    const int s = (N_L0 > 1 && t >= TL0) ? 1 : 0;
    l0off[j] = (s ? LT::value.off[N_L0 > 1 ? 1 : 0] : LT::value.off[0]) + (t - s * TL0) * NP * 256;
    l1off[j] = LT::value.off[N_L0] + t * (NP * HC * 256) + ((t - 1) / 2) * (NP * OT * 256);
  }
#pragma unroll
  for (int cl = 0; cl < CPW; ++cl) {
    int c = CPW * wave + cl;
    c = c < HC ? c : HC - 1;
    const int u = 2 * c + 2;                         // the pass whose stage carries output-layer chunk c (c < HC - 1); the drain carries the last one
    const int in_pass = LT::value.off[N_L0] + u * (NP * HC * 256) + ((u - 1) / 2) * (NP * OT * 256) + NP * HC * 256;
    outoff[cl] = c < HC - 1 ? in_pass : LT::value.off[N_L0 + HT];
  }
  // word offset of sequence position `pos` (a constant after unrolling)
  auto frag_off = [&](int pos) -> int {
    if (pos < F0) return l0off[pos / NP] + (pos % NP) * 256;
    pos -= F0;
    if (pos < F1) {
#if GBNF_COOP_L1_JOINT
      const int c = pos / (TPW * NP), j = (pos / NP) % TPW, q = pos % NP;
      return l1off[j] + (c * NP + q) * 256;
#else
      const int grp = pos / (HC * 2 * NP), rem = pos % (HC * 2 * NP);
      const int c = rem / (2 * NP), jj = (rem / NP) % 2, q = rem % NP;
      return l1off[2 * grp + jj] + (c * NP + q) * 256;
#endif
    }
    pos -= F1;
    return outoff[pos / (OT * NP)] + (pos % (OT * NP)) * 256;
  };
  u32x4 ring[R];
  const unsigned lane_w4 = (unsigned)lane * 4u, lane_b16 = (unsigned)lane * 16u;
  // a fragment load = one buffer_load_dwordx4: descriptor of the blob, the lane's 16 bytes as the vector offset, the fragment's byte
  // offset as the SCALAR offset -- no vector address arithmetic per load (the global_load form cost two VALU adds per fragment)
  const auto blob_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.blobs[p.c_begin + comp], 0, 0x7fffffff, 0x00020000);
  auto fetch_from = [&](int net_words, int pos) {                    // pos < F; net_words: the net block's word offset in the blob (uniform)
#ifdef GBNF_COOP_ABLATE_LOADS      // diagnostic: every fragment is the blob's first one, loaded once (timing only)
    if (net_words != SMALL_WORDS) return;
    ring[pos % R] = __builtin_amdgcn_raw_buffer_load_b128(blob_rsrc, lane_b16, SMALL_WORDS * 4, 0);
#else
    ring[pos % R] = __builtin_amdgcn_raw_buffer_load_b128(blob_rsrc, lane_b16, (net_words + frag_off(pos)) * 4, 0);
#endif
  };

  // ---- biases of one step (all nets) -> BIAS[buf]; 16 bytes per thread and trip
  auto bias_load = [&](int step, u32x4 (&regs)[NNETS][(BIAS_WORDS + 1023) / 1024]) {
#pragma unroll
    for (int net = 0; net < NNETS; ++net) {
      const gwords src = blob + (size_t)step * STEP_WORDS + SMALL_WORDS + (size_t)net * NET_WORDS;
#pragma unroll
      for (int k = 0; k < (BIAS_WORDS + 1023) / 1024; ++k) {
        const int w = (int)threadIdx.x * 4 + k * 1024;
        regs[net][k] = *reinterpret_cast<gu32x4>(reinterpret_cast<gbytes>(src) + 4u * (unsigned)(w < BIAS_WORDS ? w : 0));
      }
    }
  };
  auto bias_store = [&](int buf, const u32x4 (&regs)[NNETS][(BIAS_WORDS + 1023) / 1024]) {
#pragma unroll
    for (int net = 0; net < NNETS; ++net)
#pragma unroll
      for (int k = 0; k < (BIAS_WORDS + 1023) / 1024; ++k) {
        const int w = (int)threadIdx.x * 4 + k * 1024;
        if (w < BIAS_WORDS) *reinterpret_cast<u32x4*>(BIAS + (buf * NNETS + net) * BIAS_WORDS + w) = regs[net][k];
      }
  };

  Stamps st;                       // (diagnostic builds, -DGBNF_STAMPS: tools/coop_stamps.py) 0 prologue | 1 net input + layer 0 | 2 barrier 1 | 3 hidden
  st.start();                      // layer | 4 output layer + partials | 5 barrier 2 | 6 epilogue + barrier 3 | 7 tail
  // ---- prologue: the first fragments in flight, tables + biases of step 0 + the x tile -> LDS
#pragma unroll
  for (int pos = 0; pos < R; ++pos) fetch_from(SMALL_WORDS, pos);     // (step 0, net 0)
  {
    u32x4 b0[NNETS][(BIAS_WORDS + 1023) / 1024];
    bias_load(0, b0);
    if (lds_tables) {
      for (int s = 0; s < p.n_steps; ++s) {
        const gwords src = blob + (size_t)s * STEP_WORDS;
        for (int w = (int)threadIdx.x * 4; w < SMALL_WORDS; w += 64 * WAVES * 4)
          *reinterpret_cast<u32x4*>(SM + s * SMALL_WORDS + w) = *reinterpret_cast<gu32x4>(src + w);
      }
    }
    if (lane < d) {
      constexpr int RPW = (16 * NT + WAVES - 1) / WAVES;               // rows per wave
      float xv[RPW];
      const int64_t last = p.n - 1;
#pragma unroll
      for (int k = 0; k < RPW; ++k) {
        const int64_t n = row0 + wave + WAVES * k;
        xv[k] = xin[(n < p.n ? n : last) * d + lane];
      }
#pragma unroll
      for (int k = 0; k < RPW; ++k) {
        const int r = wave + WAVES * k;
        if (r < 16 * NT) Z[lane * ZS + r] = (row0 + r < p.n) ? xv[k] : 0.0f;
      }
    }
    bias_store(0, b0);
    if (threadIdx.x < NT) SATW[threadIdx.x] = 0u;
  }
  __syncthreads();
  st.mark(0);

  // log-det partials per OUTPUT TILE and sample tile (kept by the wave that owns the unit's epilogue; summed over o in a fixed
  // order at the end: a sample's result does not depend on the tile size or on which wave served it)
  float ld[OT][NT], ld2[OT][NT];
  bool sat[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    sat[nt] = false;
#pragma unroll
    for (int o = 0; o < OT; ++o) { ld[o][nt] = 0.0f; ld2[o][nt] = 0.0f; }
  }
  float ld_const = 0.0f;

  using lf32 = __attribute__((address_space(3))) float;
  const unsigned zb_ = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)Z + 4u * (unsigned)i;
  auto zoff = [&](int slot) { return zb_ + (unsigned)__mul24(slot, 4 * ZS); };       // &Z[slot * ZS + i]
  auto zld = [&](unsigned a, int nt) { return *reinterpret_cast<lf32*>((uintptr_t)(a + 64u * (unsigned)nt)); };
  auto zst = [&](unsigned a, int nt, float v_) { *reinterpret_cast<lf32*>((uintptr_t)(a + 64u * (unsigned)nt)) = v_; };

  // the per-step lane tables are constants of the blob: a step's input table is loaded while the step before it is in its epilogue,
  // its output table at the top of the step -- neither sits behind a barrier on the step's critical path
  LaneTable tin;
  float ldc_next;
  auto load_tin = [&](int step_) {
    if (lds_tables) {
      ldc_next = as_f32(SM[step_ * SMALL_WORDS + 1]);
      tin.load(SM + step_ * SMALL_WORDS + SMALL_HDR + g * NENT);
    } else {
      const gwords sp_ = blob + (size_t)step_ * STEP_WORDS;
      ldc_next = as_f32(sp_[1]);
      tin.load((const uint32_t*)sp_ + SMALL_HDR + g * NENT);
    }
  };
  load_tin(0);
  for (int step = 0; step < p.n_steps; ++step) {
    const gwords sp = blob + (size_t)step * STEP_WORDS;
    const uint32_t* tabs_l = SM + step * SMALL_WORDS;
    const bool more_steps = step + 1 < p.n_steps;
    // the next step's biases: requested now, stored into the other buffer behind this step's first barrier
    u32x4 bnext[NNETS][(BIAS_WORDS + 1023) / 1024];
    bias_load(more_steps ? step + 1 : step, bnext);

    // ---- the net input: normalised in registers by EVERY wave (the same values), split into layer 0's B operand; wave 0 writes
    //      the normalised values back behind the first barrier (the other waves read the raw state until then)
    u32x4 zp[NT][NP];
    float vin[NT][NENT];
    unsigned zin[NENT];
    LaneTable tout;
#if GBNF_COOP_HOIST_TABLES
    if (lds_tables) tout.load(tabs_l + SMALL_HDR + 160 + g * NENT);
    else tout.load((const uint32_t*)sp + SMALL_HDR + 160 + g * NENT);
#else
    if (step > 0) load_tin(step);
#endif
    {
      ld_const += ldc_next;
#pragma unroll
      for (int e = 0; e < NENT; ++e) {
        const bool live = tin.slot[e] >= 0;
        zin[e] = zoff(live ? tin.slot[e] : d);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float t = norm_fn<KIND>(zld(zin[e], nt), tin.p0[e], tin.p1[e], tin.p2[e], tin.p3[e]);
          vin[nt][e] = t;                          // (dead entries: whatever the spare slot holds, written back to the spare slot)
          sat[nt] = sat[nt] || (live && !(__builtin_fabsf(t) <= 65504.0f));
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool l0 = tin.slot[2 * q] >= 0, l1 = tin.slot[2 * q + 1] >= 0;
          const float a0 = l0 ? __builtin_amdgcn_fmed3f(vin[nt][2 * q], -65504.0f, 65504.0f) : 0.0f;
          const float a1 = l1 ? __builtin_amdgcn_fmed3f(vin[nt][2 * q + 1], -65504.0f, 65504.0f) : 0.0f;
          unsigned pc[NP];
          split_pair<NP>(a0, a1, pc);
#pragma unroll
          for (int k = 0; k < NP; ++k) zp[nt][k][q] = pc[k];
        }
    }

    f32x4 outS[NNETS][OT][NT];        // this wave's K-partials of the nets' outputs
#pragma unroll
    for (int net = 0; net < NNETS; ++net) {
      constexpr int ACT_A = ACTA, ACT_B = ACTB;
      const int ACTN = net == 0 ? ACT_A : ACT_B;
      const bool relu_rt = ACTN == 3 && __builtin_amdgcn_readfirstlane((int)sp[2 + net]) != 0;
      const int net_base = step * STEP_WORDS + SMALL_WORDS + net * NET_WORDS;
      // the (step, net) behind this one: where the ring's look-ahead reads once it runs past this net's last fragment
      const int net_next = (net + 1 < NNETS) ? net_base + NET_WORDS : (more_steps ? (step + 1) * STEP_WORDS + SMALL_WORDS : net_base);
      int pos = 0;                      // sequence position (a constant at every use after unrolling)
      auto take = [&]() -> u32x4 { return ring[pos % R]; };
      auto refill = [&]() {            // the slot of position `pos` is free: fetch position pos + R into it
        const int ahead = pos + R;
        if (ahead < F) fetch_from(net_base, ahead);
        else if (ahead >= FP && ahead - FP < F) fetch_from(net_next, ahead - FP);
        ++pos;
      };
      const uint32_t* bb = BIAS + ((step & 1) * NNETS + net) * BIAS_WORDS + g * 4;
      auto ldb = [&](int tile_) { return *reinterpret_cast<const f32x4*>(bb + tile_ * 16); };
      float amax[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) amax[nt] = 0.0f;
      auto act_split = [&](const f32x4& raw, int hp, int nt, unsigned (&pc)[NP]) {
#ifdef GBNF_COOP_ABLATE_ACT        // diagnostic: no activation, no split (results wrong, timing only)
        pc[0] = __builtin_bit_cast(unsigned, raw[2 * hp]); pc[1] = __builtin_bit_cast(unsigned, raw[2 * hp + 1]);
        (void)nt;
        return;
#endif
        float a0, a1;
        if (ACTN == GBNF_ACT_TANH) {
          f32x2 e = {__builtin_amdgcn_exp2f(raw[2 * hp]), __builtin_amdgcn_exp2f(raw[2 * hp + 1])};
          e = e + f32x2{1.0f, 1.0f};
          a0 = __builtin_amdgcn_rcpf(e[0]);
          a1 = __builtin_amdgcn_rcpf(e[1]);
        } else if (ACTN == GBNF_ACT_RELU) {
          a0 = __builtin_fmaxf(raw[2 * hp], 0.0f);
          a1 = __builtin_fmaxf(raw[2 * hp + 1], 0.0f);
          amax[nt] = __builtin_fmaxf(amax[nt], __builtin_fmaxf(a0, a1));
          a0 = __builtin_fminf(a0, 65504.0f);
          a1 = __builtin_fminf(a1, 65504.0f);
        } else {                       // per step: the packer folded the tanh pre-scale into this net's layers only if it IS a tanh net
          const float r0 = __builtin_fmaxf(raw[2 * hp], 0.0f), r1 = __builtin_fmaxf(raw[2 * hp + 1], 0.0f);
          amax[nt] = __builtin_fmaxf(amax[nt], relu_rt ? __builtin_fmaxf(r0, r1) : 0.0f);
          const float t0 = tanh_hx3(raw[2 * hp]), t1 = tanh_hx3(raw[2 * hp + 1]);
          a0 = relu_rt ? __builtin_fminf(r0, 65504.0f) : t0;
          a1 = relu_rt ? __builtin_fminf(r1, 65504.0f) : t1;
        }
        split_pair<NP>(a0, a1, pc);
      };
      // the three products of one weight tile (pieces w[0] = hi, w[1] = mid) with the B operands of the NT sample tiles
      auto mac = [&](const u32x4 (&w)[NP], const u32x4 (&x)[NT][NP], f32x4 (&acc)[NT]) {
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[nt] = mfma_narrow<0>(w[Products<2>::W[pr]], x[nt][Products<2>::X[pr]], acc[nt]);
            MFMA_ORDER_FENCE();
          }
      };

      // ---- layer 0: this wave's TPW tiles -> ACT
      u32x4 own[CPW][NT][NP];
      if constexpr (PRO) {
      // product-outer order: the three products of a weight tile go to the same accumulator, and a dependent v_mfma whose destination
      // the register allocator did not keep in place waits out the whole pipeline (+ s_nop); with the products of SEVERAL tiles
      // interleaved a chain's next link is G MFMAs away
      {
        f32x4 acc0[TPW][NT];
        u32x4 w0[TPW][NP];
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          const int t = t_first + j;
#pragma unroll
          for (int q = 0; q < NP; ++q) { w0[j][q] = take(); refill(); }
          const f32x4 bias = ldb(t < HT ? t : 0);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc0[j][nt] = bias;
        }
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              acc0[j][nt] = mfma_narrow<0>(w0[j][Products<2>::W[pr]], zp[nt][Products<2>::X[pr]], acc0[j][nt]);
              MFMA_ORDER_FENCE();
            }
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          const bool real = t_first + j < HT;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
              unsigned pc[NP];
              act_split(acc0[j][nt], hp, nt, pc);
#pragma unroll
              for (int k = 0; k < NP; ++k) own[j / 2][nt][k][2 * (j & 1) + hp] = real ? pc[k] : 0u;
            }
        }
      }
      } else {
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const int t = t_first + j;
        const bool real = t < HT;
        u32x4 w[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) { w[q] = take(); refill(); }
        const f32x4 bias = ldb(real ? t : 0);
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = bias;
        mac(w, zp, acc);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int hp = 0; hp < 2; ++hp) {
            unsigned pc[NP];
            act_split(acc[nt], hp, nt, pc);
#pragma unroll
            for (int k = 0; k < NP; ++k) own[j / 2][nt][k][2 * (j & 1) + hp] = real ? pc[k] : 0u;
          }
      }
      }
      uint32_t* actb = ACT + (NNETS > 1 ? (net & 1) * ACT_WORDS : 0);
#pragma unroll
      for (int cl = 0; cl < CPW; ++cl)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int k = 0; k < NP; ++k)
            *reinterpret_cast<u32x4*>(actb + (((CPW * wave + cl) * NT + nt) * NP + k) * 256 + lane_w4) = own[cl][nt][k];
      st.mark(1);
      GBNF_COOP_SYNC();                                   // B1: every chunk of layer 0's activations is in ACT
      st.mark(2);
      if (net == 0) {
        if (wave == 0) {
#pragma unroll
          for (int e = 0; e < NENT; ++e)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) zst(zin[e], nt, vin[nt][e]);
        }
        bias_store((step + 1) & 1, bnext);
      }

#if GBNF_COOP_L1_JOINT
      // ---- hidden layer: all of this wave's tiles at once over the HC contraction chunks (a chunk's B operands are read once)
      u32x4 hO[CPW][NT][NP];
      {
        f32x4 acc[TPW][NT];
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          const int t = t_first + j;
          const f32x4 bias = ldb(HT + (t < HT ? t : 0));
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[j][nt] = bias;
        }
#pragma unroll
        for (int c = 0; c < HC; ++c) {
          u32x4 xB[NT][NP];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int k = 0; k < NP; ++k) xB[nt][k] = *reinterpret_cast<const u32x4*>(actb + ((c * NT + nt) * NP + k) * 256 + lane_w4);
      if constexpr (PRO) {
          u32x4 wj[TPW][NP];
#pragma unroll
          for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int q = 0; q < NP; ++q) { wj[j][q] = take(); refill(); }
#pragma unroll
          for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int j = 0; j < TPW; ++j)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                acc[j][nt] = mfma_narrow<0>(wj[j][Products<2>::W[pr]], xB[nt][Products<2>::X[pr]], acc[j][nt]);
                MFMA_ORDER_FENCE();
              }
      } else {
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            u32x4 w[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) { w[q] = take(); refill(); }
            mac(w, xB, acc[j]);
          }
      }
        }
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          const bool real = t_first + j < HT;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
              unsigned pc[NP];
              act_split(acc[j][nt], hp, nt, pc);
#pragma unroll
              for (int k = 0; k < NP; ++k) hO[j / 2][nt][k][2 * (j & 1) + hp] = real ? pc[k] : 0u;
            }
        }
      }

#else
      // ---- hidden layer: this wave's tiles, one chunk (two tiles) at a time, over all HC contraction chunks
      u32x4 hO[CPW][NT][NP];
#if GBNF_COOP_XB_ALL
      // one wave per SIMD (512 registers): every chunk's B operand is read ONCE, all reads in flight at the top of the layer
      constexpr bool XB_ALL = WAVES == 4;
#else
      constexpr bool XB_ALL = false;
#endif
      u32x4 xBall[XB_ALL ? HC : 1][NT][NP];
      if constexpr (XB_ALL) {
#pragma unroll
        for (int c = 0; c < HC; ++c)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int k = 0; k < NP; ++k) xBall[c][nt][k] = *reinterpret_cast<const u32x4*>(actb + ((c * NT + nt) * NP + k) * 256 + lane_w4);
        // (pinned HERE: left alone, LLVM sinks every read back down to the MFMA that consumes it)
#pragma unroll
        for (int c = 0; c < HC; ++c)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int k = 0; k < NP; ++k) asm volatile("" : "+v"(xBall[c][nt][k]));
      }
#pragma unroll
      for (int grp = 0; grp < CPW; ++grp) {
        f32x4 acc[2][NT];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int t = t_first + 2 * grp + jj;
          const f32x4 bias = ldb(HT + (t < HT ? t : 0));
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[jj][nt] = bias;
        }
#pragma unroll
        for (int c = 0; c < HC; ++c) {
          u32x4 xB[NT][NP];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int k = 0; k < NP; ++k) {
#ifdef GBNF_COOP_ABLATE_XB         // diagnostic: no LDS reads of layer 0's activations (timing only)
              xB[nt][k] = zp[nt][k];
              continue;
#endif
              if constexpr (XB_ALL) xB[nt][k] = xBall[c][nt][k];
              else xB[nt][k] = *reinterpret_cast<const u32x4*>(actb + ((c * NT + nt) * NP + k) * 256 + lane_w4);
            }
      if constexpr (PRO) {
          u32x4 wj[2][NP];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int q = 0; q < NP; ++q) { wj[jj][q] = take(); refill(); }
#pragma unroll
          for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                acc[jj][nt] = mfma_narrow<0>(wj[jj][Products<2>::W[pr]], xB[nt][Products<2>::X[pr]], acc[jj][nt]);
                MFMA_ORDER_FENCE();
              }
      } else {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            u32x4 w[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) { w[q] = take(); refill(); }
            mac(w, xB, acc[jj]);
          }
      }
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const bool real = t_first + 2 * grp + jj < HT;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
              unsigned pc[NP];
              act_split(acc[jj][nt], hp, nt, pc);
#pragma unroll
              for (int k = 0; k < NP; ++k) hO[grp][nt][k][2 * jj + hp] = real ? pc[k] : 0u;
            }
        }
      }

#endif
      st.mark(3);
      // ---- output layer, split along K: the chunks this wave produced itself
      f32x4 (&outp)[OT][NT] = outS[net];
#pragma unroll
      for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) outp[o][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int cl = 0; cl < CPW; ++cl) {
      if constexpr (PRO) {
        u32x4 wo[OT][NP];
#pragma unroll
        for (int o = 0; o < OT; ++o)
#pragma unroll
          for (int q = 0; q < NP; ++q) { wo[o][q] = take(); refill(); }
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int o = 0; o < OT; ++o)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {           // (a phantom chunk's B operand is zero: its products add nothing)
              outp[o][nt] = mfma_narrow<0>(wo[o][Products<2>::W[pr]], hO[cl][nt][Products<2>::X[pr]], outp[o][nt]);
              MFMA_ORDER_FENCE();
            }
      } else {
#pragma unroll
        for (int o = 0; o < OT; ++o) {
          u32x4 w[NP];
#pragma unroll
          for (int q = 0; q < NP; ++q) { w[q] = take(); refill(); }
          mac(w, hO[cl], outp[o]);            // (a phantom chunk's B operand is zero: its products add nothing)
        }
      }
      }
      // the padding positions of the sequence: their look-ahead fetches (the next net's first fragments)
#pragma unroll
      for (int k = F; k < FP; ++k) refill();
      if (ACTN != GBNF_ACT_TANH) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) sat[nt] = sat[nt] || !(amax[nt] <= 65504.0f);
      }
#pragma unroll
      for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          *reinterpret_cast<f32x4*>(RED + (((net * WAVES + wave) * OT + o) * NT + nt) * 256 + lane_w4) = outp[o][nt];
    }
    st.mark(4);
    GBNF_COOP_SYNC();                                     // B2: every wave's partial sums are in RED
    st.mark(5);

    // ---- coupling transform of the other half + log-det partials: output tile (o, nt) on wave (o NT + nt) % WV
#if GBNF_COOP_HOIST_TABLES
    if (more_steps) load_tin(step + 1);
#else
    if (lds_tables) tout.load(tabs_l + SMALL_HDR + 160 + g * NENT);
    else tout.load((const uint32_t*)sp + SMALL_HDR + 160 + g * NENT);
#endif
    {
      const uint32_t* bbo = BIAS + (step & 1) * NNETS * BIAS_WORDS + (2 * HT) * 16 + g * 4;
#pragma unroll
      for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if ((o * NT + nt) % WAVES != wave) continue;
          f32x4 oA = *reinterpret_cast<const f32x4*>(bbo + o * 16), oB = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          if (NNETS > 1) oB = *reinterpret_cast<const f32x4*>(bbo + BIAS_WORDS + o * 16);
#pragma unroll
          for (int w = 0; w < WAVES; ++w) {
            oA += *reinterpret_cast<const f32x4*>(RED + (((0 * WAVES + w) * OT + o) * NT + nt) * 256 + lane_w4);
            if (NNETS > 1) oB += *reinterpret_cast<const f32x4*>(RED + (((1 * WAVES + w) * OT + o) * NT + nt) * 256 + lane_w4);
          }
          if (KIND == GBNF_KIND_GLOW && !p.additive) {
            // affine coupling, models/glow.py:331-338: scale = sigmoid(raw + 2) = 1 / s1, s1 = 1 + exp(-(raw + 2)); log scale = -ln 2 log2(s1)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
              const int e = 2 * o + pp;
              if (e >= NENT) continue;
              const bool live = tout.slot[e] >= 0;
              const unsigned za = zoff(live ? tout.slot[e] : d);
              const float shift = oA[2 * pp], raw = oA[2 * pp + 1];
              const float s1 = 1.0f + __builtin_amdgcn_exp2f((raw + 2.0f) * -1.4426950408889634f);
              const float l2 = __builtin_amdgcn_logf(s1);
              float t = norm_fn<KIND>(zld(za, nt), tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
              t = (t + shift) * __builtin_amdgcn_rcpf(s1);
              ld2[o][nt] += live ? l2 : 0.0f;
              zst(za, nt, t);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int e = 4 * o + r;
              if (e >= NENT) continue;
              const bool live = tout.slot[e] >= 0;
              const unsigned za = zoff(live ? tout.slot[e] : d);
              float t = norm_fn<KIND>(zld(za, nt), tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
              if constexpr (KIND == GBNF_KIND_GLOW) {
                t = t + oA[r];                                          // additive, models/glow.py:328-329
              } else {
                const float shift = oA[r], scale = oB[r];
                t = shift + t * exp_fast(scale);                        // models/transformations.py:575
                ld[o][nt] += live ? scale : 0.0f;
              }
              zst(za, nt, t);
            }
          }
        }
    }
    GBNF_COOP_SYNC();                                     // B3: the step's state is in Z
    st.mark(6);
  }

  // ---- log-det partials and range marks of the waves meet; then base density + outputs
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int o = 0; o < OT; ++o)
      if ((o * NT + nt) % WAVES == wave) LDP[(o * NT + nt) * 64 + lane] = __builtin_fmaf(-0.69314718055994531f, ld2[o][nt], ld[o][nt]);
    const unsigned long long m = __ballot(sat[nt]);
    const unsigned rows = (unsigned)((m | (m >> 16) | (m >> 32) | (m >> 48)) & 0xffffull);
    if (rows != 0u && lane == 0) atomicOr(SATW + nt, rows);
  }
  __syncthreads();
  const gwords tail = blob + (size_t)p.n_steps * STEP_WORDS;
  unsigned bad_mask = 0u;                 // bit r: row r of the tile left the fp16 range
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bad_mask |= SATW[nt] << (16 * nt);
  const bool any_sat = bad_mask != 0u;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if (nt % WAVES != wave) continue;
    float quad = 0.0f;
    if (p.base_mean == nullptr) {
      for (int j = g; j < d; j += 4) {
        const float v = Z[j * ZS + 16 * nt + i];
        quad = __builtin_fmaf(-0.5f * v, v, quad);          // (explicit: the same rounding in every instantiation)
      }
    } else {
      for (int j = g; j < d; j += 4) {
        const int slot = (int)tail[j];
        const float mu = p.base_mean[j], sd = p.base_std[j];
        const float inv_sd = 1.0f / sd, lsd = logf(sd);
        const float v = (Z[slot * ZS + 16 * nt + i] - mu) * inv_sd;
        quad = __builtin_fmaf(-0.5f * v, v, quad) - lsd;
      }
    }
    float l = 0.0f;
#pragma unroll
    for (int o = 0; o < OT; ++o) l += LDP[(o * NT + nt) * 64 + lane];
    quad += __shfl_xor(quad, 16); quad += __shfl_xor(quad, 32);
    l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
    const bool bad = (bad_mask >> (16 * nt + i)) & 1u;
    const int64_t n = row0 + 16 * nt + i;
    if (g == 0 && n < p.n) {
      const float ldj = l + ld_const;
      const float nanv = __builtin_nanf("");
      if (p.ldj_out) p.ldj_out[out_base + n] = bad ? nanv : ldj;
      if (p.ll_out) p.ll_out[out_base + n] = bad ? nanv : __builtin_fmaf(-0.91893853320467274f, (float)d, quad) + ldj;
    }
  }
  if (p.sat != nullptr && any_sat && threadIdx.x == 0) {
    atomicAdd(p.sat, 1ull);
    atomicMax(p.sat + SAT_MARKS + p.seq % SAT_SLOTS, p.seq);      // tells the repair launch behind this one that it has work
  }
  if (p.z_out != nullptr && lane < d) {
    const int slot = (int)tail[lane];
    float* zo = p.z_out + (int64_t)comp * p.n * d;
    for (int r = wave; r < 16 * NT; r += WAVES) {
      const int64_t n = row0 + r;
      float v = Z[slot * ZS + r];
      if ((bad_mask >> r) & 1u) v = __builtin_nanf("");
      if (n < p.n) zo[n * d + lane] = v;
    }
  }
#ifdef GBNF_STAMPS
  st.mark(7);
  if (p.dbg != nullptr && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) p.dbg[((size_t)blockIdx.x * WAVES + wave) * 8 + k] = st.acc[k];
  }
#endif
}

template <int KIND, int HT, int OT, int NT, int ACTA, int ACTB, int WV>
static hipError_t coop_launch(FlowLaunch p, hipStream_t s) {
  constexpr int NNETS = (KIND == GBNF_KIND_REALNVP) ? 2 : 1;
  if (p.inverse || p.repair) return hipErrorInvalidValue;
  p.n_tiles = (int32_t)((p.n + 16 * NT - 1) / (16 * NT));
  p.lds_tables = p.n_steps <= LDS_TABLE_STEPS && flow_coop_lds_bytes(p.n_steps, NT, HT, OT, NNETS, p.d, true, WV) <= 160 * 1024;
  const size_t lds = flow_coop_lds_bytes(p.n_steps, NT, HT, OT, NNETS, p.d, p.lds_tables != 0, WV);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const long long grid = (long long)p.n_tiles * p.n_comp * p.n_batches;
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
  p.n_items = (int32_t)grid;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)flow_kernel_coop<KIND, HT, OT, NT, ACTA, ACTB, WV>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((flow_kernel_coop<KIND, HT, OT, NT, ACTA, ACTB, WV>), dim3((unsigned)grid), dim3(64 * WV), lds, s, p);
  return hipGetLastError();
}

// registry key: VariantKey{kind, ht, /*ksl*/ -13 (cooperative f16x3), 0, ot, FORM, /*depth*/ 1, act_a, act_b};
// FORM 1 = 16-sample tiles on 4 waves, 2 = 32-sample tiles on 4 waves, 3 = 32-sample tiles on 8 waves, 4 = 16-sample tiles on 8 waves
constexpr int coop_form_nt(int form) { return (form == 1 || form == 4) ? 1 : 2; }
constexpr int coop_form_waves(int form) { return form >= 3 ? 8 : 4; }
#define GBNF_INSTANTIATE_COOP(KIND, HT, OT, FORM, ACTA, ACTB)                                                   \
  namespace gbnf {                                                                                              \
  static hipError_t launch_coop_##KIND##_##HT##_##OT##_##FORM##_##ACTA##_##ACTB(const FlowLaunch& p0, unsigned, \
                                                                                hipStream_t s) {                \
    return coop_launch<KIND, HT, OT, coop_form_nt(FORM), ACTA, ACTB, coop_form_waves(FORM)>(p0, s);             \
  }                                                                                                             \
  static const int reg_coop_##KIND##_##HT##_##OT##_##FORM##_##ACTA##_##ACTB =                                   \
      (register_variant(VariantKey{KIND, HT, -13, 0, OT, FORM, 1, ACTA, ACTB},                                  \
                        launch_coop_##KIND##_##HT##_##OT##_##FORM##_##ACTA##_##ACTB,                            \
                        "flow_kernel_coop<" #KIND "," #HT "," #OT ",form " #FORM "," #ACTA "," #ACTB ">"),      \
       0);                                                                                                      \
  }

}  // namespace gbnf
