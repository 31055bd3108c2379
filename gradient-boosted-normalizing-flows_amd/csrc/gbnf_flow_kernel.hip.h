// gbnf_flow_kernel.hip.h -- the fused per-component flow kernel for gfx950 (MI355X, CDNA4).
//
// One 64-lane wavefront (= one workgroup) owns a tile of 16*NT samples of ONE boosted
// component and carries it through all K flow steps without ever leaving the CU:
//
//   * the sample tile's d features live in a wave-private LDS array Z[slot][sample]
//     (<= 8.4 KB); permutations / half swaps are pure slot renames folded into index
//     tables at pack time, so no data moves for Permute1d or for RealNVP's flip;
//   * the coupling network runs "transposed" on the matrix cores with exact-f32 MFMA
//     (v_mfma_f32_16x16x4_f32):  H^T(units x samples) = W(units x k) . act^T(k x samples).
//     With that orientation the accumulator tile of layer l IS the B operand of layer
//     l+1 (lane (i,g) holds units 16t+4g+r, r=0..3, of sample i -- exactly what k-step r
//     of chunk t wants), so hidden activations never touch LDS or HBM: the whole
//     Linear->tanh->Linear->tanh->Linear chain is register-resident;
//   * weights are the A operand, pre-tiled at pack time into the exact lane order so
//     every wave-load is one contiguous 1 KiB global_load_dwordx4 streamed from L2
//     (one component per XCD => ~1.2 MB of weights stays in that XCD's 4 MiB L2);
//   * with ~330 registers a wave runs alone on its SIMD, so the instruction stream itself
//     interleaves MFMA, tanh (VALU) and weight prefetch (software pipelining by hand,
//     fenced with sched_barrier so the compiler cannot sink the prefetches);
//   * log|det J| partial sums stay in lanes and are folded across the 4 lane groups
//     with two DPP/shuffle steps at the very end, together with sum_j z_j^2.
//
// Reference semantics implemented (file:line of the reference):
//   FlowStep.encode            models/glow.py:317-342     (actnorm -> permute -> affine/additive coupling)
//   _ActNorm.forward           models/layers.py:488-533   ((x + bias) * exp(logs); ld += sum(logs))
//   Permute1d.forward          models/layers.py:661-668   (x[:, indices])
//   TanhNet / ReLUNet          models/layers.py:208-243
//   RealNVP.forward            models/transformations.py:560-579 (flipped => halves swap)
//   BatchNorm.forward (eval)   models/layers.py:337-358
//   log_normal_standard        utils/distributions.py:44-60
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gbnf.h"

namespace gbnf {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;

constexpr int KS1MAX = 8;    // first-layer k-steps of 4  => coupling-net input width <= 32
constexpr int ZSLOTS = 64;   // features per sample <= 64
constexpr int NENT = 8;      // per-lane table entries (in: k-steps, out: coupled features)
constexpr int SMALL_HDR = 16;
constexpr int SMALL_WORDS = SMALL_HDR + 10 * 4 * NENT;  // header + {slot,p0..p3} x {in,out}
constexpr int LDS_TABLE_STEPS = 24;  // per-step tables are staged in LDS when K <= this and they fit (round 6: 12 -> 24, the chained backward sweep needs them there)

// Packed-parameter layout of one coupling network, in 32-bit words (host packer and kernel
// share it).  HT = hidden tiles of 16 units, KS1 = first-layer k-steps, OT = output tiles of
// 16 rows, LMID = number of hidden->hidden layers (coupling_network_depth).
// "+1"/"+4" rows are zero pads that the software prefetch may touch past the end.
struct NetLayoutRT {
  int KQ, W1, W1_TILE, B1, MID0, MID_W, MID_STRIDE, W3, B3, NET_WORDS;
  constexpr NetLayoutRT(int HT, int KS1, int OT, int LMID)
      : KQ((KS1 + 3) / 4),
        W1(0),
        W1_TILE(64 * ((KS1 + 3) / 4) * 4),                       // [t][lane][KQ*4] f32
        B1(W1 + (HT + 4) * W1_TILE),                              // [HT+4][4 g][4 r]
        MID0(B1 + (HT + 4) * 16),                                 // LMID x { W [HT+1][HT][64][4], B [HT+1][4][4] }
        MID_W((HT + 1) * HT * 256),
        MID_STRIDE(MID_W + (HT + 1) * 16),
        W3(MID0 + LMID * MID_STRIDE),                             // [HT+1][OT][64][4]
        B3(W3 + (HT + 1) * OT * 256),                             // [OT][4][4]
        NET_WORDS(B3 + OT * 16) {}
};

// The LMID template / variant-key field: 0..2 = coupling_network_depth of a TanhNet / ReLUNet; 10 + B = a ResidualNet of
// B blocks (models/layers.py:246-301: Linear, B x [x + Linear(relu(Linear(relu(x))))], Linear -- the same Linear shapes
// as a plain net of depth 2B, other activation placement and a skip connection).
constexpr bool lmid_is_residual(int lmid) { return lmid >= 10; }
constexpr int lmid_mid_layers(int lmid) { return lmid >= 10 ? 2 * (lmid - 10) : lmid; }

constexpr int MAX_BATCHES = 32;  // batches one launch can serve

struct FlowLaunch {
  const uint32_t* const* blobs;  // device array: packed parameter blob per component
  const float* xs[MAX_BATCHES];  // n_batches inputs of (n, d) each; log-densities of batch b go to columns [b*n, (b+1)*n)
  float* z_out;                  // (n_comp, n, d) or null
  float* ldj_out;                // (n_comp, n)    or null
  float* ll_out;                 // (n_comp, n)    or null
  const float* base_mean;        // (d,) or null  -> N(0,1)
  const float* base_std;         // (d,) or null
  int64_t n;
  int64_t out_stride;              // floats between consecutive components' rows of ldj_out / ll_out (>= n)
  int32_t d;
  int32_t n_steps;
  int32_t c_begin;
  int32_t n_comp;
  int32_t n_tiles;               // ceil(n / (16*NT))
  int32_t additive;              // glow: additive coupling
  int32_t n_batches;             // 1..MAX_BATCHES (z_out / ldj_out only with 1)
  int32_t inverse;               // f32 kernel only: run the flow backwards (xs = z in, z_out = x out, ldj_out = log|det dx/dz|)
  unsigned long long* dbg;       // diagnostic builds (-DGBNF_STAMPS) only: per-block phase cycle sums
  unsigned long long* sat;       // split-f16 kernel: [0] counter of waves that stored an operand beyond the fp16 range, marks behind it
  int32_t ring;                  // hx3 kernels: stage slots of the LDS weight ring (2..HX3_MAX_RING), set by the launcher
  int32_t repair;                // hx3 kernels (bf16x6 instantiations): 1 = the pass behind an f16x3 launch: only workgroups owning a
                                 //   sample whose outputs are NaN run (samples the f16x3 launch marked as out of range) -- or every
                                 //   workgroup once *guard != 0; 2 = conditional full launch: returns at once unless *guard != 0
  int32_t lds_tables;            // hx3 kernels: per-step tables staged in LDS (else read from the blob), set by the launcher
  int32_t stagger;               // hx3 kernels, 4-wave workgroups in pairs per CU: sleeps of 2048 cycles for the one in the odd wave slots
  unsigned long long seq;        // hx3 kernels: serial number of an f16x3 launch and its repair launch (process-wide, never 0, never
                                 //   reused): the f16x3 launch raises mark slot seq % SAT_SLOTS to it (atomicMax) when it marks a sample;
                                 //   the repair launch returns at once when the slot is below its seq (nobody from its launch on has
                                 //   marked), runs when it is equal, and also runs when it is above (a later launch that shares the
                                 //   slot has marked and may have hidden this launch's mark)
  const unsigned* guard;         // hx3 kernels: the handle's numerics-guard word (device) or null: non-zero = the f16x3 arithmetic failed
                                 //   the check on the caller's data, every repair launch re-evaluates its whole work list in bf16x6
  int32_t n_items;               // hx3 kernels: work items (component, batch, tile group) of the launch; a repair launch walks them
                                 //   with a grid of at most a few workgroups per CU
  // TRAIN instantiations of the hx3 kernel (the training path's forward sweep, csrc/gbnf_train.hip): besides z / ldj the
  // kernel saves what the backward pass needs, so that nothing is recomputed there
  float* trace_out;              //   [n_steps][d slots][np]: every step's normalised state (slot layout)
  float* acts_out;               //   the trainer's operand workspace: per (step, net) a region of `net_rows` rows of np floats,
                                 //   tiled [16-sample tile][row][16]: net input (ip rows) | hidden activations (2 x hp) | the
                                 //   backward's gradient-side rows (2 x hp + op) | net output (op rows, at row offset ip + 4 hp + op)
  int64_t np;                    //   samples padded to whole 32-sample tiles
  int32_t tr_ip, tr_hp, tr_op;   //   padded net-input / hidden / net-output rows (multiples of 16)
  int32_t net_rows;              //   ip + 4 hp + 2 op
  // the backward sweep of the training path (bwd_kernel_hx3, gbnf_train_bwd.hip.h); acts_out is read (saved activations and
  // net outputs) and written (gradient-side operands), np / tr_* as above
  const uint32_t* const* blobs_bwd;   // device array [1]: the packed TRANSPOSED weights (BwdLayout), steps 0 .. K-1
  const int32_t* bwd_tab;        //   [n_steps][2 (in | out)][4 g][NENT]: index of the entry's feature in the step's parameter vectors, -1 = none
  const int64_t* bwd_goff;       //   [n_steps][2]: float offsets of the step's two normalisation-parameter gradients in `grads`
  const float* trace_in;         //   [n_steps][d][np]
  const float* g_z;              //   (n, d) upstream gradient or null
  const float* g_ldj;            //   (n,) or null
  float* g_x;                    //   (n, d) or null
  float* grads;                  //   flat parameter-gradient buffer (ActNorm / BatchNorm entries are accumulated here)
  const unsigned* gmax;          //   bits of the largest |upstream entry| (gradient scaling)
  float* partials;               //   [workgroups][n_steps][2][64]: every workgroup's sums of the normalisation-parameter gradients
                                 //   (bwd_param_reduce_kernel adds them up in a fixed order: no atomics, bit-reproducible)
  // step RANGES of the two training sweeps (round 4: BatchNorm on batch statistics -- a step's statistics need the whole batch, so
  // the sweep is cut in front of every BatchNorm step and the state is parked in HBM in slot layout [d][np] between the launches)
  int32_t k_begin, k_end;        //   steps [k_begin, k_end) of this launch; k_end == 0: all n_steps
  const float* state_in;         //   forward: state at the input of step k_begin (null: the rows of xs); backward: the SCALED gradient
                                 //   state behind step k_end - 1 (null: g_z / g_ldj rows)
  float* state_out;              //   forward: state behind step k_end - 1 (null: the flow ends here: z_out / ldj_out); backward: gradient
                                 //   state in front of step k_begin (null: g_x rows)
  int32_t ldj_accumulate;        //   forward: ldj_out += this range's log-det instead of =
};
// Per-device saturation words (64-bit): [0] counter of waves that marked a sample, [1] unused, [2 ..] launch marks
constexpr int SAT_MARKS = 2;
constexpr int SAT_SLOTS = 1024;

__device__ __forceinline__ float as_f32(uint32_t u) { return __builtin_bit_cast(float, u); }

// In-kernel phase stamps (cdna_hip_programming.md section 7).  Diagnostic builds only: the shipped
// kernel is compiled without GBNF_STAMPS and then no stamp executes.
struct Stamps {
#ifdef GBNF_STAMPS
  unsigned long long last = 0;
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int cur = 0;                   // bucket of the phase in progress (split kernels: stage_end books its wait separately)
#endif
  __device__ __forceinline__ void set(int k) {
#ifdef GBNF_STAMPS
    cur = k;
#else
    (void)k;
#endif
  }
  __device__ __forceinline__ void start() {
#ifdef GBNF_STAMPS
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last)::"memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
  __device__ __forceinline__ void mark(int k) {
#ifdef GBNF_STAMPS
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    acc[k] += t - last;
    last = t;
#else
    (void)k;
#endif
  }
};

// tanh with absolute error ~1e-7: 1 - 2/(e^{2x}+1) on v_exp_f32 / v_rcp_f32 (1 ulp each); saturates
// correctly at +-inf.  Hidden activations feed f32 dot products of O(1) terms, so absolute (not
// relative) accuracy is what the log-likelihood sees; parity tests pin the end-to-end 1e-5 bar.
__device__ __forceinline__ float tanh_act(float x) {
#ifdef GBNF_TANH_LIBM
  return tanhf(x);
#else
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // e^{2x}
  const float r = __builtin_amdgcn_rcpf(e + 1.0f);
  return __builtin_fmaf(-2.0f, r, 1.0f);
#endif
}

// ACT: GBNF_ACT_TANH, GBNF_ACT_RELU, or 3 (GBNF_ACT_PER_STEP, gbnf_internal.h) = chosen per step and net (`relu`, uniform): both are computed and one is
// selected -- straight-line code, the interleaving with the MFMAs stays what it is
template <int ACT>
__device__ __forceinline__ float act_fn(float v, bool relu) {
  if constexpr (ACT == GBNF_ACT_TANH) {
    return tanh_act(v);
  } else if constexpr (ACT == GBNF_ACT_RELU) {
    return __builtin_fmaxf(v, 0.0f);
  } else {
    const float t = tanh_act(v), r = __builtin_fmaxf(v, 0.0f);
    return relu ? r : t;
  }
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Scheduling fence that only MFMAs may not cross (VALU / SALU / transcendental / VMEM / DS may):
// keeps the hand-written MFMA order -- accumulator chains alternating, so a dependent
// v_mfma_f32_16x16x4_f32 (40-cycle latency, 32-cycle issue) is never issued back to back; left alone,
// the scheduler groups each chain's MFMAs together and every one of them stalls on its predecessor.
#define MFMA_ORDER_FENCE() __builtin_amdgcn_sched_barrier(0x7F6)

// ---------------------------------------------------------------------------------
// Register-resident coupling network for one wave.
//   zb  : first-layer B operands, zb[s][nt] = normalised input feature 4s+g of sample (nt, i)
//   out : OT x NT accumulator tiles of the last Linear (bias included)
// KSL = live k-steps (1..4) in the LAST hidden tile (hidden width padded to a k-step multiple;
// the padded units sit in whole trailing k-steps so they are skipped, not multiplied by zero).
// ---------------------------------------------------------------------------------
template <int HT, int KSL, int KS1, int OT, int NT, int LMID, int ACT>
__device__ __forceinline__ void coupling_net(const uint32_t* __restrict__ net, const float (&zb)[KS1][NT],
                                             int lane, int g, f32x4 (&out)[OT][NT], Stamps& st, bool relu) {
  constexpr NetLayoutRT L(HT, KS1, OT, lmid_mid_layers(LMID));
  constexpr bool RES = lmid_is_residual(LMID);
  constexpr int MIDS = lmid_mid_layers(LMID);
  constexpr int KQ = L.KQ;
  const f32x4* w1 = reinterpret_cast<const f32x4*>(net + L.W1) + lane * KQ;   // + t*64*KQ
  const f32x4* b1 = reinterpret_cast<const f32x4*>(net + L.B1) + g;           // + t*4
  const f32x4* w3 = reinterpret_cast<const f32x4*>(net + L.W3) + lane;
  const f32x4* b3 = reinterpret_cast<const f32x4*>(net + L.B3) + g;

  f32x4 hA[HT][NT];

  // one tile of layer 0 (in -> hidden): KS1 k-steps, NT independent accumulator chains
  auto layer0_tile = [&](const f32x4 (&w)[KQ], f32x4 bias, f32x4 (&h)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) h[nt] = bias;
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        h[nt] = mfma4(w[s >> 2][s & 3], zb[s][nt], h[nt]);
        MFMA_ORDER_FENCE();
      }
    }
  };
  auto load_w1 = [&](int t, f32x4 (&w)[KQ]) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) w[q] = w1[t * 64 * KQ + q];
  };

  if constexpr (LMID != 1) {
    // ---- generic (not hand-pipelined) forms for depth 0 and depth >= 2
    {
      f32x4 wq[2][KQ];
      f32x4 bq[2];
      load_w1(0, wq[0]);
      bq[0] = b1[0];
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        load_w1(t + 1, wq[(t + 1) & 1]);
        bq[(t + 1) & 1] = b1[(t + 1) * 4];
        layer0_tile(wq[t & 1], bq[t & 1], hA[t]);
        if constexpr (!RES) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) hA[t][nt][r] = act_fn<ACT>(hA[t][nt][r], relu);
        }
      }
    }
    st.mark(1);
    if constexpr (RES) {
      // ---- residual blocks: hA is the running (un-activated) state; block = hA += W_b . relu(W_a . relu(hA) + b_a) + b_b
#pragma unroll
      for (int blk = 0; blk < MIDS / 2; ++blk) {
        const f32x4* wa = reinterpret_cast<const f32x4*>(net + L.MID0 + (2 * blk) * L.MID_STRIDE) + lane;
        const f32x4* ba = reinterpret_cast<const f32x4*>(net + L.MID0 + (2 * blk) * L.MID_STRIDE + L.MID_W) + g;
        const f32x4* wb = reinterpret_cast<const f32x4*>(net + L.MID0 + (2 * blk + 1) * L.MID_STRIDE) + lane;
        const f32x4* bb = reinterpret_cast<const f32x4*>(net + L.MID0 + (2 * blk + 1) * L.MID_STRIDE + L.MID_W) + g;
        f32x4 hB[HT][NT];
#pragma unroll
        for (int u = 0; u < HT; ++u) {
          const f32x4 bias = ba[u * 4];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) hB[u][nt] = bias;
#pragma unroll
          for (int t = 0; t < HT; ++t) {
            const f32x4 a = wa[(u * HT + t) * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (t < HT - 1 || r < KSL) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) hB[u][nt] = mfma4(a[r], __builtin_fmaxf(hA[t][nt][r], 0.0f), hB[u][nt]);
              }
            }
          }
        }
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) hB[t][nt][r] = __builtin_fmaxf(hB[t][nt][r], 0.0f);
#pragma unroll
        for (int u = 0; u < HT; ++u) {
          f32x4 acc[NT];
          const f32x4 bias = bb[u * 4];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt] = bias;
#pragma unroll
          for (int t = 0; t < HT; ++t) {
            const f32x4 a = wb[(u * HT + t) * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (t < HT - 1 || r < KSL) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma4(a[r], hB[t][nt][r], acc[nt]);
              }
            }
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) hA[u][nt] += acc[nt];      // the skip connection (hA[u] is no input of this layer)
        }
      }
    }
#pragma unroll
    for (int m = 0; m < (RES ? 0 : MIDS); ++m) {
      const f32x4* w = reinterpret_cast<const f32x4*>(net + L.MID0 + m * L.MID_STRIDE) + lane;
      const f32x4* b = reinterpret_cast<const f32x4*>(net + L.MID0 + m * L.MID_STRIDE + L.MID_W) + g;
      f32x4 hB[HT][NT];
#pragma unroll
      for (int u = 0; u < HT; ++u) {
        f32x4 bb = b[u * 4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) hB[u][nt] = bb;
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          f32x4 a = w[(u * HT + t) * 64];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (t < HT - 1 || r < KSL) {
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) hB[u][nt] = mfma4(a[r], hA[t][nt][r], hB[u][nt]);
            }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) hA[t][nt][r] = act_fn<ACT>(hB[t][nt][r], relu);
    }
    st.mark(2);
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      f32x4 b = b3[o * 4];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) out[o][nt] = b;
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) {
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        f32x4 a = w3[(t * OT + o) * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (t < HT - 1 || r < KSL) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) out[o][nt] = mfma4(a[r], hA[t][nt][r], out[o][nt]);
          }
        }
      }
    }
    st.mark(4);
  } else {
    // ---- depth 1 (the reference default): Linear -> act -> Linear -> act -> Linear as ONE
    //      hand-pipelined stream.  The wave runs alone on its SIMD, so MFMA, VALU (tanh) and the
    //      weight prefetch are interleaved in program order:
    //        pass u = 0 : region t = { 4*NT*.. MFMAs of hidden-layer tile 0, k-chunk t }  +
    //                     { layer-0 tile t+1: KS1*NT MFMAs, then its tanh }  + prefetches
    //        pass u >= 1: region t = { MFMAs of hidden tile u, k-chunk t } + one tanh value of tile
    //                     u-1 per region; after the last tanh, tile u-1 is consumed as k-chunk u-1
    //                     of the output layer.  Tile u never exists outside 4*NT registers.
    //      Every region re-issues the weight tile it just consumed for the NEXT pass into the other
    //      register set (ping-pong => no copies, a full pass of prefetch distance); biases are
    //      fetched one pass ahead; sched_barrier fences keep the compiler from sinking the loads.
    const f32x4* w2 = reinterpret_cast<const f32x4*>(net + L.MID0) + lane;           // [u][t][64]
    const f32x4* b2 = reinterpret_cast<const f32x4*>(net + L.MID0 + L.MID_W) + g;    // [u][4]

    f32x4 A0[HT], A1[HT];
    f32x4 A3[OT];
    f32x4 wq[4][KQ];   // layer-0 weight tiles, ring of 4 (tile k lives in wq[k & 3])
    f32x4 bq[4];
    f32x4 bias_c, bias_n;

    // prologue loads: everything the first pass needs
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      load_w1(k, wq[k]);          // tiles >= HT are zero pads
      bq[k] = b1[k * 4];
    }
    bias_c = b2[0];
#pragma unroll
    for (int t = 0; t < HT; ++t) A0[t] = w2[t * 64];
#pragma unroll
    for (int o = 0; o < OT; ++o) A3[o] = w3[o * 64];
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      f32x4 b = b3[o * 4];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) out[o][nt] = b;
    }

    // tanh of tile u-1 is sprinkled over regions R0 .. R0+NTR-1 of pass u (not region 0: its inputs
    // are the previous pass's last MFMA results), then tile u-1 feeds the output layer in region TG
    constexpr int NV = 4 * NT;                          // raw sums per hidden tile per lane
    constexpr int R0 = (HT > 1) ? 1 : 0;
    constexpr int VPT = (NV + (HT - R0) - 1) / (HT - R0);   // tanh values per region
    constexpr int NTR = (NV + VPT - 1) / VPT;
    constexpr int TG = (R0 + NTR < HT) ? R0 + NTR : HT - 1;

    f32x4 pre[NT];   // raw sums of hidden tile u-1
    f32x4 hb[NT];    // act(pre)

    // k-chunk t of the current hidden tile; the NEXT pass's weight tiles are requested two per
    // region, i.e. all of them in the first half of the pass, so that nothing young is in flight
    // at the loop back-edge (the compiler drains vmcnt there)
    auto region = [&](int t, f32x4 (&Ac)[HT], f32x4 (&An)[HT], const f32x4* wn, f32x4 (&acc)[NT]) {
      if (2 * t < HT) An[2 * t] = wn[(2 * t) * 64];
      if (2 * t + 1 < HT) An[2 * t + 1] = wn[(2 * t + 1) * 64];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (t < HT - 1 || r < KSL) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[nt] = mfma4(Ac[t][r], hA[t][nt][r], acc[nt]);
            MFMA_ORDER_FENCE();
          }
        } else {
          // the skipped k-steps' weight registers must stay "used" until here: otherwise the
          // allocator recycles them while the prefetch that writes them is still in flight and
          // has to drain vmcnt(0) in the middle of a pass
          asm volatile("" ::"v"(Ac[t][r]));
        }
      }
    };
    auto out_chunk = [&](const f32x4* w3n, int nk) {
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        f32x4 a = A3[o];
        A3[o] = w3n[o * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (r < nk) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              out[o][nt] = mfma4(a[r], hb[nt][r], out[o][nt]);
              MFMA_ORDER_FENCE();
            }
          }
        }
      }
    };
    auto pass = [&](int u, f32x4 (&Ac)[HT], f32x4 (&An)[HT]) {
      f32x4 acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = bias_c;
      bias_n = b2[(u + 1) * 4];
      const f32x4* wn = w2 + (u + 1) * HT * 64;
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        region(t, Ac, An, wn, acc);
        if (t >= R0 && t - R0 < NTR) {
#pragma unroll
          for (int v = (t - R0) * VPT; v < (t - R0 + 1) * VPT && v < NV; ++v)
            hb[v >> 2][v & 3] = act_fn<ACT>(pre[v >> 2][v & 3], relu);
        }
        if (t == TG) out_chunk(w3 + u * OT * 64, 4);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) pre[nt] = acc[nt];
      bias_c = bias_n;
    };

    // layer-0 tiles 0 and 1 (tile 1 stays raw: it is activated during region 0)
    f32x4 hraw[2][NT];
    {
      f32x4 h0[NT];
      layer0_tile(wq[0], bq[0], h0);
      if (HT > 1) layer0_tile(wq[1], bq[1], hraw[1]);
      // ring slots 0 and 1 are free again: tiles 4 and 5 (region t refills slot (t+2)&3 with tile t+6)
      if (4 < HT) {
        load_w1(4, wq[0]);
        bq[0] = b1[4 * 4];
      }
      if (5 < HT) {
        load_w1(5, wq[1]);
        bq[1] = b1[5 * 4];
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) hA[0][nt][r] = act_fn<ACT>(h0[nt][r], relu);
    }
    __builtin_amdgcn_sched_barrier(0);
    st.mark(1);

    // pass u = 0 with layer 0 folded in, two tiles of lookahead:
    //   region t = { layer-0 MFMAs of tile t+2 } + { hidden-layer MFMAs, k-chunk t } + { tanh of tile t+1 }
    {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) pre[nt] = bias_c;
      bias_n = b2[4];
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        if (t + 2 < HT) {
          layer0_tile(wq[(t + 2) & 3], bq[(t + 2) & 3], hraw[t & 1]);
          if (t + 6 < HT) {
            load_w1(t + 6, wq[(t + 2) & 3]);
            bq[(t + 2) & 3] = b1[(t + 6) * 4];
          }
        }
        region(t, A0, A1, w2 + HT * 64, pre);
        if (t + 1 < HT) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) hA[t + 1][nt][r] = act_fn<ACT>(hraw[(t + 1) & 1][nt][r], relu);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      bias_c = bias_n;
    }
    st.mark(2);
    {
      int u = 1;
#pragma unroll 1
      for (; u + 1 < HT; u += 2) {
        pass(u, A1, A0);
        pass(u + 1, A0, A1);
      }
      if (u < HT) pass(u, A1, A0);
    }
    st.mark(3);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) hb[nt][r] = act_fn<ACT>(pre[nt][r], relu);
    out_chunk(w3 + HT * OT * 64, KSL);
    st.mark(4);
  }
}

// y = norm(v): ActNorm for Glow, eval-mode BatchNorm for RealNVP (identity params when absent)
template <int KIND>
__device__ __forceinline__ float norm_fn(float v, float p0, float p1, float p2, float p3) {
  if constexpr (KIND == GBNF_KIND_GLOW) {
    return (v + p0) * p1;                 // (x + bias) * exp(logs), models/layers.py:488-533
  } else {
    float xhat = (v - p0) / p1;           // (x - mean) / sqrt(var + eps), models/layers.py:353
    return p2 * xhat + p3;                // exp(log_gamma) * x_hat + beta, models/layers.py:354
  }
}

// x = norm^-1(y).  ActNorm reverse: y * exp(-logs) - bias (models/layers.py:493-533, the packer stores exp(-logs) in
// p2); BatchNorm.inverse: (y - beta) * exp(-log_gamma) * sqrt(var + eps) + mean (models/layers.py:360-372).
template <int KIND>
__device__ __forceinline__ float invnorm_fn(float y, float p0, float p1, float p2, float p3) {
  if constexpr (KIND == GBNF_KIND_GLOW) {
    return y * p2 - p0;
  } else {
    return ((y - p3) / p2) * p1 + p0;
  }
}

// scale = sigmoid(v) and log(scale) from one exp: e = exp(-v); scale = 1/(1+e); log scale = -log(1+e).
// (the reference takes log() of the rounded sigmoid, models/glow.py:333-338; both are within an ulp or
// two of the exact value).  v_exp_f32 / v_rcp_f32 / v_log_f32 are 1-ulp instructions.
__device__ __forceinline__ void sigmoid_logsigmoid(float v, float& sc, float& lsc) {
#ifdef GBNF_EPILOGUE_LIBM
  sc = 1.0f / (1.0f + expf(-v));
  lsc = logf(sc);
#else
  const float e = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
  const float s1 = 1.0f + e;
  sc = __builtin_amdgcn_rcpf(s1);
  lsc = -0.69314718055994531f * __builtin_amdgcn_logf(s1);
#endif
}

// exp(v) on v_exp_f32 with the rounding error of v*log2(e) folded back in (keeps ~1-2 ulp for |v| < 64)
__device__ __forceinline__ float exp_fast(float v) {
#ifdef GBNF_EPILOGUE_LIBM
  return expf(v);
#else
  const float t = v * 1.4426950408889634f;
  const float lo = __builtin_fmaf(v, 1.4426950408889634f, -t) + v * 1.9259629911266175e-8f;  // log2(e) lo part
  const float e = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(e, lo * 0.69314718055994531f, e);
#endif
}

// per-lane view of one {slot, p0..p3} table (8 entries): works for LDS and global pointers
struct LaneTable {
  int slot[NENT];
  float p0[NENT], p1[NENT], p2[NENT], p3[NENT];
  template <typename P>
  __device__ __forceinline__ void load(P tab) {   // tab -> [5 arrays][4 g][NENT], already offset by g*NENT
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      i32x4 s4 = *reinterpret_cast<const i32x4*>(tab + h * 4);
      f32x4 a4 = *reinterpret_cast<const f32x4*>(tab + 32 + h * 4);
      f32x4 b4 = *reinterpret_cast<const f32x4*>(tab + 64 + h * 4);
      f32x4 c4 = *reinterpret_cast<const f32x4*>(tab + 96 + h * 4);
      f32x4 d4 = *reinterpret_cast<const f32x4*>(tab + 128 + h * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        slot[h * 4 + e] = s4[e]; p0[h * 4 + e] = a4[e]; p1[h * 4 + e] = b4[e];
        p2[h * 4 + e] = c4[e]; p3[h * 4 + e] = d4[e];
      }
    }
  }
};

template <int KIND, int HT, int KSL, int KS1, int OT, int NT, int LMID, int ACTA, int ACTB>
__global__ void __launch_bounds__(64) flow_kernel(const FlowLaunch p) {
  constexpr int ZS = 16 * NT + 1;   // +1: conflict-free transposed x load / z store
  constexpr int NNETS = (KIND == GBNF_KIND_REALNVP) ? 2 : 1;
  constexpr NetLayoutRT L(HT, KS1, OT, lmid_mid_layers(LMID));
  constexpr int STEP_WORDS = SMALL_WORDS + NNETS * L.NET_WORDS;

  // one dynamic LDS block (16-byte aligned base): [ per-step tables | Z ]
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const bool lds_tables = p.n_steps <= LDS_TABLE_STEPS;
  uint32_t* SM = lds;
  float* Z = reinterpret_cast<float*>(lds + (lds_tables ? p.n_steps * SMALL_WORDS : 0));

  const int lane = threadIdx.x;
  const int i = lane & 15;
  const int g = lane >> 4;

  // ---- XCD-aware block -> (component, sample tile).  Blocks are dealt round-robin over the
  //      8 XCDs; give each XCD a contiguous run of the (component-major) work list so one
  //      component's weights stay in one XCD's L2.  Bijective for any grid size.
  int comp, tile, batch;
  {
    const int total = gridDim.x;
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int base = total >> 3, rem = total & 7;
    const int q = xcd * base + (xcd < rem ? xcd : rem) + j;
    const int per_comp = p.n_tiles * p.n_batches;   // work list: component-major, then batch, then tile
    comp = q / per_comp;
    const int r = q - comp * per_comp;
    batch = r / p.n_tiles;
    tile = r - batch * p.n_tiles;
  }
  const uint32_t* __restrict__ blob = p.blobs[p.c_begin + comp];
  const int d = p.d;
  const int64_t row0 = (int64_t)tile * (16 * NT);
  const float* __restrict__ xin = p.xs[batch];
  const uint32_t* tail = blob + (size_t)p.n_steps * STEP_WORDS;   // final slot of logical feature j
  const bool inv = p.inverse != 0;

  // ---- per-step index/normalisation tables -> LDS (1.3 KB per step), x tile -> Z[feature][sample]
  //      (inverse: the input is z in the reference's output order, so feature j starts in its FINAL slot)
  if (lds_tables) {
    for (int s = 0; s < p.n_steps; ++s) {
      const uint32_t* src = blob + (size_t)s * STEP_WORDS;
      for (int w = lane * 4; w < SMALL_WORDS; w += 256)
        *reinterpret_cast<i32x4*>(SM + s * SMALL_WORDS + w) = *reinterpret_cast<const i32x4*>(src + w);
    }
  }
  if (lane < d) {
    const int zslot = inv ? (int)tail[lane] : lane;
    // every row's load in flight before the first LDS store (rows past the batch re-read its last row and are zeroed)
    float xv[16 * NT];
    const int64_t last = p.n - 1;
#pragma unroll
    for (int r = 0; r < 16 * NT; ++r) {
      const int64_t n = row0 + r;
      xv[r] = xin[(n < p.n ? n : last) * d + lane];
    }
#pragma unroll
    for (int r = 0; r < 16 * NT; ++r) Z[zslot * ZS + r] = (row0 + r < p.n) ? xv[r] : 0.0f;
  }
  __syncthreads();

  float ld[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) ld[nt] = 0.0f;
  float ld_const = 0.0f;
  Stamps st;
  st.start();

  for (int it = 0; it < p.n_steps; ++it) {
    const int step = inv ? p.n_steps - 1 - it : it;      // the inverse undoes the steps last to first
    const uint32_t* __restrict__ sp = blob + (size_t)step * STEP_WORDS;
    LaneTable tin, tout;
    float step_ld;
    // per-step activation variants only: 1 = relu, from the step header (uniform)
    const bool relu_a = ACTA == 3 && __builtin_amdgcn_readfirstlane(sp[2]) != 0;
    const bool relu_b = ACTB == 3 && __builtin_amdgcn_readfirstlane(sp[3]) != 0;
    if (lds_tables) {
      const uint32_t* sm = SM + step * SMALL_WORDS;
      step_ld = as_f32(sm[1]);
      tin.load(sm + SMALL_HDR + g * NENT);
      tout.load(sm + SMALL_HDR + 160 + g * NENT);
    } else {
      step_ld = as_f32(sp[1]);
      tin.load(sp + SMALL_HDR + g * NENT);
      tout.load(sp + SMALL_HDR + 160 + g * NENT);
    }
    ld_const += inv ? -step_ld : step_ld;

    // ---- normalise the coupling net's inputs in place; they are the first layer's B operands
    float zb[KS1][NT];
#pragma unroll
    for (int e = 0; e < KS1; ++e) {
      const bool live = tin.slot[e] >= 0;
      const int zoff = (live ? tin.slot[e] : 0) * ZS + i;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float v = Z[zoff + 16 * nt];
        if (!inv) {
          v = norm_fn<KIND>(v, tin.p0[e], tin.p1[e], tin.p2[e], tin.p3[e]);
          if (live) Z[zoff + 16 * nt] = v;
        } else if (live) {
          // the stored value IS the normalised input of the net; un-normalise it for the preceding step
          Z[zoff + 16 * nt] = invnorm_fn<KIND>(v, tin.p0[e], tin.p1[e], tin.p2[e], tin.p3[e]);
        }
        zb[e][nt] = live ? v : 0.0f;
      }
    }

    // ---- coupling network(s) on the matrix cores
    st.mark(0);
    f32x4 outA[OT][NT];
    coupling_net<HT, KSL, KS1, OT, NT, LMID, ACTA>(sp + SMALL_WORDS, zb, lane, g, outA, st, relu_a);
    f32x4 outB[OT][NT];
    if constexpr (KIND == GBNF_KIND_REALNVP) {
      coupling_net<HT, KSL, KS1, OT, NT, LMID, ACTB>(sp + SMALL_WORDS + L.NET_WORDS, zb, lane, g, outB, st, relu_b);
    }

    // The epilogue's first accumulator reads may sit directly behind a (direction) branch, 2-3 issue slots after the
    // last MFMA; the compiler's MFMA-result hazard padding was seen to miss that path (gfx950, ROCm 7.2: the inverse
    // direction read the last output row one k-step short).  16 wait states cover an 8-pass MFMA on every path.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7\n\ts_nop 7");
    __builtin_amdgcn_sched_barrier(0);

    // ---- coupling transform of the other half, in place, + per-lane log-det partials
    if (KIND == GBNF_KIND_GLOW && !p.additive) {
      // affine, "cross" split: rows 2j / 2j+1 of the last Linear are (shift_j, raw_j) and sit
      // in adjacent accumulator registers of the same lane.  models/glow.py:331-338.
#pragma unroll
      for (int e = 0; e < 2 * OT && e < NENT; ++e) {
        const int o = e >> 1, pp = e & 1;
        const bool live = tout.slot[e] >= 0;
        const int zoff = (live ? tout.slot[e] : 0) * ZS + i;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float v = Z[zoff + 16 * nt];
          const float shift = outA[o][nt][2 * pp], raw = outA[o][nt][2 * pp + 1];
          float sc, lsc;
          sigmoid_logsigmoid(raw + 2.0f, sc, lsc);
          if (!inv) {
            v = norm_fn<KIND>(v, tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
            v = (v + shift) * sc;
          } else {
            v = v / sc - shift;                                   // FlowStep.decode, models/glow.py:352-355
            v = invnorm_fn<KIND>(v, tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
            lsc = -lsc;
          }
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
          if (!inv) v = norm_fn<KIND>(v, tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
          if constexpr (KIND == GBNF_KIND_GLOW) {
            v = inv ? v - outA[o][nt][r] : v + outA[o][nt][r];      // additive, models/glow.py:328-329 / 349-350
            if (inv) v = invnorm_fn<KIND>(v, tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);
            if (live) Z[zoff + 16 * nt] = v;
          } else {
            const float shift = outA[o][nt][r], scale = outB[o][nt][r];
            if (!inv) {
              v = shift + v * exp_fast(scale);            // models/transformations.py:575
            } else {
              v = (v - shift) * exp_fast(-scale);         // the true inverse of :575 (the reference's own
              v = invnorm_fn<KIND>(v, tout.p0[e], tout.p1[e], tout.p2[e], tout.p3[e]);   // .inverse is not: SURVEY S3)
            }
            if (live) {
              Z[zoff + 16 * nt] = v;
              ld[nt] += inv ? -scale : scale;             // models/transformations.py:577
            }
          }
        }
      }
    }
    __syncthreads();
    st.mark(5);
  }

  // ---- base log-density + log|det J|, folded over the 4 lane groups
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
      if (p.ll_out) p.ll_out[o] = (q - 0.91893853320467274f * (float)d) + ldj;   // -d/2 log(2 pi)
    }
  }

  // ---- z in the reference's feature order (post-permutation, post-swap); inverse: x, feature j sits in slot j
  if (p.z_out != nullptr && lane < d) {
    const int slot = inv ? lane : (int)tail[lane];
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
    for (int k = 0; k < 8; ++k) p.dbg[(size_t)blockIdx.x * 8 + k] = st.acc[k];
  }
#endif
}

// dynamic LDS bytes a launch needs (tables for K steps when they fit + Z)
inline size_t flow_lds_bytes(int n_steps, int nt) {
  const size_t tables = n_steps <= LDS_TABLE_STEPS ? (size_t)n_steps * SMALL_WORDS : 0;
  return (tables + (size_t)ZSLOTS * (16 * nt + 1)) * 4;
}

// ---------------------------------------------------------------------------------
// variant registry: each variant translation unit instantiates one kernel and registers a launcher
// ---------------------------------------------------------------------------------
struct VariantKey {
  int kind, ht, ksl, ks1, ot, nt, lmid, act_a, act_b;
  bool operator==(const VariantKey& o) const {
    return kind == o.kind && ht == o.ht && ksl == o.ksl && ks1 == o.ks1 && ot == o.ot && nt == o.nt &&
           lmid == o.lmid && act_a == o.act_a && act_b == o.act_b;
  }
};
using LaunchFn = hipError_t (*)(const FlowLaunch&, unsigned grid, hipStream_t);
void register_variant(const VariantKey& key, LaunchFn fn, const char* name);
int tuning_wg_pairs();      // launch policy of the split kernels' 4-wave workgroup pairs (gbnf_api.hip, gbnf_tuning_set "wg_pairs")

#define GBNF_INSTANTIATE(KIND, HT, KSL, KS1, OT, NT, LMID, ACTA, ACTB)                                      \
  namespace gbnf {                                                                                          \
  static hipError_t launch_##KIND##_##HT##_##KSL##_##KS1##_##OT##_##NT##_##LMID##_##ACTA##_##ACTB(          \
      const FlowLaunch& p, unsigned grid, hipStream_t s) {                                                  \
    hipLaunchKernelGGL((flow_kernel<KIND, HT, KSL, KS1, OT, NT, LMID, ACTA, ACTB>), dim3(grid), dim3(64),   \
                       flow_lds_bytes(p.n_steps, NT), s, p);                                                \
    return hipGetLastError();                                                                               \
  }                                                                                                         \
  static const int reg_##KIND##_##HT##_##KSL##_##KS1##_##OT##_##NT##_##LMID##_##ACTA##_##ACTB =             \
      (register_variant(VariantKey{KIND, HT, KSL, KS1, OT, NT, LMID, ACTA, ACTB},                           \
                        launch_##KIND##_##HT##_##KSL##_##KS1##_##OT##_##NT##_##LMID##_##ACTA##_##ACTB,      \
                        "flow_kernel<" #KIND "," #HT "," #KSL "," #KS1 "," #OT "," #NT "," #LMID "," #ACTA  \
                        "," #ACTB ">"),                                                                     \
       0);                                                                                                  \
  }

}  // namespace gbnf
