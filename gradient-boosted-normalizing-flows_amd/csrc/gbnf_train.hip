// gbnf_train.hip -- the training path of one boosted component: forward on the LIVE
// parameters and the backward pass (recompute-in-backward), for gfx950.
//
// Why a second kernel family: the evaluation kernels read parameters from a packed, immutable blob -- right for
// density evaluation, wrong for training, where every optimiser step changes every weight.  The trainer therefore
// binds the DEVICE addresses of the caller's parameter tensors once and reads them in their natural nn.Linear
// (out, in) row-major layout; an optimiser that updates in place needs no repacking at all.
//
//   prep_kernel          at the start of every call: the LIVE f32 weights -> split-f16 MFMA A fragments (hi, mid), both
//                        orientations (W forward, W^T backward), zero padded, plus zero-padded bias copies.
//   train_kernel<KIND, MODE, NT>
//                        one workgroup of 8 waves = NT 16-sample tiles of one component.  Activations that feed a
//                        dense layer are LDS-resident as split rows [sample][hi | mid] (the B operand of
//                        v_mfma_f32_16x16x32_f16 is one ds_read_b128); the waves split a layer's output tiles in
//                        pairs and stream the fragments from L2 through a register ring (three MFMAs per product,
//                        f32 accumulation).
//                        MODE 0: x -> z, ldj (+ the trace of every step's normalised state).   MODE 1: the steps in
//                        reverse: recompute the nets, coupling / normalisation backward, net backward (dgrad chain
//                        in place of the activations), emit the two operands of every weight gradient to a
//                        workspace, atomically accumulate the ActNorm / BatchNorm parameter gradients.
//   wgrad_kernel         all weight/bias gradients of a component in ONE launch: dW = D . A^T contracted over
//                        the samples on the f16 pipe (operands split in registers; split over sample chunks,
//                        atomic accumulation).
//   bn_stats_kernel, bn_bwd_fix_kernel, rows_to_slots_kernel, slots_to_rows_kernel
//                        train-mode BatchNorm (batch statistics need the whole batch: one launch per step).
//
// Reference semantics differentiated (the forward is the one of gbnf_flow_kernel.hip.h):
//   FlowStep.encode models/glow.py:317-342, _ActNorm.forward models/layers.py:488-533, TanhNet/ReLUNet
//   models/layers.py:208-243, RealNVP.forward models/transformations.py:560-579, BatchNorm.forward (running
//   statistics) models/layers.py:337-358.  Caller: loss.backward() of density_experiment.py:366-374 through
//   compute_kl_pq_loss (density_experiment.py:606-660).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../include/gbnf.h"
#include "gbnf_internal.h"


namespace gbnf {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int TR_MAX_NT = 2;        // 16-sample tiles per workgroup (they share every weight fragment a wave loads)
constexpr int TR_MAX_LAYERS = 6;    // Linear layers per coupling net (depth <= 2; a ResidualNet of 2 blocks has 6)
constexpr int TR_MAX_IN = 32;       // coupling-net input / coupled-half width
constexpr int TR_MAX_HIDDEN = 512;  // hidden width (32 output tiles = 16 tile pairs per layer)
constexpr int TR_LDS_BYTES = 160 * 1024;
constexpr int TR_WS_SLACK_ROWS = 320;   // workspace rows behind the last operand region: a block of wgrad_kernel reads up to 256 rows
#ifndef GBNF_TR_WAVES
#define GBNF_TR_WAVES 8
#endif
constexpr int TR_PD = 4;            // weight-prefetch distance of the dense layers, in k-chunks (even: B ping-pong)
static_assert(TR_PD % 2 == 0, "the LDS operand ping-pong follows the ring index");
constexpr int TR_WAVES = GBNF_TR_WAVES;         // waves per workgroup; they share the workgroup's sample tiles and split every layer's output tiles

struct TrLayer {
  const float* W;      // (rows, cols) row-major = nn.Linear.weight (out, in)
  const float* b;      // (rows,)
  int64_t gW, gb;      // float offsets into the flat gradient buffer
  int64_t fw, bw;      // u32x4 offsets of the split fragments of W (forward) / W^T (backward) in the trainer's fragment buffer
  int64_t fb;          // float offset (same buffer) of the bias, zero-padded to whole 16-unit tiles
  int rows, cols;
};
struct TrNet {
  TrLayer layer[TR_MAX_LAYERS];
  int n_layers, act;
};
struct TrStep {
  int in_f, out_f, has_norm, pad0;
  const float* na;     // glow: actnorm bias   | realnvp: bn log_gamma
  const float* nb;     // glow: actnorm logs   | realnvp: bn beta
  const float* mean;   // realnvp: bn running_mean
  const float* var;    // realnvp: bn running_var
  float* bmean;        // realnvp: batch statistics of this step's input (batch-statistics mode), (d,) each
  float* bvar;
  float eps;
  int pad1;
  int64_t g_na, g_nb;  // float offsets into the flat gradient buffer
  int in_slot[TR_MAX_IN], out_slot[TR_MAX_IN];
  int feat[64];        // feat[slot] = index into na/nb/mean/var of the feature that lives in that slot at this step
  TrNet net[2];        // glow: net[0] = block; realnvp: net[0] = t_net (shift), net[1] = s_net (log-scale)
};

struct TrainLaunch {
  const TrStep* steps;
  const int* tail;     // final slot of logical feature j
  const float* x;      // (n, d)
  float* z_out;        // MODE 0
  float* ldj_out;      // MODE 0
  float* trace_out;    // MODE 0, optional: every step's normalised state [K][d][np], saved for the backward
  const float* trace;  // MODE 1, optional: that buffer (else the forward sweep is recomputed)
  // step-range launches (batch-statistics BatchNorm needs a grid-wide reduction between steps): the state / its gradient
  // travel between launches in slot layout [d][np]
  int k_begin, k_end;        // steps [k_begin, k_end) of the K
  int batch_stats;           // BatchNorm normalises with TrStep::bmean / bvar instead of the running statistics
  int ldj_accumulate;        // MODE 0: ldj_out += instead of =
  const float* state_in;     // MODE 0: state before step k_begin (null: x, row-major)
  float* state_out;          // MODE 0: state after step k_end - 1 (null: z_out / none)
  const float* gstate_in;    // MODE 1: gradient w.r.t. the state after step k_end - 1 (null: g_z through the final map)
  float* gstate_out;         // MODE 1: gradient w.r.t. the state before step k_begin (null: g_x, row-major)
  const float* g_z;    // MODE 1 (n, d) or null
  const float* g_ldj;  // MODE 1 (n,) or null
  const unsigned* gmax;  // MODE 1: bits of max(|g_z|, |g_ldj|) over the batch (gmax_kernel): fixes the gradient scale
  float* g_x;          // MODE 1 (n, d) or null
  float* grads;        // MODE 1 flat parameter-gradient buffer
  unsigned long long* dbg;   // diagnostic builds only
  unsigned* sat;             // library-wide counter: waves that stored a split operand beyond the fp16 range (it saturates)
  float* ws;           // MODE 1 workspace: operands of the weight gradients
  const u32x4* frag;   // split-f16 weight fragments of this call (prep_kernel)
  int64_t n, np;       // samples, samples rounded up to whole workgroups (16 * TR_MAX_NT)
  int d, K, kind, additive;
  int residual;        // coupling nets are ResidualNets (models/layers.py:246-301): n_hidden = 2 blocks + 1
  int n_hidden;        // hidden layers per net = depth + 1
  int hp, ip, op;      // padded hidden / net-input / net-output rows (multiples of 16)
  int hw, xw, ow;      // the same padded to 32: widths of the split activation rows
  int64_t net_rows;    // workspace rows (of np floats) per (step, net)
};

__device__ __forceinline__ float tr_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}
// four values at a time: `act` is uniform, so this is one scalar branch around straight-line code
__device__ __forceinline__ f32x4 tr_act4(int act, f32x4 v) {
#ifdef GBNF_TR_ABLATE_ACT           // diagnostic: no activation
  return v;
#endif
  f32x4 h;
  if (act == GBNF_ACT_TANH) {
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = tr_tanh(v[r]);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = fmaxf(v[r], 0.0f);
  }
  return h;
}
__device__ __forceinline__ f32x4 tr_dact4_mul(int act, f32x4 h, f32x4 v) {   // v * act'(pre-activation), through the output h
  f32x4 o;
  if (act == GBNF_ACT_TANH) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = v[r] * __builtin_fmaf(-h[r], h[r], 1.0f);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = h[r] > 0.0f ? v[r] : 0.0f;
  }
  return o;
}
// Accumulators are read right behind a loop exit: the compiler's MFMA-result hazard padding does not look across that
// branch on gfx950 / ROCm 7.2 (same finding as in gbnf_flow_kernel.hip.h: the last k-step went missing), so pad by hand.
// The accumulators are operands of the padding ("+a": they stay in AGPRs), so every read of them is ordered behind it.
__device__ __forceinline__ void tr_mfma_drain(f32x4& c0, f32x4& c1) {
  asm volatile("s_nop 7\n\ts_nop 7" : "+a"(c0), "+a"(c1));
}
// ... and the other direction: freshly written accumulators feed an MFMA in the next basic block.
__device__ __forceinline__ void tr_acc_settle(f32x4& c0, f32x4& c1) {
  asm volatile("s_nop 3" : "+a"(c0), "+a"(c1));
}
template <int NT>
__device__ __forceinline__ void tr_mfma_drain(f32x4 (&c0)[NT], f32x4 (&c1)[NT]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) tr_mfma_drain(c0[nt], c1[nt]);
}
template <int NT>
__device__ __forceinline__ void tr_acc_settle(f32x4 (&c0)[NT], f32x4 (&c1)[NT]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) tr_acc_settle(c0[nt], c1[nt]);
}

// Parameter tensors are global memory: say so, or every load is a flat_load that also counts against lgkmcnt.
typedef const float __attribute__((address_space(1)))* gptr;
__device__ __forceinline__ gptr tr_global(const float* p) { return (gptr)(p); }

// In-kernel phase stamps, diagnostic builds only (-DGBNF_TRAIN_STAMPS; tools/train_stamps.py).
struct TrStamps {
#ifdef GBNF_TRAIN_STAMPS
  unsigned long long last = 0, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  __device__ __forceinline__ void mark(int k) {
#ifdef GBNF_TRAIN_STAMPS
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    if (k >= 0) acc[k] += t - last;
    last = t;
#else
    (void)k;
#endif
  }
};

__device__ __forceinline__ int tr_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class P>
__device__ __forceinline__ P tr_uniform_ptr(P p) {          // same, for any pointer type
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (P)(((unsigned long long)hi << 32) | lo);
}
// A pointer that is the same in every lane, moved to scalar registers: address arithmetic on it then does not depend
// on the vector-memory load that fetched it from the step table (which would drag a vmcnt(0) into every prefetch).
__device__ __forceinline__ gptr tr_uniform(gptr p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (gptr)(((unsigned long long)hi << 32) | lo);
}

// ---- split-f16 dense layers: an f32 operand is two fp16 pieces x ~ hi + mid and a product
// is three v_mfma_f32_16x16x32_f16 with f32 accumulation.  Weights: prep_kernel splits the LIVE f32 parameter tensors
// into MFMA A fragments ([o][c][hi|mid][64 lanes][8 halfs], k = 32c + 8g + j, zero padded: no edge handling in the
// stream at all) at the start of every call -- both orientations (W for the forward, W^T for the backward), 5 us.
// Activations that feed a dense layer live in LDS as rows [sample][hi: width halfs | mid: width halfs] (+16 B pad: the 16
// lanes of a group hit 16 different bank quads), so a lane's B operand (8 consecutive units of its sample) is ONE
// ds_read_b128; they are split where they are produced (layer epilogues, the element-wise stages), never in the stream.
typedef const u32x4 __attribute__((address_space(1)))* gfrag;

__device__ __forceinline__ f32x4 tr_mfma16(u32x4 a, u32x4 b, f32x4 c) {
#ifdef GBNF_TR_ABLATE_MFMA          // diagnostic: one cheap VALU op instead of the MFMA (keeps every value live)
  c[0] += __builtin_bit_cast(float, a[0] ^ b[0]);
  return c;
#endif
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// hi = f16(x) (toward zero), mid = f16(x - hi) for a pair, clamped to the fp16 range
__device__ __forceinline__ void tr_split_pair(float x0, float x1, unsigned& hi, unsigned& mid) {
  x0 = __builtin_amdgcn_fmed3f(x0, -65504.0f, 65504.0f);
  x1 = __builtin_amdgcn_fmed3f(x1, -65504.0f, 65504.0f);
  const auto h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
  hi = __builtin_bit_cast(unsigned, h);
  const auto m = __builtin_amdgcn_cvt_pkrtz(x0 - (float)h[0], x1 - (float)h[1]);
  mid = __builtin_bit_cast(unsigned, m);
}
// Workgroup barrier that publishes LDS traffic only: the epilogues' global stores (weight-gradient operands) and the
// fragment prefetches stay in flight across it -- __syncthreads() would drain them (vmcnt(0)) at the end of every layer.
__device__ __forceinline__ void tr_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// LDS pointers carry their address space: through a generic pointer every access is a flat_load / flat_store, which
// counts against vmcnt AND lgkmcnt -- the B-operand reads of the dense stream would then wait for every weight fragment
// in flight (s_waitcnt vmcnt(0) per iteration: no prefetch at all).
#define TR_LDS __attribute__((address_space(3)))
typedef float TR_LDS* lfp;
typedef const float TR_LDS* lcfp;
typedef int TR_LDS* lip;
typedef unsigned TR_LDS* lup;
typedef unsigned char TR_LDS* lbp;

// a split activation buffer in LDS: 16 NT rows (samples) of [hi: w halfs][mid: w halfs] + 16 B
struct TrSplit {
  lbp base;
  int w;                                                  // channels (multiple of 32)
  __device__ __forceinline__ int rs() const { return 4 * w + 16; }
  __device__ __forceinline__ void put1(int i, int ch, float v) const {       // one element (element-wise stages)
    unsigned h, m;
    tr_split_pair(v, 0.0f, h, m);
    lbp q = base + i * rs() + 2 * ch;
    *reinterpret_cast<unsigned short TR_LDS*>(q) = (unsigned short)h;
    *reinterpret_cast<unsigned short TR_LDS*>(q + 2 * w) = (unsigned short)m;
  }
  __device__ __forceinline__ void put4(int i, int ch0, f32x4 v) const {       // 4 consecutive channels (dense epilogues)
    unsigned h01, m01, h23, m23;
    tr_split_pair(v[0], v[1], h01, m01);
    tr_split_pair(v[2], v[3], h23, m23);
    lbp q = base + i * rs() + 2 * ch0;
    *reinterpret_cast<u32x2 TR_LDS*>(q) = u32x2{h01, h23};
    *reinterpret_cast<u32x2 TR_LDS*>(q + 2 * w) = u32x2{m01, m23};
  }
  __device__ __forceinline__ f32x4 get4(int i, int ch0) const {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const unsigned TR_LDS* qh = reinterpret_cast<const unsigned TR_LDS*>(base + i * rs() + 2 * ch0);
    const unsigned TR_LDS* qm = reinterpret_cast<const unsigned TR_LDS*>(base + i * rs() + 2 * ch0 + 2 * w);
    const unsigned h0 = qh[0], h1 = qh[1], m0 = qm[0], m1 = qm[1];
    const h2 h01 = __builtin_bit_cast(h2, h0), h23 = __builtin_bit_cast(h2, h1);
    const h2 m01 = __builtin_bit_cast(h2, m0), m23 = __builtin_bit_cast(h2, m1);
    f32x4 r;
    r[0] = (float)h01[0] + (float)m01[0];
    r[1] = (float)h01[1] + (float)m01[1];
    r[2] = (float)h23[0] + (float)m23[0];
    r[3] = (float)h23[1] + (float)m23[1];
    return r;
  }
};

// out[u][s] = epi(u0, acc + bias) for the 16*out_tiles output units: A = pre-split fragments (kc32 chunks of 32 k per
// output tile), `in` = split activation rows.  The waves of the workgroup split the output tiles in pairs (wave w owns
// pairs w, w + TR_WAVES, ...); a wave's (pair, chunk) iterations form ONE software-pipelined stream with TR_PD
// iterations of fragments in flight (unconditional loads, static ring indices); a pair's bias is requested up front and
// added in its epilogue (`bias`: the zero-padded copy in the fragment buffer, or null).  A fragment is used for all NT sample tiles of the workgroup (rows i + 16 nt of `in`).
// epi(u0, nt, v): units u0..u0+3 of sample i of tile nt; units past the layer's width arrive as exact zeros.
template <int NT, class Epi>
__device__ __forceinline__ void tr_dense(gfrag A, gptr bias, int kc32, int frag_tiles, const TrSplit in,
                                         int out_tiles, int lane, int wave, TrStamps& stamps, Epi epi) {
  const int i = lane & 15, g = lane >> 4;
  const int n_pairs = (out_tiles + 1) >> 1;
  const int my_pairs = wave < n_pairs ? (n_pairs - wave + TR_WAVES - 1) / TR_WAVES : 0;
  const int T = my_pairs * kc32;
  typedef const f32x4 __attribute__((address_space(1)))* gv4;
  auto load_bias = [&](int o) -> f32x4 {          // the padded copy prep_kernel wrote: units past the layer are zeros
    if (bias == nullptr) return f32x4{0.f, 0.f, 0.f, 0.f};
    return *(gv4)(bias + 16 * (o < out_tiles ? o : 0) + 4 * g);
  };
  u32x4 r0h[TR_PD], r0m[TR_PD], r1h[TR_PD], r1m[TR_PD];
  int pl = wave, cl = 0;                       // load cursor (pair, chunk)
  auto issue = [&](u32x4& a0h, u32x4& a0m, u32x4& a1h, u32x4& a1m) {
    const int o0 = 2 * pl < frag_tiles ? 2 * pl : 0, o1 = 2 * pl + 1 < frag_tiles ? 2 * pl + 1 : 0;   // past the end: tile 0 again
    const gfrag f0 = A + ((size_t)o0 * kc32 + cl) * 128 + lane, f1 = A + ((size_t)o1 * kc32 + cl) * 128 + lane;
#ifndef GBNF_TR_ABLATE_LOADS        // (diagnostic builds: tools/build_train_ablations.sh -- timing only, results wrong)
    a0h = f0[0]; a0m = f0[64];
    a1h = f1[0]; a1m = f1[64];
#else
    (void)f0; (void)f1;
#endif
    if (++cl == kc32) { cl = 0; pl += TR_WAVES; }
  };
  constexpr int MAXP = (TR_MAX_HIDDEN / 32 + TR_WAVES - 1) / TR_WAVES;   // tile pairs per wave
  f32x4 bq0[MAXP], bq1[MAXP];
#pragma unroll
  for (int q = 0; q < MAXP; ++q) {
    bq0[q] = load_bias(2 * (wave + q * TR_WAVES));
    bq1[q] = load_bias(2 * (wave + q * TR_WAVES) + 1);
  }
#pragma unroll
  for (int j = 0; j < TR_PD; ++j) issue(r0h[j], r0m[j], r1h[j], r1m[j]);
  stamps.mark(5);
  int pair = wave, c = 0, q = 0;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc0[NT], acc1[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { acc0[nt] = zero; acc1[nt] = zero; }
  const lbp inl = in.base + i * in.rs() + 16 * g;
  const int mid_off = 2 * in.w, tile_off = 16 * in.rs();
  // B operands are read one iteration ahead (ping-pong on the ring index): a chunk's ds_read_b128s are issued behind
  // the previous chunk's MFMAs instead of in front of its own (the LDS round trip was exposed in every iteration).
  // B depends on the chunk only, not on the output tile: after a pair's last chunk comes chunk 0 again.
  u32x4 Bh[2][NT], Bm[2][NT];
  auto load_b = [&](int cc, u32x4 (&h)[NT], u32x4 (&m)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      h[nt] = *reinterpret_cast<const u32x4 TR_LDS*>(inl + nt * tile_off + 64 * cc);
      m[nt] = *reinterpret_cast<const u32x4 TR_LDS*>(inl + nt * tile_off + 64 * cc + mid_off);
    }
  };
  load_b(0, Bh[0], Bm[0]);
  auto body = [&](const u32x4& a0h, const u32x4& a0m, const u32x4& a1h, const u32x4& a1m, const u32x4 (&bh)[NT],
                  const u32x4 (&bm)[NT], u32x4 (&nh)[NT], u32x4 (&nm)[NT]) {
    load_b(c + 1 == kc32 ? 0 : c + 1, nh, nm);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc0[nt] = tr_mfma16(a0m, bh[nt], acc0[nt]);
      acc1[nt] = tr_mfma16(a1m, bh[nt], acc1[nt]);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc0[nt] = tr_mfma16(a0h, bm[nt], acc0[nt]);
      acc1[nt] = tr_mfma16(a1h, bm[nt], acc1[nt]);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc0[nt] = tr_mfma16(a0h, bh[nt], acc0[nt]);
      acc1[nt] = tr_mfma16(a1h, bh[nt], acc1[nt]);
    }
    if (++c == kc32) {
      tr_mfma_drain(acc0, acc1);
      f32x4 bias0 = bq0[0], bias1 = bq1[0];
#pragma unroll
      for (int qq = 1; qq < MAXP; ++qq) {
        bias0 = q == qq ? bq0[qq] : bias0;
        bias1 = q == qq ? bq1[qq] : bias1;
      }
      const int o = 2 * pair;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4 v0, v1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v0[r] = acc0[nt][r] + bias0[r];             // padded units: zero fragments + zero bias = exactly zero
          v1[r] = acc1[nt][r] + bias1[r];
        }
        epi(16 * o + 4 * g, nt, v0);
        if (o + 1 < out_tiles) epi(16 * o + 16 + 4 * g, nt, v1);
        acc0[nt] = zero; acc1[nt] = zero;
      }
      tr_acc_settle(acc0, acc1);
      c = 0; pair += TR_WAVES; ++q;
    }
  };
  int t = 0;
  for (; t + TR_PD <= T; t += TR_PD) {
#pragma unroll
    for (int j = 0; j < TR_PD; ++j) {
      body(r0h[j], r0m[j], r1h[j], r1m[j], Bh[j & 1], Bm[j & 1], Bh[(j + 1) & 1], Bm[(j + 1) & 1]);
      issue(r0h[j], r0m[j], r1h[j], r1m[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < TR_PD - 1; ++j)
    if (t + j < T) body(r0h[j], r0m[j], r1h[j], r1m[j], Bh[j & 1], Bm[j & 1], Bh[(j + 1) & 1], Bm[(j + 1) & 1]);
  stamps.mark(6);
  tr_lds_barrier();    // the layer's output (LDS) is complete for every wave
  stamps.mark(7);
}

// ---- gradient scaling.  The split-f16 operands carry f32 accuracy only for magnitudes between ~0.1 and 65504 (the mid
// piece of a smaller value is an fp16 subnormal: absolute floor 2^-24).  Activations are O(1) by construction; gradients
// are as large as the caller's loss makes them (a mean over 65536 samples hands in 1.5e-5 per sample).  The backward
// pass is linear in the upstream gradient, so it runs on alpha * gradient with alpha = the power of two that puts the
// largest upstream entry of the batch at 2^2, and every result (g_x, parameter gradients) is multiplied by 1 / alpha --
// exact scalings.  2^2 leaves a factor 16 000 of headroom for what the chain multiplies a sample's gradient by (RealNVP:
// exp(scale)) before the fp16 range saturates; entries far below the largest one keep an ABSOLUTE accuracy of 2^-24 * 4
// relative to it, i.e. what f32 rounding of the large entries costs anyway.  gmax_kernel finds that largest entry (non-negative floats order like their bit patterns).
__device__ __forceinline__ void tr_grad_scale(unsigned max_bits, float& alpha, float& inv_alpha) {
  const int e = (int)((max_bits >> 23) & 255u);          // biased exponent of the largest |upstream gradient|
  if (max_bits == 0u || e == 255) { alpha = 1.0f; inv_alpha = 1.0f; return; }
  int k = 2 - (e - 127);                                   // alpha = 2^k: the largest upstream entry lands in [4, 8)
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  alpha = __builtin_bit_cast(float, (unsigned)(k + 127) << 23);
  inv_alpha = __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
}

__global__ void __launch_bounds__(256) gmax_kernel(const float* __restrict__ g_z, const float* __restrict__ g_ldj, int64_t n, int d,
                                                   unsigned* __restrict__ out) {
  unsigned m = 0u;
  const int64_t stride = (int64_t)gridDim.x * 256, t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g_z != nullptr) {
    const int64_t total = n * d;
    const int64_t n4 = (reinterpret_cast<uintptr_t>(g_z) & 15u) == 0 ? total / 4 : 0;      // 16-byte loads where the buffer allows
    const u32x4* g4 = reinterpret_cast<const u32x4*>(g_z);
#pragma unroll 4
    for (int64_t e = t0; e < n4; e += stride) {
      const u32x4 v = g4[e];
      m = max(max(m, v[0] & 0x7fffffffu), max(max(v[1] & 0x7fffffffu, v[2] & 0x7fffffffu), v[3] & 0x7fffffffu));
    }
    for (int64_t e = 4 * n4 + t0; e < total; e += stride) m = max(m, __builtin_bit_cast(unsigned, g_z[e]) & 0x7fffffffu);
  }
  if (g_ldj != nullptr)
    for (int64_t e = t0; e < n; e += stride) m = max(m, __builtin_bit_cast(unsigned, g_ldj[e]) & 0x7fffffffu);
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) m = max(m, (unsigned)__shfl_xor((int)m, s));
  // one atomic per workgroup: 2752 waves on one address took 30 of this kernel's 36 us at N = 65536
  __shared__ unsigned wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
    if (m != 0u) atomicMax(out, m);
  }
}

__device__ __forceinline__ float tr_group_sum(float v) {   // sum over the 16 lanes of a lane group (all active)
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}

template <int KIND, int MODE, int NT>
__global__ void __launch_bounds__(64 * TR_WAVES) train_kernel(const TrainLaunch p) {
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  const lfp lds = (lfp)lds_raw;
  constexpr int TS = 16 * NT;                            // samples per workgroup
  constexpr int S = TS + 1;                              // LDS row stride (floats) of the f32 arrays: samples + 1 pad
  constexpr int GS = 4 * TR_WAVES;                       // lane groups of 16 in the workgroup: elementwise loop stride
  const int lane = threadIdx.x & 63, i = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // scalar: tile loops / branches on it are uniform
  const int g = 4 * wave + (lane >> 4);                  // lane-group index (elementwise work); MFMA code uses lane, wave
  const int d = p.d, K = p.K;
  const int64_t row0 = (int64_t)blockIdx.x * TS;
  const int64_t tile0 = (int64_t)blockIdx.x * NT;        // first 16-sample tile (the workspace is tiled by 16 samples)
  // a lane works on sample i of each of the NT tiles: local row ii = i + 16 nt, global row row0 + ii
#define TR_NT _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)

  TrStamps stamps;
  stamps.mark(-1);
  // a split operand saturates at +-65504: remember it, count the wave once at the end
  bool sat = false;
  auto watch = [&](float v) { sat = sat || !(__builtin_fabsf(v) <= 65504.0f); };
  auto watch4 = [&](f32x4 v) { watch(v[0]); watch(v[1]); watch(v[2]); watch(v[3]); };
  auto report = [&]() {
    if (p.sat != nullptr && __any(sat) && lane == 0) atomicAdd(p.sat, 1u);
  };
  // per-step tables, staged once: slot maps and the normalisation constants of every slot
  // (everything a step needs lives in LDS from here on: a read of the step table in global memory inside the step loop
  // is a vector-memory load, and waiting for it drains the weight prefetch)
  const lip TI = reinterpret_cast<lip>(lds);               // [K][128]: in_slot[32], out_slot[32], feat[64]
  const lfp TP = lds + K * 128;                       // [K][4][64]: p0..p3 per slot
  for (int k = 0; k < K; ++k) {
    const TrStep& st = p.steps[k];
    for (int t = threadIdx.x; t < 192; t += 64 * TR_WAVES)
    if (t < 32) TI[k * 128 + t] = st.in_slot[t];
    else if (t < 64) TI[k * 128 + t] = st.out_slot[t - 32];
    else if (t >= 128) TI[k * 128 + t - 64] = st.feat[t - 128];
    else {
      const int sl = t - 64;
      float p0 = 0.f, p1 = 1.f, p2 = 0.f, p3 = 0.f;
      if (sl < d) {
        const int f = st.feat[sl];
        if constexpr (KIND == GBNF_KIND_GLOW) {
          p3 = st.nb[f];                                   // logs: per-sample logdet term, models/layers.py:506-512
          p0 = st.na[f];                                   // y = (x + bias) * exp(logs)
          p1 = __expf(p3);
        } else if (st.has_norm) {
          const float* mu = p.batch_stats ? st.bmean : st.mean;
          const float* vr = p.batch_stats ? st.bvar : st.var;
          const float ve = vr[f] + st.eps, lg = st.na[f];
          p0 = mu[f];                                      // y = (x - mean) * [exp(log_gamma) / sqrt(var + eps)] + beta
          p1 = __expf(lg) / sqrtf(ve);
          p2 = st.nb[f];
          p3 = lg - 0.5f * __logf(ve);                     // models/layers.py:357-358
        }
      }
      const lfp tp = TP + k * 256 + sl;
      tp[0] = p0; tp[64] = p1; tp[128] = p2; tp[192] = p3;
    }
  }
  // ... and the per-layer descriptors (padded-bias offset, fragment offsets, rows, cols): 8 words per (step, net, layer)
  const lup TL = reinterpret_cast<lup>(lds + K * 384);
  for (int e = threadIdx.x; e < K * 2 * TR_MAX_LAYERS; e += 64 * TR_WAVES) {
    const int k = e / (2 * TR_MAX_LAYERS), q = (e / TR_MAX_LAYERS) & 1, l = e % TR_MAX_LAYERS;
    const TrLayer& L = p.steps[k].net[q].layer[l];
    const lup t = TL + e * 8;
    t[0] = (unsigned)L.fb; t[1] = (unsigned)(L.fb >> 32);
    t[2] = (unsigned)L.fw; t[3] = (unsigned)(L.fw >> 32);
    t[4] = (unsigned)L.bw; t[5] = (unsigned)(L.bw >> 32);
    t[6] = (unsigned)L.rows; t[7] = (unsigned)L.cols;
  }
  // ... and the per-step scalars: in_f, out_f, has_norm, activations, g_na, g_nb (two words each)
  const lup TSC = TL + K * 2 * TR_MAX_LAYERS * 8;
  for (int k = threadIdx.x; k < K; k += 64 * TR_WAVES) {
    const TrStep& st = p.steps[k];
    const lup t = TSC + k * 8;
    t[0] = (unsigned)st.in_f; t[1] = (unsigned)st.out_f; t[2] = (unsigned)st.has_norm;
    t[3] = (unsigned)st.net[0].act | ((unsigned)st.net[KIND == GBNF_KIND_GLOW ? 0 : 1].act << 8);   // per step and net
    t[4] = (unsigned)st.g_na; t[5] = (unsigned)((unsigned long long)st.g_na >> 32);
    t[6] = (unsigned)st.g_nb; t[7] = (unsigned)((unsigned long long)st.g_nb >> 32);
  }
  struct StepD { int in_f, out_f, has_norm, act[2]; long long g_na, g_nb; };
  auto step_desc = [&](int k) -> StepD {                 // from the LDS table, moved to scalar registers
    const lup t = TSC + k * 8;
    unsigned w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = __builtin_amdgcn_readfirstlane(t[e]);
    StepD D;
    D.in_f = (int)w[0]; D.out_f = (int)w[1]; D.has_norm = (int)w[2];
    D.act[0] = (int)(w[3] & 255u); D.act[1] = (int)(w[3] >> 8);
    D.g_na = (long long)(((unsigned long long)w[5] << 32) | w[4]);
    D.g_nb = (long long)(((unsigned long long)w[7] << 32) | w[6]);
    return D;
  };
  const lfp Y = lds + K * 384 + K * 2 * TR_MAX_LAYERS * 8 + K * 8;   // [K*d] normalised state of every step (MODE 1)
  const lfp Zc = Y + (MODE == 1 ? K * d * S : 0);         // [d]   running state (forward) / gradient state (backward)
  const lfp GX = Zc + d * S;                              // [ip]  gradient w.r.t. the coupling-net input
  const lfp O = GX + p.ip * S;                            // [op]  net output, then its gradient
  const lfp O2 = O + p.op * S;                            // [op]  realnvp: shift output / shift gradient
  const lfp RT = O2 + p.op * S;                           // [hp]  ResidualNets: running state t (forward) / its gradient (backward)
  const lfp RED = RT + (p.residual ? p.hp * S : 0);       // [waves][NT][16]  cross-wave scratch
  // split-f16 rows (B operands of the dense layers): net input, hidden activations (then their gradients), output gradient
  lbp sp = reinterpret_cast<lbp>(lds) + ((((RED + 16 * TR_WAVES * NT) - lds) * 4 + 15) & ~15);   // 16-byte aligned
  const TrSplit XS{sp, p.xw};
  sp += TS * XS.rs();
  // hidden layer l's rows: computed, not an array (a dynamically indexed local array lives in scratch memory, and a
  // scratch load is a vector-memory load: waiting for it drains the weight prefetch)
  const lbp hs0 = sp;
  const int hs_bytes = TS * (4 * p.hw + 16);
  auto HS = [&](int l) -> TrSplit { return TrSplit{hs0 + l * hs_bytes, p.hw}; };
  sp += p.n_hidden * hs_bytes;
  const TrSplit GOS{sp, p.ow};
  // channels between the 16-padded and the 32-padded widths are never written by an epilogue: zero the split rows once
  {
    const lbp s0 = XS.base;
    const int bytes = (int)((sp + TS * GOS.rs()) - s0);
    for (int e = 16 * threadIdx.x; e < bytes; e += 16 * 64 * TR_WAVES) *reinterpret_cast<u32x4 TR_LDS*>(s0 + e) = u32x4{0u, 0u, 0u, 0u};
  }

  const int hid_tiles = p.hp >> 4, out_tiles = p.op >> 4, in_tiles = p.ip >> 4;

  // ---- normalisation of slot s at a step (ActNorm / eval-mode BatchNorm)
  auto norm_fwd = [&](int k, int s, float v, float& logdet) -> float {
    const lcfp tp = TP + k * 256 + s;
    logdet += tp[192];
    if constexpr (KIND == GBNF_KIND_GLOW) return (v + tp[0]) * tp[64];
    else return (v - tp[0]) * tp[64] + tp[128];
  };

  struct LayerD { gptr b; long long fw, bw; int rows, cols; };
  auto layer_desc = [&](int k, int q, int l) -> LayerD {     // from the LDS table, moved to scalar registers
    const lup t = TL + ((k * 2 + q) * TR_MAX_LAYERS + l) * 8;
    unsigned w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = __builtin_amdgcn_readfirstlane(t[e]);
    LayerD D;
    D.b = (gptr)(reinterpret_cast<const float*>(p.frag) + (long long)(((unsigned long long)w[1] << 32) | w[0]));
    D.fw = (long long)(((unsigned long long)w[3] << 32) | w[2]);
    D.bw = (long long)(((unsigned long long)w[5] << 32) | w[4]);
    D.rows = (int)w[6]; D.cols = (int)w[7];
    return D;
  };
  const int nl = tr_uniform(p.steps[0].net[0].n_layers);

  // ---- coupling net forward from XS: hidden layers into HS (+ emit f32), last layer into `out` (f32; or skipped)
  auto net_forward = [&](int k, int q, int act, lfp out, float* ws_net) {
    TrSplit in = XS;
    for (int l = 0; l + 1 < nl; ++l) {
      const TrSplit Hl = HS(l);
      const bool emit = (MODE == 1) && ws_net != nullptr;    // backward sweep only: activation-side operand of dW
      // operand sub-regions are tiled: [tile of 16 samples][unit][16] -- a workgroup's rows of one operand are one
      // contiguous block, and a unit's 16 samples one 64-byte run (wgrad_kernel reads 16 units x 64 B per wave load)
      float* ws_h = emit ? ws_net + ((size_t)p.ip + (size_t)l * p.hp) * p.np + (size_t)tile0 * p.hp * 16 + i : nullptr;
      const LayerD L = layer_desc(k, q, l);
      tr_dense<NT>((gfrag)(p.frag + L.fw), L.b, in.w >> 5, hid_tiles, in, hid_tiles, lane, wave,
                   stamps, [&](int u0, int nt, f32x4 v) {
        f32x4 h;
        if (!p.residual) {
          h = tr_act4(act, v);
        } else {
          // ResidualNet: layer 0 = initial_layer, odd layers = first Linear of a block, even ones = second (+ skip).
          // Hl receives the INPUT of the next Linear: relu(t) in front of a block, relu(m) inside it, t itself in front
          // of final_layer (models/layers.py:267-273, 296-300)
          if (l & 1) {
            h = tr_act4(GBNF_ACT_RELU, v);
          } else {
            f32x4 t = v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const lfp rt = RT + (u0 + r) * S + i + 16 * nt;
              if (l > 0) t[r] += *rt;
              *rt = t[r];
            }
            h = l == nl - 2 ? t : tr_act4(GBNF_ACT_RELU, t);
          }
        }
        watch4(h);
        Hl.put4(i + 16 * nt, u0, h);
        if (emit) {
#pragma unroll
          for (int r = 0; r < 4; ++r) ws_h[((size_t)nt * p.hp + u0 + r) * 16] = h[r];
        }
      });
      in = Hl;
    }
    if (out != nullptr) {
      const LayerD L = layer_desc(k, q, nl - 1);
      tr_dense<NT>((gfrag)(p.frag + L.fw), L.b, in.w >> 5, out_tiles, in, out_tiles, lane, wave,
                   stamps, [&](int u0, int nt, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(u0 + r) * S + i + 16 * nt] = v[r];
      });
    }
  };

  // ---- coupling net backward: cur = gradient w.r.t. the net output (LDS f32, [op]); leaves d(loss)/d(net input) in GX
  auto net_backward = [&](int k, int q, int act, lcfp cur, float* ws_net, bool accumulate) {
    const int nh = p.n_hidden;
    // gradient-side operand of the last layer's weight gradient (f32 to the workspace) + its split copy for the dense chain
    {
      float* ws_d = ws_net + ((size_t)p.ip + 2 * (size_t)nh * p.hp) * p.np + (size_t)tile0 * p.op * 16 + i;
      for (int u = g; u < p.ow; u += GS) {
        TR_NT {
          const float v = u < p.op ? cur[u * S + i + 16 * nt] : 0.0f;
          if (u < p.op) ws_d[((size_t)nt * p.op + u) * 16] = v;
          watch(v);
          GOS.put1(i + 16 * nt, u, v);
        }
      }
    }
    tr_lds_barrier();
    TrSplit in = GOS;
    for (int l = nl - 1; l >= 1; --l) {
      const TrSplit Hl = HS(l - 1);                          // activations of hidden layer l-1 -> overwritten by its gradient
      float* ws_d = ws_net + ((size_t)p.ip + (size_t)nh * p.hp + (size_t)(l - 1) * p.hp) * p.np + (size_t)tile0 * p.hp * 16 + i;
      const LayerD L = layer_desc(k, q, l);                  // A = W^T: output units = cols, k = rows
      tr_dense<NT>((gfrag)(p.frag + L.bw), gptr(nullptr), in.w >> 5, hid_tiles, in, hid_tiles, lane, wave, stamps,
                   [&](int u0, int nt, f32x4 v) {
        const f32x4 h = Hl.get4(i + 16 * nt, u0);
        f32x4 gpre;
        if (!p.residual) {
          gpre = tr_dact4_mul(act, h, v);
        } else if (l == nl - 1) {
          gpre = v;                                  // d/dt behind the last block: final_layer's input is t itself
#pragma unroll
          for (int r = 0; r < 4; ++r) RT[(u0 + r) * S + i + 16 * nt] = v[r];
        } else if (!(l & 1)) {
          gpre = tr_dact4_mul(GBNF_ACT_RELU, h, v);   // through the relu between a block's two Linears
        } else {
          // through a block's entry relu, plus the skip connection: d/dt in front of the block
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const lfp rt = RT + (u0 + r) * S + i + 16 * nt;
            gpre[r] = *rt + (h[r] > 0.0f ? v[r] : 0.0f);
            *rt = gpre[r];
          }
        }
        watch4(gpre);
        Hl.put4(i + 16 * nt, u0, gpre);
#pragma unroll
        for (int r = 0; r < 4; ++r) ws_d[((size_t)nt * p.hp + u0 + r) * 16] = gpre[r];
      });
      in = Hl;
    }
    const LayerD L0 = layer_desc(k, q, 0);
    tr_dense<NT>((gfrag)(p.frag + L0.bw), gptr(nullptr), in.w >> 5, in_tiles, in, in_tiles, lane, wave, stamps,
                 [&](int u0, int nt, f32x4 v) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const lfp gx = GX + (u0 + r) * S + i + 16 * nt;
        *gx = accumulate ? *gx + v[r] : v[r];
      }
    });
  };

  // ---- x tile -> Zc (slot j = feature j), or the state a previous step-range launch left in slot layout
  if (MODE == 0 && p.state_in != nullptr) {
    for (int s = g; s < d; s += GS) TR_NT Zc[s * S + i + 16 * nt] = p.state_in[(size_t)s * p.np + row0 + i + 16 * nt];
  } else {
    for (int j = g; j < d; j += GS) TR_NT {
      const int64_t r = row0 + i + 16 * nt;
      Zc[j * S + i + 16 * nt] = r < p.n ? p.x[r * d + j] : 0.0f;
    }
  }
  tr_lds_barrier();    // (also: the tables are complete)
  stamps.mark(0);

  // =============================== forward through all steps
  float ld[NT];      // per-lane partials of log|det J| (every lane group adds its own slots / features)
  TR_NT ld[nt] = 0.0f;
  const bool have_trace = (MODE == 1) && p.trace != nullptr;
  const int kb = p.k_begin, ke = p.k_end;
  if (have_trace) {  // the forward call saved every step's normalised state: just bring this tile's columns in
    for (int e = g + kb * d; e < ke * d; e += GS) TR_NT Y[e * S + i + 16 * nt] = p.trace[(size_t)e * p.np + row0 + i + 16 * nt];
  }
  for (int k = kb; k < (have_trace ? kb : ke); ++k) {
    const StepD st = step_desc(k);
    for (int s = g; s < d; s += GS) TR_NT {
      const int ii = i + 16 * nt;
      const float y = norm_fwd(k, s, Zc[s * S + ii], ld[nt]);
      Zc[s * S + ii] = y;
      if (MODE == 1) Y[(k * d + s) * S + ii] = y;
      if (MODE == 0 && p.trace_out != nullptr) p.trace_out[((size_t)k * d + s) * p.np + row0 + ii] = y;
    }
    if (MODE == 1 && k == ke - 1) break;                   // the last step's outputs are not needed for the backward
    tr_lds_barrier();
    const lip ti = TI + k * 128;
    for (int kk = g; kk < p.xw; kk += GS) TR_NT {
      const float v = kk < st.in_f ? Zc[ti[kk] * S + i + 16 * nt] : 0.0f;
      watch(v);
      XS.put1(i + 16 * nt, kk, v);
    }
    tr_lds_barrier();
    stamps.mark(1);
    if constexpr (KIND == GBNF_KIND_GLOW) {
      net_forward(k, 0, st.act[0], O, nullptr);
      stamps.mark(-1);
      for (int j = g; j < st.out_f; j += GS) {
        const int slot = ti[32 + j];
        TR_NT {
          const int ii = i + 16 * nt;
          const float y2 = Zc[slot * S + ii];
          if (p.additive) {
            Zc[slot * S + ii] = y2 + O[j * S + ii];                                  // models/glow.py:328-329
          } else {
            // scale = sigmoid(raw + 2) and log(scale) from one exponential, as the evaluation kernels do
            // (sigmoid_logsigmoid in gbnf_flow_kernel.hip.h): e = exp(-(raw + 2)), scale = 1 / (1 + e), log = -log(1 + e)
            const float e = __builtin_amdgcn_exp2f((O[(2 * j + 1) * S + ii] + 2.0f) * -1.4426950408889634f);
            const float s1 = 1.0f + e;
            const float sc = __builtin_amdgcn_rcpf(s1);
            Zc[slot * S + ii] = (y2 + O[(2 * j) * S + ii]) * sc;                     // models/glow.py:333-336
            ld[nt] += -0.69314718055994531f * __builtin_amdgcn_logf(s1);             // log(scale), models/glow.py:338
          }
        }
      }
    } else {
      net_forward(k, 0, st.act[0], O2, nullptr);
      net_forward(k, 1, st.act[1], O, nullptr);
      for (int j = g; j < st.out_f; j += GS) {
        const int slot = ti[32 + j];
        TR_NT {
          const int ii = i + 16 * nt;
          const float scale = O[j * S + ii];
          Zc[slot * S + ii] = O2[j * S + ii] + Zc[slot * S + ii] * __expf(scale);     // models/transformations.py:575
          ld[nt] += scale;                                                           // models/transformations.py:577
        }
      }
    }
    tr_lds_barrier();
    stamps.mark(3);
  }

  if constexpr (MODE == 0) {
    TR_NT {
      ld[nt] += __shfl_xor(ld[nt], 16);
      ld[nt] += __shfl_xor(ld[nt], 32);
      if ((lane >> 4) == 0) RED[(wave * NT + nt) * 16 + i] = ld[nt];        // fold the waves' partial sums
    }
    tr_lds_barrier();
    TR_NT {
      const int ii = i + 16 * nt;
      const int64_t r = row0 + ii;
      if (r < p.n) {
        if (p.ldj_out != nullptr && g == 0) {
          float t = p.ldj_accumulate ? p.ldj_out[r] : 0.0f;
          for (int w = 0; w < TR_WAVES; ++w) t += RED[(w * NT + nt) * 16 + i];
          p.ldj_out[r] = t;
        }
        if (p.z_out != nullptr && ke == K)
          for (int j = g; j < d; j += GS) p.z_out[r * d + j] = Zc[p.tail[j] * S + ii];
      }
      if (p.state_out != nullptr)
        for (int s = g; s < d; s += GS) p.state_out[(size_t)s * p.np + r] = Zc[s * S + ii];
    }
    stamps.mark(4);
    report();
#ifdef GBNF_TRAIN_STAMPS
    if (p.dbg != nullptr && threadIdx.x == 0)
      for (int q = 0; q < 8; ++q) p.dbg[(size_t)blockIdx.x * 8 + q] = stamps.acc[q];
#endif
    return;
  } else {
    // =============================== backward (on alpha * the upstream gradient, see tr_grad_scale)
    tr_lds_barrier();
    float alpha = 1.0f, inv_alpha = 1.0f;
    if (p.gmax != nullptr) tr_grad_scale(__builtin_amdgcn_readfirstlane(*p.gmax), alpha, inv_alpha);
    float gl[NT];
    TR_NT {
      const int64_t r = row0 + i + 16 * nt;
      gl[nt] = (r < p.n && p.g_ldj != nullptr) ? p.g_ldj[r] * alpha : 0.0f;
    }
    if (p.gstate_in != nullptr) {
      for (int s = g; s < d; s += GS) TR_NT Zc[s * S + i + 16 * nt] = p.gstate_in[(size_t)s * p.np + row0 + i + 16 * nt];
    } else {
      for (int j = g; j < d; j += GS) TR_NT {
        const int64_t r = row0 + i + 16 * nt;
        Zc[p.tail[j] * S + i + 16 * nt] = (r < p.n && p.g_z != nullptr) ? p.g_z[r * d + j] * alpha : 0.0f;
      }
    }
    tr_lds_barrier();
    const lfp G = Zc;
    const int nnets = (KIND == GBNF_KIND_GLOW) ? 1 : 2;

    // normalisation backward for one slot and one sample: returns d(loss)/d(pre-norm value) and adds the sample's terms
    // of the two parameter gradients to ga / gb; norm_commit folds them over the workgroup's samples of that slot
    auto norm_bwd = [&](const StepD& st, int k, int s, float gy, float y, float gldj, float& ga, float& gb) -> float {
      const lcfp tp = TP + k * 256 + s;
      const float gx = gy * tp[64];
      if constexpr (KIND == GBNF_KIND_GLOW) {
        ga += gx;                   // d/d bias
        gb += gy * y + gldj;        // d/d logs: y = (x + bias) e^logs, and logdet += logs for every sample
      } else {
        if (!st.has_norm) return gy;
        ga += gy * (y - tp[128]) + gldj;   // d/d log_gamma
        gb += gy;                          // d/d beta
      }
      return gx;
    };
    auto norm_commit = [&](const StepD& st, int k, int s, float ga, float gb) {
      if (KIND != GBNF_KIND_GLOW && !st.has_norm) return;
      const int f = TI[k * 128 + 64 + s];
      ga = tr_group_sum(ga);
      gb = tr_group_sum(gb);
      if (i == 0) {
        atomicAdd(p.grads + st.g_na + f, ga * inv_alpha);
        atomicAdd(p.grads + st.g_nb + f, gb * inv_alpha);
      }
    };

    for (int k = ke - 1; k >= kb; --k) {
      const StepD st = step_desc(k);
      const lcfp Yk = Y + k * d * S;
      float* ws_step = p.ws + (size_t)k * nnets * p.net_rows * p.np;
      const lip ti = TI + k * 128;
      // net input (also the activation-side operand of the first layer's weight gradient)
      for (int kk = g; kk < p.xw; kk += GS) TR_NT {
        const int ii = i + 16 * nt;
        const float v = kk < st.in_f ? Yk[ti[kk] * S + ii] : 0.0f;
        watch(v);
        XS.put1(ii, kk, v);
        if (kk < p.ip)
          for (int q = 0; q < nnets; ++q)
            (ws_step + (size_t)q * p.net_rows * p.np)[((size_t)(tile0 + nt) * p.ip + kk) * 16 + i] = v;
      }
      tr_lds_barrier();
      if constexpr (KIND == GBNF_KIND_GLOW) {
        net_forward(k, 0, st.act[0], O, ws_step);
        for (int j = g; j < st.out_f; j += GS) {
          const int slot = ti[32 + j];
          float ga = 0.0f, gb = 0.0f;
          TR_NT {
            const int ii = i + 16 * nt;
            const float y2 = Yk[slot * S + ii], g2 = G[slot * S + ii];
            float gy;
            if (p.additive) {
              gy = g2;
              O[j * S + ii] = g2;
            } else {
              const float shift = O[(2 * j) * S + ii];
              const float e = __expf(-(O[(2 * j + 1) * S + ii] + 2.0f));
              const float sc = 1.0f / (1.0f + e);
              const float omsc = e < 1e30f ? e * sc : 1.0f;               // 1 - scale
              gy = g2 * sc;
              O[(2 * j) * S + ii] = gy;                                     // d/d shift
              O[(2 * j + 1) * S + ii] = (g2 * (y2 + shift) * sc + gl[nt]) * omsc;   // d/d raw: z2' = (y2+shift) s, ld += log s
            }
            G[slot * S + ii] = norm_bwd(st, k, slot, gy, y2, gl[nt], ga, gb);
          }
          norm_commit(st, k, slot, ga, gb);
        }
        tr_lds_barrier();
        net_backward(k, 0, st.act[0], O, ws_step, false);
      } else {
        float* ws_t = ws_step;
        float* ws_s = ws_step + (size_t)p.net_rows * p.np;
        net_forward(k, 1, st.act[1], O, ws_s);                                 // log-scale net
        for (int u = g; u < p.op; u += GS) TR_NT O2[u * S + i + 16 * nt] = 0.0f;
        tr_lds_barrier();
        for (int j = g; j < st.out_f; j += GS) {
          const int slot = ti[32 + j];
          float ga = 0.0f, gb = 0.0f;
          TR_NT {
            const int ii = i + 16 * nt;
            const float y2 = Yk[slot * S + ii], g2 = G[slot * S + ii];
            const float es = __expf(O[j * S + ii]);
            O2[j * S + ii] = g2;                                            // d/d shift
            O[j * S + ii] = g2 * y2 * es + gl[nt];                          // d/d scale: z2' = shift + y2 e^scale, ld += scale
            G[slot * S + ii] = norm_bwd(st, k, slot, g2 * es, y2, gl[nt], ga, gb);
          }
          norm_commit(st, k, slot, ga, gb);
        }
        tr_lds_barrier();
        net_backward(k, 1, st.act[1], O, ws_s, false);
        net_forward(k, 0, st.act[0], nullptr, ws_t);                           // shift net: only its activations are needed
        net_backward(k, 0, st.act[0], O2, ws_t, true);
      }
      for (int kk = g; kk < st.in_f; kk += GS) {
        const int slot = ti[kk];
        float ga = 0.0f, gb = 0.0f;
        TR_NT {
          const int ii = i + 16 * nt;
          G[slot * S + ii] = norm_bwd(st, k, slot, G[slot * S + ii] + GX[kk * S + ii], Yk[slot * S + ii], gl[nt], ga, gb);
        }
        norm_commit(st, k, slot, ga, gb);
      }
      tr_lds_barrier();
    }
    TR_NT {
      const int ii = i + 16 * nt;
      const int64_t r = row0 + ii;
      if (p.gstate_out != nullptr) {
        for (int s = g; s < d; s += GS) p.gstate_out[(size_t)s * p.np + r] = G[s * S + ii];
      } else if (r < p.n && p.g_x != nullptr) {
        for (int j = g; j < d; j += GS) p.g_x[r * d + j] = G[j * S + ii] * inv_alpha;
      }
    }
    report();
  }
#undef TR_NT
}

// ---------------------------------------------------------------------------------------------------------------
// Weight / bias gradients: C (M x N, row-major) += D (M rows of np samples) . A (N rows of np samples)^T, bias += row sums of D.
// Round 4: a workgroup of 8 waves owns a block of bm x bn of C (bm, bn = 64 | 128 | 256: a 224 x 224 layer is ONE block) over one
// chunk of samples, and the operand rows of a k-step are staged once per workgroup in LDS, split into (hi, mid) fp16 operands on
// the way.  Why one block: the kernel runs at the rate of its operand loads (the loads alone, nothing else in the loop: 330 of
// 365 us at N = 65536 with round 3's 128 x 128 blocks, which read every row of the 224 x 224 layer twice -- from memory: the two
// readers of a row do not meet in an L2, 60 streaming workgroups per XCD turn over its 4 MB in two k-steps).  Rounds 1-3: every
// wave read its own 64 + 64 rows (each slab four times over), then 4-wave workgroups of 128 x 128 with f32 rows in LDS and every
// wave splitting its own fragments.  Branch-free: operand rows past the padded matrix belong to the next workspace region (or
// the slack rows behind the last one) and only feed output rows / columns that are never stored.
// ---------------------------------------------------------------------------------------------------------------
struct WgProblem {
  int64_t d_row, a_row;   // first row (of np floats) of the two operands' sub-regions in the workspace
  int64_t c_off, b_off;   // float offsets of dW / db in the flat gradient buffer
  int M, N, blk_begin, nb;
  int d_rows, a_rows;     // rows of the two sub-regions (tiled as [tile][row][16 samples])
  int bm, bn;             // the block: 64 | 128 | 256 rows of D, of A
  int wm, wn;             // the workgroup's 8 waves: wm x wn, every wave (bm / wm) x (bn / wn) of the block
};
constexpr int WG_ROWS_MAX = 512;       // operand rows per k-step: bm + bn
#ifndef GBNF_WG_ABL
#define GBNF_WG_ABL 0
#endif
constexpr int WG_THREADS = 512;        // 8 waves, two per SIMD: 256 registers each
constexpr size_t WG_LDS_BYTES = (size_t)2 * WG_ROWS_MAX * 8 * 16;     // two k-steps of (hi, mid) rows: 128 KB

// hi = f16(x) toward zero, mid = f16(x - hi) for 8 consecutive samples of one operand row -> one MFMA operand each.
// No clamp: toward-zero conversion saturates at the largest finite fp16, so a value beyond the fp16 range degrades to
// a finite wrong number (<= 131008) instead of an infinity -- the forward path saturates such values anyway.
// 16 VALU instructions: the residual x - hi is ONE v_fma_mix_f32 per value (the f16 half is widened inside the instruction;
// exact, like the v_cvt_f32_f16 + v_sub_f32 pair LLVM writes for `x - (float)hi`).  (v_fma_mixlo/hi_f16 would convert and pack
// the residuals too, 8 instructions -- but round to nearest: an out-of-range value's residual becomes an infinity.)
__device__ __forceinline__ void wg_split8(f32x4 lo, f32x4 hi4, u32x4& h, u32x4& m) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4& v = q < 2 ? lo : hi4;
    const int e = 2 * (q & 1);
    const unsigned hw = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[e], v[e + 1]));
    h[q] = hw;
    // IN PLACE (an asm output must never be a fresh register: gbnf_flow_kernel_hx3.hip.h, split_pair_f16)
    float r0 = v[e], r1 = v[e + 1];
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(r0) : "v"(hw));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r1) : "v"(hw));
    m[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
  }
}

// The contraction runs on the f16 pipe with split operands (section 4.1: three v_mfma_f32_16x16x32_f16 per product, f32
// accumulation): one k = 32 step = 32 samples = two 16-sample tiles of the workspace.  A thread carries "double pieces": samples
// 4 q .. 4 q + 3 of BOTH tiles of one row (q = tid & 3; row tid / 4 of a 128-row slab) -- two 16-byte global loads, each of
// which a wave issues over 1 KB of contiguous memory (16 rows x 64 bytes of one tile).  The 8 values are k group q of the step:
// the contraction does not care in which order the 32 samples of a k-step are numbered as long as D and A number them alike,
// and both go through this staging.  They are split into one hi and one mid MFMA operand and written to LDS as 16-byte slots:
// row r owns 8 slots (128 bytes = one pass over the banks), operand (q, hi | mid) sits in slot (2 q + mid) ^ (r & 7) -- the 8 rows
// that 8 neighbouring lanes read at once hit 8 different slots, and so do the 2 rows x 4 groups they write at once.  A 64-row
// operand has half a piece per thread: the upper 4 waves carry the lower waves' pieces again (same loads, same LDS slots, same
// values) so that every wave issues the same loads -- the compiler's s_waitcnt vmcnt(N) in front of a split then names exactly
// the loads of the OTHER register set; with a conditional fetch anywhere it has to assume the worst and waits for everything
// in flight.  The bias gradients (row sums of D) are summed by the staging threads from the f32 values they hold anyway.
template <int ND, int NA, int TM, int TN, bool TWO>
__device__ __forceinline__ void wgrad_block(const WgProblem& P, const float* __restrict__ ws, float* __restrict__ grads, int64_t np,
                                            int64_t s_begin, int64_t s_end, int m0, int n0, float inv_alpha, u32x4* wg_stage_raw) {
  typedef const f32x4 __attribute__((address_space(1)))* gv4;
  constexpr int NPC = ND + NA;
  u32x4 (*stage)[WG_ROWS_MAX * 8] = reinterpret_cast<u32x4 (*)[WG_ROWS_MAX * 8]>(wg_stage_raw);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wmi = wave / P.wn, wni = wave - wmi * P.wn;
  // (the per-thread offsets are recomputed where they are used, from a copy of tid the compiler cannot see through: kept in
  // registers across the loop they are the values that get spilled -- and a scratch reload inside the loop is a vmcnt(0) wait,
  // i.e. the end of the prefetch)
  // (and tid itself comes from the wave number, a scalar, and the lane count: not a register that lives across the loop either)
  auto opaque_tid = [&]() {
    int t = (wave << 6) + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(t));
    return t;
  };
  // piece j < ND: D rows 128 j + t / 4 (a 64-row operand: (t / 4) & 63, the upper waves repeat the lower ones'); then the A pieces
  auto piece_row = [&](int t, int j) {
    const bool isd = j < ND;
    const int jj = isd ? j : j - ND;
    const int rows = isd ? P.bm : P.bn;
    return 128 * jj + ((t >> 2) & (rows < 128 ? 63 : 127));
  };
  auto piece_src = [&](int t, int j, int64_t s) -> gv4 {
    const int q = t & 3;
    const bool isd = j < ND;
    const int R = isd ? P.d_rows : P.a_rows;
    const int64_t base = (isd ? P.d_row : P.a_row) * np + (int64_t)(isd ? m0 : n0) * 16 + s * R;      // (scalar)
    return (gv4)(ws + base + (unsigned)(piece_row(t, j) * 16 + q * 4));
  };
  auto piece_tile = [&](int j) { return (j < ND ? P.d_rows : P.a_rows) * 4; };     // f32x4 steps from tile 0 to tile 1 of the k-step
  auto piece_slot = [&](int t, int j) {                   // the hi operand's LDS slot (mid: ^ 1)
    const int r = (j < ND ? 0 : P.bm) + piece_row(t, j);
    return r * 8 + ((2 * (t & 3)) ^ (r & 7));
  };
  f32x4 pre0[NPC][2], pre1[TWO ? NPC : 1][2];
  float bs[ND];               // row sums of this thread's D pieces (the bias gradient)
#pragma unroll
  for (int j = 0; j < ND; ++j) bs[j] = 0.0f;
  auto fetch = [&](auto& pre, int64_t s) {
    const int t = opaque_tid();
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const gv4 src = piece_src(t, j, s);
      pre[j][0] = src[0];
      pre[j][1] = src[piece_tile(j)];
    }
  };
  auto stash = [&](const auto& pre, int b, bool count) {
    const int t = opaque_tid();
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
#if GBNF_WG_ABL == 5                                  // (diagnostic: the loads alone -- no split, no LDS traffic, no MFMA)
      bs[0] += pre[j][0][0] + pre[j][1][3];
      continue;
#endif
      u32x4 h, m;
      wg_split8(pre[j][0], pre[j][1], h, m);
      const int sl = piece_slot(t, j);
      stage[b][sl] = h;
      stage[b][sl ^ 1] = m;
      if (j < ND && count) {                               // (uniform: the blocks that own the bias, real k-steps)
        const f32x4 u = pre[j][0] + pre[j][1];
        bs[j] += (u[0] + u[1]) + (u[2] + u[3]);
      }
    }
  };
  f32x4 acc[TM][TN];
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int x = 0; x < TM; ++x)
#pragma unroll
    for (int y = 0; y < TN; ++y) acc[x][y] = zero;
  auto compute = [&](int b) {
    // LDS slots of this lane's fragments: D tile x -> local row 16 (TM wmi + x) + i; A tile y -> bm + 16 (TN wni + y) + i
    // (the XOR term of a row 16 t + i is i & 7: tile t + 1 sits 128 slots behind tile t)
    const int tl = opaque_tid(), li = tl & 15, lg = (tl >> 4) & 3;
    const int dix0 = (16 * TM * wmi + li) * 8 + ((2 * lg) ^ (li & 7));
    const int aix0 = (P.bm + 16 * TN * wni + li) * 8 + ((2 * lg) ^ (li & 7));
    // (at most two A tiles at a time, the D fragments one tile row at a time: 24 operand registers live -- the accumulators
    // and the two prefetch sets need the room at two waves per SIMD)
    constexpr int YS = TN < 2 ? TN : 2;
#pragma unroll
    for (int yh = 0; yh < TN / YS; ++yh) {
      u32x4 ah[YS], am[YS];
#pragma unroll
      for (int y = 0; y < YS; ++y) {
        ah[y] = stage[b][aix0 + 128 * (YS * yh + y)];
        am[y] = stage[b][(aix0 ^ 1) + 128 * (YS * yh + y)];
      }
      // (the next tile row's D fragments are read under this one's MFMAs, and no further ahead: left to itself the scheduler
      // hoists all TM rows' reads to the top -- 64 registers for the 128 x 64 wave tile, which then spills its prefetch set)
      u32x4 dh[2], dm[2];
      dh[0] = stage[b][dix0];
      dm[0] = stage[b][dix0 ^ 1];
#pragma unroll
      for (int x = 0; x < TM; ++x) {
        const int c = x & 1;
        if (x + 1 < TM) {
          dh[c ^ 1] = stage[b][dix0 + 128 * (x + 1)];
          dm[c ^ 1] = stage[b][(dix0 ^ 1) + 128 * (x + 1)];
        }
#pragma unroll
        for (int y = 0; y < YS; ++y) acc[x][YS * yh + y] = tr_mfma16(dm[c], ah[y], acc[x][YS * yh + y]);
#pragma unroll
        for (int y = 0; y < YS; ++y) acc[x][YS * yh + y] = tr_mfma16(dh[c], am[y], acc[x][YS * yh + y]);
#pragma unroll
        for (int y = 0; y < YS; ++y) acc[x][YS * yh + y] = tr_mfma16(dh[c], ah[y], acc[x][YS * yh + y]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // Two k-steps of loads in flight (two register sets, used in turn; one set for the 128 x 64 wave tile, whose 128 accumulator
  // registers leave no room for the second).  Every path through the loop issues the same loads in the same order: past the end
  // of the chunk its last k-step again, staged into a buffer nobody computes on.
  const int64_t s_last = s_end - 32;
  auto at = [&](int64_t s) { return s < s_end ? s : s_last; };
  const bool own_bias = n0 == 0;
  if constexpr (TWO) {
    fetch(pre0, s_begin);
    fetch(pre1, at(s_begin + 32));
    stash(pre0, 0, own_bias);
    fetch(pre0, at(s_begin + 64));
    __syncthreads();
    // at the top: LDS buffer 0 holds k-step s; pre1 = k-step s + 32 and pre0 = k-step s + 64 are in flight
    int64_t s = s_begin;
    for (; s + 32 < s_end; s += 64) {
#if GBNF_WG_ABL != 2 && GBNF_WG_ABL != 5            // (diagnostic builds: 1 = no atomics, 2 = no MFMA loop, 5 = loads only)
      compute(0);
#endif
      stash(pre1, 1, own_bias);
      fetch(pre1, at(s + 96));
      __syncthreads();
#if GBNF_WG_ABL != 2 && GBNF_WG_ABL != 5
      compute(1);
#endif
      stash(pre0, 0, own_bias && s + 64 < s_end);
      fetch(pre0, at(s + 128));
      __syncthreads();
    }
#if GBNF_WG_ABL != 2 && GBNF_WG_ABL != 5
    if (s < s_end) compute(0);                          // an odd number of k-steps: the last one is in buffer 0
#endif
  } else {
    fetch(pre0, s_begin);
    stash(pre0, 0, own_bias);
    fetch(pre0, at(s_begin + 32));
    __syncthreads();
    int b = 0;
    for (int64_t s = s_begin; s < s_end; s += 32, b ^= 1) {
#if GBNF_WG_ABL != 2 && GBNF_WG_ABL != 5
      compute(b);
#endif
      stash(pre0, b ^ 1, own_bias && s + 32 < s_end);
      fetch(pre0, at(s + 64));
      __syncthreads();
    }
  }
#pragma unroll
  for (int x = 0; x < TM; ++x) {
    if constexpr (TN >= 2) {
#pragma unroll
      for (int y = 0; y < TN; y += 2) tr_mfma_drain(acc[x][y], acc[x][y + 1]);
    } else {
      tr_mfma_drain(acc[x][0], acc[x][0]);
    }
  }
  if (n0 == 0) {     // db[m] = sum over samples of D[m][.]: the 4 threads of a row (k groups 0..3) are neighbours
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      float v = bs[j];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      const int te = opaque_tid();
      const int rl = 128 * j + (te >> 2);                 // (a 64-row operand: the upper waves hold copies)
      const int m = m0 + rl;
      if ((te & 3) == 0 && rl < P.bm && m < P.M) atomicAdd(grads + P.b_off + m, v * inv_alpha);
    }
  }
#if GBNF_WG_ABL == 1
  if (np > 0) return;
#endif
  float* C = grads + P.c_off;
  const int lane = opaque_tid() & 63, i = lane & 15, g = lane >> 4;
  const int mw = m0 + 16 * TM * wmi, nw = n0 + 16 * TN * wni;
#pragma unroll
  for (int x = 0; x < TM; ++x)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = mw + 16 * x + 4 * g + r;
#pragma unroll
      for (int y = 0; y < TN; ++y) {
        const int n = nw + 16 * y + i;
        if (m < P.M && n < P.N) atomicAdd(C + (size_t)m * P.N + n, acc[x][y][r] * inv_alpha);
      }
    }
}

__global__ void __launch_bounds__(WG_THREADS) wgrad_kernel(const WgProblem* __restrict__ probs, int n_probs,
                                                    const float* __restrict__ ws, float* __restrict__ grads, int64_t np, int chunk,
                                                    const unsigned* __restrict__ gmax, int n_blocks, const LiveReduce red) {
  extern __shared__ __attribute__((aligned(16))) u32x4 wg_stage_raw[];      // WG_LDS_BYTES (dynamic: > 64 KB)
  // Launch index L = y G + x is dealt out in order (round-robin over the XCDs, a workgroup to every CU as it falls free); work
  // item L is chunk L mod Y of block L / Y, and the host numbers the blocks heaviest first (a 256 x 256 block streams 64 KB per
  // k-step, a 256 x 64 one 40 KB: a CU pulls ~24 GB/s whatever it does, so the time of a block is its bytes): the light blocks
  // fill in behind the heavy ones instead of the heavy ones finishing alone.
  int bx, by;
  {
    const int Y = gridDim.y, L = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
    bx = L / Y;
    by = L - bx * Y;
  }
  // blocks behind the dW blocks (first sample chunk only): the fixed-order sum of the backward kernel's per-workgroup
  // ActNorm / BatchNorm gradient partials, grads[goff[kw] + j] += sum_b partials[b][kw][j]  (kw = step * 2 + which)
  if (bx >= n_blocks) {
    if (by != 0 || red.partials == nullptr || bx - n_blocks >= 2 * red.K) return;
    float (*rsum)[64] = reinterpret_cast<float (*)[64]>(wg_stage_raw);       // (no static LDS)
    const int kw = bx - n_blocks, j = threadIdx.x & 63, part = threadIdx.x >> 6;      // 8 parts
    if ((red.skip_steps >> (kw >> 1)) & 1u) return;           // added already (a BatchNorm step of a batch-statistics sweep)
    const float* src = red.partials + kw * 64 + j;
    const int64_t stride = (int64_t)red.K * 128;
    // (16 loads in flight per thread: with 4 the 512 partials of an N = 65536 sweep were 16 dependent round trips, 11.8 us)
    float a[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = 0.0f;
    int b = part;
    for (; b + 8 * 15 < red.n_wg; b += 8 * 16) {
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q] += src[(int64_t)(b + 8 * q) * stride];
    }
    for (; b < red.n_wg; b += 8) a[0] += src[(int64_t)b * stride];
    rsum[part][j] = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) +
                    (((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15])));
    __syncthreads();
    if (part == 0 && j < red.d)
      grads[red.goff[kw] + j] += ((rsum[0][j] + rsum[1][j]) + (rsum[2][j] + rsum[3][j])) + ((rsum[4][j] + rsum[5][j]) + (rsum[6][j] + rsum[7][j]));
    return;
  }
  float alpha = 1.0f, inv_alpha = 1.0f;          // the gradient-side operands were emitted on the scaled gradient
  if (gmax != nullptr) tr_grad_scale(__builtin_amdgcn_readfirstlane(*gmax), alpha, inv_alpha);
  int pi = 0;
  while (pi + 1 < n_probs && bx >= probs[pi + 1].blk_begin) ++pi;
  const WgProblem P = probs[pi];
  const int blk = bx - P.blk_begin;
  const int m0 = (blk / P.nb) * P.bm, n0 = (blk % P.nb) * P.bn;
  const int64_t s_begin = (int64_t)by * chunk;       // `chunk` samples per block (a multiple of 32)
  const int64_t s_end = (s_begin + chunk < np) ? s_begin + chunk : np;
  // pieces per thread (a 64-row operand: one, repeated by the upper waves) and the wave's tile of the block, see wg_shape()
#define WG_CASE(BM, BN, ND, NA, TM, TN) \
  if (P.bm == BM && P.bn == BN) return wgrad_block<ND, NA, TM, TN, (TM * TN < 32)>(P, ws, grads, np, s_begin, s_end, m0, n0, inv_alpha, wg_stage_raw)
  WG_CASE(256, 256, 2, 2, 8, 4);
  WG_CASE(256, 128, 2, 1, 4, 4);
  WG_CASE(256, 64, 2, 1, 2, 4);
  WG_CASE(128, 256, 1, 2, 4, 4);
  WG_CASE(64, 256, 1, 2, 4, 2);
  WG_CASE(128, 128, 1, 1, 4, 2);
  WG_CASE(128, 64, 1, 1, 2, 2);
  WG_CASE(64, 128, 1, 1, 2, 2);
  WG_CASE(64, 64, 1, 1, 2, 1);
#undef WG_CASE
}
// the block of a problem and its 8 waves (wm x wn; every wave TM x TN tiles of 16 x 16 with TM = bm / (16 wm), TN = bn / (16 wn))
// (cap: largest block edge -- 256 for big batches, where every operand row must be loaded once; 128 for small ones, where the
// operands fit the L2s and what counts is the number of workgroups)
static void wg_shape(int M, int N, WgProblem& P, int cap = 256) {
  P.bm = M <= 64 ? 64 : (M <= 128 || cap <= 128 ? 128 : 256);
  P.bn = N <= 64 ? 64 : (N <= 128 || cap <= 128 ? 128 : 256);
  struct S { int bm, bn, wm, wn; };
  static const S table[] = {{256, 256, 2, 4}, {256, 128, 4, 2}, {256, 64, 8, 1}, {128, 256, 2, 4}, {64, 256, 1, 8},
                            {128, 128, 2, 4}, {128, 64, 4, 2},  {64, 128, 2, 4}, {64, 64, 2, 4}};
  for (const S& e : table)
    if (e.bm == P.bm && e.bn == P.bn) { P.wm = e.wm; P.wn = e.wn; }
}

// ---- batch-statistics BatchNorm (models/layers.py:338-358 in train mode): the pieces that need the whole batch ----------
// (n, d) row-major <-> slot layout [d][np] (slot j = feature j at the input of step 0)
__global__ void __launch_bounds__(256) rows_to_slots_kernel(const float* __restrict__ x, float* __restrict__ st, int64_t n, int64_t np, int d) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= np) return;
  for (int j = 0; j < d; ++j) st[(size_t)j * np + s] = s < n ? x[s * d + j] : 0.0f;
}
__global__ void __launch_bounds__(256) slots_to_rows_kernel(const float* __restrict__ st, float* __restrict__ x, int64_t n, int64_t np, int d,
                                                            const unsigned* __restrict__ gmax) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  float alpha = 1.0f, inv_alpha = 1.0f;          // the gradient state travels scaled (tr_grad_scale)
  if (gmax != nullptr) tr_grad_scale(*gmax, alpha, inv_alpha);
  for (int j = 0; j < d; ++j) x[s * d + j] = st[(size_t)j * np + s] * inv_alpha;
}

__device__ __forceinline__ float tr_block_sum(float v, float* red) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// One block per slot: mean and UNBIASED variance over the n valid samples of the state (two passes, fixed order).
__global__ void __launch_bounds__(256) bn_stats_kernel(const TrStep* __restrict__ steps, int k, const float* __restrict__ st, int64_t n, int64_t np) {
  __shared__ float red[4];
  const TrStep& S = steps[k];
  const int slot = blockIdx.x, f = S.feat[slot];
  const float* col = st + (size_t)slot * np;
  float a = 0.0f;
  for (int64_t s = threadIdx.x; s < n; s += 256) a += col[s];
  const float mean = tr_block_sum(a, red) / (float)n;
  float b = 0.0f;
  for (int64_t s = threadIdx.x; s < n; s += 256) {
    const float c = col[s] - mean;
    b += c * c;
  }
  const float var = tr_block_sum(b, red) / (float)(n - 1);
  if (threadIdx.x == 0) {
    S.bmean[f] = mean;
    S.bvar[f] = var;
  }
}

// The same statistics for large batches (round 4): one block per slot reads a 65536-row column in 87 us -- four of them were 23 % of
// the HEPMASS train() step.  Three launches over (slot, row chunk) blocks, every sum in a fixed order (bit-reproducible):
//   pass 1: chunk sums -> ps[slot][NB];  pass 2: mean from them (every block adds the NB sums in the same order), chunk sums of
//   the centred squares -> pq[slot][NB];  finalize: mean, unbiased variance -> the caller's buffers.
constexpr int BN_CHUNKS = 64;
__global__ void __launch_bounds__(256) bn_stats_part_kernel(const float* __restrict__ st, int64_t n, int64_t np, int nb, int pass,
                                                            const float* __restrict__ ps, float* __restrict__ out) {
  __shared__ float red[4];
  const int slot = blockIdx.x, b = blockIdx.y;
  const float* col = st + (size_t)slot * np;
  const int64_t per = (n + nb - 1) / nb, s0 = (int64_t)b * per, s1 = s0 + per < n ? s0 + per : n;
  float mean = 0.0f;
  if (pass == 2) {
    float tot = 0.0f;
    for (int q = 0; q < nb; ++q) tot += ps[slot * nb + q];
    mean = tot / (float)n;
  }
  float a = 0.0f;
  for (int64_t s = s0 + threadIdx.x; s < s1; s += 256) {
    const float c = col[s] - mean;
    a += pass == 2 ? c * c : c;
  }
  const float tot = tr_block_sum(a, red);
  if (threadIdx.x == 0) out[slot * nb + b] = tot;
}
__global__ void __launch_bounds__(64) bn_stats_final_kernel(const TrStep* __restrict__ steps, int k, int d, int64_t n, int nb,
                                                            const float* __restrict__ ps, const float* __restrict__ pq) {
  const int slot = threadIdx.x;
  if (slot >= d) return;
  const TrStep& S = steps[k];
  float a = 0.0f, b = 0.0f;
  for (int q = 0; q < nb; ++q) { a += ps[slot * nb + q]; b += pq[slot * nb + q]; }
  const int f = S.feat[slot];
  S.bmean[f] = a / (float)n;
  S.bvar[f] = b / (float)(n - 1);
}

// Backward through the batch statistics of step k, applied to the gradient state the step's backward launch left behind
// (which treated mean / var as constants):   g_x -= (gamma / sigma) * S1 / n  +  x_hat / (sigma (n - 1)) * S2,
// S1 = sum g_y = d/d beta,  S2 = gamma sum g_y x_hat + sum g_ldj = d/d log_gamma -- both already in `grads`.
__global__ void __launch_bounds__(256) bn_bwd_fix_kernel(const TrStep* __restrict__ steps, int k, int d, const float* __restrict__ trace, const float* __restrict__ grads,
                                                         float* __restrict__ gst, int64_t n, int64_t np, const unsigned* __restrict__ gmax) {
  float alpha = 1.0f, inv_alpha = 1.0f;          // gst is the SCALED gradient state, S1 / S2 come from the (unscaled) gradient buffer
  if (gmax != nullptr) tr_grad_scale(*gmax, alpha, inv_alpha);
  const TrStep& S = steps[k];
  const int slot = blockIdx.y, f = S.feat[slot];
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  const float gamma = __expf(S.na[f]), beta = S.nb[f], sigma = sqrtf(S.bvar[f] + S.eps);
  const float s1 = grads[S.g_nb + f], s2 = grads[S.g_na + f];
  const float xhat = (trace[((size_t)k * d + slot) * np + s] - beta) / gamma;
  gst[(size_t)slot * np + s] -= alpha * (gamma / sigma * s1 / (float)n + xhat / (sigma * (float)(n - 1)) * s2);
}

// Split every Linear's LIVE f32 weight into f16x3 MFMA A fragments, both orientations, one launch: a 64-thread block
// = one fragment pair (hi, mid) of [o][c]; lane (i,g) element j = M[16 o + i][32 c + 8 g + j], zero outside the matrix.
struct PrepProblem {
  const float* W;      // (rows, cols) row-major
  int64_t off;         // u32x4 offset of this orientation's fragments
  int rows, cols;      // of W
  int trans;           // 0: M = W (out units = rows, k = cols); 1: M = W^T
  int tiles, kc;       // fragment grid: output tiles x 32-wide k chunks
  int blk_begin;
  const float* bias;   // forward orientation only: (rows,) -> zero-padded copy at float offset boff of the fragment buffer
  int64_t boff;
};

__global__ void __launch_bounds__(64) prep_kernel(const PrepProblem* __restrict__ probs, int n_probs, u32x4* __restrict__ frag) {
  int pi = 0;
  while (pi + 1 < n_probs && (int)blockIdx.x >= probs[pi + 1].blk_begin) ++pi;
  const PrepProblem P = probs[pi];
  const int blk = blockIdx.x - P.blk_begin;
  const int o = blk / P.kc, c = blk - o * P.kc;
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  const int m = 16 * o + i;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 32 * c + 8 * g + j;
    float x = 0.0f;
    if (!P.trans) {
      if (m < P.rows && k < P.cols) x = P.W[(size_t)m * P.cols + k];
    } else {
      if (m < P.cols && k < P.rows) x = P.W[(size_t)k * P.cols + m];
    }
    v[j] = x;
  }
  unsigned h[4], md[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {       // weights: hi rounded to nearest, mid = f16(x - hi)
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 hh = {(_Float16)v[2 * q], (_Float16)v[2 * q + 1]};
    const h2 mm = {(_Float16)(v[2 * q] - (float)hh[0]), (_Float16)(v[2 * q + 1] - (float)hh[1])};
    h[q] = __builtin_bit_cast(unsigned, hh);
    md[q] = __builtin_bit_cast(unsigned, mm);
  }
  u32x4* dst = frag + P.off + ((size_t)o * P.kc + c) * 128 + lane;
  dst[0] = u32x4{h[0], h[1], h[2], h[3]};
  dst[64] = u32x4{md[0], md[1], md[2], md[3]};
  if (P.bias != nullptr && c == 0 && lane < 16)
    reinterpret_cast<float*>(frag)[P.boff + m] = m < P.rows ? P.bias[m] : 0.0f;
}

}  // namespace gbnf

using namespace gbnf;

struct gbnf_trainer {
  int kind = 0, d = 0, K = 0, additive = 0, n_hidden = 0, hp = 0, ip = 0, op = 0, nnets = 1;
  int64_t net_rows = 0, grad_floats = 0;
  size_t lds_fwd[TR_MAX_NT + 1] = {0, 0, 0}, lds_bwd[TR_MAX_NT + 1] = {0, 0, 0};   // dynamic LDS bytes by tiles per workgroup
  TrStep* steps_dev = nullptr;
  int* tail_dev = nullptr;
  WgProblem* probs_dev = nullptr;
  WgProblem* probs_small_dev = nullptr;     // the same problems cut into blocks of at most 128 x 128 (small batches)
  int wg_blocks_small = 0;
  PrepProblem* prep_dev = nullptr;
  u32x4* frag_dev = nullptr;           // split-f16 weight fragments, rebuilt by prep_kernel at the start of every call
  int n_probs = 0, wg_blocks = 0, n_prep = 0, prep_blocks = 0;
  int hw = 0, xw = 0, ow = 0;
  int residual = 0;                    // ResidualNet coupling networks
  unsigned* gmax_dev = nullptr;        // bits of the largest |upstream gradient| of the current backward call (gradient scaling)
  float* bn_part_dev = nullptr;        // [2][64 slots][BN_CHUNKS]: chunk sums of the batch statistics (bn_stats_part_kernel)
  int batch_stats = 0;                 // BatchNorm on batch statistics (the reference's train() mode)
  std::vector<int> has_norm;           // per step
  std::vector<char> stats_bound;       // per step: bmean / bvar bound by the caller
  std::vector<float*> bmean_ptr, bvar_ptr;   // per step: the caller's device buffers (host copies of TrStep::bmean / bvar)
  LiveBlob* live = nullptr;            // round 3: the forward sweep on flow_kernel_hx3<TRAIN> (depth-1 TanhNet / ReLUNet), else null
  mutable int last_fwd_ranges = 0, last_bwd_ranges = 0;     // (tests) launches of the register-chained kernels by the last forward / backward call; 0 = the round-1 kernels ran
};

// tuning / test knob: GBNF_TRAIN_PATH=old keeps the round-1 kernels for every call
static bool tr_fast_path_enabled() {
  static const bool on = [] { const char* e = getenv("GBNF_TRAIN_PATH"); return !(e && !strcmp(e, "old")); }();
  return on;
}

static int ceil16(int v) { return (v + 15) / 16 * 16; }
static int64_t tr_padded(int64_t n) { return (n + 16 * TR_MAX_NT - 1) / (16 * TR_MAX_NT) * (16 * TR_MAX_NT); }   // np: whole workgroups for every NT

// 16-sample tiles per workgroup.  Two tiles halve the weight-fragment traffic per sample (the bound of these kernels,
// DESIGN.md section 4.7) at twice the LDS per workgroup: worth it once there are enough workgroups to cover the chip.
static int pick_nt(const gbnf_trainer* t, int64_t np, int mode) {
  static const int forced = [] { const char* e = getenv("GBNF_TRAIN_NT"); return e ? atoi(e) : 0; }();
  const size_t* lds = mode == 0 ? t->lds_fwd : t->lds_bwd;
  // measured (MINIBOONE Glow): one tile per workgroup is the faster of the two up to 1024 samples (a lone workgroup's
  // pass through the layers is the whole latency there), two tiles from 2048 on even while they leave CUs idle
  int nt = np / 16 > 96 ? TR_MAX_NT : 1;
  if (forced >= 1 && forced <= TR_MAX_NT) nt = forced;
  while (nt > 1 && lds[nt] > (size_t)TR_LDS_BYTES) --nt;
  return nt;
}

template <int MODE>
static void launch_train(const gbnf_trainer* t, const TrainLaunch& p, hipStream_t s) {
  const int nt = pick_nt(t, p.np, MODE);
  const dim3 grid((unsigned)(p.np / (16 * nt))), blk(64 * TR_WAVES);
  const size_t lds = MODE == 0 ? t->lds_fwd[nt] : t->lds_bwd[nt];
  const bool glow = t->kind == GBNF_KIND_GLOW;
  if (nt == 2) {
    if (glow) hipLaunchKernelGGL((train_kernel<GBNF_KIND_GLOW, MODE, 2>), grid, blk, lds, s, p);
    else hipLaunchKernelGGL((train_kernel<GBNF_KIND_REALNVP, MODE, 2>), grid, blk, lds, s, p);
  } else {
    if (glow) hipLaunchKernelGGL((train_kernel<GBNF_KIND_GLOW, MODE, 1>), grid, blk, lds, s, p);
    else hipLaunchKernelGGL((train_kernel<GBNF_KIND_REALNVP, MODE, 1>), grid, blk, lds, s, p);
  }
}


extern "C" {

int gbnf_trainer_create(const gbnf_flow_desc* desc, gbnf_trainer** out) {
  if (out == nullptr) return fail(GBNF_ERR_INVALID, "gbnf_trainer_create: out is null");
  *out = nullptr;
  // Validate the descriptor with the evaluation path's own checks (shape rules are identical); that call dereferences
  // only the HOST fields (perm_indices, sizes), never the parameter arrays, when asked to validate only.
  int rc = gbnf_flow_validate(desc);
  if (rc) return rc;
  const int d = desc->d, K = desc->n_steps;
  const bool glow = desc->kind == GBNF_KIND_GLOW;
  const bool additive = glow && desc->coupling == GBNF_COUPLING_ADDITIVE;
  const int d1 = d / 2, d2 = d - d1;
  const gbnf_net& n0 = glow ? desc->glow_steps[0].block : desc->realnvp_steps[0].t_net;
  const int h = n0.layers[0].out_features, nl = n0.n_layers;
  const bool residual = n0.activation == GBNF_ACT_RESIDUAL_RELU;
  if (nl > TR_MAX_LAYERS) return fail(GBNF_ERR_UNSUPPORTED, "gbnf_trainer_create: %d Linear layers per net > %d", nl, TR_MAX_LAYERS);
  if (d2 > TR_MAX_IN) return fail(GBNF_ERR_UNSUPPORTED, "gbnf_trainer_create: half width %d > %d", d2, TR_MAX_IN);

  gbnf_trainer* t = new gbnf_trainer();
  t->kind = desc->kind; t->d = d; t->K = K; t->additive = additive ? 1 : 0;
  t->nnets = glow ? 1 : 2;
  t->residual = residual ? 1 : 0;
  t->n_hidden = nl - 1;
  t->hp = ceil16(h);
  // (round 5) a width no TRAIN kernel variant is compiled for trains on the next wider one: the operand rows follow the VARIANT's
  // hidden tiles (the extra units have zero weights: their activations, gradients and operand rows are zeros)
  const bool chained_shape = (residual ? (nl == 4 || nl == 6) : (nl >= 2 && nl <= 4)) && tr_fast_path_enabled();
  int hp_wide = 0;
  if (chained_shape) {
    const int rows = live_blob_train_rows(desc);
    if (rows > t->hp && rows <= TR_MAX_HIDDEN) hp_wide = rows;
  }
  t->ip = ceil16(d2);
  t->op = ceil16(glow && !additive ? 2 * d2 : d2);
  // operand rows per (step, net): net input | hidden activations | hidden gradients | output gradient | (round 3) the net's
  // output as the forward sweep saved it for the backward sweep
  t->net_rows = (int64_t)t->ip + 2LL * t->n_hidden * t->hp + 2LL * t->op;
  t->xw = (d2 + 31) / 32 * 32; t->ow = ((glow && !additive ? 2 * d2 : d2) + 31) / 32 * 32;
  if (t->hp > TR_MAX_HIDDEN) {
    delete t;
    return fail(GBNF_ERR_UNSUPPORTED, "gbnf_trainer_create: hidden width %d > %d", h, TR_MAX_HIDDEN);
  }
  const size_t tables = (size_t)K * (384 + 2 * TR_MAX_LAYERS * 8 + 8) * 4;
  auto size_lds = [&](int hp) {        // the per-step kernels' LDS at a padded hidden width
    t->hp = hp;
    t->hw = (hp + 31) / 32 * 32;
    const size_t common = (size_t)d + (size_t)t->ip + 2 * (size_t)t->op + (residual ? t->hp : 0);   // f32 rows: state, GX, O, O2, (RT)
    for (int nt = 1; nt <= TR_MAX_NT; ++nt) {
      const size_t S = 16 * nt + 1;
      const size_t split = 16 * nt * ((size_t)(4 * t->xw + 16) + (size_t)t->n_hidden * (4 * t->hw + 16) + (size_t)(4 * t->ow + 16));
      t->lds_fwd[nt] = tables + common * S * 4 + 64 * TR_WAVES * nt + 16 + split;
      t->lds_bwd[nt] = tables + (common + (size_t)K * d) * S * 4 + 64 * TR_WAVES * nt + 16 + split;
    }
  };
  // (the variant's wider rows only if the per-step kernels -- untraced forward calls, the fall-backs -- still fit with them)
  const int hp_own = t->hp;
  if (hp_wide) size_lds(hp_wide);
  if (!hp_wide || t->lds_bwd[1] > (size_t)TR_LDS_BYTES) size_lds(hp_own);
  t->net_rows = (int64_t)t->ip + 2LL * t->n_hidden * t->hp + 2LL * t->op;
  if (t->lds_bwd[1] > (size_t)TR_LDS_BYTES) {
    const size_t need = t->lds_bwd[1];
    delete t;
    return fail(GBNF_ERR_UNSUPPORTED, "gbnf_trainer_create: K*d = %d needs %zu bytes of LDS per workgroup (> %d)", K * d, need,
                TR_LDS_BYTES);
  }

  std::vector<TrStep> steps(K);
  std::vector<WgProblem> probs;
  std::vector<PrepProblem> preps;
  int64_t frag_off = 0;
  int prep_blocks = 0;
  std::vector<int> sigma(d), prev(d);
  for (int j = 0; j < d; ++j) sigma[j] = j;
  int64_t goff = 0;
  int blocks = 0;
  for (int s = 0; s < K; ++s) {
    TrStep& st = steps[s];
    std::memset(&st, 0, sizeof(st));
    prev = sigma;
    int in_f, out_f;
    st.g_na = goff; goff += d;
    st.g_nb = goff; goff += d;
    const gbnf_net* nets[2] = {nullptr, nullptr};
    if (glow) {
      const gbnf_glow_step& g = desc->glow_steps[s];
      st.has_norm = 1;
      st.na = g.actnorm_bias; st.nb = g.actnorm_logs;
      for (int j = 0; j < d; ++j) {
        const int m = (int)g.perm_indices[j];       // z = actnorm(x)[:, perm]
        sigma[j] = prev[m];
        st.feat[sigma[j]] = m;
      }
      in_f = d1; out_f = d2;
      nets[0] = &g.block;
    } else {
      const gbnf_realnvp_step& r = desc->realnvp_steps[s];
      st.has_norm = r.has_batch_norm ? 1 : 0;
      st.na = r.bn_log_gamma; st.nb = r.bn_beta; st.mean = r.bn_running_mean; st.var = r.bn_running_var;
      st.eps = r.bn_eps;
      for (int j = 0; j < d; ++j) {
        const int m = r.flipped ? ((j < d2) ? d1 + j : j - d2) : j;
        sigma[j] = prev[m];
        st.feat[sigma[j]] = m;
      }
      in_f = r.flipped ? d2 : d1; out_f = r.flipped ? d1 : d2;
      nets[0] = &r.t_net; nets[1] = &r.s_net;
    }
    t->has_norm.push_back(st.has_norm);
    t->stats_bound.push_back(0);
    st.in_f = in_f; st.out_f = out_f;
    for (int j = 0; j < in_f; ++j) st.in_slot[j] = sigma[j];
    for (int j = 0; j < out_f; ++j) st.out_slot[j] = sigma[in_f + j];
    for (int q = 0; q < t->nnets; ++q) {
      TrNet& net = st.net[q];
      net.n_layers = nl; net.act = residual ? GBNF_ACT_RELU : nets[q]->activation;
      const int64_t base_row = ((int64_t)s * t->nnets + q) * t->net_rows;
      for (int l = 0; l < nl; ++l) {
        const gbnf_linear& lin = nets[q]->layers[l];
        TrLayer& L = net.layer[l];
        L.W = lin.weight; L.b = lin.bias; L.rows = lin.out_features; L.cols = lin.in_features;
        L.gW = goff; goff += (int64_t)L.rows * L.cols;
        L.gb = goff; goff += L.rows;
        // fragment grids follow the LDS buffers the layer reads / writes: output tiles of the destination buffer, k chunks of
        // the (32-padded) source rows
        const bool first = l == 0, last = l == nl - 1;
        for (int trans = 0; trans < 2; ++trans) {
          PrepProblem P{};
          P.W = L.W; P.rows = L.rows; P.cols = L.cols; P.trans = trans;
          if (!trans) {            // forward: out = rows, k = cols
            P.tiles = (last ? t->op : t->hp) / 16;
            P.kc = (first ? t->xw : t->hw) / 32;
            L.fw = frag_off;
            P.bias = L.b;
            P.boff = (frag_off + (int64_t)P.tiles * P.kc * 128) * 4;      // right behind this orientation's fragments
            L.fb = P.boff;
          } else {                 // backward: out = cols, k = rows
            P.tiles = (first ? t->ip : t->hp) / 16;
            P.kc = (last ? t->ow : t->hw) / 32;
            L.bw = frag_off;
          }
          P.off = frag_off;
          P.blk_begin = prep_blocks;
          prep_blocks += P.tiles * P.kc;
          frag_off += (int64_t)P.tiles * P.kc * 128 + (trans ? 0 : P.tiles * 4);   // (+ 16 bias floats per tile)
          preps.push_back(P);
        }
        WgProblem P{};
        P.M = L.rows; P.N = L.cols;
        // gradient-side operand: D of this layer's output; activation-side operand: this layer's input
        P.d_row = base_row + (l == nl - 1 ? (int64_t)t->ip + 2LL * t->n_hidden * t->hp
                                          : (int64_t)t->ip + (int64_t)t->n_hidden * t->hp + (int64_t)l * t->hp);
        P.a_row = base_row + (l == 0 ? 0 : (int64_t)t->ip + (int64_t)(l - 1) * t->hp);
        P.c_off = L.gW; P.b_off = L.gb;
        P.d_rows = (l == nl - 1) ? t->op : t->hp;
        P.a_rows = (l == 0) ? t->ip : t->hp;
        wg_shape(P.M, P.N, P);
        P.nb = (P.N + P.bn - 1) / P.bn;
        P.blk_begin = blocks;
        blocks += ((P.M + P.bm - 1) / P.bm) * P.nb;
        probs.push_back(P);
      }
    }
  }
  // heaviest blocks first (see wgrad_kernel): the problems in the order of the operand rows a block streams per k-step
  std::stable_sort(probs.begin(), probs.end(), [](const WgProblem& a, const WgProblem& b) { return a.bm + a.bn > b.bm + b.bn; });
  blocks = 0;
  for (WgProblem& P : probs) {
    P.blk_begin = blocks;
    blocks += ((P.M + P.bm - 1) / P.bm) * P.nb;
  }
  t->grad_floats = goff;
  t->n_probs = (int)probs.size();
  t->wg_blocks = blocks;
  t->n_prep = (int)preps.size();
  t->prep_blocks = prep_blocks;
  std::vector<int> tail(64, 0);
  for (int j = 0; j < d; ++j) tail[j] = sigma[j];

  hipError_t e = hipMalloc((void**)&t->steps_dev, sizeof(TrStep) * K);
  if (e == hipSuccess) e = hipMemcpy(t->steps_dev, steps.data(), sizeof(TrStep) * K, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&t->tail_dev, sizeof(int) * 64);
  if (e == hipSuccess) e = hipMemcpy(t->tail_dev, tail.data(), sizeof(int) * 64, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&t->probs_dev, sizeof(WgProblem) * probs.size());
  if (e == hipSuccess) e = hipMemcpy(t->probs_dev, probs.data(), sizeof(WgProblem) * probs.size(), hipMemcpyHostToDevice);
  {
    std::vector<WgProblem> small = probs;
    int b = 0;
    for (WgProblem& P : small) {
      wg_shape(P.M, P.N, P, 128);
      P.nb = (P.N + P.bn - 1) / P.bn;
      P.blk_begin = b;
      b += ((P.M + P.bm - 1) / P.bm) * P.nb;
    }
    t->wg_blocks_small = b;
    if (e == hipSuccess) e = hipMalloc((void**)&t->probs_small_dev, sizeof(WgProblem) * small.size());
    if (e == hipSuccess) e = hipMemcpy(t->probs_small_dev, small.data(), sizeof(WgProblem) * small.size(), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&t->prep_dev, sizeof(PrepProblem) * preps.size());
  if (e == hipSuccess) e = hipMemcpy(t->prep_dev, preps.data(), sizeof(PrepProblem) * preps.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&t->frag_dev, (size_t)frag_off * 16);
  if (e == hipSuccess) e = hipMalloc((void**)&t->gmax_dev, sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc((void**)&t->bn_part_dev, sizeof(float) * 2 * 64 * BN_CHUNKS);
  if (e == hipSuccess) {
    const void* fns[8] = {(const void*)train_kernel<GBNF_KIND_GLOW, 0, 1>, (const void*)train_kernel<GBNF_KIND_GLOW, 1, 1>,
                          (const void*)train_kernel<GBNF_KIND_REALNVP, 0, 1>, (const void*)train_kernel<GBNF_KIND_REALNVP, 1, 1>,
                          (const void*)train_kernel<GBNF_KIND_GLOW, 0, 2>, (const void*)train_kernel<GBNF_KIND_GLOW, 1, 2>,
                          (const void*)train_kernel<GBNF_KIND_REALNVP, 0, 2>, (const void*)train_kernel<GBNF_KIND_REALNVP, 1, 2>};
    for (int k = 0; k < 8 && e == hipSuccess; ++k)
      e = hipFuncSetAttribute(fns[k], hipFuncAttributeMaxDynamicSharedMemorySize, TR_LDS_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_LDS_BYTES);
  }
  if (e != hipSuccess) {
    gbnf_trainer_destroy(t);
    return fail(GBNF_ERR_HIP, "gbnf_trainer_create: %s", hipGetErrorString(e));
  }
  // the register-chained forward sweep where a TRAIN variant of the evaluation kernel covers the geometry (else the
  // kernels of this file do the forward too)
  if (chained_shape) {       // TanhNet / ReLUNet of depth 0, 1, 2; RealNVP ResidualNets of one or two blocks (live_choose refuses what has no TRAIN variant)
    LiveBlob* lb = nullptr;
    std::vector<int64_t> goff(2 * (size_t)K);
    for (int k = 0; k < K; ++k) { goff[2 * k] = steps[k].g_na; goff[2 * k + 1] = steps[k].g_nb; }
    // (32-bit offsets inside one step's operand region: net_rows * np must stay below 2^31 -- checked per call)
    // (the kernels save one row per hidden unit of their COMPILED width: a variant wider than this flow's padded width -- the
    //  nearest compiled one for an unlisted geometry -- would write past the rows the workspace has; such flows keep the old path)
    if (live_blob_create(desc, goff.data(), &lb) == GBNF_OK) {
      if (live_blob_hidden_rows(lb) == t->hp) t->live = lb;
      else live_blob_destroy(lb);
    }
  }
  *out = t;
  return GBNF_OK;
}

int gbnf_trainer_destroy(gbnf_trainer* t) {
  if (t == nullptr) return GBNF_OK;
  if (t->steps_dev) (void)hipFree(t->steps_dev);
  if (t->tail_dev) (void)hipFree(t->tail_dev);
  if (t->probs_dev) (void)hipFree(t->probs_dev);
  if (t->probs_small_dev) (void)hipFree(t->probs_small_dev);
  if (t->prep_dev) (void)hipFree(t->prep_dev);
  if (t->frag_dev) (void)hipFree(t->frag_dev);
  if (t->gmax_dev) (void)hipFree(t->gmax_dev);
  if (t->bn_part_dev) (void)hipFree(t->bn_part_dev);
  live_blob_destroy(t->live);
  delete t;
  return GBNF_OK;
}

int gbnf_trainer_grad_floats(const gbnf_trainer* t, int64_t* n_floats) {
  if (!t || !n_floats) return fail(GBNF_ERR_INVALID, "gbnf_trainer_grad_floats: null argument");
  *n_floats = t->grad_floats;
  return GBNF_OK;
}

int gbnf_trainer_workspace_bytes(const gbnf_trainer* t, int64_t n, int64_t* bytes) {
  if (!t || !bytes || n < 0) return fail(GBNF_ERR_INVALID, "gbnf_trainer_workspace_bytes: bad argument");
  const int64_t np = tr_padded(n);
  // operand regions + 256 slack rows (a 256-row block of wgrad_kernel may run past the last region)
  // ... + the gradient state of step-by-step launches (batch-statistics BatchNorm)
  *bytes = (((int64_t)t->K * t->nnets * t->net_rows + TR_WS_SLACK_ROWS) * np + (int64_t)t->d * np) * 4;
  return GBNF_OK;
}

#ifdef GBNF_TRAIN_STAMPS
static unsigned long long* g_train_stamp_buf = nullptr;
extern "C" void gbnf_debug_set_train_stamp_buffer(unsigned long long* p) { g_train_stamp_buf = p; }
#endif

static void fill_launch(const gbnf_trainer* t, TrainLaunch& p, const float* x, int64_t n) {
  std::memset(&p, 0, sizeof(p));
#ifdef GBNF_TRAIN_STAMPS
  p.dbg = g_train_stamp_buf;
#endif
  p.steps = t->steps_dev; p.tail = t->tail_dev; p.x = x;
  p.sat = training_saturation_counter();
  p.n = n; p.np = tr_padded(n);
  p.d = t->d; p.K = t->K; p.kind = t->kind; p.additive = t->additive;
  p.residual = t->residual;
  p.n_hidden = t->n_hidden; p.hp = t->hp; p.ip = t->ip; p.op = t->op; p.net_rows = t->net_rows;
  p.hw = t->hw; p.xw = t->xw; p.ow = t->ow; p.frag = t->frag_dev;
  p.k_begin = 0; p.k_end = t->K;
}

// batch mean / unbiased variance of step k's input state (slot layout) into the caller's buffers
static void launch_bn_stats(const gbnf_trainer* t, int k, const float* state, int64_t n, int64_t np, hipStream_t s) {
  if (n < 16384) {
    hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)t->d), dim3(256), 0, s, (const TrStep*)t->steps_dev, k, state, n, np);
    return;
  }
  const int nb = BN_CHUNKS;
  float* ps = t->bn_part_dev;
  float* pq = t->bn_part_dev + 64 * BN_CHUNKS;
  hipLaunchKernelGGL(bn_stats_part_kernel, dim3((unsigned)t->d, (unsigned)nb), dim3(256), 0, s, state, n, np, nb, 1, (const float*)nullptr, ps);
  hipLaunchKernelGGL(bn_stats_part_kernel, dim3((unsigned)t->d, (unsigned)nb), dim3(256), 0, s, state, n, np, nb, 2, (const float*)ps, pq);
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(1), dim3(64), 0, s, (const TrStep*)t->steps_dev, k, t->d, n, nb, (const float*)ps, (const float*)pq);
}

// batch-statistics mode is on and some step has a BatchNorm whose statistics must come from the batch
static bool needs_step_launches(const gbnf_trainer* t) {
  if (!t->batch_stats) return false;
  for (int v : t->has_norm)
    if (v) return true;
  return false;
}

int gbnf_trainer_set_batch_stats(gbnf_trainer* t, int32_t on) {
  if (!t) return fail(GBNF_ERR_INVALID, "gbnf_trainer_set_batch_stats: trainer is null");
  // (the step ranges of a batch-statistics sweep are tracked in a 32-bit mask: LiveReduce::skip_steps)
  if (on && t->K > 32) return fail(GBNF_ERR_UNSUPPORTED, "gbnf_trainer_set_batch_stats: batch statistics support at most 32 steps (K = %d)", t->K);
  if (on && t->kind == GBNF_KIND_REALNVP)
    for (size_t k = 0; k < t->has_norm.size(); ++k)
      if (t->has_norm[k] && !t->stats_bound[k])
        return fail(GBNF_ERR_INVALID, "gbnf_trainer_set_batch_stats: step %d has a BatchNorm but no batch-statistics buffers "
                    "(gbnf_trainer_bind_batch_stats)", (int)k);
  t->batch_stats = on ? 1 : 0;
  return GBNF_OK;
}

int gbnf_trainer_bind_batch_stats(gbnf_trainer* t, int32_t step, float* mean_dev, float* var_dev) {
  if (!t || step < 0 || step >= t->K || !mean_dev || !var_dev)
    return fail(GBNF_ERR_INVALID, "gbnf_trainer_bind_batch_stats: bad argument");
  if (!t->has_norm[step] || t->kind != GBNF_KIND_REALNVP)
    return fail(GBNF_ERR_INVALID, "gbnf_trainer_bind_batch_stats: step %d has no BatchNorm", step);
  float* ptrs[2] = {mean_dev, var_dev};
  const hipError_t e = hipMemcpy(reinterpret_cast<char*>(t->steps_dev + step) + offsetof(TrStep, bmean), ptrs, sizeof(ptrs),
                                 hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_trainer_bind_batch_stats: %s", hipGetErrorString(e));
  t->stats_bound[step] = 1;
  if ((int)t->bmean_ptr.size() != t->K) { t->bmean_ptr.assign(t->K, nullptr); t->bvar_ptr.assign(t->K, nullptr); }
  t->bmean_ptr[step] = mean_dev;
  t->bvar_ptr[step] = var_dev;
  return GBNF_OK;
}

int gbnf_trainer_trace_floats(const gbnf_trainer* t, int64_t n, int64_t* n_floats) {
  if (!t || !n_floats || n < 0) return fail(GBNF_ERR_INVALID, "gbnf_trainer_trace_floats: bad argument");
  *n_floats = ((int64_t)t->K + 1) * t->d * tr_padded(n);     // K normalised states + the running state
  if (t->live)       // + the operand workspace the forward sweep fills (and the slack rows wgrad_kernel may read behind it)
    *n_floats += ((int64_t)t->K * t->nnets * t->net_rows + TR_WS_SLACK_ROWS) * tr_padded(n);
  return GBNF_OK;
}

int gbnf_trainer_forward(const gbnf_trainer* t, const float* x, int64_t n, float* z, float* ldj, float* trace,
                         void* stream) {
  if (!t) return fail(GBNF_ERR_INVALID, "gbnf_trainer_forward: trainer is null");
  if (n < 0) return fail(GBNF_ERR_INVALID, "gbnf_trainer_forward: n < 0");
  if (n == 0) return GBNF_OK;
  if (!x) return fail(GBNF_ERR_INVALID, "gbnf_trainer_forward: x is null");
  TrainLaunch p;
  fill_launch(t, p, x, n);
  p.z_out = z; p.ldj_out = ldj; p.trace_out = trace;
  p.batch_stats = t->batch_stats;
  hipStream_t s = (hipStream_t)stream;
  if (t->live != nullptr && trace != nullptr && (int64_t)t->nnets * t->net_rows * p.np < (1LL << 31) &&
      (!needs_step_launches(t) || (ldj != nullptr && n >= 2 && (int)t->bmean_ptr.size() == t->K))) {
    // round 3: the forward sweep on the evaluation kernel (register-chained, weights staged once per workgroup): re-pack
    // the live parameters on the device, then x -> z, ldj + trace + the activation-side operands of the weight gradients
    float* acts = trace + ((int64_t)t->K + 1) * t->d * p.np;
    t->last_fwd_ranges = 1;
    if (!needs_step_launches(t)) {
      const int rc = live_blob_forward(t->live, x, n, z, ldj, trace, acts, p.np, t->ip, t->hp, t->op, stream);
      if (rc) return rc;
    } else {
      t->last_fwd_ranges = 0;
      // round 4: BatchNorm on batch statistics (the reference's train() mode, models/layers.py:338-346) on the same kernels: the
      // sweep is cut in front of every BatchNorm step -- its statistics need the whole batch: one column reduction of the parked
      // state (bn_stats_kernel), the step's table entries re-derived from them (live_norm_step_kernel) -- and the state is parked
      // in HBM in slot layout between the launches (the last d * np floats of the trace buffer)
      float* state = trace + (int64_t)t->K * t->d * p.np;
      const unsigned nb = (unsigned)((p.np + 255) / 256);
      hipLaunchKernelGGL(rows_to_slots_kernel, dim3(nb), dim3(256), 0, s, x, state, n, p.np, t->d);
      bool first = true;
      for (int k0 = 0; k0 < t->K;) {
        int k1 = k0 + 1;
        while (k1 < t->K && !t->has_norm[k1]) ++k1;
        LiveRange rg{k0, k1, state, k1 < t->K ? state : nullptr, k0 > 0 ? 1 : 0, first, nullptr, nullptr};
        if (t->has_norm[k0]) {
          launch_bn_stats(t, k0, state, n, p.np, s);
          rg.bmean = t->bmean_ptr[k0]; rg.bvar = t->bvar_ptr[k0];
        }
        const int rc = live_blob_forward(t->live, x, n, z, ldj, trace, acts, p.np, t->ip, t->hp, t->op, stream, &rg);
        if (rc) return rc;
        ++t->last_fwd_ranges;
        first = false;
        k0 = k1;
      }
    }
    // without a matching backward variant the backward kernels of this file run, on prep_kernel's fragments (valid while
    // the parameters are what they are now: the trace contract)
    if (!live_blob_has_backward(t->live))
      hipLaunchKernelGGL(prep_kernel, dim3((unsigned)t->prep_blocks), dim3(64), 0, s, (const PrepProblem*)t->prep_dev, t->n_prep, t->frag_dev);
    const hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_trainer_forward launch: %s", hipGetErrorString(e2));
    return GBNF_OK;
  }
  // the parameters may have changed since the last call: split them into this call's MFMA fragments
  t->last_fwd_ranges = 0;
  hipLaunchKernelGGL(prep_kernel, dim3((unsigned)t->prep_blocks), dim3(64), 0, s, (const PrepProblem*)t->prep_dev, t->n_prep, t->frag_dev);
  if (!needs_step_launches(t)) {
    launch_train<0>(t, p, s);
  } else {
    // BatchNorm on batch statistics: a step's statistics need the whole batch, so the steps run one launch at a time
    // with the state parked in HBM in slot layout (the last d*np floats of the trace buffer)
    if (!trace || !ldj) return fail(GBNF_ERR_INVALID, "gbnf_trainer_forward: batch-statistics mode needs the trace buffer and ldj");
    if (n < 2) return fail(GBNF_ERR_INVALID, "gbnf_trainer_forward: batch statistics need n >= 2");
    float* state = trace + (int64_t)t->K * t->d * p.np;
    const unsigned nb = (unsigned)((p.np + 255) / 256);
    hipLaunchKernelGGL(rows_to_slots_kernel, dim3(nb), dim3(256), 0, s, x, state, n, p.np, t->d);
    for (int k = 0; k < t->K; ++k) {
      if (t->has_norm[k])
        launch_bn_stats(t, k, state, n, p.np, s);
      TrainLaunch q = p;
      q.k_begin = k; q.k_end = k + 1;
      q.state_in = state; q.state_out = state;
      q.ldj_accumulate = k > 0;
      launch_train<0>(t, q, s);
    }
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_trainer_forward launch: %s", hipGetErrorString(e));
  return GBNF_OK;
}

int gbnf_trainer_backward(const gbnf_trainer* t, const float* x, int64_t n, const float* trace, const float* g_z,
                          const float* g_ldj, float* g_x, float* grads, void* workspace, int64_t workspace_bytes,
                          void* stream) {
  if (!t) return fail(GBNF_ERR_INVALID, "gbnf_trainer_backward: trainer is null");
  if (n < 0) return fail(GBNF_ERR_INVALID, "gbnf_trainer_backward: n < 0");
  if (n == 0) return GBNF_OK;
  if (!x || !grads || !workspace) return fail(GBNF_ERR_INVALID, "gbnf_trainer_backward: x / grads / workspace is null");
  int64_t need = 0;
  gbnf_trainer_workspace_bytes(t, n, &need);
  if (workspace_bytes < need)
    return fail(GBNF_ERR_INVALID, "gbnf_trainer_backward: workspace of %lld bytes < %lld", (long long)workspace_bytes,
                (long long)need);
  TrainLaunch p;
  fill_launch(t, p, x, n);
  p.g_z = g_z; p.g_ldj = g_ldj; p.g_x = g_x; p.grads = grads; p.ws = (float*)workspace; p.trace = trace;
  p.batch_stats = t->batch_stats;
  float* gstate = (float*)workspace + ((int64_t)t->K * t->nnets * t->net_rows + TR_WS_SLACK_ROWS) * p.np;   // [d][np]: gradient state between step launches
  hipStream_t s = (hipStream_t)stream;
  // the scale of this call's gradients: the largest upstream entry (tr_grad_scale)
  (void)hipMemsetAsync(t->gmax_dev, 0, sizeof(unsigned), s);
  {
    const int64_t work = n * (g_z ? t->d : 1);
    const unsigned gb = (unsigned)((work + 256 * 16 - 1) / (256 * 16) < 256 ? (work + 256 * 16 - 1) / (256 * 16) : 256);
    hipLaunchKernelGGL(gmax_kernel, dim3(gb ? gb : 1), dim3(256), 0, s, g_z, g_ldj, n, t->d, t->gmax_dev);
  }
  p.gmax = t->gmax_dev;
  if (live_blob_has_backward(t->live) && trace != nullptr && (int64_t)t->nnets * t->net_rows * p.np < (1LL << 31) &&
      (!needs_step_launches(t) || (int)t->bmean_ptr.size() == t->K)) {
    // round 3: the register-chained backward sweep on what the forward sweep saved behind the trace, then the weight
    // gradients from the operand workspace that now lives there too
    float* acts = const_cast<float*>(trace) + ((int64_t)t->K + 1) * t->d * p.np;
    LiveReduce red{};
    t->last_bwd_ranges = 1;
    if (!needs_step_launches(t)) {
      const int rc = live_blob_backward(t->live, n, trace, acts, p.np, t->ip, t->hp, t->op, g_z, g_ldj, g_x, grads, t->gmax_dev, stream, &red);
      if (rc) return rc;
    } else {
      t->last_bwd_ranges = 0;
      // round 4, the mirror image of the forward: one launch per step range, last range first; behind a range that starts with a
      // BatchNorm step its two parameter sums are added up (they are that step's d/d beta and d/d log_gamma: exactly the two batch
      // sums the correction needs) and the parked gradient state is corrected for the dependence of the batch statistics on
      // every sample (bn_bwd_fix_kernel); the last launch's state goes out as g_x rows
      std::vector<int> starts;
      for (int k0 = 0; k0 < t->K;) { starts.push_back(k0); int k1 = k0 + 1; while (k1 < t->K && !t->has_norm[k1]) ++k1; k0 = k1; }
      unsigned done = 0u;
      for (int ri = (int)starts.size() - 1; ri >= 0; --ri) {
        const int k0 = starts[ri], k1 = ri + 1 < (int)starts.size() ? starts[ri + 1] : t->K;
        LiveRange rg{k0, k1, k1 < t->K ? gstate : nullptr, gstate, 0, false, nullptr, nullptr};
        const int rc = live_blob_backward(t->live, n, trace, acts, p.np, t->ip, t->hp, t->op, g_z, g_ldj, nullptr, grads, t->gmax_dev, stream, &red, &rg);
        if (rc) return rc;
        ++t->last_bwd_ranges;
        if (t->has_norm[k0]) {
          LiveReduce one = red;
          one.skip_steps = ~(1u << k0);
          hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)(2 * red.K), 1u), dim3(WG_THREADS), 2048, s, t->probs_dev, t->n_probs, (const float*)acts, grads,
                             p.np, 512, (const unsigned*)t->gmax_dev, 0, one);
          done |= 1u << k0;
          const dim3 fg((unsigned)((n + 255) / 256), (unsigned)t->d);
          hipLaunchKernelGGL(bn_bwd_fix_kernel, fg, dim3(256), 0, s, (const TrStep*)t->steps_dev, k0, t->d, trace, (const float*)grads,
                             gstate, n, p.np, (const unsigned*)t->gmax_dev);
        }
      }
      if (g_x != nullptr)
        hipLaunchKernelGGL(slots_to_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)gstate, g_x, n, p.np, t->d,
                           (const unsigned*)t->gmax_dev);
      red.skip_steps = done;
    }
    // samples per block: the largest power of two in [128, 2048] that still leaves ~a block per CU (>= 240 blocks; one 128 KB
    // workgroup of 8 waves per CU).  Measured (MINIBOONE step, 15 blocks per chunk): N = 65536 with 1024 / 2048 / 4096 samples
    // per block 64.9 / 66.7 / 63.8 M samples/s, N = 16384 with 512 / 1024 / 2048: 46.1 / 49.2 / 41.2 M, N = 4096 with 128 / 256 /
    // 512: 16.5 / 19.1 / 18.8 M; every further chunk adds a bm x bn tile of float atomics per block of dW, every chunk less leaves
    // CUs idle.
    // (up to 16 k rows the operands fit the L2s: blocks of at most 128 x 128 -- twice the workgroups)
    static const int small_np = [] { const char* e = getenv("GBNF_WG_SMALL_NP"); return e ? atoi(e) : 16384; }();
    const bool small = p.np <= small_np && t->probs_small_dev != nullptr;
    const WgProblem* wg_probs = small ? t->probs_small_dev : t->probs_dev;
    const int wg_blocks = small ? t->wg_blocks_small : t->wg_blocks;
    int chunk2 = 128;
    while (chunk2 < 2048 && (int64_t)wg_blocks * (p.np / (2 * chunk2)) >= 240) chunk2 *= 2;
    if (const char* e = getenv("GBNF_WG_CHUNK")) { if (atoi(e) > 0) chunk2 = atoi(e); }      // (A/B runs)
    // (+ 2 K blocks: the sums of the backward kernel's parameter-gradient partials ride in this launch)
    const dim3 wgrid2((unsigned)(wg_blocks + 2 * red.K), (unsigned)((p.np + chunk2 - 1) / chunk2));
    hipLaunchKernelGGL(wgrad_kernel, wgrid2, dim3(WG_THREADS), WG_LDS_BYTES, s, wg_probs, t->n_probs,
                       (const float*)acts, grads, p.np, chunk2, (const unsigned*)t->gmax_dev, wg_blocks, red);
    const hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_trainer_backward launch: %s", hipGetErrorString(e2));
    return GBNF_OK;
  }
  // a trace is valid only while the parameters are what they were in the forward call that wrote it (include/gbnf.h):
  // that call split them into this trainer's fragment buffer, so the fragments are still the right ones
  t->last_bwd_ranges = 0;
  if (trace == nullptr)
    hipLaunchKernelGGL(prep_kernel, dim3((unsigned)t->prep_blocks), dim3(64), 0, s, (const PrepProblem*)t->prep_dev, t->n_prep, t->frag_dev);
  if (!needs_step_launches(t)) {
    launch_train<1>(t, p, s);
  } else {
    // the mirror image of the forward: one launch per step; after a BatchNorm step the gradient state is corrected for
    // the dependence of the batch statistics on every sample (its two batch sums are that step's d/d beta, d/d log_gamma)
    if (!trace) return fail(GBNF_ERR_INVALID, "gbnf_trainer_backward: batch-statistics mode needs the forward call's trace");
    for (int k = t->K - 1; k >= 0; --k) {
      TrainLaunch q = p;
      q.k_begin = k; q.k_end = k + 1;
      q.gstate_in = (k == t->K - 1) ? nullptr : gstate;
      q.gstate_out = gstate;
      launch_train<1>(t, q, s);
      if (t->has_norm[k]) {
        const dim3 fg((unsigned)((n + 255) / 256), (unsigned)t->d);
        hipLaunchKernelGGL(bn_bwd_fix_kernel, fg, dim3(256), 0, s, (const TrStep*)t->steps_dev, k, t->d, trace, (const float*)grads,
                           gstate, n, p.np, (const unsigned*)t->gmax_dev);
      }
    }
    if (g_x != nullptr)
      hipLaunchKernelGGL(slots_to_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)gstate, g_x, n, p.np, t->d,
                         (const unsigned*)t->gmax_dev);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_trainer_backward launch: %s", hipGetErrorString(e));
  // samples per block: every block ends in 4096 atomic adds, so as many samples as still leave a few thousand waves
  static const int forced_chunk = [] { const char* e = getenv("GBNF_WGRAD_CHUNK"); return e ? atoi(e) : 0; }();
  int chunk = 512;
  while (chunk < 4096 && (int64_t)t->wg_blocks * (p.np / (2 * chunk)) >= 768) chunk *= 2;     // (a block = 4 waves)
  if (forced_chunk >= 32 && forced_chunk % 32 == 0) chunk = forced_chunk;
  const dim3 wgrid((unsigned)t->wg_blocks, (unsigned)((p.np + chunk - 1) / chunk));
  hipLaunchKernelGGL(wgrad_kernel, wgrid, dim3(WG_THREADS), WG_LDS_BYTES, s, t->probs_dev, t->n_probs, (const float*)workspace, grads, p.np, chunk,
                     (const unsigned*)t->gmax_dev, t->wg_blocks, LiveReduce{});
  e = hipGetLastError();
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_trainer_backward wgrad launch: %s", hipGetErrorString(e));
  return GBNF_OK;
}

// (tests) how the last forward / backward call of a trainer ran: launches of the register-chained kernels (1 = the whole sweep in one
// launch, > 1 = one per step range of a batch-statistics BatchNorm sweep), 0 = the round-1 per-step kernels
int gbnf_debug_trainer_last_path(const gbnf_trainer* t, int32_t* fwd_ranges, int32_t* bwd_ranges) {
  if (!t) return fail(GBNF_ERR_INVALID, "gbnf_debug_trainer_last_path: trainer is null");
  if (fwd_ranges) *fwd_ranges = t->last_fwd_ranges;
  if (bwd_ranges) *bwd_ranges = t->last_bwd_ranges;
  return GBNF_OK;
}

// (tests) the live blob of a trainer after a device re-pack from the current parameter values
int gbnf_debug_trainer_blob(const gbnf_trainer* t, uint32_t* out_host, int64_t* n_words) {
  if (!t) return fail(GBNF_ERR_INVALID, "gbnf_debug_trainer_blob: trainer is null");
  if (!t->live) return fail(GBNF_ERR_UNSUPPORTED, "gbnf_debug_trainer_blob: this trainer has no live blob");
  return live_blob_words(t->live, out_host, n_words);
}

}  // extern "C"
