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

// Packed-parameter layout of one coupling network, in 32-bit words (host packer and
// kernel share it).  HT = hidden tiles of 16 units, OT = output tiles of 16 rows,
// LMID = number of hidden->hidden layers (coupling_network_depth).
template <int HT, int OT, int LMID>
struct NetLayout {
  static constexpr int W1 = 0;                                   // [KS1MAX+1][HT][64]        f32
  static constexpr int B1 = W1 + (KS1MAX + 1) * HT * 64;         // [HT][4 g][4 r]
  static constexpr int MID0 = B1 + HT * 16;                      // LMID x { W [HT+1][HT][64][4], B [HT][4][4] }
  static constexpr int MID_W = (HT + 1) * HT * 256;
  static constexpr int MID_STRIDE = MID_W + HT * 16;
  static constexpr int W3 = MID0 + LMID * MID_STRIDE;            // [HT+1][OT][64][4]
  static constexpr int B3 = W3 + (HT + 1) * OT * 256;            // [OT][4][4]
  static constexpr int NET_WORDS = B3 + OT * 16;
};

struct FlowLaunch {
  const uint32_t* const* blobs;  // device array: packed parameter blob per component
  const float* x;                // (n, d)
  float* z_out;                  // (n_comp, n, d) or null
  float* ldj_out;                // (n_comp, n)    or null
  float* ll_out;                 // (n_comp, n)    or null
  const float* base_mean;        // (d,) or null  -> N(0,1)
  const float* base_std;         // (d,) or null
  int64_t n;
  int32_t d;
  int32_t n_steps;
  int32_t c_begin;
  int32_t n_comp;
  int32_t n_tiles;               // ceil(n / (16*NT))
  int32_t additive;              // glow: additive coupling
};

__device__ __forceinline__ float as_f32(uint32_t u) { return __builtin_bit_cast(float, u); }

// tanh with absolute error ~1e-7 (1 - 2/(e^{2|x|}+1) on v_exp_f32 / v_rcp_f32, 1 ulp each).
// Hidden activations feed f32 dot products of O(1) terms, so absolute (not relative)
// accuracy is what the log-likelihood sees; parity tests pin the end-to-end 1e-5 bar.
__device__ __forceinline__ float tanh_act(float x) {
#ifdef GBNF_TANH_LIBM
  return tanhf(x);
#else
  float ax = __builtin_fabsf(x);
  float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);  // e^{2|x|}
  float r = __builtin_amdgcn_rcpf(e + 1.0f);
  float t = __builtin_fmaf(-2.0f, r, 1.0f);
  return __builtin_copysignf(t, x);
#endif
}

template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
  if constexpr (ACT == GBNF_ACT_TANH) {
    return tanh_act(v);
  } else {
    return __builtin_fmaxf(v, 0.0f);
  }
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------
// Register-resident coupling network for one wave.
//   zb      : wave-private LDS scratch, zb[(s*NT+nt)*64 + lane] = B operand of first-layer
//             k-step s (normalised z1 values), rows 0..ks1 valid (row ks1 is a zero pad)
//   out     : OT x NT accumulator tiles of the last Linear (bias included)
// KSL = live k-steps (1..4) in the LAST hidden tile (hidden width padded to a k-step
// multiple; the padded units sit in whole trailing k-steps so they are skipped, not
// multiplied by zero).
// ---------------------------------------------------------------------------------
template <int HT, int KSL, int OT, int NT, int LMID, int ACT>
__device__ __forceinline__ void coupling_net(const uint32_t* __restrict__ net, const float* zb,
                                             int ks1, int lane, int g, f32x4 (&out)[OT][NT]) {
  using L = NetLayout<HT, OT, LMID>;

  // ---- layer 0: in -> hidden.  Rolled over k-steps, all HT*NT accumulators independent.
  f32x4 hA[HT][NT];
  {
    const float* w1 = reinterpret_cast<const float*>(net + L::W1) + lane;
    const f32x4* b1 = reinterpret_cast<const f32x4*>(net + L::B1) + g;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      f32x4 b = b1[t * 4];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) hA[t][nt] = b;
    }
    float wc[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) wc[t] = w1[t * 64];
    float zc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) zc[nt] = zb[nt * 64 + lane];
#pragma unroll 1
    for (int s = 0; s < ks1; ++s) {
      float zn[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) zn[nt] = zb[((s + 1) * NT + nt) * 64 + lane];
      const float* wn = w1 + (s + 1) * HT * 64;
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        float w = wc[t];
        wc[t] = wn[t * 64];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) hA[t][nt] = mfma4(w, zc[nt], hA[t][nt]);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) zc[nt] = zn[nt];
    }
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) hA[t][nt][r] = act_fn<ACT>(hA[t][nt][r]);
  }

  // ---- hidden -> hidden layers that are NOT the last one: fully unrolled (their output
  //      tiles must land in a statically indexed register array).
#pragma unroll
  for (int m = 0; m < LMID - 1; ++m) {
    const f32x4* w = reinterpret_cast<const f32x4*>(net + L::MID0 + m * L::MID_STRIDE) + lane;
    const f32x4* b = reinterpret_cast<const f32x4*>(net + L::MID0 + m * L::MID_STRIDE + L::MID_W) + g;
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
        for (int r = 0; r < 4; ++r) hA[t][nt][r] = act_fn<ACT>(hB[t][nt][r]);
  }

  // ---- output accumulators start at the last Linear's bias
  const f32x4* w3 = reinterpret_cast<const f32x4*>(net + L::W3) + lane;
  {
    const f32x4* b3 = reinterpret_cast<const f32x4*>(net + L::B3) + g;
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      f32x4 b = b3[o * 4];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) out[o][nt] = b;
    }
  }

  if constexpr (LMID == 0) {
    // hidden -> out straight from hA
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
  } else {
    // ---- last hidden->hidden layer fused with the output layer.  Rolled over output tile u:
    //      tile u of the hidden layer is finished (all k), activated, and immediately
    //      consumed as k-chunk u of the output layer -- it never exists outside 4*NT VGPRs.
    //      Weight tiles for u+1 are requested right after tile u's registers are consumed
    //      (one full tile time of prefetch distance; the blob has one pad row for u = HT).
    const f32x4* w2 = reinterpret_cast<const f32x4*>(net + L::MID0 + (LMID - 1) * L::MID_STRIDE) + lane;
    const f32x4* b2 = reinterpret_cast<const f32x4*>(net + L::MID0 + (LMID - 1) * L::MID_STRIDE + L::MID_W) + g;
    f32x4 A[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) A[t] = w2[t * 64];
    f32x4 A3[OT];
#pragma unroll
    for (int o = 0; o < OT; ++o) A3[o] = w3[o * 64];

    auto hidden_tile = [&](int u, f32x4 (&hb)[NT]) {
      f32x4 bb = b2[u * 4];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) hb[nt] = bb;
      const f32x4* wn = w2 + (u + 1) * HT * 64;
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        f32x4 a = A[t];
        A[t] = wn[t * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (t < HT - 1 || r < KSL) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) hb[nt] = mfma4(a[r], hA[t][nt][r], hb[nt]);
          }
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) hb[nt][r] = act_fn<ACT>(hb[nt][r]);
    };

#pragma unroll 1
    for (int u = 0; u < HT - 1; ++u) {
      f32x4 hb[NT];
      hidden_tile(u, hb);
      const f32x4* w3n = w3 + (u + 1) * OT * 64;
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        f32x4 a = A3[o];
        A3[o] = w3n[o * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) out[o][nt] = mfma4(a[r], hb[nt][r], out[o][nt]);
      }
    }
    {
      f32x4 hb[NT];
      hidden_tile(HT - 1, hb);
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        f32x4 a = A3[o];
#pragma unroll
        for (int r = 0; r < KSL; ++r)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) out[o][nt] = mfma4(a[r], hb[nt][r], out[o][nt]);
      }
    }
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

__device__ __forceinline__ float sigmoid_acc(float v) { return 1.0f / (1.0f + expf(-v)); }

template <int KIND, int HT, int KSL, int OT, int NT, int LMID, int ACTA, int ACTB>
__global__ void __launch_bounds__(64) flow_kernel(const FlowLaunch p) {
  constexpr int ZS = 16 * NT + 1;   // +1: conflict-free transposed x load / z store
  constexpr int NNETS = (KIND == GBNF_KIND_REALNVP) ? 2 : 1;
  using L = NetLayout<HT, OT, LMID>;
  constexpr int STEP_WORDS = SMALL_WORDS + NNETS * L::NET_WORDS;

  __shared__ float Z[ZSLOTS * ZS];
  __shared__ float ZB[(KS1MAX + 1) * NT * 64];

  const int lane = threadIdx.x;
  const int i = lane & 15;
  const int g = lane >> 4;

  // ---- XCD-aware block -> (component, sample tile).  Blocks are dealt round-robin over the
  //      8 XCDs; give each XCD a contiguous run of the (component-major) work list so one
  //      component's weights stay in one XCD's L2.  Bijective for any grid size.
  int comp, tile;
  {
    const int total = gridDim.x;
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int base = total >> 3, rem = total & 7;
    const int q = xcd * base + (xcd < rem ? xcd : rem) + j;
    comp = q / p.n_tiles;
    tile = q - comp * p.n_tiles;
  }
  const uint32_t* __restrict__ blob = p.blobs[p.c_begin + comp];
  const int d = p.d;
  const int64_t row0 = (int64_t)tile * (16 * NT);

  // ---- x tile -> Z[feature][sample]  (rows of x are contiguous: coalesced 4*d-byte runs)
  if (lane < d) {
#pragma unroll 8
    for (int r = 0; r < 16 * NT; ++r) {
      const int64_t n = row0 + r;
      float v = 0.0f;
      if (n < p.n) v = p.x[n * d + lane];
      Z[lane * ZS + r] = v;
    }
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) ZB[(KS1MAX * NT + nt) * 64 + lane] = 0.0f;
  __syncthreads();

  float ld[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) ld[nt] = 0.0f;
  float ld_const = 0.0f;

  for (int step = 0; step < p.n_steps; ++step) {
    const uint32_t* __restrict__ sp = blob + (size_t)step * STEP_WORDS;
    const int ks1 = __builtin_amdgcn_readfirstlane((int)sp[0]);
    ld_const += as_f32(sp[1]);
    const uint32_t* tab = sp + SMALL_HDR + g * NENT;

    // ---- normalise the coupling net's inputs in place and stage them as first-layer B operands
    {
      int slot[NENT];
      float p0[NENT], p1[NENT], p2[NENT], p3[NENT];
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
#pragma unroll
      for (int e = 0; e < NENT; ++e) {
        if (e < ks1) {
          const bool live = slot[e] >= 0;
          const int zoff = (live ? slot[e] : 0) * ZS + i;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            float v = Z[zoff + 16 * nt];
            v = norm_fn<KIND>(v, p0[e], p1[e], p2[e], p3[e]);
            if (live) Z[zoff + 16 * nt] = v;
            ZB[(e * NT + nt) * 64 + lane] = live ? v : 0.0f;
          }
        }
      }
    }

    // ---- coupling network(s) on the matrix cores
    f32x4 outA[OT][NT];
    coupling_net<HT, KSL, OT, NT, LMID, ACTA>(sp + SMALL_WORDS, ZB, ks1, lane, g, outA);
    f32x4 outB[OT][NT];
    if constexpr (KIND == GBNF_KIND_REALNVP) {
      coupling_net<HT, KSL, OT, NT, LMID, ACTB>(sp + SMALL_WORDS + L::NET_WORDS, ZB, ks1, lane, g, outB);
    }

    // ---- coupling transform of the other half, in place, + per-lane log-det partials
    {
      const uint32_t* otab = tab + 160;
      int slot[NENT];
      float p0[NENT], p1[NENT], p2[NENT], p3[NENT];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        i32x4 s4 = *reinterpret_cast<const i32x4*>(otab + h * 4);
        f32x4 a4 = *reinterpret_cast<const f32x4*>(otab + 32 + h * 4);
        f32x4 b4 = *reinterpret_cast<const f32x4*>(otab + 64 + h * 4);
        f32x4 c4 = *reinterpret_cast<const f32x4*>(otab + 96 + h * 4);
        f32x4 d4 = *reinterpret_cast<const f32x4*>(otab + 128 + h * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          slot[h * 4 + e] = s4[e]; p0[h * 4 + e] = a4[e]; p1[h * 4 + e] = b4[e];
          p2[h * 4 + e] = c4[e]; p3[h * 4 + e] = d4[e];
        }
      }
      if (KIND == GBNF_KIND_GLOW && !p.additive) {
        // affine, "cross" split: rows 2j / 2j+1 of the last Linear are (shift_j, raw_j) and sit
        // in adjacent accumulator registers of the same lane.  models/glow.py:331-338.
#pragma unroll
        for (int e = 0; e < 2 * OT && e < NENT; ++e) {
          const int o = e >> 1, pp = e & 1;
          const bool live = slot[e] >= 0;
          const int zoff = (live ? slot[e] : 0) * ZS + i;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            float v = Z[zoff + 16 * nt];
            v = norm_fn<KIND>(v, p0[e], p1[e], p2[e], p3[e]);
            const float shift = outA[o][nt][2 * pp], raw = outA[o][nt][2 * pp + 1];
            const float sc = sigmoid_acc(raw + 2.0f);
            v = (v + shift) * sc;
            if (live) {
              Z[zoff + 16 * nt] = v;
              ld[nt] += logf(sc);
            }
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4 * OT && e < NENT; ++e) {
          const int o = e >> 2, r = e & 3;
          const bool live = slot[e] >= 0;
          const int zoff = (live ? slot[e] : 0) * ZS + i;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            float v = Z[zoff + 16 * nt];
            v = norm_fn<KIND>(v, p0[e], p1[e], p2[e], p3[e]);
            if constexpr (KIND == GBNF_KIND_GLOW) {
              v = v + outA[o][nt][r];                       // additive, models/glow.py:328-329
              if (live) Z[zoff + 16 * nt] = v;
            } else {
              const float shift = outA[o][nt][r], scale = outB[o][nt][r];
              v = shift + v * expf(scale);                  // models/transformations.py:575
              if (live) {
                Z[zoff + 16 * nt] = v;
                ld[nt] += scale;                            // models/transformations.py:577
              }
            }
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- base log-density + log|det J|, folded over the 4 lane groups
  const uint32_t* tail = blob + (size_t)p.n_steps * STEP_WORDS;   // final slot of logical feature j
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
      const int64_t o = (int64_t)comp * p.n + n;
      if (p.ldj_out) p.ldj_out[o] = ldj;
      if (p.ll_out) p.ll_out[o] = (q - 0.91893853320467274f * (float)d) + ldj;   // -d/2 log(2 pi)
    }
  }

  // ---- z in the reference's feature order (post-permutation, post-swap)
  if (p.z_out != nullptr && lane < d) {
    const int slot = (int)tail[lane];
    float* zo = p.z_out + (int64_t)comp * p.n * d;
#pragma unroll 8
    for (int r = 0; r < 16 * NT; ++r) {
      const int64_t n = row0 + r;
      if (n < p.n) zo[n * d + lane] = Z[slot * ZS + r];
    }
  }
}

// ---------------------------------------------------------------------------------
// variant registry: each variants/*.hip instantiates one kernel and registers a launcher
// ---------------------------------------------------------------------------------
struct VariantKey {
  int kind, ht, ksl, ot, nt, lmid, act_a, act_b;
  bool operator==(const VariantKey& o) const {
    return kind == o.kind && ht == o.ht && ksl == o.ksl && ot == o.ot && nt == o.nt &&
           lmid == o.lmid && act_a == o.act_a && act_b == o.act_b;
  }
};
using LaunchFn = hipError_t (*)(const FlowLaunch&, unsigned grid, hipStream_t);
void register_variant(const VariantKey& key, LaunchFn fn, const char* name);

#define GBNF_INSTANTIATE(KIND, HT, KSL, OT, NT, LMID, ACTA, ACTB)                                   \
  namespace gbnf {                                                                                  \
  static hipError_t launch_##KIND##_##HT##_##KSL##_##OT##_##NT##_##LMID##_##ACTA##_##ACTB(          \
      const FlowLaunch& p, unsigned grid, hipStream_t s) {                                          \
    hipLaunchKernelGGL((flow_kernel<KIND, HT, KSL, OT, NT, LMID, ACTA, ACTB>), dim3(grid), dim3(64), \
                       0, s, p);                                                                    \
    return hipGetLastError();                                                                       \
  }                                                                                                 \
  static const int reg_##KIND##_##HT##_##KSL##_##OT##_##NT##_##LMID##_##ACTA##_##ACTB =             \
      (register_variant(VariantKey{KIND, HT, KSL, OT, NT, LMID, ACTA, ACTB},                        \
                        launch_##KIND##_##HT##_##KSL##_##OT##_##NT##_##LMID##_##ACTA##_##ACTB,      \
                        "flow_kernel<" #KIND "," #HT "," #KSL "," #OT "," #NT "," #LMID "," #ACTA   \
                        "," #ACTB ">"),                                                             \
       0);                                                                                          \
  }

}  // namespace gbnf
