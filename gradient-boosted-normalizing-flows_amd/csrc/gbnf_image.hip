// gbnf_image.hip -- density evaluation of ONE image Glow component (BASELINE.json
// configs[3]: CIFAR-10 3x32x32 multi-scale Boosted-Glow) for gfx950.
//
//   x (n,3,H,W) in [0,1]  --dequantise, logit-->  L levels of { squeeze, K x FlowStep, Split2d }  -->  z, log|det|, ll
//
// Every convolution is an implicit GEMM on the matrix cores (exact-f32 v_mfma_f32_16x16x4_f32), "transposed" like the
// tabular path:  OUT(channels x pixels) = sum_tap W_tap(out x in) . IN_tap(in x pixels).  One workgroup (4 waves) owns a
// strip of IMG_R rows of one image: the strip plus its halo, all input channels, is staged ONCE in LDS with the zero
// 'same' padding materialised, and is the B operand of every tap straight from LDS (a tap is an address offset); the
// weights are the A operand, pre-tiled at pack time into lane order (one 16-byte load per lane per 4 k-steps) with the
// per-channel scales of ActNorm2d / Conv2dZeros folded in, streamed from L2 through a ring of loads in flight.
//   * ActNorm2d + InvertibleConv1x1 / Permute2d of a FlowStep collapse into ONE 1x1 convolution
//     (W_eff = W_perm . diag(exp(logs)), b_eff = W_eff . bias); its log-determinant is a per-handle constant.
//   * the coupling (shift/scale "cross" rows are adjacent accumulator registers), the Split2d prior density and the
//     ReLU / bias epilogues are fused into the producing convolution; log-det partials are reduced with wave shuffles.
//   * convolutions with fewer output tiles than waves (the last 3x3 of a coupling net: 256 -> 12/24 channels) split
//     their contraction (taps x channel chunks) over the 4 waves and fold the partial tiles through LDS.
//
// Reference semantics (file:line):  Glow.encode / dequantize / to_logits models/glow.py:92-110, 125-179;  FlowNet image
// branch :192-252;  FlowStep.encode :317-342;  _ActNorm (x H*W log-det) models/layers.py:488-533;  InvertibleConv1x1
// :722-796;  Permute2d :671-680;  ConvNet :304-317;  Conv2d / Conv2dZeros :577-630;  Split2d :685-705;  squeeze2d
// utils/utilities.py:107-119;  Glow.prior models/glow.py:62-84;  log_normal_diag utils/distributions.py:13-21;
// ll = log_normal_diag(z, mu, var) + logdet image_experiment.py:227.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/gbnf.h"
#include "gbnf_internal.h"
#include "gbnf_image_net.h"

namespace gbnf {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int IMG_R = 4;          // rows of one image per workgroup
constexpr int IMG_WAVES = 4;
constexpr int IMG_PD = 4;         // weight-prefetch distance (iterations of 4 k-steps)
constexpr int IMG_MAX_PT = 4;     // pixel tiles of 16 per strip (IMG_R * W / 16, W <= 16)

// (the epilogue kinds EPI_* live in gbnf_image_net.h: shared with the fused coupling-net kernel's translation unit)

struct ConvLaunch {
  const float* in;        // (n, *, H, W): first input channel of image 0
  int64_t in_img;         // floats between images
  const float* wp;        // packed weights [OT][taps][KC][64][4]
  const float* bias;      // [OT*16]
  float* out;             // EPI_RELU / EPI_STORE: (n, *, H, W) first output channel of image 0
  int64_t out_img;
  float* st;              // EPI_COUPLE_* / EPI_SPLIT: the coupled half z2 (n, *, H, W), first channel of image 0
  int64_t st_img;
  float* ldj;             // (n,) accumulated with atomics (EPI_COUPLE_AFFINE / EPI_SPLIT)
  int cin, cout, H, W, ks, n_strips;
  int Hv, Wv;             // the map proper: rows < Hv, columns < Wv of the H x W storage (a 14 x 14 map lives in 16 x 16 storage;
                          // everything outside is ZERO in every tensor in HBM -- the 'same' padding of the map -- and stays so)
  float temperature;      // EPI_SPLIT_INV: z2 = mean + exp(log-var) * temperature * eps (models/layers.py:697)
  int c_chunk;            // input channels staged per pass (a multiple of 16; 0 = all: 3 x 3 convolutions from > 256 channels do not
                          // fit the LDS at once and stage their input in two halves; split-contraction form only)
  int o_split;            // workgroups sharing one strip, each with 1/o_split of the output tiles (fills the chip at small batch)
  // fused producer (1x1 convolutions only): the input of this convolution is relu(conv3x3(pre_in) + pre_bias), computed
  // for the strip straight into LDS instead of being read from `in` (the ConvNet's first layer never touches HBM)
  const float* pre_in;    // (n, *, H, W) first input channel of image 0, or null
  int64_t pre_in_img;
  const float* pre_wp;    // packed 3x3 weights [cin/16 tiles][9][pre_kc][64][4]
  const float* pre_bias;
  int pre_cin;
};

__device__ __forceinline__ f32x4 img_mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// see gbnf_train.hip: the compiler's MFMA-result hazard padding does not look across branches on this toolchain
template <int N>
__device__ __forceinline__ void img_drain(f32x4 (&c)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) asm volatile("s_nop 7\n\ts_nop 7" : "+a"(c[k]));
}

// EPI: see enum.  PT: pixel tiles per strip (2 or 4).  KS: 1 or 3 (a 1x1 convolution stages no halo: half the LDS,
// two workgroups per CU).
// The work of one workgroup: strip `strip` of image `n`, share `osp` of the output tiles.  A device function so that the repair
// kernel (img_repair_kernel: one workgroup walks a marked image through the whole exact-f32 sequence) runs the same code.
template <int EPI, int PT, int KS>
__device__ __forceinline__ void img_conv_body(const ConvLaunch& p, const int n, const int strip, const int osp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef const float __attribute__((address_space(1)))* gptr;
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int W = 16 * PT / IMG_R;                        // the strip is IMG_R full rows: W = 16 (PT = 4) or 8 (PT = 2)
  constexpr int HALO = KS >> 1;
  constexpr int WP = W + 2 * HALO, RP = IMG_R + 2 * HALO, CS = RP * WP;   // padded row / rows / channel stride in LDS
  const int H = p.H, r0 = strip * IMG_R;
  const int kc = (p.cin + 15) >> 4, cin_pad = kc * 16;
  constexpr int taps = KS * KS, half = HALO;
  const int OT = (p.cout + 15) >> 4;

  if (KS == 1 && p.pre_in != nullptr) {
    // ---- fused first layer: stage its small input with a halo behind the main buffer, run the 3x3 on the matrix cores,
    //      leave relu(. + bias) in the main buffer as this convolution's input
    constexpr int WPz = W + 2, RPz = IMG_R + 2, CSz = RPz * WPz;
    float* zin = lds + cin_pad * CS;
    const int pkc = (p.pre_cin + 15) >> 4;
    {
      const float* src = p.pre_in + (int64_t)n * p.pre_in_img;
      constexpr int Q = W / 4;
      const int q = threadIdx.x % Q, rid = threadIdx.x / Q;
      constexpr int ROWS_PER_PASS = 64 * IMG_WAVES / Q;
      for (int idx = rid; idx < pkc * 16 * RPz; idx += ROWS_PER_PASS) {
        const int ci = idx / RPz, rr = idx - ci * RPz;
        const int row = r0 + rr - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ci < p.pre_cin && row >= 0 && row < H) v = *reinterpret_cast<const f32x4*>(src + ((int64_t)ci * H + row) * W + 4 * q);
        float* dst = zin + ci * CSz + rr * WPz + 1 + 4 * q;
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        if (q == 0) dst[-1] = 0.0f;
        if (q == Q - 1) dst[4] = 0.0f;
      }
    }
    __syncthreads();
    typedef const float __attribute__((address_space(1)))* gptr0;
    gptr0 pw = (gptr0)p.pre_wp, pb = (gptr0)p.pre_bias;
    // (pre_cin <= 16: one k-chunk, 9 taps.)  All 9 tap fragments of a tile are requested at once, the next tile's while
    // this tile's MFMAs run.
    auto load9 = [&](int o, f32x4 (&af)[9]) {
      const int oo = o < kc ? o : 0;                          // past the end: a valid tile again (unused)
#pragma unroll
      for (int t = 0; t < 9; ++t)
        af[t] = *reinterpret_cast<const f32x4 __attribute__((address_space(1)))*>(pw + ((size_t)oo * 9 + t) * 256 + lane * 4);
    };
    int zoff[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int lin = 16 * pt + i, pr = lin / W, pc = lin % W;
      zoff[pt] = (4 * g) * CSz + (pr + 1) * WPz + pc + 1;
    }
    f32x4 af[2][9];
    load9(wave, af[0]);
    auto pre_tile = [&](int o, const f32x4 (&a)[9]) {
      f32x4 acc[PT];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) acc[pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) acc[pt] = img_mfma(a[t][r], zin[zoff[pt] + r * CSz + dy * WPz + dx], acc[pt]);
        }
      }
      img_drain(acc);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int lin = 16 * pt + i;                         // main buffer (no halo): [channel][IMG_R * W]
        const bool in_img = r0 + lin / W < H;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = 16 * o + 4 * g + r;
          lds[co * CS + lin] = (co < p.cin && in_img) ? fmaxf(acc[pt][r] + pb[co < p.cin ? co : 0], 0.0f) : 0.0f;
        }
      }
    };
    for (int o = wave; o < kc; o += 2 * IMG_WAVES) {         // the main convolution's input tiles, two per round
      load9(o + IMG_WAVES, af[1]);
      pre_tile(o, af[0]);
      load9(o + 2 * IMG_WAVES, af[0]);
      if (o + IMG_WAVES < kc) pre_tile(o + IMG_WAVES, af[1]);
    }
  }
  // ---- stage the strip (+ halo, zero padded) of the input channels [cb, cb + cn): one image row = W/4 16-byte loads
  const int c_chunk = (p.c_chunk > 0 && p.c_chunk < cin_pad) ? p.c_chunk : cin_pad;
  auto stage_channels = [&](int cb, int cn) {
    const float* src = p.in + (int64_t)n * p.in_img;
    constexpr int Q = W / 4;                                 // float4s per image row
    const int q = threadIdx.x % Q, rid = threadIdx.x / Q;
    constexpr int ROWS_PER_PASS = 64 * IMG_WAVES / Q;
    for (int idx = rid; idx < cn * RP; idx += ROWS_PER_PASS) {
      const int cl = idx / RP, rr = idx - cl * RP, ci = cb + cl;
      const int row = r0 + rr - HALO;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ci < p.cin && row >= 0 && row < H) v = *reinterpret_cast<const f32x4*>(src + ((int64_t)ci * H + row) * W + 4 * q);
      float* dst = lds + cl * CS + rr * WP + HALO + 4 * q;
      dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
      if (HALO && q == 0) dst[-1] = 0.0f;                    // left / right zero columns
      if (HALO && q == Q - 1) dst[4] = 0.0f;
    }
  };
  if (!(KS == 1 && p.pre_in != nullptr)) stage_channels(0, c_chunk);
  __syncthreads();

  // pixel of lane i in tile pt: linear index 16 pt + i inside the strip (row-major over IMG_R x W)
  int boff[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int lin = 16 * pt + i, pr = lin / W, pc = lin % W;
    boff[pt] = (4 * g) * CS + (pr + HALO) * WP + pc + HALO;
  }

  const bool splitk = OT < IMG_WAVES;                        // few output tiles: the waves split the contraction instead
  const int T_all = taps * kc;                               // iterations (tap, chunk) of one output tile
  gptr wp = (gptr)p.wp;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  // one output tile over the iterations [t_begin, t_end) with stride t_step: acc[pt] += ...
  // (iterations t = (tap, c) over the kcc 16-channel groups staged at the moment, the first of which is group cb16 of the
  // convolution: the weight fragment of an iteration is (tap, cb16 + c) of the tile)
  auto run_tile = [&](int o, int t_begin, int t_end, int t_step, f32x4 (&acc)[PT], int cb16 = 0, int kcc_ = -1) {
    const int kcc = kcc_ < 0 ? kc : kcc_;
    const gptr wo = wp + ((size_t)o * T_all) * 256 + lane * 4;
    auto widx = [&](int t) { const int tap = t / kcc; return tap * kc + cb16 + (t - tap * kcc); };
    f32x4 ring[IMG_PD];
    int tl = t_begin;
#pragma unroll
    for (int j = 0; j < IMG_PD; ++j) {                       // unconditional: past the end re-reads the first fragment
      const int tt = tl < t_end ? tl : 0;
      ring[j] = *reinterpret_cast<const f32x4 __attribute__((address_space(1)))*>(wo + (size_t)widx(tt) * 256);
      tl += t_step;
    }
    auto body = [&](const f32x4& a, int t) {
      const int tap = t / kcc, c = t - tap * kcc;
      const int dy = tap / KS - half, dx = tap % KS - half;
      const float* b0 = lds + (16 * c) * CS + dy * WP + dx;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[pt] = img_mfma(a[r], b0[boff[pt] + r * CS], acc[pt]);
      }
    };
    int t = t_begin;
    for (; t + (IMG_PD - 1) * t_step < t_end; ) {
#pragma unroll
      for (int j = 0; j < IMG_PD; ++j) {
        body(ring[j], t);
        const int tt = tl < t_end ? tl : 0;
        ring[j] = *reinterpret_cast<const f32x4 __attribute__((address_space(1)))*>(wo + (size_t)widx(tt) * 256);
        tl += t_step;
        t += t_step;
      }
    }
#pragma unroll
    for (int j = 0; j < IMG_PD - 1; ++j) {
      if (t < t_end) {
        body(ring[j], t);
        t += t_step;
      }
    }
  };

  float ld = 0.0f;
  // epilogue of one (output tile o, pixel tile pt): lane (i,g) holds channels 16o+4g+r of pixel 16pt+i
  auto epilogue = [&](int o, int pt, const f32x4& acc) {
    gptr bias = (gptr)p.bias;
    const int lin = 16 * pt + i, pr = lin / W, pc = lin % W;
    const int row = r0 + pr;
    const bool in_img = row < H;
    const bool valid = row < p.Hv && pc < p.Wv;            // outside the map: stores write zero, couplings and log-dets skip
    const int64_t pix = (int64_t)row * W + pc;
    if constexpr (EPI == EPI_RELU || EPI == EPI_STORE) {
      float* out = p.out + (int64_t)n * p.out_img;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 16 * o + 4 * g + r;
        float v = acc[r] + bias[co];
        if (EPI == EPI_RELU) v = fmaxf(v, 0.0f);
        if (co < p.cout && in_img) out[(int64_t)co * H * W + pix] = valid ? v : 0.0f;
      }
    } else if constexpr (EPI == EPI_COUPLE_ADD || EPI == EPI_COUPLE_ADD_INV) {
      float* st = p.st + (int64_t)n * p.st_img;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 16 * o + 4 * g + r;
        const float h = acc[r] + bias[co];
        if (co < p.cout && valid) st[(int64_t)co * H * W + pix] += (EPI == EPI_COUPLE_ADD) ? h : -h;   // models/glow.py:328-329, :349-350
      }
    } else {
      // "cross" rows: (2j, 2j+1) = (shift_j, raw_j) for the coupling, (mean_j, log-var_j) for the Split2d prior
      float* st = p.st + (int64_t)n * p.st_img;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int co = 16 * o + 4 * g + 2 * q, j = co >> 1;
        if (co + 1 < p.cout && valid) {
          const float h0 = acc[2 * q] + bias[co], h1 = acc[2 * q + 1] + bias[co + 1];
          float* zp = st + (int64_t)j * H * W + pix;
          const float z2 = *zp;
          if constexpr (EPI == EPI_COUPLE_AFFINE) {
            const float e = __expf(-(h1 + 2.0f));                   // scale = sigmoid(raw + 2), models/glow.py:333
            const float sc = 1.0f / (1.0f + e);
            *zp = (z2 + h0) * sc;                                   // models/glow.py:334-335
            ld += -log1pf(e);                                       // log(scale), models/glow.py:338
          } else if constexpr (EPI == EPI_COUPLE_AFFINE_INV) {
            const float e = __expf(-(h1 + 2.0f));                   // models/glow.py:352-355: z2 / scale - shift
            *zp = z2 * (1.0f + e) - h0;
          } else if constexpr (EPI == EPI_SPLIT_INV) {              // models/layers.py:695-699: the slot holds eps on entry
            *zp = h0 + __expf(h1) * p.temperature * z2;
          } else {                                                  // Split2d: log_normal_diag(z2; mean, log-var)
            const float dlt = z2 - h0;
            ld += -0.5f * (h1 + dlt * dlt * __expf(-h1));           // utils/distributions.py:14, models/layers.py:703
          }
        }
      }
    }
  };

  if (!splitk) {
    const int o_per = (OT + p.o_split - 1) / p.o_split, o_end = min(OT, (osp + 1) * o_per);
    for (int o = osp * o_per + wave; o < o_end; o += IMG_WAVES) {
      f32x4 acc[PT];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) acc[pt] = zero;
      run_tile(o, 0, T_all, 1, acc);
      img_drain(acc);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) epilogue(o, pt, acc[pt]);
    }
  } else {
    // every wave: ALL (<= 3) output tiles over its share of the (tap, chunk) iterations; the partial tiles meet in LDS
    // once the strip is dead, and the (tile, pixel tile) epilogues are dealt round-robin to the waves
    constexpr int MAXO = IMG_WAVES - 1;
    f32x4 part[MAXO][PT];
#pragma unroll
    for (int o = 0; o < MAXO; ++o)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) part[o][pt] = zero;
    for (int cb = 0; cb < cin_pad; cb += c_chunk) {          // (one pass unless the input channels are staged in halves)
      const int cn = cb + c_chunk <= cin_pad ? c_chunk : cin_pad - cb;
      if (cb > 0) {
        __syncthreads();                                     // everybody is done with the previous channels
        stage_channels(cb, cn);
        __syncthreads();
      }
#pragma unroll
      for (int o = 0; o < MAXO; ++o)
        if (o < OT) run_tile(o, wave, taps * (cn >> 4), IMG_WAVES, part[o], cb >> 4, cn >> 4);
    }
#pragma unroll
    for (int o = 0; o < MAXO; ++o)
      if (o < OT) img_drain(part[o]);
    __syncthreads();                                         // nobody reads the strip any more
    f32x4* red = reinterpret_cast<f32x4*>(lds);              // [wave][o][pt][64]
#pragma unroll
    for (int o = 0; o < MAXO; ++o)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
        if (o < OT) red[((wave * MAXO + o) * PT + pt) * 64 + lane] = part[o][pt];
    __syncthreads();
    for (int q = wave; q < OT * PT; q += IMG_WAVES) {
      const int o = q / PT, pt = q % PT;
      f32x4 acc = zero;
#pragma unroll
      for (int w = 0; w < IMG_WAVES; ++w) {
        const f32x4 v = red[((w * MAXO + o) * PT + pt) * 64 + lane];
        acc += v;
      }
      epilogue(o, pt, acc);
    }
  }

  if constexpr (EPI == EPI_COUPLE_AFFINE || EPI == EPI_SPLIT) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) ld += __shfl_xor(ld, m);
    if (lane == 0) atomicAdd(p.ldj + n, ld);
  }
}

template <int EPI, int PT, int KS>
__global__ void __launch_bounds__(64 * IMG_WAVES) img_conv_kernel(const ConvLaunch p) {
  const int osp = blockIdx.x % p.o_split, bid = blockIdx.x / p.o_split;
  img_conv_body<EPI, PT, KS>(p, bid / p.n_strips, bid % p.n_strips, osp);
}

// ---- dequantise + logit + first squeeze (models/glow.py:125-179; utils/utilities.py:107-119) -------------------
// x (n, C, Hi, Wi) in [0,1] (+ uniform noise or null) -> squeezed logits (n, 4C, H/2, W/2) in H x W STORAGE (Hi <= H, Wi <= W:
// a 28 x 28 input lives in the top-left 14 x 14 of 16 x 16 maps, zero outside); ldj[n] = its log-det over the real pixels.
__device__ __forceinline__ void img_pre_body(const float* __restrict__ x, const float* __restrict__ noise, float* __restrict__ out,
                                             float* __restrict__ ldj, int C, int H, int W, int Hi, int Wi, float bounds, float ld_const, const int n) {
  const int chw = C * H * W, xchw = C * Hi * Wi;
  const float* xi = x + (int64_t)n * xchw;
  float* oi = out + (int64_t)n * chw;
  const float soft_c = log1pf((1.0f - bounds) / bounds);     // softplus(log(1-b) - log(b))
  float ld = 0.0f;
  const int H2 = H / 2, W2 = W / 2;
  for (int e = threadIdx.x; e < chw; e += 256) {             // e: an element of the squeezed storage (oc, y2, x2)
    const int oc = e / (H2 * W2), rem = e % (H2 * W2), y2 = rem / W2, x2 = rem % W2;
    const int c = oc >> 2, y = 2 * y2 + ((oc >> 1) & 1), xx = 2 * x2 + (oc & 1);
    float logit = 0.0f;
    if (y < Hi && xx < Wi) {
      const int src = (c * Hi + y) * Wi + xx;
      float v = xi[src];
      v = (255.0f * v + (noise ? noise[(int64_t)n * xchw + src] : 0.0f)) / 256.0f;    // models/glow.py:136
      v = ((v * 2.0f - 1.0f) * bounds + 1.0f) * 0.5f;                                   // models/glow.py:164-168
      logit = logf(v) - logf(1.0f - v);                                                 // :171
      // softplus(l) + softplus(-l) = -log(v) - log(1-v) for l = logit(v)
      ld += -logf(v) - logf(1.0f - v) - soft_c;                                         // :174-175
    }
    oi[e] = logit;
  }
  __shared__ float red[256];
  red[threadIdx.x] = ld;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) ldj[n] = red[0] + ld_const;
  __syncthreads();                                           // (red is reused by the caller's next phase in the repair kernel)
}
__global__ void __launch_bounds__(256) img_pre_kernel(const float* __restrict__ x, const float* __restrict__ noise, float* __restrict__ out,
                                                      float* __restrict__ ldj, int C, int H, int W, int Hi, int Wi, float bounds, float ld_const) {
  img_pre_body(x, noise, out, ldj, C, H, W, Hi, Wi, bounds, ld_const, (int)blockIdx.x);
}

// ---- ActNorm2d + invertible 1x1 convolution / Permute2d of a FlowStep (models/glow.py:319-322; layers.py:488-533, 756-796):
// out[co][pixel] = sum_ci Weff[co][ci] in[ci][pixel] + beff[co], one thread per pixel, the C inputs in registers, an ordered
// fmaf chain per output (exact f32).  3 MB in, 3 MB out per 256-image launch at the 16-wide level: the implicit-GEMM kernel
// (img_conv_kernel<EPI_STORE, ., 1>: 1024 workgroups that stage a strip in LDS for 12 x 12 MFMAs) took 7.6 us for it, 13 % of the
// path's GPU time for its two levels.  CP: channels padded to the instantiation (4 .. 64).
template <int CP>
__global__ void __launch_bounds__(64) img_mix_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ wb /* [C][CP] | [C] */,
                                                     int C, int HW, int W, int Hv, int Wv, int64_t total /* n HW */) {
  const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;          // (one wave per workgroup: a 64-image batch still fills the chip)
  if (t >= total) return;
  const int64_t n = t / HW;
  const int pix = (int)(t - n * HW);
  const float* xi = in + n * (int64_t)C * HW + pix;
  float* oi = out + n * (int64_t)C * HW + pix;
  float v[CP];
#pragma unroll
  for (int ci = 0; ci < CP; ++ci) v[ci] = ci < C ? xi[(int64_t)ci * HW] : 0.0f;
  const bool valid = pix / W < Hv && pix % W < Wv;           // outside the map proper the state stays zero
  // the matrix is the same for every lane: rows padded to CP with zeros, read through the scalar cache (s_load), no LDS
  // (a staged copy in LDS cost one dependent ds_read per multiply: 25 us for C = 24)
  // (four rows per trip: their scalar loads are in flight together -- one row per trip waited ~0.3 us for each)
  for (int co0 = 0; co0 < C; co0 += 4) {
    float acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = co0 + q < C ? co0 + q : C - 1;
      const float* wr = wb + co * CP;
      acc[q] = wb[C * CP + co];
#pragma unroll
      for (int ci = 0; ci < CP; ++ci) acc[q] = fmaf(wr[ci], v[ci], acc[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (co0 + q < C) oi[(int64_t)(co0 + q) * HW] = valid ? acc[q] : 0.0f;
  }
}
static int mix_pad(int C) { return C <= 4 ? 4 : C <= 8 ? 8 : C <= 12 ? 12 : C <= 16 ? 16 : C <= 24 ? 24 : C <= 32 ? 32 : C <= 48 ? 48 : 64; }
static bool launch_mix(const float* in, float* out, const float* wb, int C, int H, int W, int Hv, int Wv, int64_t n, hipStream_t s) {
  const int64_t total = n * H * W;
  const dim3 grid((unsigned)((total + 63) / 64)), blk(64);
#define GBNF_MIX(CPV) hipLaunchKernelGGL((img_mix_kernel<CPV>), grid, blk, 0, s, in, out, wb, C, H * W, W, Hv, Wv, total)
  if (C <= 4) GBNF_MIX(4);
  else if (C <= 8) GBNF_MIX(8);
  else if (C <= 12) GBNF_MIX(12);
  else if (C <= 16) GBNF_MIX(16);
  else if (C <= 24) GBNF_MIX(24);
  else if (C <= 32) GBNF_MIX(32);
  else if (C <= 48) GBNF_MIX(48);
  else if (C <= 64) GBNF_MIX(64);
  else return false;
#undef GBNF_MIX
  return true;
}

// squeeze2d of the first `C` channels of (n, Cin_total, H, W) -> (n, 4C, Ho, Wo) storage with Ho >= H / 2, Wo >= W / 2 (the kernels
// work on maps at least 8 wide: the 4 x 4 map of a third level lives in the corner of 8 x 8 storage), zero outside
__device__ __forceinline__ void img_squeeze_body(const float* __restrict__ in, int64_t in_img, float* __restrict__ out, int C, int H, int W, int Ho,
                                                 int Wo, const int n) {
  const int ochw = 4 * C * Ho * Wo;
  const float* xi = in + (int64_t)n * in_img;
  float* oi = out + (int64_t)n * ochw;
  for (int e = threadIdx.x; e < ochw; e += 256) {
    const int oc = e / (Ho * Wo), rem = e % (Ho * Wo), y2 = rem / Wo, x2 = rem % Wo;
    const int c = oc >> 2, y = 2 * y2 + ((oc >> 1) & 1), xx = 2 * x2 + (oc & 1);
    oi[e] = (y < H && xx < W) ? xi[((int64_t)c * H + y) * W + xx] : 0.0f;
  }
}
__global__ void __launch_bounds__(256) img_squeeze_kernel(const float* __restrict__ in, int64_t in_img, float* __restrict__ out, int C, int H, int W,
                                                          int Ho, int Wo) {
  img_squeeze_body(in, in_img, out, C, H, W, Ho, Wo, (int)blockIdx.x);
}

// unsqueeze2d (utils/utilities.py:121-135) of (n, 4C, H/2, W/2) into the first C channels of (n, Ctot, H, W); the next
// `C_eps` channels are filled from eps (n, C_eps, H, W) (the standard-normal draws Split2d's reverse scales in place)
// (body: xi / oi / ei = this image's input, output and eps)
__device__ __forceinline__ void img_unsqueeze_body(const float* __restrict__ xi, float* __restrict__ oi, int C, int H, int W,
                                                   const float* __restrict__ ei, int C_eps, int Hv, int Wv, int Hs, int Ws) {
  const int chw = C * H * W;                                 // in: (4C, Hs, Ws) storage with Hs >= H / 2, Ws >= W / 2
  for (int e = threadIdx.x; e < chw; e += 256) {
    const int c = e / (H * W), rem = e % (H * W), y = rem / W, xx = rem % W;
    const int ic = c * 4 + (y & 1) * 2 + (xx & 1);
    oi[e] = xi[((int64_t)ic * Hs + (y >> 1)) * Ws + (xx >> 1)];      // (zero outside the map proper, like its input)
  }
  if (ei != nullptr) {                                       // eps (n, C_eps, Hv, Wv) has the map's own size: zero around it
    const int ehw = C_eps * H * W;
    for (int e = threadIdx.x; e < ehw; e += 256) {
      const int c = e / (H * W), rem = e % (H * W), y = rem / W, xx = rem % W;
      oi[chw + e] = (y < Hv && xx < Wv) ? ei[((int64_t)c * Hv + y) * Wv + xx] : 0.0f;
    }
  }
}
__global__ void __launch_bounds__(256) img_unsqueeze_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t out_img, int C, int H,
                                                            int W, const float* __restrict__ eps, int C_eps, int Hv, int Wv, int Hs, int Ws) {
  const int64_t n = blockIdx.x;
  img_unsqueeze_body(in + n * 4 * C * Hs * Ws, out + n * out_img, C, H, W, eps ? eps + n * C_eps * Hv * Wv : nullptr, C_eps, Hv, Wv, Hs, Ws);
}

// z (n, C, Hv, Wv) of the map's own size into the top-left corner of zeroed (n, C, H, W) storage
__device__ __forceinline__ void img_embed_body(const float* __restrict__ zi, float* __restrict__ oi, int C, int H, int W, int Hv, int Wv) {
  const int chw = C * H * W;
  for (int e = threadIdx.x; e < chw; e += 256) {
    const int c = e / (H * W), rem = e % (H * W), y = rem / W, xx = rem % W;
    oi[e] = (y < Hv && xx < Wv) ? zi[((int64_t)c * Hv + y) * Wv + xx] : 0.0f;
  }
}
__global__ void __launch_bounds__(256) img_embed_kernel(const float* __restrict__ z, float* __restrict__ out, int C, int H, int W, int Hv, int Wv) {
  const int64_t n = blockIdx.x;
  img_embed_body(z + n * C * Hv * Wv, out + n * C * H * W, C, H, W, Hv, Wv);
}

// the last unsqueeze + to_logits(reverse=True) (models/glow.py:151-158): (n, 4C, H/2, W/2) logits -> x (n, C, H, W)
__device__ __forceinline__ void img_post_body(const float* __restrict__ xi, float* __restrict__ oi, int C, int H, int W, int Hi, int Wi, float bounds) {
  const int xchw = C * Hi * Wi;                              // storage H x W, x (C, Hi, Wi)
  for (int e = threadIdx.x; e < xchw; e += 256) {
    const int c = e / (Hi * Wi), rem = e % (Hi * Wi), y = rem / Wi, xx = rem % Wi;
    const int ic = c * 4 + (y & 1) * 2 + (xx & 1);
    const float v = xi[((int64_t)ic * (H / 2) + (y >> 1)) * (W / 2) + (xx >> 1)];
    const float sg = 1.0f / (__expf(-v) + 1.0f);
    oi[e] = ((sg * 2.0f - 1.0f) / bounds + 1.0f) * 0.5f;
  }
}
__global__ void __launch_bounds__(256) img_post_kernel(const float* __restrict__ in, float* __restrict__ x, int C, int H, int W, int Hi, int Wi,
                                                       float bounds) {
  const int64_t n = blockIdx.x;
  img_post_body(in + n * C * H * W, x + n * C * Hi * Wi, C, H, W, Hi, Wi, bounds);
}

// ll[n] = sum -0.5 (log-var + (z - mean)^2 exp(-log-var)) + ldj[n]  with per-channel prior constants (Glow.prior on zeros:
// Conv2dZeros(0) = bias * exp(3 logs), models/glow.py:62-84); optional copies of z / mean / log-var.
// (body: zi = this image's state, zo = its compact z output or null; returns the prior's log-density sum in every thread)
__device__ __forceinline__ float img_final_body(const float* __restrict__ zi, const float* __restrict__ prior /* [2C] */, float* __restrict__ zo,
                                                int C, int H, int W, int Hv, int Wv) {
  float acc = 0.0f;
  for (int e = threadIdx.x; e < C * Hv * Wv; e += 256) {      // e: an element of the compact z (C, Hv, Wv); storage is H x W
    const int c = e / (Hv * Wv), rem = e % (Hv * Wv), y = rem / Wv, xx = rem % Wv;
    const float mu = prior[c], lv = prior[C + c], v = zi[((int64_t)c * H + y) * W + xx], dlt = v - mu;
    acc += -0.5f * (lv + dlt * dlt * __expf(-lv));
    if (zo) zo[e] = v;
  }
  __shared__ float red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}
__global__ void __launch_bounds__(256) img_final_kernel(const float* __restrict__ z, int64_t z_img, const float* __restrict__ prior /* [2C] */,
                                                        const float* __restrict__ ldj, float* __restrict__ ll, float* __restrict__ z_out,
                                                        int C, int H, int W, int Hv, int Wv) {
  const int64_t n = blockIdx.x;
  const float r = img_final_body(z + n * z_img, prior, z_out ? z_out + n * C * Hv * Wv : nullptr, C, H, W, Hv, Wv);
  if (threadIdx.x == 0 && ll) ll[n] = r + ldj[n];
}



// ---- data-dependent ActNorm2d initialisation (models/layers.py:473-486): per-channel statistics over (n, the map proper) of the
// tensor that reaches an ActNorm2d: mean[c] = mean(t), var[c] = mean((t - mean)^2) (two passes, like the reference; one block
// per channel, fixed-order sums: bit-reproducible).  t: (n, C, H, W) storage with image stride t_img.
__global__ void __launch_bounds__(256) img_channel_stats_kernel(const float* __restrict__ t, int64_t t_img, int64_t n, int H, int W, int Hv, int Wv,
                                                                float* __restrict__ mean_out, float* __restrict__ var_out) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  const int64_t per = (int64_t)Hv * Wv, total = n * per;
  auto at = [&](int64_t e) {
    const int64_t img = e / per;
    const int rem = (int)(e - img * per), y = rem / Wv, xx = rem - y * Wv;
    return t[img * t_img + ((int64_t)c * H + y) * W + xx];
  };
  auto block_sum = [&](float v) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
  };
  float a = 0.0f;
  for (int64_t e = threadIdx.x; e < total; e += 256) a += at(e);
  const float mean = block_sum(a) / (float)total;
  float b = 0.0f;
  for (int64_t e = threadIdx.x; e < total; e += 256) {
    const float d = at(e) - mean;
    b += d * d;
  }
  const float var = block_sum(b) / (float)total;
  if (threadIdx.x == 0) {
    mean_out[c] = mean;
    var_out[c] = var;
  }
}

// ---- numerics protocol of the split-f16 coupling nets ---------------------------------------------------------------
// A fused coupling-net workgroup that meets an operand beyond the fp16 range raises mark[image].  Behind the f16x3 pass, on the
// same stream, in the same call (round 5: no capacity, no "armed from the next call on"):
//   img_compact_kernel   the marked images' indices, in order, into list[0 .. count) (every one of them); on the handle's first
//                        launch and every `check_every`-th after it, also up to IMG_CHECK_MAX unmarked images into the check list
//   img_repair_kernel    ONE launch that returns at once when its list is empty.  Otherwise workgroup b walks image list[b]
//                        through the WHOLE exact-f32 sequence -- every launch of the exact-f32 pass as a recorded op (RepairOp),
//                        the image's tensors in a workgroup-private workspace, the strips of a convolution one after the other
//                        through the very body the stand-alone kernels run (img_conv_body) -- and writes z / ldj / ll (or x, on
//                        the way back) of that image over what the split-f16 pass left there.  Slow per image (one workgroup
//                        instead of the chip) and rare by construction: a model that drives ordinary data out of range is
//                        demoted by the create-time probe.
//   the same kernel in CHECK mode in front of it: the check list's images on exact f32, compared with the split-f16 result in
//   place; a difference beyond the tolerance raises the handle's device word `force_all` (the repair launch of this and every
//   later call then re-evaluates ALL images: nothing of a failed mode reaches the caller, also under graph replay) and a pinned
//   host word (later calls run the exact-f32 kernels directly).
constexpr int IMG_CHECK_MAX = 2;
constexpr int IMG_REPAIR_WGS = 256;        // workgroups (= private workspaces) of a repair launch
constexpr uintptr_t IMG_REC_BASE = 0x100000000000ull;       // workspace address the ops are recorded against

enum { ROP_PRE = 0, ROP_CONV, ROP_SQUEEZE, ROP_FINAL, ROP_EMBED, ROP_UNSQUEEZE, ROP_POST };
struct RepairOp {
  int kind, epi, pt, ks;
  ConvLaunch p;           // ROP_CONV: workspace pointers relative to IMG_REC_BASE, parameter pointers absolute
  int64_t a, b, c;        // the other ops: float offsets into the private workspace (source, destination, log-det slot)
  int i[8];
  float f[2];
  const float* prior;     // ROP_FINAL
  int64_t eps_img, eps_lvl;   // ROP_UNSQUEEZE: eps of this level = eps + eps_lvl * n_batch + image * eps_img (0 / 0: none)
};

struct RepairArgs {
  const RepairOp* ops;
  int n_ops, check;
  const unsigned* list;       // image indices
  const unsigned* count;      // entries of list
  unsigned* force_all;        // device word: non-zero = every image of the call (repair mode reads, check mode raises)
  const unsigned* mark;       // check mode: (n,) marks of the split-f16 pass
  int64_t n;                  // images of the call
  float* ws;
  int64_t ws_stride;          // floats of one workgroup's workspace
  const float* x; const float* noise; float* z; float* ldj; float* ll;      // forward
  const float* zin; const float* eps; float* xout; float temperature;      // z -> x
  int64_t xchw, zsz;
  float tol;
  unsigned* host_words;       // pinned: [2] checks completed, [3] checks failed, [4] worst relative difference (float bits)
};

__global__ void __launch_bounds__(256) img_compact_kernel(const unsigned* __restrict__ mark, int n, unsigned* __restrict__ list,
                                                          unsigned* __restrict__ count /* [0] marked, [1] check entries */,
                                                          unsigned* __restrict__ check_list, unsigned* __restrict__ dev_state /* [0] launches */,
                                                          unsigned* host_words, int check_every) {
  // in order: one wave scans (n is a batch size: a few thousand at most)
  if (threadIdx.x >= 64) return;
  unsigned base = 0, nchk = 0;
  bool check = false;
  if (check_list != nullptr) {
    unsigned serial = 0;
    if (threadIdx.x == 0) { serial = dev_state[0]; dev_state[0] = serial + 1u; }
    serial = __shfl(serial, 0);
    check = check_every >= 0 && (serial == 0u || (check_every > 0 && serial % (unsigned)check_every == 0u));
  }
  for (int b0 = 0; b0 < n; b0 += 64) {
    const int k = b0 + (int)threadIdx.x;
    const bool m = k < n && mark[k] != 0u;
    const unsigned long long bal = __ballot(m);
    if (m) list[base + __popcll(bal & ((1ull << threadIdx.x) - 1ull))] = (unsigned)k;
    base += __popcll(bal);
    if (check && nchk < (unsigned)IMG_CHECK_MAX) {
      const unsigned long long free = __ballot(k < n && !m);
      const unsigned before = __popcll(free & ((1ull << threadIdx.x) - 1ull));
      if (k < n && !m && nchk + before < (unsigned)IMG_CHECK_MAX) check_list[nchk + before] = (unsigned)k;
      nchk = min(nchk + (unsigned)__popcll(free), (unsigned)IMG_CHECK_MAX);
    }
  }
  if (threadIdx.x == 0) {
    count[0] = base;
    count[1] = nchk;
    if (base != 0u && host_words != nullptr) {
      atomicAdd_system(host_words, 1u);             // calls that marked an image
      atomicAdd_system(host_words + 1, base);       // images re-evaluated on exact f32
    }
  }
}

template <int EPI>
__device__ __forceinline__ void img_repair_conv(const ConvLaunch& p, const int pt, const int ks) {
  for (int strip = 0; strip < p.n_strips; ++strip) {
    if (pt == 4 && ks == 3) img_conv_body<EPI, 4, 3>(p, 0, strip, 0);
    else if (pt == 4) { if constexpr (EPI == EPI_RELU || EPI == EPI_STORE) img_conv_body<EPI, 4, 1>(p, 0, strip, 0); }
    else if (ks == 3) img_conv_body<EPI, 2, 3>(p, 0, strip, 0);
    else { if constexpr (EPI == EPI_RELU || EPI == EPI_STORE) img_conv_body<EPI, 2, 1>(p, 0, strip, 0); }
    __syncthreads();                                         // the next strip restages the LDS
  }
}

// INV: the z -> x sequence (its own instantiation: each carries only the epilogues its direction uses)
template <bool INV>
__global__ void __launch_bounds__(64 * IMG_WAVES) img_repair_kernel(const RepairArgs a) {
  unsigned cnt = *a.count;
  bool all = false;
  if (!a.check && *a.force_all != 0u) { cnt = (unsigned)a.n; all = true; }
  if (cnt == 0u) return;
  float* wsb = a.ws + (int64_t)blockIdx.x * a.ws_stride;
  const int64_t delta = reinterpret_cast<const char*>(wsb) - reinterpret_cast<const char*>(IMG_REC_BASE);
  auto rb = [&](const float* q) -> float* { return q ? reinterpret_cast<float*>(reinterpret_cast<uintptr_t>(q) + delta) : nullptr; };
  for (unsigned b = blockIdx.x; b < cnt; b += gridDim.x) {
    const int64_t img = all ? (int64_t)b : (int64_t)a.list[b];
    for (int k = 0; k < a.n_ops; ++k) {
      const RepairOp& op = a.ops[k];
      switch (op.kind) {
        case ROP_CONV: {
          ConvLaunch p = op.p;
          p.in = rb(p.in); p.out = rb(p.out); p.st = rb(p.st); p.pre_in = rb(p.pre_in); p.ldj = rb(p.ldj);
          p.temperature = a.temperature;
          if constexpr (!INV) {
            if (op.epi == EPI_RELU) img_repair_conv<EPI_RELU>(p, op.pt, op.ks);
            else if (op.epi == EPI_STORE) img_repair_conv<EPI_STORE>(p, op.pt, op.ks);
            else if (op.epi == EPI_COUPLE_AFFINE) img_repair_conv<EPI_COUPLE_AFFINE>(p, op.pt, op.ks);
            else if (op.epi == EPI_COUPLE_ADD) img_repair_conv<EPI_COUPLE_ADD>(p, op.pt, op.ks);
            else img_repair_conv<EPI_SPLIT>(p, op.pt, op.ks);
          } else {
            if (op.epi == EPI_RELU) img_repair_conv<EPI_RELU>(p, op.pt, op.ks);
            else if (op.epi == EPI_STORE) img_repair_conv<EPI_STORE>(p, op.pt, op.ks);
            else if (op.epi == EPI_COUPLE_AFFINE_INV) img_repair_conv<EPI_COUPLE_AFFINE_INV>(p, op.pt, op.ks);
            else if (op.epi == EPI_COUPLE_ADD_INV) img_repair_conv<EPI_COUPLE_ADD_INV>(p, op.pt, op.ks);
            else img_repair_conv<EPI_SPLIT_INV>(p, op.pt, op.ks);
          }
          break;
        }
        case ROP_PRE:
          if constexpr (!INV)
            img_pre_body(a.x + img * a.xchw, a.noise ? a.noise + img * a.xchw : nullptr, wsb + op.b, wsb + op.c, op.i[0], op.i[1], op.i[2],
                         op.i[3], op.i[4], op.f[0], op.f[1], 0);
          break;
        case ROP_SQUEEZE:
          if constexpr (!INV) img_squeeze_body(wsb + op.a, 0, wsb + op.b, op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], 0);
          break;
        case ROP_FINAL:
          if constexpr (!INV) {
            float* zo = (a.check || a.z == nullptr) ? nullptr : a.z + img * a.zsz;
            const float r = img_final_body(wsb + op.a, op.prior, zo, op.i[0], op.i[1], op.i[2], op.i[3], op.i[4]);
            if (threadIdx.x == 0) {
              const float ldv = __hip_atomic_load(wsb + op.c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), llv = r + ldv;
              if (!a.check) {
                a.ldj[img] = ldv;
                if (a.ll) a.ll[img] = llv;
              } else if (a.mark[img] == 0u) {                 // (a marked image is repaired anyway)
                const float got = a.ll ? a.ll[img] : a.ldj[img], want = a.ll ? llv : ldv;
                const float err = fabsf(got - want) / fmaxf(fabsf(want), 1.0f);
                atomicAdd_system(a.host_words + 2, 1u);
                atomicMax_system(a.host_words + 4, __float_as_uint(err == err ? err : INFINITY));
                if (!(err <= a.tol)) {
                  atomicExch(a.force_all, 1u);
                  atomicAdd_system(a.host_words + 3, 1u);
                }
              }
            }
          }
          break;
        case ROP_EMBED:
          if constexpr (INV) img_embed_body(a.zin + img * a.zsz, wsb + op.b, op.i[0], op.i[1], op.i[2], op.i[3], op.i[4]);
          break;
        case ROP_UNSQUEEZE:
          if constexpr (INV)
            img_unsqueeze_body(wsb + op.a, wsb + op.b, op.i[0], op.i[1], op.i[2],
                               op.eps_img ? a.eps + op.eps_lvl * a.n + img * op.eps_img : nullptr, op.i[3], op.i[4], op.i[5], op.i[6], op.i[7]);
          break;
        case ROP_POST:
          if constexpr (INV) img_post_body(wsb + op.a, a.xout + img * a.xchw, op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.f[0]);
          break;
        default: break;
      }
      __threadfence();                                       // the next op reads what this one wrote (other waves, global memory)
      __syncthreads();
    }
  }
}

}  // namespace gbnf

#include "gbnf_image_hx3.hip.h"

namespace gbnf {

// ---------------------------------------------------------------------------------------------------------------
// host side: packing + launch sequence
// ---------------------------------------------------------------------------------------------------------------
struct PackedConv {
  int cin = 0, cout = 0, ks = 1;
  size_t w_off = 0, b_off = 0;   // float offsets into the handle's parameter blob
  size_t x_off = 0;              // f16x3 fragments (wide convolutions; first 3x3: taps folded into k), 0 = none
  size_t k_off = 0;              // first 3x3: im2col offset table
  int kc = 0;                    // first 3x3: 32-wide chunks of the folded contraction
  size_t p_off = 0;              // 1x1 mixes: the plain row-major [cout][cin] matrix and [cout] bias behind it (img_mix_kernel), 0 = none
};

}  // namespace gbnf

using namespace gbnf;

struct gbnf_image_flow {
  int C = 0, H = 0, W = 0, L = 0, K = 0, hidden = 0, additive = 0, depth = 0;   // H x W: the STORAGE of the input (32 x 32)
  int64_t state_img = 0;                     // floats of the largest state tensor of one image over the levels (storage)
  int Hi = 0, Wi = 0;                        // the input proper (Hi <= H, Wi <= W: 32 x 32, 28 x 28, 28 x 20 ...)
  float bounds = 0.9f;
  double ld_const = 0;                       // dequantisation + every ActNorm2d / invconv log-det (per image)
  std::vector<int> level_steps;              // FlowSteps per level
  std::vector<PackedConv> mix, split;        // [sum steps] / [L-1]
  std::vector<PackedConv> mix_inv;           // [sum steps]: (invconv / Permute2d)^-1 then ActNorm2d reverse, as one C x C matrix
  std::vector<std::vector<PackedConv>> net;  // [L*K][depth + 2]
  size_t prior_off = 0;                      // [2 Cz] top prior (mean, log-var per channel)
  float* blob_dev = nullptr;
  int zC = 0, zH = 0, zW = 0;
  double macs = 0;                           // multiply-adds per image
  int math_mode = 0;                         // GBNF_MATH_F32 or GBNF_MATH_F16X3 (the coupling nets' two wide convolutions)
  int chp = 0;                               // hidden width padded to 32 (split-f16 activation layout)
  // numerics protocol of the split-f16 coupling nets (round 4)
  float probe_rel_err = 0.0f;                // create-time probe: largest relative difference of ll, f16x3 vs exact f32, on 4 images
  bool probed = false;
  // pinned, device-visible: [0] calls that marked an image, [1] images re-evaluated on exact f32, [2] on-data checks completed,
  // [3] on-data checks failed (non-zero: the handle runs on the exact-f32 kernels from the next call on), [4] worst relative
  // difference a check has seen (float bits)
  unsigned* host_words = nullptr;
  unsigned* dev_state = nullptr;             // device: [0] launches so far (the check schedule), [1] force_all (a check failed)
  RepairOp* ops_dev = nullptr;               // the exact-f32 sequence of ONE image as recorded ops: forward [0, n_ops_fwd), then z -> x
  int n_ops_fwd = 0, n_ops_inv = 0;
  size_t repair_lds_fwd = 0, repair_lds_inv = 0;
  bool on_data_demoted() const { return host_words != nullptr && ((volatile unsigned*)host_words)[3] != 0u; }
};

namespace {

struct Packer {
  std::vector<float> blob;
  // weight w[co][ci][ky][kx] * row_scale[co] -> fragments [OT][taps][KC][64 lanes][4]; bias[co] -> [OT*16]
  PackedConv add(const float* w, int cout, int cin, int ks, const std::vector<double>& row_scale, const std::vector<double>& bias) {
    PackedConv pc;
    pc.cin = cin; pc.cout = cout; pc.ks = ks;
    const int OT = (cout + 15) / 16, KC = (cin + 15) / 16, taps = ks * ks;
    pc.w_off = blob.size();
    blob.resize(blob.size() + (size_t)OT * taps * KC * 256, 0.0f);
    for (int o = 0; o < OT; ++o)
      for (int tap = 0; tap < taps; ++tap)
        for (int c = 0; c < KC; ++c)
          for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 4; ++r) {
              const int co = 16 * o + (lane & 15), ci = 16 * c + 4 * (lane >> 4) + r;
              float v = 0.0f;
              if (co < cout && ci < cin) v = (float)(row_scale[co] * (double)w[((size_t)co * cin + ci) * taps + tap]);
              blob[pc.w_off + ((((size_t)o * taps + tap) * KC + c) * 64 + lane) * 4 + r] = v;
            }
    pc.b_off = blob.size();
    blob.resize(blob.size() + (size_t)OT * 16, 0.0f);
    for (int co = 0; co < cout; ++co) blob[pc.b_off + co] = (float)bias[co];
    return pc;
  }
};

unsigned short f16_bits(float x) {
  const _Float16 h = static_cast<_Float16>(x);      // round to nearest even
  unsigned short b;
  std::memcpy(&b, &h, 2);
  return b;
}

// f16x3 A fragments of a convolution: [o][tap][c][hi|mid][64 lanes][8 halfs], k = 32c + 8g + j, c over the input channels
// padded to cin_pad (the activation layout's padded hidden width: zero fragments past cin); returns the float offset
size_t pack_hx3(Packer& P, const float* w, int cout, int cin, int cin_pad, int ks, const std::vector<double>& row_scale) {
  const int OT = (cout + 15) / 16, KC = (cin_pad + 31) / 32, taps = ks * ks;
  while (P.blob.size() % 4) P.blob.push_back(0.0f);                     // 16-byte aligned fragments
  const size_t off = P.blob.size();
  P.blob.resize(off + (size_t)OT * taps * KC * 2 * 64 * 4, 0.0f);
  unsigned short* dst = reinterpret_cast<unsigned short*>(P.blob.data() + off);
  for (int o = 0; o < OT; ++o)
    for (int tap = 0; tap < taps; ++tap)
      for (int c = 0; c < KC; ++c)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int co = 16 * o + (lane & 15), ci = 32 * c + 8 * (lane >> 4) + j;
            float v = 0.0f;
            if (co < cout && ci < cin) v = (float)(row_scale[co] * (double)w[((size_t)co * cin + ci) * taps + tap]);
            const _Float16 hi = static_cast<_Float16>(v);
            const float mid = v - (float)hi;
            const size_t base = ((((size_t)o * taps + tap) * KC + c) * 2) * 64 * 8;
            dst[base + (size_t)lane * 8 + j] = f16_bits((float)hi);
            dst[base + 64 * 8 + (size_t)lane * 8 + j] = f16_bits(mid);
          }
  return off;
}

// The first 3x3 (<= 16 input channels) as one GEMM with its taps folded into the contraction, k = tap * cin + ci:
// f16x3 A fragments [tile][kc][hi|mid][64 lanes][8 halfs] followed by the im2col offset table int[32 * kc]
// (float offset ci * CSz + (dy+1) * WPz + (dx+1) from the corner of the pixel's 3x3 window in the staged strip; -1 = padding).
size_t pack_folded(Packer& P, const float* w, int cout_pad16, int cout, int cin, int W, const std::vector<double>& row_scale,
                   size_t* koff_out, int* kc_out) {
  const int tiles = cout_pad16 / 16, K = 9 * cin, KC = (K + 31) / 32;
  while (P.blob.size() % 4) P.blob.push_back(0.0f);
  const size_t off = P.blob.size();
  P.blob.resize(off + (size_t)tiles * KC * 2 * 64 * 4, 0.0f);
  unsigned short* dst = reinterpret_cast<unsigned short*>(P.blob.data() + off);
  for (int o = 0; o < tiles; ++o)
    for (int c = 0; c < KC; ++c)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int co = 16 * o + (lane & 15), k = 32 * c + 8 * (lane >> 4) + j;
          float v = 0.0f;
          if (co < cout && k < K) {
            const int tap = k / cin, ci = k % cin;
            v = (float)(row_scale[co] * (double)w[((size_t)co * cin + ci) * 9 + tap]);
          }
          const _Float16 hi = static_cast<_Float16>(v);
          const size_t base = (((size_t)o * KC + c) * 2) * 64 * 8;
          dst[base + (size_t)lane * 8 + j] = f16_bits((float)hi);
          dst[base + 64 * 8 + (size_t)lane * 8 + j] = f16_bits(v - (float)hi);
        }
  const size_t koff = P.blob.size();
  P.blob.resize(koff + (size_t)32 * KC, 0.0f);
  const int WPz = W + 2, CSz = (IMG_R + 2) * WPz;
  for (int k = 0; k < 32 * KC; ++k) {
    int v = -1;
    if (k < K) {
      const int tap = k / cin, ci = k % cin;
      v = ci * CSz + (tap / 3) * WPz + (tap % 3);
    }
    std::memcpy(&P.blob[koff + k], &v, 4);
  }
  *koff_out = koff;
  *kc_out = KC;
  return off;
}

// log|det A| of an n x n matrix (double, partial pivoting)
double logabsdet(std::vector<double> a, int n) {
  double acc = 0;
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int r = k + 1; r < n; ++r)
      if (std::fabs(a[r * n + k]) > std::fabs(a[piv * n + k])) piv = r;
    if (a[piv * n + k] == 0.0) return -INFINITY;
    if (piv != k)
      for (int c = 0; c < n; ++c) std::swap(a[k * n + c], a[piv * n + c]);
    acc += std::log(std::fabs(a[k * n + k]));
    for (int r = k + 1; r < n; ++r) {
      const double f = a[r * n + k] / a[k * n + k];
      for (int c = k; c < n; ++c) a[r * n + c] -= f * a[k * n + c];
    }
  }
  return acc;
}

// A^-1 of an n x n matrix (double, Gauss-Jordan with partial pivoting); false if singular
bool invert(std::vector<double> a, int n, std::vector<double>* out) {
  std::vector<double>& b = *out;
  b.assign((size_t)n * n, 0.0);
  for (int k = 0; k < n; ++k) b[(size_t)k * n + k] = 1.0;
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int r = k + 1; r < n; ++r)
      if (std::fabs(a[r * n + k]) > std::fabs(a[piv * n + k])) piv = r;
    if (a[piv * n + k] == 0.0) return false;
    if (piv != k)
      for (int c = 0; c < n; ++c) { std::swap(a[k * n + c], a[piv * n + c]); std::swap(b[k * n + c], b[piv * n + c]); }
    const double d = 1.0 / a[k * n + k];
    for (int c = 0; c < n; ++c) { a[k * n + c] *= d; b[k * n + c] *= d; }
    for (int r = 0; r < n; ++r) {
      if (r == k) continue;
      const double f = a[r * n + k];
      if (f == 0.0) continue;
      for (int c = 0; c < n; ++c) { a[r * n + c] -= f * a[k * n + c]; b[r * n + c] -= f * b[k * n + c]; }
    }
  }
  return true;
}

int check_conv(const gbnf_conv& c, int cin, int cout, int ks, bool want_an, bool want_zeros, const char* what) {
  if (!c.weight) return fail(GBNF_ERR_INVALID, "%s: null weight", what);
  if (c.in_channels != cin || c.out_channels != cout || c.kernel_size != ks)
    return fail(GBNF_ERR_INVALID, "%s: is %dx%d k=%d, expected %dx%d k=%d", what, c.out_channels, c.in_channels, c.kernel_size,
                cout, cin, ks);
  if (want_an && (!c.actnorm_bias || !c.actnorm_logs)) return fail(GBNF_ERR_INVALID, "%s: needs its ActNorm2d arrays", what);
  if (want_zeros && (!c.bias || !c.logs)) return fail(GBNF_ERR_INVALID, "%s: Conv2dZeros needs bias and logs", what);
  return GBNF_OK;
}

// Conv2d + ActNorm2d: (conv + an_bias) * exp(an_logs);  Conv2dZeros: (conv + bias) * exp(3 logs)
PackedConv pack_conv(Packer& P, const gbnf_conv& c, std::vector<double>* scale_out = nullptr) {
  std::vector<double> scale(c.out_channels, 1.0), bias(c.out_channels, 0.0);
  for (int co = 0; co < c.out_channels; ++co) {
    double b = c.bias ? c.bias[co] : 0.0;
    if (c.actnorm_bias) {
      scale[co] = std::exp((double)c.actnorm_logs[co]);
      b = (b + c.actnorm_bias[co]) * scale[co];
    }
    if (c.logs) {
      const double s3 = std::exp(3.0 * (double)c.logs[co]);
      scale[co] *= s3;
      b *= s3;
    }
    bias[co] = b;
  }
  if (scale_out) *scale_out = scale;
  return P.add(c.weight, c.out_channels, c.in_channels, c.kernel_size, scale, bias);
}

#ifdef GBNF_IMG_STAMPS
unsigned long long* g_img_stamp_buf = nullptr;
#endif

// image_forward_impl / image_inverse_impl in RECORD mode: no launches; every launch of the exact-f32 sequence for one image
// becomes a RepairOp (workspace = IMG_REC_BASE) for img_repair_kernel
struct Recorder {
  std::vector<RepairOp> ops;
  size_t lds = 0;
  const float* ws = nullptr;
  RepairOp& add(int kind) {
    ops.emplace_back();
    std::memset(&ops.back(), 0, sizeof(RepairOp));
    ops.back().kind = kind;
    return ops.back();
  }
};

template <int EPI>
void launch_conv(const ConvLaunch& p, int n, hipStream_t s, Recorder* rec = nullptr) {
  const int halo = p.ks >> 1;
  const int per_ch = (IMG_R + 2 * halo) * (p.W + 2 * halo);
  const int kc = (p.cin + 15) / 16, OT = (p.cout + 15) / 16, PT = IMG_R * p.W / 16;
  size_t lds = (size_t)kc * 16 * per_ch * 4;
  ConvLaunch q = p;
  q.c_chunk = 0;
  if (lds > 150 * 1024 && OT < IMG_WAVES && p.pre_in == nullptr) {      // a 3 x 3 from 512 channels: its strip in two halves
    q.c_chunk = ((kc + 1) / 2) * 16;
    lds = (size_t)q.c_chunk * per_ch * 4;
  }
  if (p.pre_in != nullptr) lds += (size_t)((p.pre_cin + 15) / 16) * 16 * (IMG_R + 2) * (p.W + 2) * 4;
  if (OT < IMG_WAVES) lds = std::max(lds, (size_t)IMG_WAVES * (IMG_WAVES - 1) * PT * 64 * 16);
  q.o_split = 1;
  if (OT >= 2 * IMG_WAVES && (int64_t)n * p.n_strips < 512) q.o_split = 2;      // < 2 workgroups per CU otherwise
  if (OT >= 4 * IMG_WAVES && (int64_t)n * p.n_strips < 256) q.o_split = 4;
  if (rec != nullptr) {
    RepairOp& op = rec->add(ROP_CONV);
    q.o_split = 1;
    op.epi = EPI; op.pt = PT; op.ks = p.ks; op.p = q;
    rec->lds = std::max(rec->lds, lds);
    return;
  }
  const dim3 grid((unsigned)(n * p.n_strips * q.o_split)), blk(64 * IMG_WAVES);
  if (PT == 4 && p.ks == 3) hipLaunchKernelGGL((img_conv_kernel<EPI, 4, 3>), grid, blk, lds, s, q);
  else if (PT == 4) hipLaunchKernelGGL((img_conv_kernel<EPI, 4, 1>), grid, blk, lds, s, q);
  else if (p.ks == 3) hipLaunchKernelGGL((img_conv_kernel<EPI, 2, 3>), grid, blk, lds, s, q);
  else hipLaunchKernelGGL((img_conv_kernel<EPI, 2, 1>), grid, blk, lds, s, q);
}

template <int EPI>
hipError_t allow_lds() {
  hipError_t e = hipSuccess;
  const void* fns[4] = {(const void*)img_conv_kernel<EPI, 4, 3>, (const void*)img_conv_kernel<EPI, 4, 1>,
                        (const void*)img_conv_kernel<EPI, 2, 3>, (const void*)img_conv_kernel<EPI, 2, 1>};
  for (int k = 0; k < 4 && e == hipSuccess; ++k)
    e = hipFuncSetAttribute(fns[k], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return e;
}

}  // namespace

extern "C" {

#ifdef GBNF_IMG_STAMPS
void gbnf_debug_set_image_stamp_buffer(unsigned long long* p) { g_img_stamp_buf = p; }
#endif

static int64_t image_state_floats(const gbnf_image_flow* f, int64_t n);
struct ActNormStats;
static int image_record_repair_ops(gbnf_image_flow* f);
static int image_forward_impl(const gbnf_image_flow* f, const float* x, const float* noise, int64_t n, float* z, float* ldj, float* ll,
                              float* workspace, hipStream_t s, bool force_f32, unsigned* mark, ActNormStats* stats = nullptr,
                              Recorder* rec = nullptr);

static int image_probe(gbnf_image_flow* f) {
  constexpr int PN = 4;
  const int64_t chw = (int64_t)f->C * f->Hi * f->Wi;
  std::vector<float> host((size_t)2 * PN * chw);
  uint64_t st = 0x9E3779B97F4A7C15ull;
  for (float& v : host) {                      // splitmix64 -> [0, 1): pixels and dequantisation noise
    st += 0x9E3779B97F4A7C15ull;
    uint64_t zz = st;
    zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ull;
    zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBull;
    zz ^= zz >> 31;
    v = (float)((double)(zz >> 11) / 9007199254740992.0);
  }
  float* dev = nullptr;
  const int64_t wsf = image_state_floats(f, PN);
  const size_t total = (size_t)(2 * PN * chw + 4 * PN + wsf) * 4;
  if (hipMalloc((void**)&dev, total) != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_image_flow_create: probe allocation failed");
  float* x = dev; float* noise = dev + PN * chw; float* ldj = noise + PN * chw; float* lla = ldj + PN; float* ldjb = lla + PN;
  float* llb = ldjb + PN; float* ws = llb + PN;
  int rc = GBNF_OK;
  hipError_t e = hipMemcpy(dev, host.data(), host.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = image_forward_impl(f, x, noise, PN, nullptr, ldj, lla, ws, nullptr, false, nullptr);
    if (!rc) rc = image_forward_impl(f, x, noise, PN, nullptr, ldjb, llb, ws, nullptr, true, nullptr);
    float a[PN], b[PN];
    if (!rc) e = hipMemcpy(a, lla, sizeof(a), hipMemcpyDeviceToHost);
    if (!rc && e == hipSuccess) e = hipMemcpy(b, llb, sizeof(b), hipMemcpyDeviceToHost);
    if (!rc && e == hipSuccess) {
      float worst = 0.0f;
      for (int k = 0; k < PN; ++k) {
        const float err = std::fabs(a[k] - b[k]) / std::fmax(std::fabs(b[k]), 1.0f);
        worst = (err == err) ? std::fmax(worst, err) : INFINITY;
      }
      f->probe_rel_err = worst;
      f->probed = true;
      if (!(worst <= 2.5e-6f)) f->math_mode = GBNF_MATH_F32;
    }
  }
  (void)hipFree(dev);
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_image_flow_create: probe failed: %s", hipGetErrorString(e));
  return rc;
}

int gbnf_image_flow_create(const gbnf_image_flow_desc* d, gbnf_image_flow** out) { return gbnf_image_flow_create_mode(d, GBNF_MATH_DEFAULT, out); }

int gbnf_image_flow_create_mode(const gbnf_image_flow_desc* d, int32_t math_mode, gbnf_image_flow** out) {
  if (!out) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_create: out is null");
  if (math_mode != GBNF_MATH_DEFAULT && math_mode != GBNF_MATH_F32 && math_mode != GBNF_MATH_F16X3)
    return fail(GBNF_ERR_INVALID, "gbnf_image_flow_create_mode: math mode %d (image components: DEFAULT, F32, F16X3)", math_mode);
  *out = nullptr;
  if (!d || !d->levels) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_create: null descriptor");
  if (d->n_levels < 1 || d->n_levels > 4) return fail(GBNF_ERR_UNSUPPORTED, "n_levels=%d outside [1,4]", d->n_levels);
  if (d->coupling != GBNF_COUPLING_AFFINE && d->coupling != GBNF_COUPLING_ADDITIVE)
    return fail(GBNF_ERR_INVALID, "unknown coupling %d", d->coupling);
  // The kernels work on 16- and 8-wide square maps (a 32 x 32 input after one and two squeezes).  A smaller input -- the
  // reference's 1 x 28 x 28 (MNIST, Omniglot, Caltech) and 1 x 28 x 20 (Frey faces) loaders, utils/load_data.py:389-529 -- lives in
  // the top-left corner of the same 32 x 32 STORAGE: every tensor in HBM is zero outside the map proper (that IS the map's 'same'
  // padding), every kernel that stores masks what lies outside, log-dets and priors sum over the map proper.
  int C = d->channels, H = 32, W = 32, Hv = d->height, Wv = d->width;
  if (C < 1 || Hv < 2 || Wv < 2 || Hv > H || Wv > W)
    return fail(GBNF_ERR_UNSUPPORTED, "input %dx%dx%d: at most 32 x 32 pixels", C, Hv, Wv);
  if (!(d->bounds > 0.5f && d->bounds < 1.0f)) return fail(GBNF_ERR_INVALID, "bounds must be in (0.5, 1)");
  auto* f = new gbnf_image_flow();
  f->C = C; f->H = H; f->W = W; f->Hi = Hv; f->Wi = Wv; f->L = d->n_levels; f->state_img = (int64_t)C * H * W; f->additive = d->coupling == GBNF_COUPLING_ADDITIVE;
  f->bounds = d->bounds; f->hidden = d->hidden;
  const char* env_math = getenv("GBNF_MATH");                // "f32": exact-f32 MFMA everywhere (tuning / test knob)
  // (hidden widths above 256 -- the usual Glow width is 512 -- run on the exact-f32 convolutions: the split-f16 kernels keep a
  // strip's hidden activation in LDS, 150 KB at 256 channels)
  // (round 5: the fused split-f16 kernel takes hidden widths to 512 -- the usual Glow width -- in two halves of the hidden channels,
  //  gbnf_image_net.hip; such a width is padded to a multiple of 64: both halves whole 32-channel chunks)
  const bool use_hx3 = math_mode != GBNF_MATH_F32 && !(env_math && !strcmp(env_math, "f32")) && d->hidden <= 512;
  f->math_mode = use_hx3 ? GBNF_MATH_F16X3 : GBNF_MATH_F32;
  f->chp = d->hidden > 256 ? (d->hidden + 63) / 64 * 64 : (d->hidden + 31) / 32 * 32;
  Packer P;
  double ld_const = -std::log(256.0) * C * Hv * Wv;            // dequantisation, models/glow.py:137
  char what[96];
  int rc = GBNF_OK;
  for (int l = 0; l < d->n_levels && rc == GBNF_OK; ++l) {
    const gbnf_image_level& lv = d->levels[l];
    if (Hv % 2 || Wv % 2) { rc = fail(GBNF_ERR_INVALID, "level %d: odd spatial size %d x %d", l, Hv, Wv); break; }
    C *= 4; H /= 2; W /= 2; Hv /= 2; Wv /= 2;
    if (H < 8) H = W = 8;                                        // (a third level's 4 x 4 map: the corner of 8 x 8 storage)
    f->state_img = std::max(f->state_img, (int64_t)C * H * W);
    if (W != 16 && W != 8) { rc = fail(GBNF_ERR_UNSUPPORTED, "level %d works on %dx%d maps; compiled for widths 16 and 8 (32x32 input, <= 2 levels)", l, H, W); break; }
    if (H % IMG_R) { rc = fail(GBNF_ERR_UNSUPPORTED, "level %d: height %d not a multiple of %d", l, H, IMG_R); break; }
    if (C > 64) { rc = fail(GBNF_ERR_UNSUPPORTED, "level %d: %d channels > 64", l, C); break; }
    f->level_steps.push_back(lv.n_steps);
    if (lv.n_steps < 1 || !lv.steps) { rc = fail(GBNF_ERR_INVALID, "level %d: no steps", l); break; }
    const int c1 = C / 2, c2 = C - c1;
    for (int k = 0; k < lv.n_steps && rc == GBNF_OK; ++k) {
      const gbnf_image_step& st = lv.steps[k];
      if (!st.actnorm_bias || !st.actnorm_logs || (!st.perm_weight && !st.perm_indices)) {
        rc = fail(GBNF_ERR_INVALID, "level %d step %d: null actnorm / permutation", l, k); break;
      }
      // ActNorm2d then invconv / Permute2d as one C x C matrix
      std::vector<double> wperm((size_t)C * C, 0.0);
      if (st.perm_weight) {
        for (int e = 0; e < C * C; ++e) wperm[e] = st.perm_weight[e];
        ld_const += logabsdet(wperm, C) * Hv * Wv;               // models/layers.py:756, 790
      } else {
        std::vector<char> seen(C, 0);
        for (int j = 0; j < C; ++j) {
          const int64_t m = st.perm_indices[j];
          if (m < 0 || m >= C || seen[m]) { rc = fail(GBNF_ERR_INVALID, "level %d step %d: perm_indices is not a permutation", l, k); break; }
          seen[m] = 1;
          wperm[(size_t)j * C + m] = 1.0;                        // z[:, j] = y[:, indices[j]], models/layers.py:675-677
        }
        if (rc) break;
      }
      std::vector<float> weff((size_t)C * C);
      std::vector<double> ones(C, 1.0), beff(C, 0.0);
      for (int j = 0; j < C; ++j) {
        double b = 0;
        for (int m = 0; m < C; ++m) {
          const double e = std::exp((double)st.actnorm_logs[m]);
          const double v = wperm[(size_t)j * C + m] * e;
          weff[(size_t)j * C + m] = (float)v;
          b += v * st.actnorm_bias[m];
        }
        beff[j] = b;
      }
      for (int m = 0; m < C; ++m) ld_const += (double)st.actnorm_logs[m] * Hv * Wv;    // models/layers.py:506-508
      f->mix.push_back(P.add(weff.data(), C, C, 1, ones, beff));
      {
        PackedConv& pm = f->mix.back();                          // ... and the plain matrix + bias for img_mix_kernel
        pm.p_off = P.blob.size();
        const int CP = mix_pad(C);                               // rows padded with zeros to the kernel's instantiation
        for (int j = 0; j < C; ++j)
          for (int m = 0; m < CP; ++m) P.blob.push_back(m < C ? weff[(size_t)j * C + m] : 0.0f);
        for (int j = 0; j < C; ++j) P.blob.push_back((float)beff[j]);
      }
      {
        // the way back (FlowStep.decode, models/glow.py:360-364): x = exp(-logs) * (W^-1 y) - bias
        std::vector<double> winv;
        if (!invert(wperm, C, &winv)) { rc = fail(GBNF_ERR_INVALID, "level %d step %d: singular 1x1 convolution", l, k); break; }
        std::vector<float> wi((size_t)C * C);
        std::vector<double> rs(C), bi(C);
        for (int m = 0; m < C; ++m) {
          rs[m] = std::exp(-(double)st.actnorm_logs[m]);
          bi[m] = -(double)st.actnorm_bias[m];
          for (int j = 0; j < C; ++j) wi[(size_t)m * C + j] = (float)winv[(size_t)m * C + j];
        }
        f->mix_inv.push_back(P.add(wi.data(), C, C, 1, rs, bi));
        {
          // ... and the plain matrix diag(rs) . W^-1 with its bias for img_mix_kernel
          PackedConv& pm = f->mix_inv.back();
          pm.p_off = P.blob.size();
          const int CP = mix_pad(C);
          for (int m = 0; m < C; ++m)
            for (int j = 0; j < CP; ++j) P.blob.push_back(j < C ? (float)(rs[m] * (double)wi[(size_t)m * C + j]) : 0.0f);
          for (int m = 0; m < C; ++m) P.blob.push_back((float)bi[m]);
        }
      }
      // ConvNet: Conv2d 3x3 (+ActNorm2d), [Conv2d 1x1 (+ActNorm2d)] x depth, Conv2dZeros 3x3
      if (st.n_convs < 2 || st.n_convs > 5 || !st.convs) { rc = fail(GBNF_ERR_INVALID, "level %d step %d: needs 2..5 convolutions", l, k); break; }
      const int hdim = st.convs[0].out_channels;
      if (hdim < 1 || hdim > 512) { rc = fail(GBNF_ERR_UNSUPPORTED, "hidden width %d outside [1,512]", hdim); break; }
      std::vector<PackedConv> net;
      for (int q = 0; q < st.n_convs && rc == GBNF_OK; ++q) {
        const bool first = q == 0, last = q == st.n_convs - 1;
        snprintf(what, sizeof(what), "level %d step %d conv %d", l, k, q);
        rc = check_conv(st.convs[q], first ? c1 : hdim, last ? (f->additive ? c2 : 2 * c2) : hdim, (first || last) ? 3 : 1, !last, last, what);
        if (rc == GBNF_OK) {
          std::vector<double> sc;
          PackedConv pc = pack_conv(P, st.convs[q], &sc);
          // split-f16 path (coupling_network_depth 1 at every width, 0 / 2 to h = 256 since round 6; first 3x3 with <= 24 input channels -- the 48-channel third level of a
          // 3 x 32 x 32 input, round 5 -- its folded contraction 9 cin <= 224 = 7 chunks): extra fragment sets
          if (use_hx3 && c1 <= 24 && (st.n_convs == 3 || ((st.n_convs == 2 || st.n_convs == 4) && hdim <= 256))) {
            const int chp = hdim > 256 ? (hdim + 63) / 64 * 64 : (hdim + 31) / 32 * 32;
            if (q == 0) pc.x_off = pack_folded(P, st.convs[q].weight, chp, hdim, c1, W, sc, &pc.k_off, &pc.kc);
            else pc.x_off = pack_hx3(P, st.convs[q].weight, st.convs[q].out_channels, st.convs[q].in_channels, chp,
                                     st.convs[q].kernel_size, sc);
          }
          net.push_back(pc);
          f->macs += (double)st.convs[q].in_channels * st.convs[q].out_channels * st.convs[q].kernel_size * st.convs[q].kernel_size * Hv * Wv;
        }
      }
      f->net.push_back(net);
      f->depth = st.n_convs - 2;
      f->macs += (double)C * C * Hv * Wv;
    }
    if (rc) break;
    if (l < d->n_levels - 1) {
      if (!lv.split_prior) { rc = fail(GBNF_ERR_INVALID, "level %d: missing Split2d prior", l); break; }
      snprintf(what, sizeof(what), "level %d split prior", l);
      rc = check_conv(*lv.split_prior, C / 2, C, 3, false, true, what);
      if (rc) break;
      if (C % 2) { rc = fail(GBNF_ERR_UNSUPPORTED, "Split2d on an odd channel count"); break; }
      f->split.push_back(pack_conv(P, *lv.split_prior));
      f->macs += (double)(C / 2) * C * 9 * Hv * Wv;
      C /= 2;
    }
  }
  if (rc == GBNF_OK) {
    f->zC = C; f->zH = Hv; f->zW = Wv;
    // top prior: Conv2dZeros applied to zeros = bias * exp(3 logs) per channel (models/glow.py:62-84); else zeros
    f->prior_off = P.blob.size();
    P.blob.resize(P.blob.size() + 2 * (size_t)C, 0.0f);
    if (d->learn_top) {
      const gbnf_conv& t = *d->learn_top;
      if (t.out_channels != 2 * C || !t.bias || !t.logs) rc = fail(GBNF_ERR_INVALID, "learn_top must be a Conv2dZeros with %d outputs", 2 * C);
      else
        for (int c = 0; c < 2 * C; ++c) P.blob[f->prior_off + c] = (float)((double)t.bias[c] * std::exp(3.0 * (double)t.logs[c]));
    }
  }
  if (rc == GBNF_OK) {
    f->ld_const = ld_const;
    hipError_t e = hipMalloc((void**)&f->blob_dev, P.blob.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(f->blob_dev, P.blob.data(), P.blob.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = allow_lds<EPI_RELU>();
    if (e == hipSuccess) e = allow_lds<EPI_STORE>();
    if (e == hipSuccess) e = allow_lds<EPI_COUPLE_AFFINE>();
    if (e == hipSuccess) e = allow_lds<EPI_COUPLE_ADD>();
    if (e == hipSuccess) e = allow_lds<EPI_SPLIT>();
    if (e == hipSuccess) e = allow_lds<EPI_COUPLE_AFFINE_INV>();
    if (e == hipSuccess) e = allow_lds<EPI_COUPLE_ADD_INV>();
    if (e == hipSuccess) e = allow_lds<EPI_SPLIT_INV>();
    {
      const void* fns[10] = {(const void*)img_mid_hx3_kernel<4, 2>, (const void*)img_mid_hx3_kernel<4, 4>,
                             (const void*)img_mid_hx3_kernel<4, 5>, (const void*)img_mid_hx3_kernel<2, 2>,
                             (const void*)img_mid_hx3_kernel<2, 4>, (const void*)img_mid_hx3_kernel<2, 5>,
                             (const void*)img_last_hx3_kernel<EPI_COUPLE_AFFINE, 4>, (const void*)img_last_hx3_kernel<EPI_COUPLE_AFFINE, 2>,
                             (const void*)img_last_hx3_kernel<EPI_COUPLE_ADD, 4>, (const void*)img_last_hx3_kernel<EPI_COUPLE_ADD, 2>};
      for (int k = 0; k < 10 && e == hipSuccess; ++k)
        e = hipFuncSetAttribute(fns[k], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (e == hipSuccess && f->math_mode == GBNF_MATH_F16X3) {
      e = hipHostMalloc((void**)&f->host_words, 8 * sizeof(unsigned), hipHostMallocMapped);
      if (e == hipSuccess) std::memset(f->host_words, 0, 8 * sizeof(unsigned));
      if (e == hipSuccess) e = hipMalloc((void**)&f->dev_state, 2 * sizeof(unsigned));
      if (e == hipSuccess) e = hipMemset(f->dev_state, 0, 2 * sizeof(unsigned));
    }
    if (e != hipSuccess) rc = fail(GBNF_ERR_HIP, "gbnf_image_flow_create: %s", hipGetErrorString(e));
    if (rc == GBNF_OK && f->math_mode == GBNF_MATH_F16X3) rc = image_record_repair_ops(f);
  }
  // ---- create-time probe (as the tabular DEFAULT mode, gbnf_api.hip): 4 synthetic images through the split-f16 coupling nets
  //      and through the exact-f32 kernels; a handle whose log-likelihoods disagree beyond 2.5e-6 runs on exact f32 from now on
  if (rc == GBNF_OK && f->math_mode == GBNF_MATH_F16X3 && !getenv("GBNF_IMAGE_NO_PROBE")) rc = image_probe(f);
  if (rc != GBNF_OK) {
    gbnf_image_flow_destroy(f);
    return rc;
  }
  *out = f;
  return GBNF_OK;
}

int gbnf_image_flow_destroy(gbnf_image_flow* f) {
  if (!f) return GBNF_OK;
  if (f->blob_dev) (void)hipFree(f->blob_dev);
  if (f->host_words) (void)hipHostFree(f->host_words);
  if (f->dev_state) (void)hipFree(f->dev_state);
  if (f->ops_dev) (void)hipFree(f->ops_dev);
  delete f;
  return GBNF_OK;
}

int gbnf_image_flow_info(const gbnf_image_flow* f, int32_t* z_channels, int32_t* z_height, int32_t* z_width, double* macs_per_image) {
  if (!f) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_info: flow is null");
  if (z_channels) *z_channels = f->zC;
  if (z_height) *z_height = f->zH;
  if (z_width) *z_width = f->zW;
  if (macs_per_image) *macs_per_image = f->macs;
  return GBNF_OK;
}

int gbnf_image_flow_prior(const gbnf_image_flow* f, float* host) {
  if (!f || !host) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_prior: null argument");
  const hipError_t e = hipMemcpy(host, f->blob_dev + f->prior_off, 2 * (size_t)f->zC * 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_image_flow_prior: %s", hipGetErrorString(e));
  return GBNF_OK;
}

// floats of the state / hidden buffers of a batch of n images
static int64_t image_state_floats(const gbnf_image_flow* f, int64_t n) {
  const int64_t chw = f->state_img;
  const int64_t hid = (int64_t)f->chp * (f->H / 2) * (f->W / 2);
  return (2 * chw + 2 * hid) * n + 64;
}

// workspace behind the main batch's state buffers (split-f16 handles): marks | list | check list, counts | the repair
// workgroups' private workspaces
static int64_t image_repair_workgroups(int64_t n) { return n < IMG_REPAIR_WGS ? n : IMG_REPAIR_WGS; }
static int64_t image_mark_floats(int64_t n) { return (n + 63) / 64 * 64; }

int gbnf_image_flow_workspace_bytes(const gbnf_image_flow* f, int64_t n, int64_t* bytes) {
  if (!f || !bytes || n < 0) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_workspace_bytes: bad argument");
  int64_t floats = image_state_floats(f, n);
  if (f->math_mode == GBNF_MATH_F16X3) floats += 2 * image_mark_floats(n) + 64 + image_repair_workgroups(n) * image_state_floats(f, 1);
  *bytes = floats * 4 + 256;
  return GBNF_OK;
}

// One pass of the launch sequence over n images.  force_f32: every convolution on the exact-f32 kernels (the repair pass and
// handles whose probe failed); mark: (n,) per-image range marks raised by the split-f16 coupling nets, or null; rec (with
// force_f32, n = 1, workspace = IMG_REC_BASE, ldj = a workspace slot): record the sequence for img_repair_kernel, launch nothing.
// `stats` (exact-f32 passes only): stop at ActNorm2d number stats->index of the component (module order: a step's own ActNorm2d,
// then the one behind each Conv2d of its coupling net) and leave the per-channel statistics of the tensor that reaches it.
struct ActNormStats {
  int index;
  float* mean;
  float* var;
  int channels;      // out: channels of that ActNorm2d (-1: index past the last one)
};
static int image_forward_impl(const gbnf_image_flow* f, const float* x, const float* noise, int64_t n, float* z, float* ldj, float* ll,
                              float* workspace, hipStream_t s, bool force_f32, unsigned* mark, ActNormStats* stats, Recorder* rec) {
  const int64_t chw = f->state_img;
  int an_index = 0;                       // ActNorm2d counter (stats)
  if (stats != nullptr) stats->channels = -1;
  const int64_t hid = (int64_t)f->chp * (f->H / 2) * (f->W / 2);
  float* SA = workspace;
  float* SB = SA + chw * n;
  float* H1 = SB + chw * n;
  float* H2 = H1 + hid * n;
  const float* blob = f->blob_dev;

  int C = f->C * 4, H = f->H / 2, W = f->W / 2, Hv = f->Hi / 2, Wv = f->Wi / 2;
  const bool padded = f->Hi != f->H || f->Wi != f->W;       // the map proper is smaller than its storage
  if (rec != nullptr) {
    RepairOp& op = rec->add(ROP_PRE);
    op.b = SA - workspace; op.c = ldj - workspace;
    op.i[0] = f->C; op.i[1] = f->H; op.i[2] = f->W; op.i[3] = f->Hi; op.i[4] = f->Wi; op.f[0] = f->bounds; op.f[1] = (float)f->ld_const;
  } else {
    hipLaunchKernelGGL(img_pre_kernel, dim3((unsigned)n), dim3(256), 0, s, x, noise, SA, ldj, f->C, f->H, f->W, f->Hi, f->Wi, f->bounds,
                       (float)f->ld_const);
  }
  float* cur = SA;      // current state (n, C, H, W), image stride = C*H*W
  float* oth = SB;
  size_t step = 0;
  for (int l = 0; l < f->L; ++l) {
    const int64_t img = (int64_t)C * H * W;
    const int n_strips = H / IMG_R;
    const int c1 = C / 2;
    const int K = f->level_steps[l];
    for (int k = 0; k < K; ++k, ++step) {
      ConvLaunch p{};
      p.H = H; p.W = W; p.Hv = Hv; p.Wv = Wv; p.n_strips = n_strips; p.ldj = ldj;
      if (stats != nullptr && an_index++ == stats->index) {      // the step's own ActNorm2d: the state as it stands
        hipLaunchKernelGGL(img_channel_stats_kernel, dim3((unsigned)C), dim3(256), 0, s, (const float*)cur, img, n, H, W, Hv, Wv, stats->mean, stats->var);
        stats->channels = C;
        return hipGetLastError() == hipSuccess ? GBNF_OK : fail(GBNF_ERR_HIP, "gbnf_image_flow_actnorm_stats: launch failed");
      }
      // ActNorm2d + permutation: cur -> oth
      const PackedConv& m = f->mix[step];
      static const bool no_mix_kernel = getenv("GBNF_IMG_NO_MIX_KERNEL") != nullptr;     // diagnostic: the implicit-GEMM form
      // (below ~8 k pixels per launch a thread per pixel leaves most of the chip idle: the implicit-GEMM form spreads the channels)
      if (rec != nullptr || no_mix_kernel || m.p_off == 0 || n * H * W < 8192 || !launch_mix(cur, oth, blob + m.p_off, C, H, W, Hv, Wv, n, s)) {
        p.in = cur; p.in_img = img; p.wp = blob + m.w_off; p.bias = blob + m.b_off; p.out = oth; p.out_img = img;
        p.cin = C; p.cout = C; p.ks = 1;
        launch_conv<EPI_STORE>(p, (int)n, s, rec);
      }
      std::swap(cur, oth);
      // coupling net on the first half
      const std::vector<PackedConv>& net = f->net[step];
      // (round 6: coupling_network_depth 0 / 2 -- nets of 2 / 4 convolutions -- on the fused kernel too, hidden widths to 256)
      const size_t NC = net.size();
      const bool fusable = H == W && (W == 16 || W == 8) && (NC == 3 || ((NC == 2 || NC == 4) && f->chp <= 256)) && net[NC - 1].x_off != 0 &&
                           img_net_hx3_lds(W, f->chp, net[0].cin, net[0].kc, net[NC - 1].cout) != 0;
      // (a padded map runs the fused kernel or the exact-f32 convolutions: the two-kernel split form does not mask)
      // (hidden widths above 256 exist on the fused kernel only)
      // (more than 5 chunks of the folded first 3x3 -- more than 17 input channels -- exist on the fused kernel only)
      if (!force_f32 && net[NC - 1].x_off != 0 && (fusable || (NC == 3 && !padded && f->chp <= 256 && net[0].kc <= 5))) {
        // split-f16 path.  Round 4: the whole coupling net in ONE kernel where a workgroup can hold the hidden activation of its
        // rows (+ halo) in LDS: the 16 x 16 and 8 x 8 maps of a 32 x 32 input (img_net_hx3_kernel, gbnf_image_hx3.hip.h)
        static const bool no_fuse = getenv("GBNF_IMG_NO_FUSE") != nullptr;        // diagnostic: the round-2 two-kernel form
        if ((!no_fuse || padded) && fusable) {
          NetLaunch q{};
          q.pre_in = cur; q.pre_in_img = img; q.pre_wp = reinterpret_cast<const unsigned*>(blob + net[0].x_off);
          q.pre_bias = blob + net[0].b_off; q.pre_kc = net[0].kc; q.pre_cin = net[0].cin;
          q.pre_koff = reinterpret_cast<const int*>(blob + net[0].k_off);
          q.n_mid = (int)NC - 2;
          q.wp = reinterpret_cast<const unsigned*>(blob + net[1].x_off); q.bias = blob + net[1].b_off;
          if (NC == 4) { q.wp2 = reinterpret_cast<const unsigned*>(blob + net[2].x_off); q.bias2 = blob + net[2].b_off; }
          q.wp3 = reinterpret_cast<const unsigned*>(blob + net[NC - 1].x_off); q.bias3 = blob + net[NC - 1].b_off;
          q.st = cur + (int64_t)c1 * H * W; q.st_img = img; q.ldj = ldj;
          q.hid = net[0].cout; q.chp = f->chp; q.cout = net[NC - 1].cout; q.H = H; q.Hv = Hv; q.Wv = Wv;
          q.sat = reinterpret_cast<unsigned long long*>(gbnf::saturation_counter());
          q.mark = mark; q.only = nullptr;
#ifdef GBNF_IMG_STAMPS
          q.dbg = (W == GBNF_IMG_STAMPS) ? g_img_stamp_buf : nullptr;      // -DGBNF_IMG_STAMPS=16 | 8: which level is stamped
#endif
          const hipError_t le = img_net_hx3_launch(q, W, f->additive != 0, n, s);
          if (le != hipSuccess) return fail(GBNF_ERR_HIP, "img_net_hx3 launch failed: %s", hipGetErrorString(le));
          continue;
        }
      }
      if (!force_f32 && NC == 3 && net[1].x_off != 0 && !padded && f->chp <= 256 && net[0].kc <= 5) {      // the round-2 two-kernel form (depth 1 only)
        const int PT = IMG_R * W / 16, OT = (net[1].cout + 15) / 16;
        MidLaunch m1{};
        m1.pre_in = cur; m1.pre_in_img = img; m1.pre_wp = reinterpret_cast<const unsigned*>(blob + net[0].x_off);
        m1.pre_koff = reinterpret_cast<const int*>(blob + net[0].k_off); m1.pre_kc = net[0].kc;
        m1.pre_bias = blob + net[0].b_off;
        m1.wp = reinterpret_cast<const unsigned*>(blob + net[1].x_off); m1.bias = blob + net[1].b_off;
        m1.h2 = reinterpret_cast<unsigned*>(H1); m1.pre_cin = net[0].cin; m1.hid = net[1].cout; m1.chp = f->chp;
        m1.H = H; m1.n_strips = n_strips;
#ifdef GBNF_IMG_STAMPS
        m1.dbg = (PT == 4) ? g_img_stamp_buf : nullptr;
#endif
        m1.o_split = 1;
        if (OT >= 2 * IMG_WAVES && (int64_t)n * n_strips < 512) m1.o_split = 2;
        if (OT >= 4 * IMG_WAVES && (int64_t)n * n_strips < 256) m1.o_split = 4;
        const int pixb = 4 * f->chp + 16;
        const size_t lds1 = (size_t)IMG_R * W * pixb + 16 * (IMG_R + 2) * (W + 2) * 4;
        const dim3 g1((unsigned)(n * n_strips * m1.o_split)), blk(64 * IMG_WAVES);
        if (16 * OT != f->chp) (void)hipMemsetAsync(H1, 0, (size_t)hid * n * 4, s);      // pad channels of the split layout
        const int pk = m1.pre_kc <= 2 ? 2 : (m1.pre_kc <= 4 ? 4 : 5);
        if (PT == 4 && pk == 2) hipLaunchKernelGGL((img_mid_hx3_kernel<4, 2>), g1, blk, lds1, s, m1);
        else if (PT == 4 && pk == 4) hipLaunchKernelGGL((img_mid_hx3_kernel<4, 4>), g1, blk, lds1, s, m1);
        else if (PT == 4) hipLaunchKernelGGL((img_mid_hx3_kernel<4, 5>), g1, blk, lds1, s, m1);
        else if (pk == 2) hipLaunchKernelGGL((img_mid_hx3_kernel<2, 2>), g1, blk, lds1, s, m1);
        else if (pk == 4) hipLaunchKernelGGL((img_mid_hx3_kernel<2, 4>), g1, blk, lds1, s, m1);
        else hipLaunchKernelGGL((img_mid_hx3_kernel<2, 5>), g1, blk, lds1, s, m1);
        LastLaunch m2{};
        m2.h2 = reinterpret_cast<const unsigned*>(H1); m2.wp = reinterpret_cast<const unsigned*>(blob + net[2].x_off);
        m2.bias = blob + net[2].b_off; m2.st = cur + (int64_t)c1 * H * W; m2.st_img = img; m2.ldj = ldj;
        m2.chp = f->chp; m2.cout = net[2].cout; m2.H = H; m2.n_strips = n_strips;
        const size_t lds2 = std::max((size_t)(IMG_R + 2) * (W + 2) * pixb, (size_t)IMG_WAVES * (IMG_WAVES - 1) * PT * 64 * 16);
        const dim3 g2((unsigned)(n * n_strips));
        if (f->additive) {
          if (PT == 4) hipLaunchKernelGGL((img_last_hx3_kernel<EPI_COUPLE_ADD, 4>), g2, blk, lds2, s, m2);
          else hipLaunchKernelGGL((img_last_hx3_kernel<EPI_COUPLE_ADD, 2>), g2, blk, lds2, s, m2);
        } else {
          if (PT == 4) hipLaunchKernelGGL((img_last_hx3_kernel<EPI_COUPLE_AFFINE, 4>), g2, blk, lds2, s, m2);
          else hipLaunchKernelGGL((img_last_hx3_kernel<EPI_COUPLE_AFFINE, 2>), g2, blk, lds2, s, m2);
        }
        continue;
      }
      const float* hin = cur;
      int64_t hin_img = img;
      float* hb[2] = {H1, H2};
      if (stats != nullptr && stats->index < an_index + (int)net.size() - 1) {
        // an ActNorm2d inside this coupling net: the convolutions in front of it one by one (relu(ActNorm2d(conv)) each), then
        // the raw output of its own convolution -- its folded scale and bias are the identity while it is un-initialised
        const int tq = stats->index - an_index;
        for (int q = 0; q <= tq; ++q) {
          const PackedConv& c = net[q];
          p.in = hin; p.in_img = hin_img; p.wp = blob + c.w_off; p.bias = blob + c.b_off;
          p.out = hb[q & 1]; p.out_img = (int64_t)c.cout * H * W; p.cin = c.cin; p.cout = c.cout; p.ks = c.ks;
          p.pre_in = nullptr;
          if (q < tq) launch_conv<EPI_RELU>(p, (int)n, s);
          else launch_conv<EPI_STORE>(p, (int)n, s);
          hin = hb[q & 1]; hin_img = p.out_img;
        }
        hipLaunchKernelGGL(img_channel_stats_kernel, dim3((unsigned)net[tq].cout), dim3(256), 0, s, hin, hin_img, n, H, W, Hv, Wv, stats->mean, stats->var);
        stats->channels = net[tq].cout;
        return hipGetLastError() == hipSuccess ? GBNF_OK : fail(GBNF_ERR_HIP, "gbnf_image_flow_actnorm_stats: launch failed");
      }
      an_index += (int)net.size() - 1;
      for (size_t q = 0; q + 1 < net.size(); ++q) {
        const PackedConv& c = net[q];
        if (q == 0 && net.size() >= 3 && c.cin <= 16) continue;     // fused into the 1x1 that follows
        p.in = hin; p.in_img = hin_img; p.wp = blob + c.w_off; p.bias = blob + c.b_off;
        p.out = hb[q & 1]; p.out_img = (int64_t)c.cout * H * W; p.cin = c.cin; p.cout = c.cout; p.ks = c.ks;
        p.pre_in = nullptr;
        if (q == 1 && net.size() >= 3 && net[0].cin <= 16) {
          p.pre_in = cur; p.pre_in_img = img; p.pre_wp = blob + net[0].w_off; p.pre_bias = blob + net[0].b_off;
          p.pre_cin = net[0].cin;
        }
        launch_conv<EPI_RELU>(p, (int)n, s, rec);
        p.pre_in = nullptr;
        hin = hb[q & 1]; hin_img = p.out_img;
      }
      const PackedConv& c = net.back();
      p.in = hin; p.in_img = hin_img; p.wp = blob + c.w_off; p.bias = blob + c.b_off; p.out = nullptr;
      p.st = cur + (int64_t)c1 * H * W; p.st_img = img; p.cin = c.cin; p.cout = c.cout; p.ks = c.ks;
      if (f->additive) launch_conv<EPI_COUPLE_ADD>(p, (int)n, s, rec);
      else launch_conv<EPI_COUPLE_AFFINE>(p, (int)n, s, rec);
    }
    if (l < f->L - 1) {
      ConvLaunch p{};
      const PackedConv& c = f->split[l];
      p.H = H; p.W = W; p.Hv = Hv; p.Wv = Wv; p.n_strips = n_strips; p.ldj = ldj;
      p.in = cur; p.in_img = img; p.wp = blob + c.w_off; p.bias = blob + c.b_off;
      p.st = cur + (int64_t)c1 * H * W; p.st_img = img; p.cin = c.cin; p.cout = c.cout; p.ks = c.ks;
      launch_conv<EPI_SPLIT>(p, (int)n, s, rec);
      const int Hn = H / 2 < 8 ? 8 : H / 2, Wn = W / 2 < 8 ? 8 : W / 2;      // storage of the next level (at least 8 wide)
      if (rec != nullptr) {
        RepairOp& op = rec->add(ROP_SQUEEZE);
        op.a = cur - workspace; op.b = oth - workspace;
        op.i[0] = c1; op.i[1] = H; op.i[2] = W; op.i[3] = Hn; op.i[4] = Wn;
      } else {
        hipLaunchKernelGGL(img_squeeze_kernel, dim3((unsigned)n), dim3(256), 0, s, (const float*)cur, img, oth, c1, H, W, Hn, Wn);
      }
      std::swap(cur, oth);
      C = c1 * 4; H = Hn; W = Wn; Hv /= 2; Wv /= 2;
    }
  }
  if (rec != nullptr) {
    RepairOp& op = rec->add(ROP_FINAL);
    op.a = cur - workspace; op.c = ldj - workspace; op.prior = blob + f->prior_off;
    op.i[0] = C; op.i[1] = H; op.i[2] = W; op.i[3] = Hv; op.i[4] = Wv;
    return GBNF_OK;
  }
  hipLaunchKernelGGL(img_final_kernel, dim3((unsigned)n), dim3(256), 0, s, (const float*)cur, (int64_t)C * H * W,
                     blob + f->prior_off, (const float*)ldj, ll, z, C, H, W, Hv, Wv);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_image_flow_forward: %s", hipGetErrorString(e));
  return GBNF_OK;
}

// GBNF_IMAGE_REPAIR=0: no pass behind the split-f16 kernels -- out-of-range images are counted (gbnf_saturation_count) but keep
// their clamped values (kernel timing only).  Anything else (default): the same-call protocol above.
static int image_repair_mode() {
  static const int mode = [] { const char* e = getenv("GBNF_IMAGE_REPAIR"); return e ? atoi(e) : 1; }();
  return mode;
}

// The part of a call's workspace behind the main batch's buffers
struct RepairSpace {
  unsigned* mark;
  unsigned* list;
  unsigned* check_list;
  unsigned* count;      // [0] marked images, [1] check entries
  float* wsr;           // image_repair_workgroups(n) private workspaces of image_state_floats(f, 1) floats
};
static RepairSpace repair_space(const gbnf_image_flow* f, float* ws, int64_t n) {
  RepairSpace r;
  float* q = ws + image_state_floats(f, n);
  r.mark = reinterpret_cast<unsigned*>(q); q += image_mark_floats(n);
  r.list = reinterpret_cast<unsigned*>(q); q += image_mark_floats(n);
  r.check_list = reinterpret_cast<unsigned*>(q); r.count = r.check_list + 16; q += 64;
  r.wsr = q;
  return r;
}

static void launch_repair(const gbnf_image_flow* f, bool inverse, RepairArgs a, int64_t n, hipStream_t s) {
  a.ops = f->ops_dev + (inverse ? f->n_ops_fwd : 0);
  a.n_ops = inverse ? f->n_ops_inv : f->n_ops_fwd;
  a.n = n;
  a.ws_stride = image_state_floats(f, 1);
  a.force_all = f->dev_state + 1;
  a.host_words = f->host_words;
  const dim3 grid((unsigned)image_repair_workgroups(n)), blk(64 * IMG_WAVES);
  if (inverse) hipLaunchKernelGGL((img_repair_kernel<true>), grid, blk, f->repair_lds_inv, s, a);
  else hipLaunchKernelGGL((img_repair_kernel<false>), grid, blk, f->repair_lds_fwd, s, a);
}

int gbnf_image_flow_forward(const gbnf_image_flow* f, const float* x, const float* noise, int64_t n, float* z, float* ldj,
                            float* ll, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!f) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_forward: flow is null");
  if (n < 0) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_forward: n < 0");
  if (n == 0) return GBNF_OK;
  if (!x || !ldj || !workspace) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_forward: x / ldj / workspace is null");
  int64_t need = 0;
  gbnf_image_flow_workspace_bytes(f, n, &need);
  if (workspace_bytes < need) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_forward: workspace of %lld bytes < %lld", (long long)workspace_bytes, (long long)need);
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  // exact f32: the handle's mode, or a split-f16 handle whose on-data check has failed (pinned word, no synchronisation)
  if (f->math_mode != GBNF_MATH_F16X3 || f->on_data_demoted()) return image_forward_impl(f, x, noise, n, z, ldj, ll, ws, s, true, nullptr);
  if (image_repair_mode() == 0) return image_forward_impl(f, x, noise, n, z, ldj, ll, ws, s, false, nullptr);
  // ---- split-f16 pass with range marks; the on-data check (first launch, every check_every-th); the marked images on exact f32
  const RepairSpace r = repair_space(f, ws, n);
  if (hipMemsetAsync(r.mark, 0, (size_t)n * 4, s) != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_image_flow_forward: memset failed");
  const int rc = image_forward_impl(f, x, noise, n, z, ldj, ll, ws, s, false, r.mark);
  if (rc) return rc;
  int32_t every = 256, tol_e9 = 2500;
  (void)gbnf_tuning_get("check_every", &every);
  (void)gbnf_tuning_get("check_tolerance_e9", &tol_e9);
  hipLaunchKernelGGL(img_compact_kernel, dim3(1), dim3(256), 0, s, (const unsigned*)r.mark, (int)n, r.list, r.count, r.check_list,
                     f->dev_state, f->host_words, (int)every);
  RepairArgs a{};
  a.mark = r.mark; a.ws = r.wsr;
  a.x = x; a.noise = noise; a.z = z; a.ldj = ldj; a.ll = ll;
  a.xchw = (int64_t)f->C * f->Hi * f->Wi; a.zsz = (int64_t)f->zC * f->zH * f->zW;      // (x and z have the map's own size)
  a.tol = 1e-9f * (float)tol_e9;
  a.check = 1; a.list = r.check_list; a.count = r.count + 1;
  launch_repair(f, false, a, n, s);
  a.check = 0; a.list = r.list; a.count = r.count;
  launch_repair(f, false, a, n, s);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_image_flow_forward (repair): %s", hipGetErrorString(e));
  return GBNF_OK;
}

int gbnf_image_flow_actnorm_stats(const gbnf_image_flow* f, const float* x, const float* noise, int64_t n, int32_t index,
                                  float* mean_dev, float* var_dev, int32_t* channels, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!f || !x || !mean_dev || !var_dev || !workspace) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_actnorm_stats: null argument");
  if (n < 1 || index < 0) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_actnorm_stats: n = %lld, index = %d", (long long)n, index);
  int64_t need = 0;
  gbnf_image_flow_workspace_bytes(f, n, &need);
  if (workspace_bytes < need) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_actnorm_stats: workspace of %lld bytes < %lld", (long long)workspace_bytes, (long long)need);
  float* ws = (float*)workspace;
  float* ldj = ws + image_state_floats(f, n);               // (scratch: the pass's own log-det accumulator)
  ActNormStats st{index, mean_dev, var_dev, -1};
  const int rc = image_forward_impl(f, x, noise, n, nullptr, ldj, nullptr, ws, (hipStream_t)stream, true, nullptr, &st);
  if (rc) return rc;
  if (st.channels < 0) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_actnorm_stats: the component has fewer than %d ActNorm2d layers", index + 1);
  if (channels) *channels = st.channels;
  return GBNF_OK;
}

int gbnf_image_flow_numerics(const gbnf_image_flow* f, gbnf_numerics_status* out) {
  if (!f || !out) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_numerics: null argument");
  out->math_mode = f->math_mode;
  // the create-time probe, or an on-data check, sent the handle to exact f32
  out->demoted = ((f->probed && f->math_mode != GBNF_MATH_F16X3) || f->on_data_demoted()) ? 1 : 0;
  if (f->on_data_demoted()) out->math_mode = GBNF_MATH_F32;
  out->checks = f->host_words ? (int64_t)((volatile unsigned*)f->host_words)[0] : 0;     // calls that marked (and repaired) an image
  out->worst_rel_err = f->probe_rel_err;
  out->tolerance = 2.5e-6f;
  return GBNF_OK;
}

int gbnf_image_flow_repair_counts(const gbnf_image_flow* f, int64_t* marked_calls, int64_t* repaired_images, int64_t* data_checks,
                                  int64_t* failed_checks, float* worst_check_rel_err) {
  if (!f) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_repair_counts: flow is null");
  volatile unsigned* w = (volatile unsigned*)f->host_words;
  if (marked_calls) *marked_calls = w ? w[0] : 0;
  if (repaired_images) *repaired_images = w ? w[1] : 0;
  if (data_checks) *data_checks = w ? w[2] : 0;
  if (failed_checks) *failed_checks = w ? w[3] : 0;
  if (worst_check_rel_err) {
    const unsigned bits = w ? w[4] : 0u;
    std::memcpy(worst_check_rel_err, &bits, 4);
  }
  return GBNF_OK;
}

int gbnf_image_flow_eps_floats(const gbnf_image_flow* f, int64_t* per_image) {
  if (!f || !per_image) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_eps_floats: null argument");
  *per_image = (int64_t)f->C * f->Hi * f->Wi - (int64_t)f->zC * f->zH * f->zW;
  return GBNF_OK;
}

// The z -> x launch sequence over n images.  fast: the fused split-f16 coupling-net kernel (marks raised in `mark`), else the
// exact-f32 convolutions; rec (exact f32, n = 1, workspace = IMG_REC_BASE): record the sequence for img_repair_kernel.
static int image_inverse_impl(const gbnf_image_flow* f, const float* z, const float* eps, float temperature, int64_t n, float* x,
                              float* workspace, hipStream_t s, bool fast, unsigned* mark, Recorder* rec) {
  const int64_t chw = f->state_img;
  const int64_t hid = (int64_t)f->chp * (f->H / 2) * (f->W / 2);
  float* SA = workspace;
  float* SB = SA + chw * n;
  float* H1 = SB + chw * n;
  float* H2 = H1 + hid * n;
  const float* blob = f->blob_dev;

  // level shapes on the way in, and where each Split2d level's eps starts (level 0 first, each (n, C_l/2, H_l, W_l))
  std::vector<int> LC(f->L), LH(f->L), LW(f->L), LHv(f->L), LWv(f->L);
  std::vector<int64_t> eps_off(f->L, 0);         // per image of the batch: the level's eps starts at eps_off * n
  {
    int C = f->C, H = f->H, W = f->W, Hv = f->Hi, Wv = f->Wi;
    int64_t off = 0;
    for (int l = 0; l < f->L; ++l) {
      C *= 4; H = H / 2 < 8 ? 8 : H / 2; W = W / 2 < 8 ? 8 : W / 2; Hv /= 2; Wv /= 2;
      LC[l] = C; LH[l] = H; LW[l] = W; LHv[l] = Hv; LWv[l] = Wv;
      if (l < f->L - 1) {
        eps_off[l] = off;
        off += (int64_t)(C / 2) * Hv * Wv;                      // (eps has the map's own size)
        C /= 2;
      }
    }
  }
  size_t step_end = 0;
  for (int l = 0; l < f->L; ++l) step_end += f->level_steps[l];

  float* cur = SA;
  float* oth = SB;
  if (rec != nullptr) {
    RepairOp& op = rec->add(ROP_EMBED);
    op.b = cur - workspace;
    op.i[0] = f->zC; op.i[1] = LH[f->L - 1]; op.i[2] = LW[f->L - 1]; op.i[3] = f->zH; op.i[4] = f->zW;
  } else {
    hipLaunchKernelGGL(img_embed_kernel, dim3((unsigned)n), dim3(256), 0, s, z, cur, f->zC, LH[f->L - 1], LW[f->L - 1], f->zH, f->zW);
  }
  for (int l = f->L - 1; l >= 0; --l) {
    const int C = LC[l], H = LH[l], W = LW[l], Hv = LHv[l], Wv = LWv[l];
    const int64_t img = (int64_t)C * H * W;
    const int n_strips = H / IMG_R;
    const int c1 = C / 2;
    const int K = f->level_steps[l];
    if (l < f->L - 1) {
      // the state of level l+1 (n, 4 c1, H/2, W/2) -> first c1 channels of this level; eps into the other half; Split2d reverse
      if (rec != nullptr) {
        RepairOp& op = rec->add(ROP_UNSQUEEZE);
        op.a = cur - workspace; op.b = oth - workspace;
        op.i[0] = c1; op.i[1] = H; op.i[2] = W; op.i[3] = C - c1; op.i[4] = Hv; op.i[5] = Wv; op.i[6] = LH[l + 1]; op.i[7] = LW[l + 1];
        op.eps_img = (int64_t)(C - c1) * Hv * Wv; op.eps_lvl = eps_off[l];
      } else {
        hipLaunchKernelGGL(img_unsqueeze_kernel, dim3((unsigned)n), dim3(256), 0, s, (const float*)cur, oth, img, c1, H, W,
                           eps + eps_off[l] * n, C - c1, Hv, Wv, LH[l + 1], LW[l + 1]);
      }
      std::swap(cur, oth);
      ConvLaunch p{};
      const PackedConv& c = f->split[l];
      p.H = H; p.W = W; p.Hv = Hv; p.Wv = Wv; p.n_strips = n_strips; p.ldj = nullptr; p.temperature = temperature;
      p.in = cur; p.in_img = img; p.wp = blob + c.w_off; p.bias = blob + c.b_off;
      p.st = cur + (int64_t)c1 * H * W; p.st_img = img; p.cin = c.cin; p.cout = c.cout; p.ks = c.ks;
      launch_conv<EPI_SPLIT_INV>(p, (int)n, s, rec);
    }
    for (int k = K - 1; k >= 0; --k) {
      const size_t step = step_end - (size_t)(K - k);
      ConvLaunch p{};
      p.H = H; p.W = W; p.Hv = Hv; p.Wv = Wv; p.n_strips = n_strips; p.ldj = nullptr;
      // coupling^-1: the net reads the first half (unchanged by the step)
      const std::vector<PackedConv>& net = f->net[step];
      const size_t NC = net.size();
      const bool fusable = fast && H == W && (W == 16 || W == 8) && (NC == 3 || ((NC == 2 || NC == 4) && f->chp <= 256)) && net[NC - 1].x_off != 0 &&
                           img_net_hx3_lds(W, f->chp, net[0].cin, net[0].kc, net[NC - 1].cout) != 0;
      bool coupled = false;
      if (fusable) {                                          // the fused split-f16 coupling-net kernel with the epilogue's way back
        NetLaunch q{};
        q.pre_in = cur; q.pre_in_img = img; q.pre_wp = reinterpret_cast<const unsigned*>(blob + net[0].x_off);
        q.pre_bias = blob + net[0].b_off; q.pre_kc = net[0].kc; q.pre_cin = net[0].cin;
        q.pre_koff = reinterpret_cast<const int*>(blob + net[0].k_off);
        q.n_mid = (int)NC - 2;
        q.wp = reinterpret_cast<const unsigned*>(blob + net[1].x_off); q.bias = blob + net[1].b_off;
        if (NC == 4) { q.wp2 = reinterpret_cast<const unsigned*>(blob + net[2].x_off); q.bias2 = blob + net[2].b_off; }
        q.wp3 = reinterpret_cast<const unsigned*>(blob + net[NC - 1].x_off); q.bias3 = blob + net[NC - 1].b_off;
        q.st = cur + (int64_t)c1 * H * W; q.st_img = img; q.ldj = nullptr;
        q.hid = net[0].cout; q.chp = f->chp; q.cout = net[NC - 1].cout; q.H = H; q.Hv = Hv; q.Wv = Wv; q.inverse = 1;
        q.sat = reinterpret_cast<unsigned long long*>(gbnf::saturation_counter());
        q.mark = mark; q.only = nullptr;
        const hipError_t le = img_net_hx3_launch(q, W, f->additive != 0, n, s);
        if (le != hipSuccess) return fail(GBNF_ERR_HIP, "img_net_hx3 launch (inverse) failed: %s", hipGetErrorString(le));
        coupled = true;
      }
      const float* hin = cur;
      int64_t hin_img = img;
      float* hb[2] = {H1, H2};
      for (size_t q = 0; !coupled && q + 1 < net.size(); ++q) {      // exact-f32 convolutions
        const PackedConv& c = net[q];
        if (q == 0 && net.size() >= 3 && c.cin <= 16) continue;     // fused into the 1x1 that follows
        p.in = hin; p.in_img = hin_img; p.wp = blob + c.w_off; p.bias = blob + c.b_off;
        p.out = hb[q & 1]; p.out_img = (int64_t)c.cout * H * W; p.cin = c.cin; p.cout = c.cout; p.ks = c.ks;
        p.pre_in = nullptr;
        if (q == 1 && net.size() >= 3 && net[0].cin <= 16) {
          p.pre_in = cur; p.pre_in_img = img; p.pre_wp = blob + net[0].w_off; p.pre_bias = blob + net[0].b_off;
          p.pre_cin = net[0].cin;
        }
        launch_conv<EPI_RELU>(p, (int)n, s, rec);
        p.pre_in = nullptr;
        hin = hb[q & 1]; hin_img = p.out_img;
      }
      if (!coupled) {
        const PackedConv& c = net.back();
        p.in = hin; p.in_img = hin_img; p.wp = blob + c.w_off; p.bias = blob + c.b_off; p.out = nullptr;
        p.st = cur + (int64_t)c1 * H * W; p.st_img = img; p.cin = c.cin; p.cout = c.cout; p.ks = c.ks;
        if (f->additive) launch_conv<EPI_COUPLE_ADD_INV>(p, (int)n, s, rec);
        else launch_conv<EPI_COUPLE_AFFINE_INV>(p, (int)n, s, rec);
      }
      // (invconv / Permute2d)^-1 + ActNorm2d reverse: cur -> oth
      const PackedConv& m = f->mix_inv[step];
      if (rec != nullptr || m.p_off == 0 || n * H * W < 8192 || !launch_mix(cur, oth, blob + m.p_off, C, H, W, Hv, Wv, n, s)) {
        p.in = cur; p.in_img = img; p.wp = blob + m.w_off; p.bias = blob + m.b_off; p.out = oth; p.out_img = img; p.st = nullptr;
        p.cin = C; p.cout = C; p.ks = 1;
        launch_conv<EPI_STORE>(p, (int)n, s, rec);
      }
      std::swap(cur, oth);
    }
    step_end -= (size_t)K;
  }
  if (rec != nullptr) {
    RepairOp& op = rec->add(ROP_POST);
    op.a = cur - workspace;
    op.i[0] = f->C; op.i[1] = f->H; op.i[2] = f->W; op.i[3] = f->Hi; op.i[4] = f->Wi; op.f[0] = f->bounds;
    return GBNF_OK;
  }
  hipLaunchKernelGGL(img_post_kernel, dim3((unsigned)n), dim3(256), 0, s, (const float*)cur, x, f->C, f->H, f->W, f->Hi, f->Wi, f->bounds);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_image_flow_inverse: %s", hipGetErrorString(e));
  return GBNF_OK;
}

// Records the exact-f32 sequences of one image (forward, then z -> x) and uploads them: img_repair_kernel's programme
static int image_record_repair_ops(gbnf_image_flow* f) {
  float* base = reinterpret_cast<float*>(IMG_REC_BASE);
  float* ldj_slot = base + image_state_floats(f, 1) - 64;       // (the spare floats behind the state buffers)
  Recorder fw, bw;
  int rc = image_forward_impl(f, nullptr, nullptr, 1, nullptr, ldj_slot, nullptr, base, nullptr, true, nullptr, nullptr, &fw);
  if (!rc) rc = image_inverse_impl(f, nullptr, nullptr, 1.0f, 1, nullptr, base, nullptr, false, nullptr, &bw);
  if (rc) return rc;
  f->n_ops_fwd = (int)fw.ops.size(); f->n_ops_inv = (int)bw.ops.size();
  f->repair_lds_fwd = fw.lds; f->repair_lds_inv = bw.lds;
  std::vector<RepairOp> all(fw.ops);
  all.insert(all.end(), bw.ops.begin(), bw.ops.end());
  hipError_t e = hipMalloc((void**)&f->ops_dev, all.size() * sizeof(RepairOp));
  if (e == hipSuccess) e = hipMemcpy(f->ops_dev, all.data(), all.size() * sizeof(RepairOp), hipMemcpyHostToDevice);
  // (the kernel also has a little static LDS -- the reduction buffers of its elementwise ops: 160 KB of dynamic LDS would not fit)
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)img_repair_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)img_repair_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
  if (e == hipSuccess && std::max(fw.lds, bw.lds) > (size_t)156 * 1024)
    return fail(GBNF_ERR_UNSUPPORTED, "gbnf_image_flow_create: a convolution of the exact-f32 sequence needs %zu bytes of LDS", std::max(fw.lds, bw.lds));
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_image_flow_create: repair programme: %s", hipGetErrorString(e));
  return GBNF_OK;
}

int gbnf_image_flow_inverse(const gbnf_image_flow* f, const float* z, const float* eps, float temperature, int64_t n, float* x,
                            void* workspace, int64_t workspace_bytes, void* stream) {
  if (!f) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_inverse: flow is null");
  if (n < 0) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_inverse: n < 0");
  if (n == 0) return GBNF_OK;
  if (!z || !x || !workspace) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_inverse: z / x / workspace is null");
  if (f->L > 1 && !eps) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_inverse: %d Split2d level(s) need eps", f->L - 1);
  int64_t need = 0;
  gbnf_image_flow_workspace_bytes(f, n, &need);
  if (workspace_bytes < need) return fail(GBNF_ERR_INVALID, "gbnf_image_flow_inverse: workspace of %lld bytes < %lld", (long long)workspace_bytes, (long long)need);
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  // split-f16 coupling nets (the fused kernel, as the forward) unless the handle runs on exact f32 (its mode, or a failed on-data
  // check of the forward direction).  An image whose hidden activation leaves the fp16 range is re-evaluated on exact f32 by the
  // repair launch behind the pass, in this call.
  const bool fast = f->math_mode == GBNF_MATH_F16X3 && !f->on_data_demoted();
  if (!fast || image_repair_mode() == 0) return image_inverse_impl(f, z, eps, temperature, n, x, ws, s, fast, nullptr, nullptr);
  const RepairSpace r = repair_space(f, ws, n);
  if (hipMemsetAsync(r.mark, 0, (size_t)n * 4, s) != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_image_flow_inverse: memset failed");
  const int rc = image_inverse_impl(f, z, eps, temperature, n, x, ws, s, true, r.mark, nullptr);
  if (rc) return rc;
  hipLaunchKernelGGL(img_compact_kernel, dim3(1), dim3(256), 0, s, (const unsigned*)r.mark, (int)n, r.list, r.count, (unsigned*)nullptr,
                     f->dev_state, f->host_words, -1);
  RepairArgs a{};
  a.mark = r.mark; a.ws = r.wsr;
  a.zin = z; a.eps = eps; a.xout = x; a.temperature = temperature;
  a.xchw = (int64_t)f->C * f->Hi * f->Wi; a.zsz = (int64_t)f->zC * f->zH * f->zW;
  a.check = 0; a.list = r.list; a.count = r.count;
  launch_repair(f, true, a, n, s);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_image_flow_inverse (repair): %s", hipGetErrorString(e));
  return GBNF_OK;
}

}  // extern "C"
