// gbnf_image_hx3.hip.h -- split-f16 ("f16x3") versions of the two wide convolutions of an image
// coupling net (the 1x1 hidden -> hidden and the last 3x3 hidden -> shift/scale), included by gbnf_image.hip.
//
// An f32 operand is two fp16 pieces x ~ hi + mid; a product is three v_mfma_f32_16x16x32_f16 with f32 accumulation
// (a_hi b_hi + a_hi b_mid + a_mid b_hi), 4.9x the f32-MFMA rate per FLOP, same accuracy to ~2^-22.  Unlike the tabular
// coupling nets (tanh + split in the MFMA stream), a ReLU ConvNet splits each activation ONCE where it is produced:
//   * weights: split at pack time (round to nearest), A fragments [o][tap][c][hi|mid][64 lanes][8 halfs];
//   * activations between the two kernels live in HBM as "NHWC split-f16": per pixel [hi: CHP halfs][mid: CHP halfs]
//     (CHP = hidden width padded to 32), so the consumer stages a pixel with plain 16-byte copies and a lane's B operand
//     (8 consecutive channels of one pixel, k = 32c + 8g + j) is ONE ds_read_b128; the pixel stride is padded by 16 B,
//     which makes the 16 lanes of a group hit 16 different bank quads.
//   img_mid_hx3_kernel : [first 3x3 (f32 MFMA, small input) -> relu -> split -> LDS] -> 1x1 (f16x3) -> relu -> split -> HBM
//   img_last_hx3_kernel: stage strip + halo -> 3x3 (f16x3, contraction split over the 4 waves) -> coupling epilogue (f32)
#pragma once

namespace gbnf {

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

__device__ __forceinline__ f32x4 img_mfma16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// hi = f16(x) (toward zero), mid = f16(x - hi) for a pair; x is clamped to the fp16 range first
__device__ __forceinline__ void img_split_pair(float x0, float x1, unsigned& hi, unsigned& mid) {
  x0 = __builtin_amdgcn_fmed3f(x0, -65504.0f, 65504.0f);
  x1 = __builtin_amdgcn_fmed3f(x1, -65504.0f, 65504.0f);
  const auto h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
  hi = __builtin_bit_cast(unsigned, h);
  // the residual x - hi as ONE v_fma_mix_f32 per value (the f16 half is widened inside the instruction), as in
  // gbnf_flow_kernel_hx3.hip.h; LLVM itself emits v_cvt_f32_f16 + v_sub_f32
  // IN PLACE: an asm output must never be a fresh register -- the allocator may hand it the dead result register of a v_mfma
  // still in flight, and the hazard recognizer does not see into asm (gbnf_flow_kernel_hx3.hip.h, split_pair_f16)
  float r0 = x0, r1 = x1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(r0) : "v"(hi));
  asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r1) : "v"(hi));
  const auto m = __builtin_amdgcn_cvt_pkrtz(r0, r1);
  mid = __builtin_bit_cast(unsigned, m);
}

// IMG_PRE_KC (template parameter of the mid kernel): 32-wide chunks of the first 3x3's folded contraction that are kept
// in registers: 2 (cin <= 7), 4 (cin <= 14) or 5 (cin <= 16)

struct MidLaunch {
  const float* pre_in;      // (n, *, H, W) f32: the coupling net's input z1
  int64_t pre_in_img;
  const unsigned* pre_wp;   // f16x3 fragments of the first 3x3 with taps folded into k: [tile][pre_kc][hi|mid][64][4 u32]
  const int* pre_koff;      // [32 * pre_kc] im2col offsets: k -> float offset ci * CSz + (dy+1) * WPz + (dx+1) from the window's corner, -1 = pad
  const float* pre_bias;    // [16 * tiles]
  int pre_kc;
  const unsigned* wp;       // f16x3 fragments of the 1x1: [o][c][hi|mid][64][4 u32]
  const float* bias;        // [16 * OT]
  unsigned* h2;             // out: NHWC split-f16 (n, H, W, 2 * chp halfs)
  int pre_cin, hid, chp, H, n_strips, o_split;
  unsigned long long* dbg;  // diagnostic builds (-DGBNF_IMG_STAMPS) only
};

#ifdef GBNF_IMG_STAMPS
#define IMG_STAMP(k)                                                                         \
  do {                                                                                       \
    unsigned long long t_;                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    if ((k) >= 0) stamp_acc[(k)] += t_ - stamp_last;                                         \
    stamp_last = t_;                                                                         \
  } while (0)
#else
#define IMG_STAMP(k) do { } while (0)
#endif

template <int PT, int IMG_PRE_KC>
__global__ void __launch_bounds__(64 * IMG_WAVES) img_mid_hx3_kernel(const MidLaunch p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  typedef const float __attribute__((address_space(1)))* gptr;
  typedef const u32x4 __attribute__((address_space(1)))* gv4;
  constexpr int W = 16 * PT / IMG_R, WPz = W + 2, RPz = IMG_R + 2, CSz = RPz * WPz, NPIX = IMG_R * W;
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int osp = blockIdx.x % p.o_split, bid = blockIdx.x / p.o_split;
  const int n = bid / p.n_strips, strip = bid % p.n_strips;
  const int H = p.H, r0 = strip * IMG_R;
  const int chp = p.chp, pixb = 4 * chp + 16;               // bytes per pixel in LDS: hi[chp] mid[chp] halfs + pad
  const int OT = (p.hid + 15) >> 4, KC = chp >> 5;
  unsigned char* HB = lds_raw;                              // [NPIX][pixb]
  float* zin = reinterpret_cast<float*>(lds_raw + (size_t)NPIX * pixb);   // [16][RPz][WPz]
#ifdef GBNF_IMG_STAMPS
  unsigned long long stamp_last = 0, stamp_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
  IMG_STAMP(-1);

  // ---- the biases this wave will need (first 3x3: its tiles wave, wave + 4, ...; 1x1: its <= 4 tiles) go in flight now: read
  //      where they are used they cost a global-memory round trip per tile with nothing to hide it (phase stamps, round 2:
  //      the first 3x3 took 49 % of this kernel for 20 % of its MFMAs)
  constexpr int PRE_T = 4;                                  // first-3x3 tiles per wave: chp / 16 / IMG_WAVES <= 4 (chp <= 256)
  f32x4 pre_b[PRE_T];
  {
    const f32x4 __attribute__((address_space(1)))* pb4 = (const f32x4 __attribute__((address_space(1)))*)p.pre_bias;
#pragma unroll
    for (int q = 0; q < PRE_T; ++q) {
      const int o = wave + q * IMG_WAVES;
      pre_b[q] = pb4[(o < OT ? o : 0) * 4 + g];              // bias rows 16 o + 4 g .. + 3 (the array is padded to OT whole tiles)
    }
  }

  // ---- stage z1 (strip + halo, zero padded)
  {
    const float* src = p.pre_in + (int64_t)n * p.pre_in_img;
    constexpr int Q = W / 4;
    const int q = threadIdx.x % Q, rid = threadIdx.x / Q;
    constexpr int ROWS_PER_PASS = 64 * IMG_WAVES / Q;
    for (int idx = rid; idx < 16 * RPz; idx += ROWS_PER_PASS) {
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
  IMG_STAMP(0);

  // ---- first 3x3 as ONE small f16x3 GEMM: its 9 taps x cin input channels are folded into the contraction
  //      (k = tap * cin + ci, padded to 32 * pre_kc), so a lane's B operand is an im2col gather of 8 values of its pixel
  //      -- built ONCE per pixel tile from a host-made offset table and reused by every output tile; relu, split -> HB
  {
    gptr pb = (gptr)p.pre_bias;
    const gv4 pw = (gv4)p.pre_wp;
    const int kcp = p.pre_kc;                               // 32-wide chunks of the folded contraction (<= IMG_PRE_KC)
    const int kt = chp >> 4;                                // 16-channel tiles of the padded hidden width
    u32x4 bh[IMG_PRE_KC][PT], bm[IMG_PRE_KC][PT];
#pragma unroll
    for (int c = 0; c < IMG_PRE_KC; ++c) {
      if (c < kcp) {
        const int* ko = p.pre_koff + 32 * c + 8 * g;        // float offsets from the 3x3 window's corner; < 0: padding
        int off[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) off[j] = ko[j];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          const int lin = 16 * pt + i, pr = lin / W, pc = lin % W;
          const float* ctr = zin + pr * WPz + pc;             // top-left corner of the pixel's 3x3 window (offsets >= 0)
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = off[j] >= 0 ? ctr[off[j]] : 0.0f;
          unsigned h[4], m[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) img_split_pair(v[2 * q], v[2 * q + 1], h[q], m[q]);
          bh[c][pt] = u32x4{h[0], h[1], h[2], h[3]};
          bm[c][pt] = u32x4{m[0], m[1], m[2], m[3]};
        }
      }
    }
    auto load_pa = [&](int o, u32x4 (&ah)[IMG_PRE_KC], u32x4 (&am)[IMG_PRE_KC]) {
      const int oo = o < kt ? o : 0;                          // past the end: a valid tile again (unused)
#pragma unroll
      for (int c = 0; c < IMG_PRE_KC; ++c) {
        const gv4 f = pw + ((size_t)oo * kcp + (c < kcp ? c : 0)) * 128 + lane;
        ah[c] = f[0];
        am[c] = f[64];
      }
    };
    auto pre_tile = [&](int o, const f32x4& bias_o, const u32x4 (&ah)[IMG_PRE_KC], const u32x4 (&am)[IMG_PRE_KC]) {
      f32x4 acc[PT];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) acc[pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < IMG_PRE_KC; ++c) {
        if (c < kcp) {
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) {
            acc[pt] = img_mfma16(am[c], bh[c][pt], acc[pt]);
            acc[pt] = img_mfma16(ah[c], bm[c][pt], acc[pt]);
            acc[pt] = img_mfma16(ah[c], bh[c][pt], acc[pt]);
          }
        }
      }
      img_drain(acc);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int lin = 16 * pt + i;
        const bool in_img = r0 + lin / W < H;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = 16 * o + 4 * g + r;
          v[r] = (co < p.hid && in_img) ? fmaxf(acc[pt][r] + bias_o[r], 0.0f) : 0.0f;
        }
        unsigned h01, m01, h23, m23;
        img_split_pair(v[0], v[1], h01, m01);
        img_split_pair(v[2], v[3], h23, m23);
        const u32x2 hi = {h01, h23}, mid = {m01, m23};
        unsigned char* px = HB + (size_t)lin * pixb + 2 * (16 * o + 4 * g);
        *reinterpret_cast<u32x2*>(px) = hi;
        *reinterpret_cast<u32x2*>(px + 2 * chp) = mid;
      }
    };
    u32x4 ah[2][IMG_PRE_KC], am[2][IMG_PRE_KC];
    load_pa(wave, ah[0], am[0]);
#pragma unroll
    for (int q = 0; q < PRE_T; q += 2) {
      const int o = wave + q * IMG_WAVES;
      if (o < kt) {
        load_pa(o + IMG_WAVES, ah[1], am[1]);
        pre_tile(o, pre_b[q], ah[0], am[0]);
        load_pa(o + 2 * IMG_WAVES, ah[0], am[0]);
        if (o + IMG_WAVES < kt) pre_tile(o + IMG_WAVES, pre_b[q + 1 < PRE_T ? q + 1 : q], ah[1], am[1]);
      }
    }
  }
  __syncthreads();
  IMG_STAMP(1);

  // ---- 1x1 hidden -> hidden, f16x3: this wave's (<= 4) output tiles x PT pixel tiles, B operands shared by the tiles
  constexpr int MAXO = 4;
  const int o_per = (OT + p.o_split - 1) / p.o_split;
  const int o_begin = osp * o_per, o_end = min(OT, o_begin + o_per);
  int ow[MAXO];
#pragma unroll
  for (int q = 0; q < MAXO; ++q) ow[q] = o_begin + wave + q * IMG_WAVES;       // tile of slot q (>= o_end: idle)
  f32x4 acc[MAXO][PT];
#pragma unroll
  for (int q = 0; q < MAXO; ++q)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[q][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 mid_b[MAXO];                                       // the 1x1's biases for this wave's tiles, in flight under the MFMA stream
  {
    const f32x4 __attribute__((address_space(1)))* b4 = (const f32x4 __attribute__((address_space(1)))*)p.bias;
#pragma unroll
    for (int q = 0; q < MAXO; ++q) mid_b[q] = b4[(ow[q] < o_end ? ow[q] : o_begin) * 4 + g];
  }
  const gv4 wp = (gv4)p.wp;
  auto load_a = [&](int c, u32x4 (&ah)[MAXO], u32x4 (&am)[MAXO]) {
    const int cc = c < KC ? c : 0;
#pragma unroll
    for (int q = 0; q < MAXO; ++q) {
      const int o = ow[q] < o_end ? ow[q] : o_begin;          // idle slots re-read a valid fragment
      const gv4 f = wp + ((size_t)o * KC + cc) * 128 + lane;
      ah[q] = f[0];
      am[q] = f[64];
    }
  };
  const unsigned char* bbase = HB + (size_t)i * pixb + 16 * g;
  auto chunk = [&](int c, const u32x4 (&ah)[MAXO], const u32x4 (&am)[MAXO]) {
    u32x4 bh[PT], bm[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const unsigned char* px = bbase + (size_t)(16 * pt) * pixb + 64 * c;
      bh[pt] = *reinterpret_cast<const u32x4*>(px);
      bm[pt] = *reinterpret_cast<const u32x4*>(px + 2 * chp);
    }
#pragma unroll
    for (int q = 0; q < MAXO; ++q) {
      if (ow[q] < o_end) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          acc[q][pt] = img_mfma16(am[q], bh[pt], acc[q][pt]);
          acc[q][pt] = img_mfma16(ah[q], bm[pt], acc[q][pt]);
          acc[q][pt] = img_mfma16(ah[q], bh[pt], acc[q][pt]);
        }
      }
    }
  };
  {
    u32x4 ah[2][MAXO], am[2][MAXO];
    load_a(0, ah[0], am[0]);
    int c = 0;
    for (; c + 2 <= KC; c += 2) {
      load_a(c + 1, ah[1], am[1]);
      chunk(c, ah[0], am[0]);
      load_a(c + 2, ah[0], am[0]);
      chunk(c + 1, ah[1], am[1]);
    }
    if (c < KC) chunk(c, ah[0], am[0]);
  }
#pragma unroll
  for (int q = 0; q < MAXO; ++q) img_drain(acc[q]);
  IMG_STAMP(2);

  // ---- relu(. + bias) -> split -> back into HB (its input role is over) -> HBM as NHWC split-f16 with 16-byte, fully
  //      coalesced copies (a lane's own 8-byte pieces would reach HBM as 32-byte fragments)
  __syncthreads();                                         // every wave is done reading HB as the 1x1's input
#pragma unroll
  for (int q = 0; q < MAXO; ++q) {
    if (ow[q] < o_end) {
      const int o = ow[q];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int lin = 16 * pt + i;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = 16 * o + 4 * g + r;
          v[r] = co < p.hid ? fmaxf(acc[q][pt][r] + mid_b[q][r], 0.0f) : 0.0f;
        }
        unsigned h01, m01, h23, m23;
        img_split_pair(v[0], v[1], h01, m01);
        img_split_pair(v[2], v[3], h23, m23);
        unsigned char* px = HB + (size_t)lin * pixb + 2 * (16 * o + 4 * g);
        *reinterpret_cast<u32x2*>(px) = u32x2{h01, h23};
        *reinterpret_cast<u32x2*>(px + 2 * chp) = u32x2{m01, m23};
      }
    }
  }
  __syncthreads();
  {
    // this workgroup's channels [16 o_begin, 16 o_end): per pixel one run in the hi half and one in the mid half
    const int run_units = (o_end - o_begin) * 2;            // 16-byte units per run (a tile = 16 halfs = 32 B)
    const int per_pix = 2 * run_units;
    unsigned char* dst0 = reinterpret_cast<unsigned char*>(p.h2) + ((int64_t)n * H + r0) * W * (int64_t)(4 * chp);
    const int rows = min(IMG_R, H - r0);
    for (int e = threadIdx.x; e < rows * W * per_pix; e += 64 * IMG_WAVES) {
      const int px = e / per_pix, u = e - px * per_pix;
      const int half = u / run_units, k = u - half * run_units;
      const int boff = half * 2 * chp + 32 * o_begin + 16 * k;
      *reinterpret_cast<u32x4*>(dst0 + (int64_t)px * (4 * chp) + boff) = *reinterpret_cast<const u32x4*>(HB + (size_t)px * pixb + boff);
    }
  }
  IMG_STAMP(3);
#ifdef GBNF_IMG_STAMPS
  if (p.dbg != nullptr && threadIdx.x == 0)
    for (int k = 0; k < 6; ++k) p.dbg[(size_t)blockIdx.x * 6 + k] = stamp_acc[k];
#endif
}

struct LastLaunch {
  const unsigned* h2;       // NHWC split-f16 input (n, H, W, 2 * chp halfs)
  const unsigned* wp;       // f16x3 fragments [o][tap][c][hi|mid][64][4 u32]
  const float* bias;        // [16 * OT]
  float* st;                // coupled half z2 (n, *, H, W) f32, first channel of image 0
  int64_t st_img;
  float* ldj;
  int chp, cout, H, n_strips;
};

// EPI: EPI_COUPLE_AFFINE or EPI_COUPLE_ADD
template <int EPI, int PT>
__global__ void __launch_bounds__(64 * IMG_WAVES) img_last_hx3_kernel(const LastLaunch p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  typedef const float __attribute__((address_space(1)))* gptr;
  typedef const u32x4 __attribute__((address_space(1)))* gv4;
  constexpr int W = 16 * PT / IMG_R, WP = W + 2, RP = IMG_R + 2;
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = blockIdx.x / p.n_strips, strip = blockIdx.x % p.n_strips;
  const int H = p.H, r0 = strip * IMG_R;
  const int chp = p.chp, pixb = 4 * chp + 16, KC = chp >> 5;
  const int OT = (p.cout + 15) >> 4;                        // <= 3
  unsigned char* HB = lds_raw;                              // [RP * WP][pixb]

  // ---- stage strip + halo: a pixel is 4*chp contiguous bytes in HBM (a wave moves one pixel per pass, lane = 16-byte
  //      unit: fully coalesced, no index arithmetic beyond the pixel's row / column); outside the image: zeros
  //      Direct-to-LDS DMA (global_load_lds_dwordx4: the pixel's bytes are lane-linear in HBM and in LDS): a wave issues
  //      all its pixels back to back and waits once -- the register-staged loop (load, wait, store per pixel) made this
  //      kernel latency-bound (74 us for 1024 workgroups at batch 256; round 2).
  {
    const int units = chp >> 2;                             // 16-byte units per pixel (<= 64: chp <= 256)
    typedef const __attribute__((address_space(1))) unsigned char* gbytes;
    typedef __attribute__((address_space(3))) void* lptr;
    gbytes src = (gbytes)(reinterpret_cast<const unsigned char*>(p.h2) + (int64_t)n * H * W * (int64_t)(4 * chp));
    for (int px = wave; px < RP * WP; px += IMG_WAVES) {
      const int rr = px / WP, cc = px - rr * WP;
      const int row = r0 + rr - 1, col = cc - 1;
      const bool inside = row >= 0 && row < H && col >= 0 && col < W;      // wave-uniform
      unsigned char* dst = HB + (size_t)px * pixb;
      if (inside) {
        if (lane < units)
          __builtin_amdgcn_global_load_lds(src + ((int64_t)row * W + col) * (int64_t)(4 * chp) + 16 * lane, (lptr)dst, 16, 0, 0);
      } else if (lane < units) {
        *reinterpret_cast<u32x4*>(dst + 16 * lane) = u32x4{0u, 0u, 0u, 0u};
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  constexpr int MAXO = IMG_WAVES - 1;
  f32x4 part[MAXO][PT];
#pragma unroll
  for (int o = 0; o < MAXO; ++o)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) part[o][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
  int boff[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int lin = 16 * pt + i, pr = lin / W, pc = lin % W;
    boff[pt] = ((pr + 1) * WP + pc + 1) * pixb + 16 * g;
  }
  const int T_all = 9 * KC;
  const gv4 wp = (gv4)p.wp;
  auto load_a = [&](int t, u32x4 (&ah)[MAXO], u32x4 (&am)[MAXO]) {
    const int tt = t < T_all ? t : 0;
#pragma unroll
    for (int o = 0; o < MAXO; ++o) {
      const gv4 f = wp + ((size_t)(o < OT ? o : 0) * T_all + tt) * 128 + lane;
      ah[o] = f[0];
      am[o] = f[64];
    }
  };
  auto iter = [&](int t, const u32x4 (&ah)[MAXO], const u32x4 (&am)[MAXO]) {
    const int tap = t / KC, c = t - tap * KC;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    const unsigned char* b0 = HB + (dy * WP + dx) * pixb + 64 * c;
    u32x4 bh[PT], bm[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      bh[pt] = *reinterpret_cast<const u32x4*>(b0 + boff[pt]);
      bm[pt] = *reinterpret_cast<const u32x4*>(b0 + boff[pt] + 2 * chp);
    }
#pragma unroll
    for (int o = 0; o < MAXO; ++o) {
      if (o < OT) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          part[o][pt] = img_mfma16(am[o], bh[pt], part[o][pt]);
          part[o][pt] = img_mfma16(ah[o], bm[pt], part[o][pt]);
          part[o][pt] = img_mfma16(ah[o], bh[pt], part[o][pt]);
        }
      }
    }
  };
  {
    u32x4 ah[2][MAXO], am[2][MAXO];
    int t = wave;
    load_a(t, ah[0], am[0]);
    for (; t + IMG_WAVES < T_all; t += 2 * IMG_WAVES) {
      load_a(t + IMG_WAVES, ah[1], am[1]);
      iter(t, ah[0], am[0]);
      load_a(t + 2 * IMG_WAVES, ah[0], am[0]);
      iter(t + IMG_WAVES, ah[1], am[1]);
    }
    if (t < T_all) iter(t, ah[0], am[0]);
  }
#pragma unroll
  for (int o = 0; o < MAXO; ++o) img_drain(part[o]);
  __syncthreads();                                         // nobody reads the strip any more
  f32x4* red = reinterpret_cast<f32x4*>(lds_raw);          // [wave][o][pt][64]
#pragma unroll
  for (int o = 0; o < MAXO; ++o)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
      if (o < OT) red[((wave * MAXO + o) * PT + pt) * 64 + lane] = part[o][pt];
  __syncthreads();

  float ld = 0.0f;
  gptr bias = (gptr)p.bias;
  float* st = p.st + (int64_t)n * p.st_img;
  for (int q = wave; q < OT * PT; q += IMG_WAVES) {
    const int o = q / PT, pt = q % PT;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < IMG_WAVES; ++w) acc += red[((w * MAXO + o) * PT + pt) * 64 + lane];
    const int lin = 16 * pt + i, row = r0 + lin / W, pc = lin % W;
    const bool in_img = row < H;
    const int64_t pix = (int64_t)row * W + pc;
    if constexpr (EPI == EPI_COUPLE_ADD) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 16 * o + 4 * g + r;
        if (co < p.cout && in_img) st[(int64_t)co * H * W + pix] += acc[r] + bias[co];        // models/glow.py:328-329
      }
    } else {
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int co = 16 * o + 4 * g + 2 * qq, j = co >> 1;
        if (co + 1 < p.cout && in_img) {
          const float h0 = acc[2 * qq] + bias[co], h1 = acc[2 * qq + 1] + bias[co + 1];
          float* zp = st + (int64_t)j * H * W + pix;
          const float e = __expf(-(h1 + 2.0f));                     // scale = sigmoid(raw + 2), models/glow.py:333
          const float sc = 1.0f / (1.0f + e);
          *zp = (*zp + h0) * sc;                                    // models/glow.py:334-335
          ld += -log1pf(e);                                         // log(scale), models/glow.py:338
        }
      }
    }
  }
  if constexpr (EPI == EPI_COUPLE_AFFINE) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) ld += __shfl_xor(ld, m);
    if (lane == 0) atomicAdd(p.ldj + n, ld);
  }
}


}  // namespace gbnf
