// gbnf_image_net.hip -- the whole coupling net of an image flow step in one kernel (round 4).  Its own translation unit:
// it is compiled with the MFMA accumulators in VGPRs (-mllvm -amdgpu-mfma-vgpr-form=1, csrc/build.py: the relu + split
// epilogues read them directly), which the exact-f32 convolution kernels of gbnf_image.hip were measured slower with.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "gbnf_image_net.h"

namespace gbnf {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

__device__ __forceinline__ f32x4 img_mfma16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// the compiler's MFMA-result hazard padding does not look across branches on this toolchain (tools/isa_hazard_lint.py checks)
// (ONE pair of s_nop for all the tiles: every tile is an operand of the same asm, so no MFMA can be scheduled behind it and no
// read in front of it; a pair per tile idled 16 cycles x 9 tiles at the end of every 1x1 pass)
__device__ __forceinline__ void img_drain(f32x4 (&c)[1]) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c[0])); }
__device__ __forceinline__ void img_drain(f32x4 (&c)[2]) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c[0]), "+v"(c[1])); }
__device__ __forceinline__ void img_drain(f32x4 (&c)[3]) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2])); }
__device__ __forceinline__ void img_drain(f32x4 (&c)[4]) {
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
}
__device__ __forceinline__ void img_drain(f32x4 (&c)[6]) {
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]));
}
__device__ __forceinline__ void img_drain(f32x4 (&c)[8]) {
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
}
__device__ __forceinline__ void img_drain(f32x4 (&c)[9]) {
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]));
}
// hi = f16(x) (toward zero), mid = f16(x - hi) for a pair, range-watched: amax = max(amax, |x0|, |x1|).  No clamp: both
// conversions round TOWARD ZERO, so a value beyond the fp16 range converts to +-65504 (never to infinity) and every piece stays
// finite; the result is wrong there, which is what the watch is for (the image is marked and re-evaluated in f32).
__device__ __forceinline__ void img_split_pair_w(float x0, float x1, unsigned& hi, unsigned& mid, float& amax) {
  amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(x0), __builtin_fabsf(x1)));
  const auto h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
  hi = __builtin_bit_cast(unsigned, h);
  // IN PLACE (an asm output must never be a fresh register: gbnf_flow_kernel_hx3.hip.h, split_pair_f16)
  float r0 = x0, r1 = x1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(r0) : "v"(hi));
  asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r1) : "v"(hi));
  const auto m = __builtin_amdgcn_cvt_pkrtz(r0, r1);
  mid = __builtin_bit_cast(unsigned, m);
}

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

// ======================================================================================================================
// Round 4: the WHOLE coupling net of an image flow step in one kernel -- first 3x3 -> relu -> 1x1 -> relu -> last 3x3 ->
// coupling epilogue (ConvNet, models/layers.py:304-317; FlowStep.encode, models/glow.py:317-342).  The 256-channel hidden
// activation never leaves the CU: rounds 1-3 wrote it to HBM between img_mid_hx3 (64 MB per dispatch at batch 256) and
// img_last_hx3 (read back with its halo: 94 MB), 70 % of the path's HBM traffic and most of both kernels' time.
//
// A workgroup of 8 waves owns RO = 8 output rows of one image: the whole 8 x 8 map of the second level, half of a 16 x 16
// map of the first.  The last 3x3 needs the hidden activation one row beyond the strip: RH = 9 hidden rows are computed for
// a 16-wide half image (the other halo row lies outside the image: zero), 12.5 % recomputation.  LDS:
//   HB   [RH * W pixels][hi: chp halfs | mid: chp halfs | 16 B pad]    the hidden activation as split f16 (144 x 1040 B = 146 KB)
//   ZP   one all-zero pixel: what a tap reads outside the image (an address select per pixel and tap, no masking of operands)
//   zin  [cin][RH + 2][W + 2] f32: z1 with its halo
// Phases (a barrier between them):
//   1  stage z1;  2  first 3x3 as a folded f16x3 GEMM (k = tap * cin + ci) -> relu -> split -> HB: wave w owns hidden tiles
//   w, w + 8 for every pixel tile (its A fragments stay in registers over the pixel groups);  3  1x1: wave w owns output tiles
//   w, w + 8 x ALL pixel tiles (18 accumulator tiles), A fragments from L2 two chunks ahead, B fragments = one ds_read_b128 per
//   pixel tile and piece, two tiles ahead; relu -> split -> back into HB (in place: every wave has finished reading);  4  last
//   3x3: the (tap, chunk) iterations are dealt to the waves, the partial tiles meet in LDS (over HB: its role is over), wave w
//   finishes pixel tile w with the f32 coupling epilogue.
// Round 5 (stamps, tools/image_stamps2.py: 49.8 k -> 46.9 k cycles per wave at W = 16): the FIRST weight fragments of every phase
// are requested a phase ahead (64 + 147 KB per workgroup through the CU's L2 port, 1 - 2.3 k cycles that every wave stood through
// at the phase's start); phase 2 is one software pipeline per wave; the 1x1's B fragments come through a ring two tiles ahead so
// that the wave that is left alone on its SIMD at the end of the phase keeps the matrix pipe busy; biases are the accumulators'
// initial values.
// Numerics: every value that is split is range-watched (one v_max3 per pair); a workgroup that met
// |value| > 65504 raises its image's mark and the per-device counter -- gbnf_api / gbnf_image.hip re-evaluate marked images
// on the exact-f32 kernels.


// FULL: the map fills its storage and the hidden width fills its padding (hid == chp) -- the CIFAR configuration: every
// per-value select on "channel exists" / "pixel belongs to the map" drops out of the tile epilogues (26 -> 22 vector and ~10
// scalar instructions fewer per 16 x 16 tile).
// KH (round 5): hidden widths 257 .. 512 -- the usual Glow width is 512 -- in KH = 2 HALVES of the hidden channels.  HB holds 256
// channels at a time (the same 1040-byte pixels): the first 3x3 writes half 0 of its output, the 1x1 accumulates its k chunks into
// ALL of the wave's output tiles (four instead of two: w, w + 8, w + 16, w + 24), the first 3x3 writes half 1 over it, the 1x1 takes
// those chunks; then the 1x1's output goes back to HB half by half, each followed by its share of the last 3x3's contraction.
// No activation is recomputed; what it costs is accumulator registers -- 4 output tiles x every pixel tile -- so a 16-wide map is
// cut into S = 4 strips of RO = 4 output rows (RH = 6 hidden rows with the halo, 96 accumulator registers) instead of two of 8.
template <int W, int IMG_PRE_KC, int EPI, int OT3, bool FULL, int KH>
__global__ void __launch_bounds__(512) img_net_hx3_kernel(const NetLaunch p) {
  static_assert(OT3 >= 1 && OT3 <= 3, "the last 3x3 has at most 48 output channels");
  static_assert(W == 16 || W == 8, "16- and 8-wide maps");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  typedef const u32x4 __attribute__((address_space(1)))* gv4;
  typedef const float __attribute__((address_space(1)))* gptr;
  constexpr int WV = 8;
  static_assert(KH == 1 || KH == 2, "one or two halves of the hidden channels");
  constexpr int RO = (W == 16 && KH == 2) ? 4 : 8;  // output rows per workgroup
  constexpr int S = W == 16 ? 16 / RO : 1;          // workgroups per image
  constexpr int RH = S == 1 ? RO : (S == 2 ? RO + 1 : RO + 2);      // hidden rows per workgroup: the strip + the halo rows inside the image (8 | 9 | 6)
  constexpr int NPH = RH * W / 16, NPO = RO * W / 16;     // pixel tiles: hidden (9 | 4), output (8 | 4)
  constexpr int ZR = RH + 2, ZW = W + 2, CSz = ZR * ZW;
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = blockIdx.x / S, strip = blockIdx.x % S;
  if (p.only != nullptr && p.only[n] == 0u) return;
  const int H = p.H, r0 = strip * RO;
  const int hr0 = S == 1 ? 0 : (r0 - 1 < 0 ? 0 : (r0 - 1 > H - RH ? H - RH : r0 - 1));     // first hidden row (image row) held in HB
  const int chp = p.chp, chh = chp / KH, pixb = 4 * chh + 16;       // chh: the channels HB holds at a time
  const int OT = (p.hid + 15) >> 4, KC = chp >> 5, kt = chp >> 4, KCH = chh >> 5;
  const int TH = chh >> 4;                                  // 16-channel tiles of one half: half hf = tiles [TH hf, TH (hf + 1))
  unsigned char* HB = lds_raw;                              // [NPH * 16][pixb]
  unsigned char* ZP = HB + (size_t)NPH * 16 * pixb;         // the zero pixel
  float* zin_f = reinterpret_cast<float*>(ZP + pixb);       // [pre_cin][ZR][ZW] (32-bit words: hi | mid << 16)
  int* koffs = reinterpret_cast<int*>(zin_f + p.pre_cin * CSz);      // [32 * pre_kc] im2col word offsets for THIS kernel's staging
  float amax = 0.0f;
  // Round 6: ConvNets of coupling_network_depth 0 and 2 (models/layers.py:304-317: [Conv2d 1x1 -> ReLU] x num_layers between the two
  // 3x3s) on this kernel -- n_mid 1x1 layers (0, 1 or 2; hidden widths to 256): none = the last 3x3 reads the first one's output as
  // it stands in HB; two = the first 1x1's output goes back into HB (relu, split) and the phase runs again on the second layer's weights.
  const int n_mid = KH == 1 ? p.n_mid : 1;
  const unsigned* wp_cur = p.wp;          // the 1x1 layer whose fragments p3_load fetches
#ifdef GBNF_IMG_STAMPS
  unsigned long long stamp_last = 0, stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  IMG_STAMP(-1);

  // ---- the first 3x3's A fragments and biases of this wave's two hidden tiles: requested HERE, a phase and two barriers ahead of
  //      their use (read where the phase starts they were a global round trip -- 1.7 k of the kernel's 50 k cycles -- with nothing to
  //      hide it; same for the 1x1's and the last 3x3's first fragments below).  Chunks past pre_kc get ZERO fragments.
  //      (a slot past the end of a half of fewer than 16 tiles repeats the half's FIRST tile, stores included: the same values to the
  //      same addresses as the wave that owns it -- no branch in the first 3x3's pipeline)
  u32x4 pre_ah[2][IMG_PRE_KC], pre_am[2][IMG_PRE_KC];
  f32x4 pre_pb[2];
  int pre_lt[2], pre_o[2];
  auto pre_load = [&](int hf) {
    const gv4 pw = (gv4)p.pre_wp;
    const int kcp = p.pre_kc;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      pre_lt[q] = wave + q * WV < TH ? wave + q * WV : 0;
      pre_o[q] = TH * hf + pre_lt[q];
#pragma unroll
      for (int c = 0; c < IMG_PRE_KC; ++c) {
        const gv4 f = pw + ((size_t)pre_o[q] * kcp + (c < kcp ? c : 0)) * 128 + lane;
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        pre_ah[q][c] = c < kcp ? f[0] : z4;
        pre_am[q][c] = c < kcp ? f[64] : z4;
      }
      pre_pb[q] = ((const f32x4 __attribute__((address_space(1)))*)p.pre_bias)[(pre_o[q] < OT ? pre_o[q] : 0) * 4 + g];
    }
  };
  if constexpr (KH == 1) pre_load(0);          // (KH = 2 keeps four output tiles of accumulators per wave: no registers to spare)

  // ---- phase 1: z1 (hidden rows + halo, zero padded) as SPLIT words hi | mid << 16 -- every element is split once here,
  //      not once per wave and tap in the im2col gather below -- and the zero pixel
  unsigned* zin = reinterpret_cast<unsigned*>(zin_f);
  {
    const float* src = p.pre_in + (int64_t)n * p.pre_in_img;
    constexpr int Q = W / 4;
    const int q = threadIdx.x % Q, rid = threadIdx.x / Q;
    constexpr int ROWS_PER_PASS = 64 * WV / Q;
    for (int idx = rid; idx < p.pre_cin * ZR; idx += ROWS_PER_PASS) {
      const int ci = idx / ZR, rr = idx - ci * ZR;
      const int row = hr0 + rr - 1;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row >= 0 && row < H) v = *reinterpret_cast<const f32x4*>(src + ((int64_t)ci * H + row) * W + 4 * q);
      unsigned h01, m01, h23, m23;
      img_split_pair_w(v[0], v[1], h01, m01, amax);
      img_split_pair_w(v[2], v[3], h23, m23, amax);
      unsigned* dst = zin + ci * CSz + rr * ZW + 1 + 4 * q;
      dst[0] = __builtin_amdgcn_perm(m01, h01, 0x05040100u);      // lo16(hi pair) | lo16(mid pair) << 16: element 0
      dst[1] = __builtin_amdgcn_perm(m01, h01, 0x07060302u);      // hi16(hi pair) | hi16(mid pair) << 16: element 1
      dst[2] = __builtin_amdgcn_perm(m23, h23, 0x05040100u);
      dst[3] = __builtin_amdgcn_perm(m23, h23, 0x07060302u);
      if (q == 0) dst[-1] = 0u;
      if (q == Q - 1) dst[4] = 0u;
    }
    for (int u = threadIdx.x; u < pixb / 16; u += 64 * WV) *reinterpret_cast<u32x4*>(ZP + 16 * u) = u32x4{0u, 0u, 0u, 0u};
    // the packed im2col table (made for the 6-row staging of img_mid_hx3: ci * 6 ZW + dy ZW + dx) re-based to this kernel's ZR
    // rows per channel; a padding slot (k >= 9 cin: -1) reads the window's corner -- any finite value will do: its WEIGHT is zero
    if ((int)threadIdx.x < 32 * p.pre_kc) {
      const int o6 = p.pre_koff[threadIdx.x];
      const int ci = o6 / (6 * ZW);
      koffs[threadIdx.x] = o6 >= 0 ? o6 + ci * (CSz - 6 * ZW) : 0;
    }
  }
  __syncthreads();
  IMG_STAMP(0);

  // ---- phase 2: first 3x3 (folded contraction) -> relu -> split -> HB
  //      (a) the im2col B operands are built ONCE per workgroup -- wave w gathers the fragment pairs w, w + 8, ... (pixel tile,
  //      chunk) from the split words of z1 and leaves them in BF as ready-made lane-linear fragments; every wave built all of
  //      them for itself before (27 % of the kernel for 14 % of its MFMAs).  BF sits behind the other buffers where LDS allows
  //      and else over the tail of HB, which the first 3x3's own output fills last: a group of pixel tiles whose output reaches
  //      into BF is computed behind a barrier that follows its B loads (every later group's fragments lie above its output).
  unsigned char* BF = lds_raw + p.bf_off;                   // [NPH][pre_kc][hi | mid][64 lanes][16 B]
  {
    const int kcp = p.pre_kc;
    for (int u = wave; u < NPH * kcp; u += WV) {
      const int pt = u / kcp, c = u - pt * kcp;
      const int* ko = koffs + 32 * c + 8 * g;
      const int lin = 16 * pt + i, pr = lin / W, pc = lin % W;
      const unsigned* ctr = zin + pr * ZW + pc;
      int o8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o8[j] = ko[j];
      unsigned v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ctr[o8[j]];
      u32x4 fh, fm;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        fh[q] = __builtin_amdgcn_perm(v[2 * q + 1], v[2 * q], 0x05040100u);      // the two hi pieces
        fm[q] = __builtin_amdgcn_perm(v[2 * q + 1], v[2 * q], 0x07060302u);      // the two mid pieces
      }
      *reinterpret_cast<u32x4*>(BF + ((size_t)u * 2) * 1024 + 16 * lane) = fh;
      *reinterpret_cast<u32x4*>(BF + ((size_t)u * 2 + 1) * 1024 + 16 * lane) = fm;
    }
  }
  __syncthreads();
#ifdef GBNF_IMG_STAMP_BF
  IMG_STAMP(0);                                              // (diagnostic: the B-fragment build counted with the staging phase)
#endif
  // ---- phase 3 state: this wave's NQ output tiles of the 1x1 x every pixel tile (it accumulates over the halves of its input)
  constexpr int NQ = 2 * KH;
  int ow[NQ];
  f32x4 acc[NQ][NPH];
  f32x4 mid_b[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    // tiles (wave, wave + 8) of half q / 2 (a half of fewer than 16 tiles leaves the upper slots idle: index OT = "no tile")
    ow[q] = (wave + (q & 1) * WV < TH && TH * (q >> 1) + wave + (q & 1) * WV < OT) ? TH * (q >> 1) + wave + (q & 1) * WV : OT;
    // (the accumulators START at the bias of their rows -- D layout: row 4 g + r -- instead of adding it in the tile epilogue: four
    //  vector instructions less per tile, in the phases that are bound by the vector issue port)
    mid_b[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (n_mid > 0) mid_b[q] = ((const f32x4 __attribute__((address_space(1)))*)p.bias)[(ow[q] < OT ? ow[q] : 0) * 4 + g];
#pragma unroll
    for (int pt = 0; pt < NPH; ++pt) acc[q][pt] = mid_b[q];
  }
  // (KH = 1) the 1x1's first two chunks of A fragments: requested under the last round of the first 3x3's pipeline
  u32x4 p3_ah[2][NQ], p3_am[2][NQ];
  auto p3_load = [&](int c, u32x4 (&ah)[NQ], u32x4 (&am)[NQ]) {            // c: chunk of the WHOLE contraction
    const gv4 wp = (gv4)wp_cur;
    const int cc = c < KC ? c : 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const gv4 f = wp + ((size_t)(ow[q] < OT ? ow[q] : 0) * KC + cc) * 128 + lane;
      ah[q] = f[0];
      am[q] = f[64];
    }
  };
  // the 1x1's output tiles of round rf (relu, split) -> HB, in place of the layer's input (every wave has finished reading it)
  auto write_back = [&](int rf) {
#pragma unroll
    for (int q = 2 * rf; q < 2 * rf + 2; ++q) {
      if (ow[q] < OT) {
        const int o = ow[q] - TH * rf;                       // the tile's place in HB
#pragma unroll
        for (int pt = 0; pt < NPH; ++pt) {
          const int lin = 16 * pt + i;
          // outside the map proper (a 14 x 14 map in 16 x 16 storage) the hidden activation the last 3 x 3 reads must be the
          // map's zero padding, not relu(bias + ...)
          const bool valid = FULL || (hr0 + lin / W < p.Hv && lin % W < p.Wv);
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = 16 * ow[q] + 4 * g + r;
            v[r] = (FULL || (co < p.hid && valid)) ? fmaxf(acc[q][pt][r], 0.0f) : 0.0f;
          }
          unsigned h01, m01, h23, m23;
          img_split_pair_w(v[0], v[1], h01, m01, amax);
          img_split_pair_w(v[2], v[3], h23, m23, amax);
          unsigned char* px = HB + (size_t)lin * pixb + 2 * (16 * o + 4 * g);
          *reinterpret_cast<u32x2*>(px) = u32x2{h01, h23};
          *reinterpret_cast<u32x2*>(px + 2 * chh) = u32x2{m01, m23};
        }
      }
    }
  };
#pragma unroll
  for (int hf = 0; hf < KH; ++hf) {
  if (hf > 0) __syncthreads();                               // every wave is done reading the previous half as the 1x1's input
  if constexpr (KH == 1) {
    // Round 5: the phase as ONE software pipeline per wave.  A step = (chunk c, pixel tile pg) of a group of G pixel tiles: its B
    // fragment pair comes from BF two steps ahead, its 6 MFMAs (two hidden tiles x three products) are followed IN PROGRAM ORDER by
    // the relu -> split -> store of one finished tile of the PREVIOUS group -- vector work that issues while the matrix pipe
    // runs.  Before, a group was [B loads, 36 MFMAs, drain, 6 tile epilogues] and both waves of a SIMD went through the same
    // sequence at the same time: the phase took the SUM of its matrix, vector and LDS times (stamps: 9.7 k cycles at W = 16
    // for 3.6 k of matrix pipe per SIMD).  Chunks past pre_kc (the compiled chunk count is the next of 2 / 4 / 5) multiply ZERO
    // A fragments with chunk 0's B fragments: no branch inside the pipeline.
    const int kcp = p.pre_kc;
    u32x4 (&ah)[2][IMG_PRE_KC] = pre_ah, (&am)[2][IMG_PRE_KC] = pre_am;
    f32x4 (&pb)[2] = pre_pb;
    int (&lt2)[2] = pre_lt, (&o2)[2] = pre_o;
#ifdef GBNF_IMG_STAMP_A
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (diagnostic: the wait for this wave's A fragments in bucket 7)
    IMG_STAMP(7);
#endif
    constexpr int G = IMG_PRE_KC <= 2 ? (W == 16 ? 3 : 2) : (W == 16 ? 1 : (IMG_PRE_KC <= 4 ? 2 : 1));      // pixel tiles per group
    constexpr int NG = (NPH + G - 1) / G, STEPS = IMG_PRE_KC * G, NS = NG * STEPS;
    constexpr int TPS = (2 * G + STEPS - 1) / STEPS;                // finished tiles handed to a step (2 G per group)
    auto load_b = [&](int t, u32x4& bh, u32x4& bm) {                // global step t = (group, chunk, pixel tile of the group)
      const int gi = t / STEPS, st = t - gi * STEPS, c = st / G, pg = st - c * G;
      const int ptc = gi * G + pg < NPH ? gi * G + pg : NPH - 1;    // (a group that runs past the last tile repeats it)
      const unsigned char* f = BF + ((size_t)(ptc * kcp + (c < kcp ? c : 0)) * 2) * 1024 + 16 * lane;
      bh = *reinterpret_cast<const u32x4*>(f);
      bm = *reinterpret_cast<const u32x4*>(f + 1024);
    };
    auto tile_out = [&](const f32x4& a, int q, int pt) {            // relu(. + bias) -> split -> HB
      if (pt >= NPH) return;                                        // (compile time)
      const int lin = 16 * pt + i;
      // (no 1x1 behind it: what the last 3x3 reads outside the map proper must be the map's zero padding -- see the 1x1's write-back)
      const bool pv = FULL || n_mid > 0 || (hr0 + lin / W < p.Hv && lin % W < p.Wv);
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 16 * o2[q] + 4 * g + r;
        v[r] = (FULL || (co < p.hid && pv)) ? fmaxf(a[r], 0.0f) : 0.0f;
      }
      unsigned h01, m01, h23, m23;
      img_split_pair_w(v[0], v[1], h01, m01, amax);
      img_split_pair_w(v[2], v[3], h23, m23, amax);
      unsigned char* px = HB + (size_t)lin * pixb + 2 * (16 * lt2[q] + 4 * g);
      *reinterpret_cast<u32x2*>(px) = u32x2{h01, h23};
      *reinterpret_cast<u32x2*>(px + 2 * chh) = u32x2{m01, m23};
    };
    u32x4 bh[3], bm[3];
    load_b(0, bh[0], bm[0]);
    if (NS > 1) load_b(1, bh[1], bm[1]);
    f32x4 cur[2][G], pend[2][G];
#pragma unroll
    for (int gi = 0; gi <= NG; ++gi) {                              // (the extra round finishes the last group's tiles)
      // the finished group's output reaches into BF: every wave holds the B fragments of that group and the ones before it first
      // (the fragments of later groups lie above its output: 16 pixb >= 2048 pre_kc wherever BF has to share HB's tail)
      if (gi > 0 && gi * G * 16 * pixb > (int)p.bf_off) __syncthreads();
      if constexpr (KH == 1) {
        if (gi == NG && n_mid > 0) { p3_load(0, p3_ah[0], p3_am[0]); p3_load(1, p3_ah[1], p3_am[1]); }
      }
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        const int t = gi * STEPS + st, c = st / G, pg = st - c * G;
        if (gi < NG) {
          if (t + 2 < NS) load_b(t + 2, bh[(t + 2) % 3], bm[(t + 2) % 3]);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            if (c == 0) cur[q][pg] = pb[q];                  // (the bias of the tile's rows: see the 1x1's accumulators)
            cur[q][pg] = img_mfma16(am[q][c], bh[t % 3], cur[q][pg]);
            cur[q][pg] = img_mfma16(ah[q][c], bm[t % 3], cur[q][pg]);
            cur[q][pg] = img_mfma16(ah[q][c], bh[t % 3], cur[q][pg]);
          }
        }
        if (gi > 0) {
#pragma unroll
          for (int k = st * TPS; k < (st + 1) * TPS && k < 2 * G; ++k) tile_out(pend[k / G][k % G], k / G, (gi - 1) * G + k % G);
        }
        // the order inside a step: the look-ahead B fragments, then an MFMA every few vector instructions, the tile's stores last
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4 * TPS, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 2 * TPS, 0);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int pg = 0; pg < G; ++pg) pend[q][pg] = cur[q][pg];
    }
  } else {
    // KH = 2 (hidden widths 257 .. 512): group by group -- B loads, MFMAs, tile epilogues -- as in round 4.  Four output tiles of 1x1
    // accumulators per wave are live across this phase; the pipelined form above spills beside them.
    const gv4 pw = (gv4)p.pre_wp;
    const int kcp = p.pre_kc;
    // this wave's two hidden tiles (of this half): A fragments and biases once, for every pixel group
    u32x4 ah[2][IMG_PRE_KC], am[2][IMG_PRE_KC];
    f32x4 pb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int lt = wave + q * WV, o = lt < TH ? TH * hf + lt : kt, oo = o < kt ? o : 0;
#pragma unroll
      for (int c = 0; c < IMG_PRE_KC; ++c) {
        const gv4 f = pw + ((size_t)oo * kcp + (c < kcp ? c : 0)) * 128 + lane;
        ah[q][c] = f[0];
        am[q][c] = f[64];
      }
      pb[q] = ((const f32x4 __attribute__((address_space(1)))*)p.pre_bias)[(o < OT ? o : 0) * 4 + g];
    }
    constexpr int G = IMG_PRE_KC <= 2 ? (W == 16 ? 3 : 2) : (W == 16 ? 1 : (IMG_PRE_KC <= 4 ? 2 : 1));      // pixel tiles per group (a group's B operands live in registers)
#pragma unroll 1
    for (int pt0 = 0; pt0 < NPH; pt0 += G) {
      u32x4 bh[IMG_PRE_KC][G], bm[IMG_PRE_KC][G];
#pragma unroll
      for (int c = 0; c < IMG_PRE_KC; ++c) {
#pragma unroll
        for (int pg = 0; pg < G; ++pg) {
          const int ptc = pt0 + pg < NPH ? pt0 + pg : NPH - 1;          // (a group that runs past the last tile repeats it)
          const unsigned char* f = BF + ((size_t)(ptc * kcp + (c < kcp ? c : 0)) * 2) * 1024 + 16 * lane;
          bh[c][pg] = *reinterpret_cast<const u32x4*>(f);
          bm[c][pg] = *reinterpret_cast<const u32x4*>(f + 1024);
        }
      }
      if ((pt0 + G) * 16 * pixb > p.bf_off) __syncthreads();           // this group's output overwrites fragments: every wave holds its B operands first
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int lt = wave + q * WV, o = lt < TH ? TH * hf + lt : kt;      // (kt: past the end)
        {                                                    // (a tile index past the end computes a valid tile again: nothing is stored)
          f32x4 acc[G];
#pragma unroll
          for (int pg = 0; pg < G; ++pg) acc[pg] = pb[q];
#pragma unroll
          for (int c = 0; c < IMG_PRE_KC; ++c) {
            if (c < kcp) {
#pragma unroll
              for (int pg = 0; pg < G; ++pg) {
                acc[pg] = img_mfma16(am[q][c], bh[c][pg], acc[pg]);
                acc[pg] = img_mfma16(ah[q][c], bm[c][pg], acc[pg]);
                acc[pg] = img_mfma16(ah[q][c], bh[c][pg], acc[pg]);
              }
            }
          }
          img_drain(acc);
#pragma unroll
          for (int pg = 0; pg < G; ++pg) {
            if (pt0 + pg >= NPH || o >= kt) continue;
            const int lin = 16 * (pt0 + pg) + i;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int co = 16 * o + 4 * g + r;
              v[r] = (FULL || co < p.hid) ? fmaxf(acc[pg][r], 0.0f) : 0.0f;
            }
            unsigned h01, m01, h23, m23;
            img_split_pair_w(v[0], v[1], h01, m01, amax);
            img_split_pair_w(v[2], v[3], h23, m23, amax);
            unsigned char* px = HB + (size_t)lin * pixb + 2 * (16 * lt + 4 * g);
            {
              *reinterpret_cast<u32x2*>(px) = u32x2{h01, h23};
              *reinterpret_cast<u32x2*>(px + 2 * chh) = u32x2{m01, m23};
            }
          }
        }
      }
    }
  }
  IMG_STAMP(1);
  __syncthreads();
  IMG_STAMP(2);

  // ---- phase 3: 1x1 hidden -> hidden: this wave's NQ output tiles x every pixel tile, the k chunks of THIS half of its input
  //      (round 6: once per 1x1 layer -- n_mid = 0, 1 or 2 for hidden widths to 256; the layers behind the first find their input in HB)
#pragma unroll 1
  for (int m = 0; m < n_mid; ++m) {
    if (KH == 1 && m > 0) {
      // the layer in front is through: its output (relu, split) replaces its input in HB; the accumulators start again at THIS layer's biases
#pragma unroll
      for (int q = 0; q < NQ; ++q) img_drain(acc[q]);
      __syncthreads();                                       // every wave is done reading HB as that layer's input
      write_back(0);
      wp_cur = p.wp2;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        mid_b[q] = ((const f32x4 __attribute__((address_space(1)))*)p.bias2)[(ow[q] < OT ? ow[q] : 0) * 4 + g];
#pragma unroll
        for (int pt = 0; pt < NPH; ++pt) acc[q][pt] = mid_b[q];
      }
      p3_load(0, p3_ah[0], p3_am[0]);
      p3_load(1, p3_ah[1], p3_am[1]);
      __syncthreads();
    }
    auto load_a = p3_load;
    const unsigned char* bbase = HB + (size_t)i * pixb + 16 * g;
    // B fragments: ONE pixel tile (hi, mid) per step, requested D steps ahead of its MFMAs into a ring of D + 1 register pairs.
    // (Round 4 read half a chunk's tiles, waited, and issued their MFMAs: a wave alone on its SIMD -- the younger of the two falls
    //  behind and finishes the phase by itself, stamps: 11.8 k vs 17.9 k cycles for the same work -- stood at every one of those
    //  waits with the matrix pipe empty.)  The ring's position is a compile-time function of the step inside an unrolled loop body
    //  (3 NPH or 2 NPH steps: multiples of D + 1); a request past the last chunk re-reads chunk 0 and is dropped.
    constexpr int D = KH == 1 ? 2 : 1, SL = D + 1;
    u32x4 rb_h[SL], rb_m[SL];
    auto load_b = [&](int c, int pt, u32x4& bh, u32x4& bm) {               // c: chunk within HB
      const unsigned char* px = bbase + (size_t)(16 * pt) * pixb + 64 * (c < KCH ? c : 0);
      bh = *reinterpret_cast<const u32x4*>(px);
      bm = *reinterpret_cast<const u32x4*>(px + 2 * chh);
    };
    auto chunk = [&](auto k_c, int c, const u32x4 (&ah)[NQ], const u32x4 (&am)[NQ]) {      // k: position in the unrolled body
      constexpr int k = decltype(k_c)::value;
#pragma unroll
      for (int pt = 0; pt < NPH; ++pt) {
        const int t = k * NPH + pt;
        load_b(c + (pt + D) / NPH, (pt + D) % NPH, rb_h[(t + D) % SL], rb_m[(t + D) % SL]);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {                       // (an idle slot repeats tile 0: no branch in the stream, nothing stored)
          acc[q][pt] = img_mfma16(am[q], rb_h[t % SL], acc[q][pt]);
          acc[q][pt] = img_mfma16(ah[q], rb_m[t % SL], acc[q][pt]);
          acc[q][pt] = img_mfma16(ah[q], rb_h[t % SL], acc[q][pt]);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * NQ, 0);
      }
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
#pragma unroll
    for (int t = 0; t < D; ++t) load_b(t / NPH, t % NPH, rb_h[t % SL], rb_m[t % SL]);
    if constexpr (KH == 1) {
      // A fragments two chunks ahead (a chunk of an 8-wide map is 24 MFMAs: shorter than an L2 round trip under load)
      u32x4 ah[3][NQ], am[3][NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) { ah[0][q] = p3_ah[0][q]; am[0][q] = p3_am[0][q]; ah[1][q] = p3_ah[1][q]; am[1][q] = p3_am[1][q]; }
      int c = 0;
#pragma unroll 1
      for (; c + 3 <= KC; c += 3) {
        load_a(c + 2, ah[2], am[2]);
        chunk(K0{}, c, ah[0], am[0]);
        load_a(c + 3, ah[0], am[0]);
        chunk(K1{}, c + 1, ah[1], am[1]);
        load_a(c + 4, ah[1], am[1]);
        chunk(K2{}, c + 2, ah[2], am[2]);
      }
      if (c < KC) chunk(K0{}, c, ah[0], am[0]);
      if (c + 1 < KC) chunk(K1{}, c + 1, ah[1], am[1]);
    } else {
      // (four output tiles per wave: the A fragments one chunk ahead -- two sets of 4 x (hi, mid) are 64 registers beside 64 .. 96
      //  accumulator registers; a chunk is 48 .. 72 MFMAs here)
      u32x4 ah[2][NQ], am[2][NQ];
      const int c0 = hf * KCH;
      load_a(c0, ah[0], am[0]);
      int c = 0;
#pragma unroll 1
      for (; c + 2 <= KCH; c += 2) {
        load_a(c0 + c + 1, ah[1], am[1]);
        chunk(K0{}, c, ah[0], am[0]);
        load_a(c + 2 < KCH ? c0 + c + 2 : c0, ah[0], am[0]);
        chunk(K1{}, c + 1, ah[1], am[1]);
      }
      if (c < KCH) chunk(K0{}, c, ah[0], am[0]);
    }
  }
  }      // halves of the hidden channels (first 3x3 + its share of the 1x1's contraction)
#pragma unroll
  for (int q = 0; q < NQ; ++q) img_drain(acc[q]);
  IMG_STAMP(3);

  // ---- the 1x1's output (relu, split) goes back to HB half by half; each half is followed by its share of the last 3x3's
  //      contraction (phase 4).  KH = 1: one round, the sequence of round 4.
  float z2v[OT3][4];        // the state values this lane's epilogue updates (pixel tile = wave)
  float b3v[OT3][4];        // ... and the last 3x3's biases of its rows
  constexpr int MAXO = OT3;                                 // output tiles of the last 3x3 (template: no branches around its MFMAs)
  f32x4 part[MAXO][NPO];
#pragma unroll
  for (int o = 0; o < MAXO; ++o)
#pragma unroll
    for (int pt = 0; pt < NPO; ++pt) part[o][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (KH = 1) the last 3x3's first A fragments -- 16-wide maps: the nine taps of this wave's chunk for output tile 0; 8-wide: the
  // first two (tap, chunk) iterations -- requested in front of the 1x1's relu / split / stores and the barrier behind them
  // (not for two output tiles on a 16-wide map: 64 accumulator registers beside the nine taps' 72)
  constexpr bool P4 = KH == 1 && (W == 8 || OT3 == 1);
  constexpr int P4N = P4 ? (W == 16 ? 9 : 2 * OT3) : 1;
  u32x4 p4_ah[P4N], p4_am[P4N];
  if constexpr (P4) {
    const gv4 wp3 = (gv4)p.wp3;
    if constexpr (W == 16) {
      const int c = wave < KCH ? wave : 0;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const gv4 f = wp3 + ((size_t)tp * KC + c) * 128 + lane;
        p4_ah[tp] = f[0];
        p4_am[tp] = f[64];
      }
    } else {
      const int T_h = 9 * KCH;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int t = wave + k * WV, tt = t < T_h ? t : 0;
        const int tap = tt / KCH, c = tt - tap * KCH;
#pragma unroll
        for (int o = 0; o < OT3; ++o) {
          const gv4 f = wp3 + ((size_t)o * (9 * KC) + tap * KC + c) * 128 + lane;
          p4_ah[k * OT3 + o] = f[0];
          p4_am[k * OT3 + o] = f[64];
        }
      }
    }
  }
#pragma unroll
  for (int rf = 0; rf < KH; ++rf) {
  {
    __syncthreads();                                         // every wave is done reading HB (the 1x1's input / the previous round's last 3x3)
    IMG_STAMP(2);
    if (n_mid > 0) write_back(rf);
  }
  IMG_STAMP(4);
  __syncthreads();
  IMG_STAMP(2);

  // ---- phase 4: last 3x3 hidden -> shift / scale, contraction (tap, chunk) dealt to the waves (this round's chunks)
  if (rf == 0) {
    const float* stq = p.st + (int64_t)n * p.st_img;
    const int lin = 16 * (wave < NPO ? wave : 0) + i, row = r0 + lin / W, pc = lin % W;
    const int64_t pix = (int64_t)row * W + pc;
#pragma unroll
    for (int o = 0; o < OT3; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 16 * o + 4 * g + r;
        const int ch = EPI == EPI_COUPLE_ADD ? co : (co >> 1);          // affine: rows (2 j, 2 j + 1) = (shift, raw scale) of channel j
        const bool use = EPI == EPI_COUPLE_ADD ? co < p.cout : ((r & 1) == 0 && co + 1 < p.cout);
        z2v[o][r] = use ? stq[(int64_t)ch * H * W + pix] : 0.0f;
        b3v[o][r] = p.bias3[co < p.cout ? co : 0];
      }
  }
  if constexpr (W == 16) {
    // 16-wide maps: a pixel tile IS an image row, so the 3 x 3 runs "input-row stationary": wave w takes the 32-channel chunks
    // c = w, w + 8, ...; every hidden row j of the strip is read ONCE per output tile (two ds_read_b128) and multiplied by all nine
    // taps' weights (resident in registers): tap (dy, dx) of input row j belongs to output row j - dy, and its column shift is a
    // DPP row shift of the B operand itself (a 16-lane DPP row = the 16 columns of one k group; the lane shifted in from outside
    // the row is zero = the 'same' padding).  27 MFMAs per two LDS reads and 16 v_mov_dpp instead of 3 per two reads: the
    // (tap, chunk) form below spent more time fetching B operands and selecting their addresses than in its MFMAs.
    const gv4 wp3 = (gv4)p.wp3;
    const int T_all = 9 * KC;
    const int sh = r0 - hr0;                                // hidden row index of output row 0 (0 | 1)
    const unsigned char* rowb = HB + (size_t)i * pixb + 16 * g;
    auto shift_cols = [&](const u32x4& b, u32x4& l, u32x4& r) {
      // l: lane col holds column col - 1 (tap dx = -1), r: column col + 1.  Inline asm: on this toolchain
      // __builtin_amdgcn_update_dpp over the elements of a vector is folded to its FIRST element for all four.
      asm volatile(
          "s_nop 1\n\t"
          "v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_mov_b32_dpp %1, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_mov_b32_dpp %2, %10 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_mov_b32_dpp %3, %11 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_mov_b32_dpp %4, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_mov_b32_dpp %5, %9 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_mov_b32_dpp %6, %10 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_mov_b32_dpp %7, %11 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
          : "=&v"(l[0]), "=&v"(l[1]), "=&v"(l[2]), "=&v"(l[3]), "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
          : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
    };
#pragma unroll 1
    for (int c = wave; c < KCH; c += WV) {                  // chunk c of HB = chunk rf KCH + c of the contraction
#pragma unroll
      for (int o = 0; o < MAXO; ++o) {
        u32x4 ah[9], am[9];
        if (P4 && o == 0 && c == wave) {                                  // (uniform) the fragments requested ahead
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) { ah[tp] = p4_ah[tp < P4N ? tp : 0]; am[tp] = p4_am[tp < P4N ? tp : 0]; }
        } else {
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
            const gv4 f = wp3 + ((size_t)o * T_all + tp * KC + rf * KCH + c) * 128 + lane;
            ah[tp] = f[0];
            am[tp] = f[64];
          }
        }
        auto load_row = [&](int j, u32x4& bh, u32x4& bm) {             // hidden row j (relative to output row 0)
          const int hr = j + sh;                                        // its index in HB; outside [0, RH): outside the image
          const unsigned char* px = rowb + (size_t)((hr < 0 ? 0 : (hr >= RH ? RH - 1 : hr)) * 16) * pixb + 64 * c;
          bh = *reinterpret_cast<const u32x4*>(px);
          bm = *reinterpret_cast<const u32x4*>(px + 2 * chh);
        };
        u32x4 bh[2], bm[2];
        load_row(-1, bh[0], bm[0]);
#pragma unroll
        for (int j = -1; j <= NPO; ++j) {
          const int cur = (j + 1) & 1;
          if (j < NPO) load_row(j + 1, bh[cur ^ 1], bm[cur ^ 1]);
          const int hr = j + sh;
          if (hr >= 0 && hr < RH && r0 + j >= 0 && r0 + j < H) {       // (uniform) the row exists
            u32x4 sbh[3], sbm[3];
            sbh[1] = bh[cur];
            sbm[1] = bm[cur];
            shift_cols(bh[cur], sbh[0], sbh[2]);
            shift_cols(bm[cur], sbm[0], sbm[2]);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
              for (int dy = -1; dy <= 1; ++dy) {
                const int orow = j - dy;
                if (orow < 0 || orow >= NPO) continue;
                const int tp = (dy + 1) * 3 + dx;
                f32x4& acc = part[o][orow];
                acc = img_mfma16(am[tp], sbh[dx], acc);
                acc = img_mfma16(ah[tp], sbm[dx], acc);
                acc = img_mfma16(ah[tp], sbh[dx], acc);
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);                              // the next tile's 18 weight fragments must not be hoisted over this one's accumulators
      }
    }
  } else {
    const int T_all = 9 * KC, T_h = 9 * KCH;                 // (tap, chunk) iterations: of the whole contraction / of this round
    const gv4 wp3 = (gv4)p.wp3;
    const unsigned hb_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)HB;
    const unsigned zp_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)ZP + 16u * g;
    // per output pixel tile: the LDS address of the lane's centre pixel (hi piece, chunk 0) and which of its 3 x 3 neighbours
    // lie inside the image (bit 3 (dy + 1) + (dx + 1)); a tap outside reads the zero pixel (an address select, no operand masking)
    unsigned ctr_a[NPO], okm[NPO];
#pragma unroll
    for (int pt = 0; pt < NPO; ++pt) {
      const int lin = 16 * pt + i, pr = lin / W, pc = lin % W;
      const int row = r0 + pr;
      ctr_a[pt] = hb_a + (unsigned)(((row - hr0) * W + pc) * pixb) + 16u * g;
      unsigned m = 0;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int rr = row + tp / 3 - 1, cc = pc + tp % 3 - 1;
        m |= (rr >= 0 && rr < H && cc >= 0 && cc < W) ? (1u << tp) : 0u;
      }
      okm[pt] = m;
    }
    auto load_a = [&](int t, u32x4 (&ah)[MAXO], u32x4 (&am)[MAXO]) {
      const int tt = t < T_h ? t : 0;
      const int tap = tt / KCH, c = tt - tap * KCH;
#pragma unroll
      for (int o = 0; o < MAXO; ++o) {
        const gv4 f = wp3 + ((size_t)o * T_all + tap * KC + rf * KCH + c) * 128 + lane;
        ah[o] = f[0];
        am[o] = f[64];
      }
    };
    // B operands of one HALF of the pixel tiles; the other half's reads are in flight under this half's MFMAs
    constexpr int HP = NPO / 2;
    auto load_b = [&](int t, int half, u32x4 (&bh)[HP], u32x4 (&bm)[HP]) {
      const int tt = t < T_h ? t : 0;
      const int tap = tt / KCH, c = tt - tap * KCH;
      const int d = ((tap / 3 - 1) * W + (tap % 3 - 1)) * pixb + 64 * c;
#pragma unroll
      for (int q = 0; q < HP; ++q) {
        const int pt = half * HP + q;
        const bool inside = (okm[pt] >> tap) & 1u;
        const unsigned a = inside ? ctr_a[pt] + (unsigned)d : zp_a;
        bh[q] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>((uintptr_t)a);
        bm[q] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>((uintptr_t)(a + (inside ? 2u * chh : 0u)));
      }
    };
    auto mac_half = [&](auto half_c, const u32x4 (&bh)[HP], const u32x4 (&bm)[HP], const u32x4 (&ah)[MAXO], const u32x4 (&am)[MAXO]) {
      constexpr int half = decltype(half_c)::value;
#pragma unroll
      for (int o = 0; o < MAXO; ++o) {
#pragma unroll
        for (int q = 0; q < HP; ++q) {
          f32x4& acc = part[o][half * HP + q];
          acc = img_mfma16(am[o], bh[q], acc);
          acc = img_mfma16(ah[o], bm[q], acc);
          acc = img_mfma16(ah[o], bh[q], acc);
        }
      }
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    // A fragments two iterations ahead (an iteration is 24 OT3 MFMAs), B operands half an iteration ahead
    u32x4 a0h[MAXO], a0m[MAXO], a1h[MAXO], a1m[MAXO], a2h[MAXO], a2m[MAXO];
    u32x4 b0h[HP], b0m[HP], b1h[HP], b1m[HP];
    int t = wave;
    if constexpr (P4) {
#pragma unroll
      for (int o = 0; o < MAXO; ++o) {
        a0h[o] = p4_ah[o < P4N ? o : 0]; a0m[o] = p4_am[o < P4N ? o : 0];
        a1h[o] = p4_ah[MAXO + o < P4N ? MAXO + o : 0]; a1m[o] = p4_am[MAXO + o < P4N ? MAXO + o : 0];
      }
    } else {
      load_a(t, a0h, a0m);
      load_a(t + WV, a1h, a1m);
    }
    load_b(t, 0, b0h, b0m);
    auto step = [&](int tt, const u32x4 (&ah)[MAXO], const u32x4 (&am)[MAXO], u32x4 (&nh)[MAXO], u32x4 (&nm)[MAXO]) {
      load_a(tt + 2 * WV, nh, nm);
      load_b(tt, 1, b1h, b1m);
      mac_half(H0{}, b0h, b0m, ah, am);
      load_b(tt + WV, 0, b0h, b0m);
      mac_half(H1{}, b1h, b1m, ah, am);
    };
#pragma unroll 1
    for (; t + 2 * WV < T_h; t += 3 * WV) {
      step(t, a0h, a0m, a2h, a2m);
      step(t + WV, a1h, a1m, a0h, a0m);
      step(t + 2 * WV, a2h, a2m, a1h, a1m);
    }
    if (t < T_h) step(t, a0h, a0m, a2h, a2m);
    if (t + WV < T_h) step(t + WV, a1h, a1m, a0h, a0m);
  }
  }      // rounds: halves of the 1x1's output and their share of the last 3x3's contraction
#pragma unroll
  for (int o = 0; o < MAXO; ++o) img_drain(part[o]);
  IMG_STAMP(5);

  // ---- the partial tiles meet in LDS (over HB), one output tile at a time; wave w finishes pixel tile w
  //      (its z2 values were requested in front of the last 3x3: a global round trip less at the tail of the workgroup)
  float ld = 0.0f;
  f32x4* red = reinterpret_cast<f32x4*>(lds_raw);          // [wave][pt][64]
  float* st = p.st + (int64_t)n * p.st_img;
#pragma unroll
  for (int o = 0; o < MAXO; ++o) {
    {
      __syncthreads();                                     // nobody reads HB (or the previous tile's partials) any more
#pragma unroll
      for (int pt = 0; pt < NPO; ++pt) red[(wave * NPO + pt) * 64 + lane] = part[o][pt];
      __syncthreads();
      if (wave < NPO) {
        const int pt = wave;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < WV; ++w) acc += red[(w * NPO + pt) * 64 + lane];
        const int lin = 16 * pt + i, row = r0 + lin / W, pc = lin % W;
        const int64_t pix = (int64_t)row * W + pc;
        const bool valid = FULL || (row < p.Hv && pc < p.Wv);      // outside the map proper the state stays zero and adds no log-det
        if constexpr (EPI == EPI_COUPLE_ADD) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = 16 * o + 4 * g + r;
            const float hh = acc[r] + b3v[o][r];
            if (co < p.cout && valid) st[(int64_t)co * H * W + pix] = p.inverse ? z2v[o][r] - hh : z2v[o][r] + hh;   // models/glow.py:328-329, 349-350
          }
        } else {
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            const int co = 16 * o + 4 * g + 2 * qq, j = co >> 1;
            if (co + 1 < p.cout && valid) {
              const float h0 = acc[2 * qq] + b3v[o][2 * qq], h1 = acc[2 * qq + 1] + b3v[o][2 * qq + 1];
              float* zp = st + (int64_t)j * H * W + pix;
              const float e = __expf(-(h1 + 2.0f));                     // scale = sigmoid(raw + 2), models/glow.py:333
              const float sc = 1.0f / (1.0f + e);
              if (p.inverse) {
                *zp = z2v[o][2 * qq] * (1.0f + e) - h0;                 // z2 / scale - shift, models/glow.py:352-355
              } else {
                *zp = (z2v[o][2 * qq] + h0) * sc;                       // models/glow.py:334-335
                ld += -0.69314718055994531f * __builtin_amdgcn_logf(1.0f + e);    // log(scale) = -log(1 + e), models/glow.py:338 (v_log_f32: 1 ulp)
              }
            }
          }
        }
      }
    }
  }
  if constexpr (EPI == EPI_COUPLE_AFFINE) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) ld += __shfl_xor(ld, m);
    if (lane == 0 && wave < NPO && !p.inverse) atomicAdd(p.ldj + n, ld);
  }
  IMG_STAMP(6);
#ifdef GBNF_IMG_STAMPS
  if (p.dbg != nullptr && lane == 0)
    for (int kk = 0; kk < 8; ++kk) p.dbg[((size_t)blockIdx.x * WV + wave) * 8 + kk] = stamp_acc[kk];
#endif
  // ---- range watch: any operand of this workgroup beyond the fp16 range -> the image is marked for the exact-f32 path
  if (__any(!(amax <= 65504.0f)) && lane == 0) {
    if (p.sat != nullptr) atomicAdd(p.sat, 1ull);
    if (p.mark != nullptr) atomicOr(p.mark + n, 1u);
  }
}

// LDS layout: HB | ZP | zin | [BF]; BF (the first 3x3's B fragments) goes behind the rest when that fits 160 KB, else over the
// tail of HB (see phase 2).  *bf_off: byte offset of BF.  Hidden widths above 256 (KH = 2: HB holds half the channels at a time and
// is written twice) need BF beside HB -- the second half's first 3x3 reads it again.
static int img_net_hx3_halves(int chp) { return chp > 256 ? 2 : 1; }
static size_t img_net_hx3_layout(int W, int chp, int cin, int pre_kc, size_t* bf_off) {
  const int KH = img_net_hx3_halves(chp);
  if (chp > 512 || (KH == 2 && chp % 64 != 0)) return 0;
  const int RO = (W == 16 && KH == 2) ? 4 : 8, S = W == 16 ? 16 / RO : 1, RH = S == 1 ? RO : (S == 2 ? RO + 1 : RO + 2);
  const int NPO = RO * W / 16, NPH = RH * W / 16;
  const size_t pixb = 4 * (size_t)(chp / KH) + 16;
  const size_t hb = (size_t)RH * W * pixb;
  const size_t work = hb + pixb + (size_t)cin * (RH + 2) * (W + 2) * 4 + (size_t)32 * pre_kc * 4;      // HB | ZP | zin | im2col table
  const size_t bf = (size_t)NPH * pre_kc * 2 * 1024;
  size_t total = work, off = (work + 15) / 16 * 16;
  if (off + bf <= 160 * 1024) {
    total = off + bf;
  } else if (bf <= hb && KH == 1) {
    off = hb - bf;                                   // (hb and bf are multiples of 16)
  } else {
    total = 0;                                       // does not fit either way: the caller keeps the two-kernel form / the exact-f32 kernels
    off = 0;
  }
  if (bf_off) *bf_off = off;
  const size_t red = (size_t)8 * NPO * 64 * 16;
  return total == 0 ? 0 : (total > red ? total : red);
}
size_t img_net_hx3_lds(int W, int chp, int cin, int pre_kc, int cout) {
  if (cout > 48 || (W == 16 && cout > 32)) return 0;
  if (pre_kc > 7 || (pre_kc > 5 && (W != 8 || chp > 256))) return 0;      // 6 / 7 chunks: the 8-wide instantiations of one half only
  return img_net_hx3_layout(W, chp, cin, pre_kc, nullptr);
}

template <int W, int EPI, int OT3, int KH>
static hipError_t img_net_hx3_launch4(const NetLaunch& q0, int64_t n, hipStream_t s) {
  NetLaunch q = q0;
  size_t bf_off = 0;
  const size_t lds = img_net_hx3_layout(W, q.chp, q.pre_cin, q.pre_kc, &bf_off);
  if (lds == 0 || lds > 160 * 1024) return hipErrorInvalidValue;
  q.bf_off = (unsigned)bf_off;
  constexpr int RO = (W == 16 && KH == 2) ? 4 : 8;
  const dim3 grid((unsigned)(n * (W == 16 ? 16 / RO : 1))), blk(512);
  const int pk = q.pre_kc <= 2 ? 2 : (q.pre_kc <= 4 ? 4 : (q.pre_kc <= 5 ? 5 : 7));
  const bool full = q.hid == q.chp && q.Hv >= q.H && q.Wv >= W;
  if constexpr (W == 8 && KH == 1) {
    // 6 / 7 chunks (18 .. 24 input channels: the third level of a 3 x 32 x 32 input, a 4 x 4 map in 8 x 8 storage -- never FULL)
    if (pk == 7) {
      static bool attr7 = false;
      if (!attr7) {
        const hipError_t e = hipFuncSetAttribute((const void*)img_net_hx3_kernel<W, 7, EPI, OT3, false, KH>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr7 = true;
      }
      hipLaunchKernelGGL((img_net_hx3_kernel<W, 7, EPI, OT3, false, KH>), grid, blk, lds, s, q);
      return hipGetLastError();
    }
  } else {
    if (pk == 7) return hipErrorInvalidValue;
  }
  static bool attr_set = false;
  if (!attr_set) {
    const void* fns[6] = {(const void*)img_net_hx3_kernel<W, 2, EPI, OT3, false, KH>, (const void*)img_net_hx3_kernel<W, 4, EPI, OT3, false, KH>,
                          (const void*)img_net_hx3_kernel<W, 5, EPI, OT3, false, KH>, (const void*)img_net_hx3_kernel<W, 2, EPI, OT3, true, KH>,
                          (const void*)img_net_hx3_kernel<W, 4, EPI, OT3, true, KH>, (const void*)img_net_hx3_kernel<W, 5, EPI, OT3, true, KH>};
    for (int k = 0; k < 6; ++k) {
      const hipError_t e = hipFuncSetAttribute(fns[k], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
    }
    attr_set = true;
  }
  if (full) {
    if (pk == 2) hipLaunchKernelGGL((img_net_hx3_kernel<W, 2, EPI, OT3, true, KH>), grid, blk, lds, s, q);
    else if (pk == 4) hipLaunchKernelGGL((img_net_hx3_kernel<W, 4, EPI, OT3, true, KH>), grid, blk, lds, s, q);
    else hipLaunchKernelGGL((img_net_hx3_kernel<W, 5, EPI, OT3, true, KH>), grid, blk, lds, s, q);
  } else {
    if (pk == 2) hipLaunchKernelGGL((img_net_hx3_kernel<W, 2, EPI, OT3, false, KH>), grid, blk, lds, s, q);
    else if (pk == 4) hipLaunchKernelGGL((img_net_hx3_kernel<W, 4, EPI, OT3, false, KH>), grid, blk, lds, s, q);
    else hipLaunchKernelGGL((img_net_hx3_kernel<W, 5, EPI, OT3, false, KH>), grid, blk, lds, s, q);
  }
  return hipGetLastError();
}
template <int W, int EPI, int OT3>
static hipError_t img_net_hx3_launch3(const NetLaunch& q, int64_t n, hipStream_t s) {
  if (img_net_hx3_halves(q.chp) == 2) return img_net_hx3_launch4<W, EPI, OT3, 2>(q, n, s);
  return img_net_hx3_launch4<W, EPI, OT3, 1>(q, n, s);
}
template <int W, int EPI>
static hipError_t img_net_hx3_launch2(const NetLaunch& q, int64_t n, hipStream_t s) {
  const int ot3 = (q.cout + 15) >> 4;
  if (ot3 == 1) return img_net_hx3_launch3<W, EPI, 1>(q, n, s);
  if (ot3 == 2) return img_net_hx3_launch3<W, EPI, 2>(q, n, s);
  // (three output tiles on 16-wide maps -- 33..48 coupling outputs, i.e. >= 32 channels at the first level -- would spill:
  //  img_net_hx3_lds answers 0 for them and gbnf_image.hip keeps the two-kernel form)
  if constexpr (W == 8) return img_net_hx3_launch3<W, EPI, 3>(q, n, s);
  return hipErrorInvalidValue;
}
hipError_t img_net_hx3_launch(const NetLaunch& q, int W, bool additive, int64_t n, hipStream_t s) {
  if (W == 16) return additive ? img_net_hx3_launch2<16, EPI_COUPLE_ADD>(q, n, s) : img_net_hx3_launch2<16, EPI_COUPLE_AFFINE>(q, n, s);
  if (W == 8) return additive ? img_net_hx3_launch2<8, EPI_COUPLE_ADD>(q, n, s) : img_net_hx3_launch2<8, EPI_COUPLE_AFFINE>(q, n, s);
  return hipErrorInvalidValue;
}

}  // namespace gbnf
