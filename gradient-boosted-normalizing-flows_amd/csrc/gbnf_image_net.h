// gbnf_image_net.h -- the fused image coupling-net kernel's launch interface (gbnf_image_net.hip), shared with gbnf_image.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace gbnf {

// epilogues of the image convolution kernels (img_conv_kernel, img_last_hx3_kernel, img_net_hx3_kernel)
enum { EPI_RELU = 0, EPI_STORE = 1, EPI_COUPLE_AFFINE = 2, EPI_COUPLE_ADD = 3, EPI_SPLIT = 4,
       // the z -> x direction (gbnf_image_flow_inverse): coupling^-1 and Split2d's re-draw of the half it dropped
       EPI_COUPLE_AFFINE_INV = 5, EPI_COUPLE_ADD_INV = 6, EPI_SPLIT_INV = 7 };

struct NetLaunch {
  const float* pre_in;      // (n, *, H, W) f32: the coupling net's input z1
  int64_t pre_in_img;
  const unsigned* pre_wp;   // f16x3 fragments of the first 3x3 with taps folded into k: [tile][pre_kc][hi|mid][64][4 u32]
  const float* pre_bias;    // [16 * tiles]
  const int* pre_koff;      // [32 * pre_kc] im2col offsets of the folded contraction for a 6-row staging (img_mid_hx3's table)
  int pre_kc, pre_cin;
  const unsigned* wp;       // 1x1: [o][c][hi|mid][64][4 u32]
  const float* bias;
  int n_mid;                // round 6: 1x1 layers between the two 3x3s (coupling_network_depth): 1, or 0 / 2 for hidden widths to 256
  const unsigned* wp2;      //   the second 1x1 (n_mid == 2), as wp
  const float* bias2;
  const unsigned* wp3;      // last 3x3: [o][tap][c][hi|mid][64][4 u32]
  const float* bias3;
  float* st;                // coupled half z2 (n, *, H, W) f32, first channel of image 0
  int64_t st_img;
  float* ldj;
  int hid, chp, cout, H;
  int inverse;              // 1: the coupling's way back (models/glow.py:349-355): z2 - h, or z2 / scale - shift; no log-det
  int Hv, Wv;               // the map proper inside its H x W storage (gbnf_image.hip, ConvLaunch): rows < Hv, columns < Wv
  unsigned bf_off;          // byte offset of the first 3x3's B-fragment buffer in LDS (set by img_net_hx3_launch)
  unsigned long long* sat;  // per-device counter of workgroups that met an operand beyond the fp16 range, or null
  unsigned* mark;           // (n,) per-image marks (non-zero: re-evaluate this image on the exact-f32 path), or null
  const unsigned* only;     // repair launches: (n,) run only images whose entry is non-zero; null = all
  unsigned long long* dbg;  // diagnostic builds (-DGBNF_IMG_STAMPS) only: [workgroup][wave][8] phase cycle sums
};

// LDS bytes of img_net_hx3_kernel for W-wide maps (16 | 8), hidden width padded to chp, cin input channels of the first 3x3
// in pre_kc 32-wide chunks of its folded contraction, cout outputs of the last 3x3; 0 = this geometry has no fused kernel
size_t img_net_hx3_lds(int W, int chp, int cin, int pre_kc, int cout);
// One launch for n images (W = H = 16 | 8): grid n * (W == 16 ? 2 : 1) workgroups of 512 threads (hidden widths above 256: n * 4
// workgroups on 16-wide maps).
hipError_t img_net_hx3_launch(const NetLaunch& q, int W, bool additive, int64_t n, hipStream_t s);

}  // namespace gbnf
