// gbnf_api.hip -- C ABI of libgbnf_hip.so (include/gbnf.h): descriptor validation, host-side
// parameter packing into MFMA fragment order, variant dispatch, the mixture log-sum-exp kernel.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string.h>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "gbnf_flow_kernel_hx3.hip.h"
#include "gbnf_train_bwd.hip.h"
#include "gbnf_internal.h"

namespace gbnf {

// ------------------------------------------------------------------ error reporting
static thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define GBNF_HIP(call)                                                                     \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) return fail(GBNF_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

// ------------------------------------------------------------------ variant registry
struct Variant {
  VariantKey key;
  LaunchFn fn;
  const char* name;
};
static std::vector<Variant>& variants() {
  static std::vector<Variant> v;
  return v;
}
void register_variant(const VariantKey& key, LaunchFn fn, const char* name) {
  variants().push_back(Variant{key, fn, name});
}

// ------------------------------------------------------------------ handles
struct NetDims {
  int in_f = 0, hidden = 0, out_f = 0, depth = 0, act = 0, residual = 0;
};

// ---- numerics guard (include/gbnf.h, gbnf_numerics_status): device-side re-check of the f16x3 choice on the caller's data
constexpr int GUARD_ROWS = 256;
struct GuardStatus {             // pinned host memory, written by guard_compare_kernel through its device alias
  unsigned long long checks;
  float worst_rel_err;
  unsigned demoted;
};
struct Guard {
  unsigned* flag_dev = nullptr;          // [0] != 0: the check failed (read by every repair launch of the handle)
  float* scratch_dev = nullptr;          // [2][n_comp][GUARD_ROWS] log-likelihoods of the check rows: f16x3 | bf16x6
  GuardStatus* status_host = nullptr;    // hipHostMalloc, mapped
  GuardStatus* status_dev = nullptr;     // the device's address of the same memory
  int n_comp = 0;
  std::atomic<long long> launches{0};
  // the check rows of a launch live in scratch_dev until guard_compare_kernel has read them: a check on ANOTHER stream while one
  // is still in flight would race on them -- such a check is skipped (the next launch checks instead)
  hipEvent_t check_done = nullptr;       // recorded behind the last check's compare launch
  hipStream_t check_stream = nullptr;
  bool check_pending = false;
  std::mutex check_mutex;
  int device = 0;
};

}  // namespace gbnf

struct gbnf_flow {
  int math_mode = 0;   // GBNF_MATH_* actually in use (F32, F16X3 or BF16X6)
  int requested_mode = 0;     // what the caller asked for (GBNF_MATH_DEFAULT = chosen by the probe)
  float probe_rel_err = -1.0f;   // DEFAULT mode: largest relative log-likelihood difference f16x3 vs bf16x6 on the probe batch
  int kind = 0, d = 0, n_steps = 0, additive = 0;
  int hidden = 0, depth = 0, act_a = 0, act_b = 0, residual = 0;
  int ht = 0, ksl = 0, ot = 0;     // tile geometry (exact)
  int ks1 = 0;
  int var_ht = 0, var_ksl = 0, var_ks1 = 0, var_ot = 0;  // geometry of the compiled variant the blob was packed for
  gbnf::LaunchFn launch_nt[3] = {nullptr, nullptr, nullptr};  // index = NT
  const char* name_nt[3] = {nullptr, nullptr, nullptr};
  // f16x3 handles of a depth-1 TanhNet / ReLUNet whose geometry has a cooperative (latency-form) variant: the same blob on
  // flow_kernel_coop (csrc/gbnf_flow_kernel_coop.hip.h), taken by launch_flow for calls of a few sample tiles (pick_coop)
  gbnf::LaunchFn launch_coop_nt[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};      // index = form: 1 = 16-sample tiles / 4 waves, 2 = 32 / 4, 3 = 32 / 8, 4 = 16 / 8
  const char* name_coop_nt[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  uint32_t* blob_dev = nullptr;
  const uint32_t** self_table_dev = nullptr;  // 1-entry blob table for single-flow launches
  size_t blob_words = 0;
  double macs = 0, padded_macs = 0;
  // split kernels: `blob_dev` / `launch_nt` belong to `math_mode`; an f16x3 handle also carries the bf16x6 packing of the
  // same parameters: the repair pass behind every f16x3 launch (samples that left the fp16 range) and the mode a
  // mixture falls back to when its components were created in different split modes
  gbnf::LaunchFn launch2_nt[3] = {nullptr, nullptr, nullptr};
  const char* name2_nt[3] = {nullptr, nullptr, nullptr};
  uint32_t* blob2_dev = nullptr;
  const uint32_t** self_table2_dev = nullptr;
  size_t blob2_words = 0;
  double padded_macs2 = 0;
  int var2_ht = 0, var2_ot = 0;
  gbnf::Guard* guard = nullptr;        // numerics guard (DEFAULT handles running f16x3 with a bf16x6 packing), else null
};

struct gbnf_mixture {
  std::vector<gbnf_flow*> flows;
  int math_mode = 0;                    // the mode every component runs in
  bool use_blob2 = false;               // components were created as f16x3 but the mixture runs their bf16x6 packing
  const uint32_t** table_dev = nullptr;
  const uint32_t** table2_dev = nullptr;   // bf16x6 packings for the repair pass of an f16x3 mixture (or null)
  float* base_dev = nullptr;  // [2][d] mean, std or null
  gbnf::Guard* guard = nullptr;        // numerics guard of an f16x3 mixture whose components are all DEFAULT handles, else null
};

namespace gbnf {

static int ceil_div(int a, int b) { return (a + b - 1) / b; }

// DEFAULT math mode keeps a component on f16x3 when its probe agrees with bf16x6 to this relative log-likelihood error
// (a quarter of the 1e-5 bar: headroom for data that is harder than the probe batch)
constexpr float PROBE_MAX_REL_ERR = 2.5e-6f;

// One counter per device (a kernel may only touch memory of the device it runs on): allocated on first use by a launch
// on that device, never freed.
constexpr int MAX_DEVICES = 64;
static std::mutex g_dev_mu;
static unsigned* g_sat_counter[MAX_DEVICES] = {};

unsigned* saturation_counter() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
  std::lock_guard<std::mutex> lk(g_dev_mu);
  if (g_sat_counter[dev] == nullptr) {
    unsigned* p = nullptr;
    const size_t bytes = sizeof(unsigned long long) * (SAT_MARKS + SAT_SLOTS);      // counter + the launch marks of the repair protocol (64-bit words)
    if (hipMalloc((void**)&p, bytes) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, bytes) != hipSuccess) { (void)hipFree(p); return nullptr; }
    g_sat_counter[dev] = p;
  }
  return g_sat_counter[dev];
}
unsigned* training_saturation_counter() {
  unsigned* p = saturation_counter();
  return p ? p + 2 : nullptr;          // 64-bit word [1] (SAT_MARKS = 2: words [0] and [1] are counters, the launch marks follow)
}

// ---- launch-policy knobs (gbnf_tuning_set / _get; initialised from the environment at first use)
struct Tuning {
  std::atomic<int> force_nt{0}, wg_pairs{-1}, repair{1}, nt2_min_waves{1024}, check_every{256}, check_tolerance_e9{2500};
  std::atomic<int> coop{-1}, coop_max_wgs{256};        // latency-form kernel: -1 automatic, 0 never, 1 / 2 / 3 always form 1 / 2 / 3
  Tuning() {
    auto env = [](const char* k) -> const char* { return getenv(k); };
    if (const char* e = env("GBNF_FORCE_NT")) force_nt = atoi(e);
    if (const char* e = env("GBNF_NO_WG_PAIRS")) wg_pairs = atoi(e) != 0 ? 0 : -1;
    if (const char* e = env("GBNF_NO_REPAIR")) repair = atoi(e) != 0 ? 0 : 1;
    if (const char* e = env("GBNF_NT2_MIN_WAVES")) nt2_min_waves = atoi(e);
    if (const char* e = env("GBNF_CHECK_EVERY")) check_every = atoi(e);
  }
};
static Tuning& tuning() {
  static Tuning t;
  return t;
}
int tuning_wg_pairs() { return tuning().wg_pairs.load(std::memory_order_relaxed); }
static std::atomic<int>* tuning_slot(const char* key) {
  Tuning& t = tuning();
  if (!key) return nullptr;
  if (!strcmp(key, "force_nt")) return &t.force_nt;
  if (!strcmp(key, "coop")) return &t.coop;
  if (!strcmp(key, "coop_max_wgs")) return &t.coop_max_wgs;
  if (!strcmp(key, "wg_pairs")) return &t.wg_pairs;
  if (!strcmp(key, "repair")) return &t.repair;
  if (!strcmp(key, "nt2_min_waves")) return &t.nt2_min_waves;
  if (!strcmp(key, "check_every")) return &t.check_every;
  if (!strcmp(key, "check_tolerance_e9")) return &t.check_tolerance_e9;
  return nullptr;
}

// Guards are POOLED per (device, component count): BoostedFlow re-packs a component after every optimiser step, and a new guard
// per re-pack meant two hipMalloc, a mapped hipHostMalloc and later their frees -- device-wide synchronisations.
static std::mutex g_guard_pool_mutex;
static std::vector<Guard*> g_guard_pool;

static void destroy_guard(Guard* g) {
  if (!g) return;
  if (g->flag_dev) (void)hipFree(g->flag_dev);
  if (g->scratch_dev) (void)hipFree(g->scratch_dev);
  if (g->status_host) (void)hipHostFree(g->status_host);
  if (g->check_done) (void)hipEventDestroy(g->check_done);
  delete g;
}
static void free_guard(Guard* g) {
  if (!g) return;
  // the flag is cleared HERE, on the null stream (by contract nothing of the dying handle is in flight): a later create takes the
  // guard from the pool behind its own synchronous parameter upload, which is ordered after this
  const bool ok = hipMemsetAsync(g->flag_dev, 0, 16, nullptr) == hipSuccess;
  std::lock_guard<std::mutex> lk(g_guard_pool_mutex);
  if (ok && g_guard_pool.size() < 64) g_guard_pool.push_back(g);
  else destroy_guard(g);
}
static hipError_t make_guard(int n_comp, Guard** out) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(g_guard_pool_mutex);
    for (size_t k = 0; k < g_guard_pool.size(); ++k) {
      Guard* g = g_guard_pool[k];
      if (g->n_comp == n_comp && g->device == dev) {
        g_guard_pool.erase(g_guard_pool.begin() + (long)k);
        g->status_host->checks = 0; g->status_host->worst_rel_err = 0.0f; g->status_host->demoted = 0;
        g->launches.store(0);
        g->check_pending = false;
        *out = g;
        return hipSuccess;
      }
    }
  }
  Guard* g = new Guard();
  g->n_comp = n_comp;
  g->device = dev;
  hipError_t e = hipMalloc((void**)&g->flag_dev, 16);
  if (e == hipSuccess) e = hipMemset(g->flag_dev, 0, 16);
  if (e == hipSuccess) e = hipMalloc((void**)&g->scratch_dev, sizeof(float) * 2 * n_comp * GUARD_ROWS);
  if (e == hipSuccess) e = hipHostMalloc((void**)&g->status_host, sizeof(GuardStatus), hipHostMallocMapped);
  if (e == hipSuccess) {
    g->status_host->checks = 0; g->status_host->worst_rel_err = 0.0f; g->status_host->demoted = 0;
    e = hipHostGetDevicePointer((void**)&g->status_dev, g->status_host, 0);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&g->check_done, hipEventDisableTiming);
  if (e != hipSuccess) { destroy_guard(g); return e; }
  *out = g;
  return hipSuccess;
}

// physical hidden position p = 16 t + 4 g + r  <->  logical unit 4*(4t + r) + g
static int phys_to_logical(int p, int h) {
  const int t = p / 16, gg = (p % 16) / 4, r = p % 4;
  const int l = 4 * (4 * t + r) + gg;
  return l < h ? l : -1;
}

static int check_net(const gbnf_net& net, int in_f, int out_f, NetDims* dims, const char* what) {
  if (net.n_layers < 2 || net.layers == nullptr)
    return fail(GBNF_ERR_INVALID, "%s: coupling network needs >= 2 Linear layers", what);
  const bool residual = net.activation == GBNF_ACT_RESIDUAL_RELU;
  if (net.activation != GBNF_ACT_TANH && net.activation != GBNF_ACT_RELU && !residual)
    return fail(GBNF_ERR_UNSUPPORTED, "%s: activation %d not supported (tanh / relu / residual-relu)", what, net.activation);
  if (residual && (net.n_layers < 4 || net.n_layers % 2 != 0))
    return fail(GBNF_ERR_INVALID, "%s: a ResidualNet has 2 + 2 B Linear layers (got %d)", what, net.n_layers);
  const int h = net.layers[0].out_features;
  for (int l = 0; l < net.n_layers; ++l) {
    const gbnf_linear& lin = net.layers[l];
    const int want_in = (l == 0) ? in_f : h;
    const int want_out = (l == net.n_layers - 1) ? out_f : h;
    if (lin.weight == nullptr || lin.bias == nullptr)
      return fail(GBNF_ERR_INVALID, "%s: layer %d has a null weight/bias", what, l);
    if (lin.in_features != want_in || lin.out_features != want_out)
      return fail(GBNF_ERR_INVALID, "%s: layer %d is %dx%d, expected %dx%d", what, l, lin.out_features,
                  lin.in_features, want_out, want_in);
  }
  dims->in_f = in_f; dims->hidden = h; dims->out_f = out_f;
  dims->depth = net.n_layers - 2;                 // hidden -> hidden Linear layers (a ResidualNet of B blocks: 2 B)
  dims->act = residual ? GBNF_ACT_RELU : net.activation;
  dims->residual = residual ? 1 : 0;
  return GBNF_OK;
}

// Pack one coupling network at word offset `base` of `blob` for variant geometry (HT, KS1, OT, LMID).
// The affine "cross" layout (shift_j, raw_j adjacent) equals natural row order of the last Linear.
static void pack_net(std::vector<uint32_t>& blob, size_t base, const gbnf_net& net, int HT, int KS1, int OT,
                     int LMID, int in_f, int h, int out_f) {
  const NetLayoutRT L(HT, KS1, OT, LMID);
  auto put = [&](size_t off, float v) { std::memcpy(&blob[base + off], &v, 4); };

  // layer 0: A fragment for tile t: lane (i,g) holds W0[unit(16t+i)][k = 4s+g] for s = 0..KS1-1
  const gbnf_linear& l0 = net.layers[0];
  for (int t = 0; t < HT; ++t)
    for (int lane = 0; lane < 64; ++lane) {
      const int i = lane & 15, gg = lane >> 4;
      const int u = phys_to_logical(16 * t + i, h);
      for (int s = 0; s < 4 * L.KQ; ++s) {
        const int k = 4 * s + gg;
        float v = 0.0f;
        if (u >= 0 && s < KS1 && k < in_f) v = l0.weight[(size_t)u * in_f + k];
        put(L.W1 + (size_t)t * L.W1_TILE + (size_t)lane * L.KQ * 4 + s, v);
      }
    }
  for (int t = 0; t < HT; ++t)
    for (int gg = 0; gg < 4; ++gg)
      for (int r = 0; r < 4; ++r) {
        const int u = phys_to_logical(16 * t + 4 * gg + r, h);
        put(L.B1 + (size_t)t * 16 + gg * 4 + r, u >= 0 ? l0.bias[u] : 0.0f);
      }
  // hidden -> hidden layers: A fragment for (out tile u, k-chunk t): lane (i,g) reg r =
  //   W[unit(16u+i)][unit(16t+4g+r)]
  for (int m = 0; m < LMID; ++m) {
    const gbnf_linear& lm = net.layers[1 + m];
    const size_t wb = L.MID0 + (size_t)m * L.MID_STRIDE;
    for (int u = 0; u < HT; ++u)
      for (int t = 0; t < HT; ++t)
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, gg = lane >> 4;
          const int uo = phys_to_logical(16 * u + i, h);
          for (int r = 0; r < 4; ++r) {
            const int ui = phys_to_logical(16 * t + 4 * gg + r, h);
            float v = 0.0f;
            if (uo >= 0 && ui >= 0) v = lm.weight[(size_t)uo * h + ui];
            put(wb + (((size_t)u * HT + t) * 64 + lane) * 4 + r, v);
          }
        }
    for (int u = 0; u < HT; ++u)
      for (int gg = 0; gg < 4; ++gg)
        for (int r = 0; r < 4; ++r) {
          const int uo = phys_to_logical(16 * u + 4 * gg + r, h);
          put(wb + L.MID_W + (size_t)u * 16 + gg * 4 + r, uo >= 0 ? lm.bias[uo] : 0.0f);
        }
  }
  // last layer: A fragment for (k-chunk u, out tile o): lane (i,g) reg r = W[row 16o+i][unit(16u+4g+r)]
  const gbnf_linear& ll = net.layers[net.n_layers - 1];
  for (int u = 0; u < HT; ++u)
    for (int o = 0; o < OT; ++o)
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, gg = lane >> 4;
        const int row = 16 * o + i;
        for (int r = 0; r < 4; ++r) {
          const int ui = phys_to_logical(16 * u + 4 * gg + r, h);
          float v = 0.0f;
          if (row < out_f && ui >= 0) v = ll.weight[(size_t)row * h + ui];
          put(L.W3 + (((size_t)u * OT + o) * 64 + lane) * 4 + r, v);
        }
      }
  for (int o = 0; o < OT; ++o)
    for (int gg = 0; gg < 4; ++gg)
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * o + 4 * gg + r;
        put(L.B3 + (size_t)o * 16 + gg * 4 + r, row < out_f ? ll.bias[row] : 0.0f);
      }
}

static size_t net_words(int HT, int KS1, int OT, int LMID) { return (size_t)NetLayoutRT(HT, KS1, OT, LMID).NET_WORDS; }

// ---- split-operand packing (f16x3 / bf16x6) ------------------------------------------------------------------
static uint16_t f16_bits(float x) {
  const _Float16 h = static_cast<_Float16>(x);      // round to nearest even
  uint16_t b;
  std::memcpy(&b, &h, 2);
  return b;
}
static float f16_to_f32(uint16_t b) {
  _Float16 h;
  std::memcpy(&h, &b, 2);
  return static_cast<float>(h);
}
static uint16_t bf16_bits(float x) {                  // round to nearest even (finite inputs)
  uint32_t u;
  std::memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf16_to_f32(uint16_t b) {
  const uint32_t u = (uint32_t)b << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// The NP pieces of one weight tile at word offset `off` (NP consecutive fragments, largest piece first): 16 rows x 32
// k-slots.  elem(i, g, j) returns the f32 weight of row i and k-slot (g, j); lane (i,g) stores its 8 narrow values as
// 4 words, element 2q in the low half.  NP = 2: fp16 (hi, mid); NP = 3: bf16 (p0, p1, p2); every piece rounded to nearest.
// Returns false if a weight does not fit the narrow type's range (fp16: |w| > 65504).
template <typename F>
static bool put_tile(std::vector<uint32_t>& blob, size_t off, int NP, F elem) {
  bool ok = true;
  for (int lane = 0; lane < 64; ++lane) {
    const int i = lane & 15, gg = lane >> 4;
    for (int q = 0; q < 4; ++q) {
      uint32_t w[3] = {0, 0, 0};
      for (int e = 0; e < 2; ++e) {
        float r = elem(i, gg, 2 * q + e);
        if (NP == 2 && !(std::fabs(r) <= 65504.0f)) ok = false;
        for (int k = 0; k < NP; ++k) {
          const uint16_t b = NP == 2 ? f16_bits(r) : bf16_bits(r);
          w[k] |= (uint32_t)b << (16 * e);
          r -= NP == 2 ? f16_to_f32(b) : bf16_to_f32(b);
        }
      }
      for (int k = 0; k < NP; ++k) blob[off + (size_t)k * 256 + (size_t)lane * 4 + q] = w[k];
    }
  }
  return ok;
}

// k-slot (g, j) of hidden chunk c  <->  hidden unit 16*(2c + (j>>2)) + 4g + (j&3): the D-register layout of two
// consecutive 16-unit accumulator tiles read as one k = 32 B operand
static int hx3_hidden_unit(int c, int gg, int j) { return 16 * (2 * c + (j >> 2)) + 4 * gg + (j & 3); }

static bool pack_net_hx3(std::vector<uint32_t>& blob, size_t base, const gbnf_net& net, int HT, int OT, int NP, int depth,
                         int in_f, int h, int out_f) {
  const Hx3Layout L(HT, OT, NP, depth);
  auto put = [&](size_t off, float v) { std::memcpy(&blob[base + off], &v, 4); };
  const gbnf_linear& l0 = net.layers[0];
  const gbnf_linear& lout = net.layers[depth + 1];
  bool ok = true;
  // tanh networks (gbnf_flow_kernel_hx3.hip.h, tanh_hx3): a layer whose output goes through tanh carries the factor
  // T = 2*log2(e) of tanh(x) = 1 - 2/(2^(T x) + 1); the kernel hands on r = 1/(2^(T x) + 1), and the layer that
  // consumes t = 1 - 2 r folds the affine map:  W.t + b = (-2 W).r + (b + W.1)   (row sums in double)
  const bool tanh_net = net.activation == GBNF_ACT_TANH;
  const float T = tanh_net ? 2.8853900817779268f : 1.0f;
  const float R = tanh_net ? -2.0f : 1.0f;                  // factor on the weights of a layer fed by an activation
  auto folded_bias = [&](const gbnf_linear& l, int row, int n_in) {
    double b = l.bias[row];
    if (tanh_net)
      for (int k = 0; k < n_in; ++k) b += (double)l.weight[(size_t)row * n_in + k];
    return (float)b;
  };
  for (int t = 0; t < HT; ++t)
    for (int k = 0; k < 16; ++k) {
      const int u = 16 * t + k;
      put((size_t)t * 16 + k, u < h ? T * l0.bias[u] : 0.0f);
      for (int j = 1; j <= depth; ++j)
        put((size_t)(j * HT + t) * 16 + k, u < h ? T * folded_bias(net.layers[j], u, h) : 0.0f);
    }
  for (int o = 0; o < OT; ++o)
    for (int k = 0; k < 16; ++k) {
      const int r = 16 * o + k;
      put((size_t)((depth + 1) * HT + o) * 16 + k, r < out_f ? folded_bias(lout, r, h) : 0.0f);
    }
  const size_t TW = (size_t)NP * 256;                        // words per weight tile
  auto out_tile = [&](size_t off, int c, int o) {             // output-layer tile o of hidden chunk c
    ok &= put_tile(blob, off, NP, [&](int i, int gg, int j) {
      const int row = 16 * o + i, ui = hx3_hidden_unit(c, gg, j);
      return (row < out_f && ui < h) ? R * lout.weight[(size_t)row * h + ui] : 0.0f;
    });
  };
  int s = 0;
  for (int i0 = 0; i0 < L.N_L0; ++i0, ++s) {               // layer 0: k-slot (g,j) = input feature 8g + j
    const int t0 = i0 * L.TL0;
    for (int tl = 0; tl < L.nf[s] / NP; ++tl) {
      const int t = t0 + tl;
      ok &= put_tile(blob, base + L.off[s] + (size_t)tl * TW, NP, [&](int i, int gg, int j) {
        const int u = 16 * t + i, k = 8 * gg + j;
        return (u < h && k < in_f) ? T * l0.weight[(size_t)u * in_f + k] : 0.0f;
      });
    }
  }
  for (int jl = 1; jl <= depth; ++jl) {
    const gbnf_linear& lj = net.layers[jl];
    for (int u = 0; u < HT; ++u, ++s) {                       // hidden row u (+ output chunk (u-2)/2 in the last hidden layer)
      for (int c = 0; c < L.HC; ++c)
        ok &= put_tile(blob, base + L.off[s] + (size_t)c * TW, NP, [&](int i, int gg, int j) {
          const int uo = 16 * u + i, ui = hx3_hidden_unit(c, gg, j);
          return (uo < h && ui < h) ? T * R * lj.weight[(size_t)uo * h + ui] : 0.0f;
        });
      if (jl == depth && u % 2 == 0 && u >= 2)
        for (int o = 0; o < OT; ++o) out_tile(base + L.off[s] + (size_t)(L.HC + o) * TW, (u - 2) / 2, o);
    }
  }
  if (depth >= 1) {                                            // drain: output chunk HC-1
    for (int o = 0; o < OT; ++o) out_tile(base + L.off[s] + (size_t)o * TW, L.HC - 1, o);
  } else {                                                     // no hidden layer: output stages of CG chunks
    for (int k = 0; k < L.N_OUT; ++k, ++s) {
      const int c0 = k * L.CG, cnt = std::min(L.CG, L.HC - c0);
      for (int cc = 0; cc < cnt; ++cc)
        for (int o = 0; o < OT; ++o) out_tile(base + L.off[s] + (size_t)(cc * OT + o) * TW, c0 + cc, o);
    }
  }
  return ok;
}

static const Variant* find_variant(const VariantKey& k) {
  for (const Variant& v : variants())
    if (v.key == k) return &v;
  return nullptr;
}

}  // namespace gbnf

using namespace gbnf;

extern "C" {

int gbnf_version(void) { return GBNF_ABI_VERSION; }

int gbnf_saturation_count(int64_t* count, int32_t reset) {
  if (count == nullptr) return fail(GBNF_ERR_INVALID, "gbnf_saturation_count: count is null");
  *count = 0;
  unsigned* dev = gbnf::saturation_counter();                // the CURRENT device's counter
  if (dev == nullptr) return GBNF_OK;
  unsigned long long host = 0;
  // every stream of the device (torch's side streams are non-blocking ones: the null-stream copy alone would not
  // order against their kernels)
  unsigned long long both[2] = {0, 0};      // [0] evaluation kernels (repaired in the same call), [1] training kernels (saturated)
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(both, dev, sizeof(both), hipMemcpyDeviceToHost);
  if (e == hipSuccess && reset) e = hipMemset(dev, 0, sizeof(both));
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_saturation_count: %s", hipGetErrorString(e));
  host = both[0] + both[1];
  *count = (int64_t)host;
  return GBNF_OK;
}

int gbnf_training_saturation_count(int64_t* count, int32_t reset) {
  if (count == nullptr) return fail(GBNF_ERR_INVALID, "gbnf_training_saturation_count: count is null");
  *count = 0;
  unsigned* dev = gbnf::training_saturation_counter();
  if (dev == nullptr) return GBNF_OK;
  unsigned long long host = 0;
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(&host, dev, sizeof(host), hipMemcpyDeviceToHost);
  if (e == hipSuccess && reset) e = hipMemset(dev, 0, sizeof(host));
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_training_saturation_count: %s", hipGetErrorString(e));
  *count = (int64_t)host;
  return GBNF_OK;
}

const char* gbnf_last_error(void) { return g_err.c_str(); }

static int default_math_mode() {
  static const int mode = [] {
    const char* e = getenv("GBNF_MATH");          // "f32" | "f16x3": tuning / test knob
    if (e && !strcmp(e, "f32")) return (int)GBNF_MATH_F32;
    if (e && !strcmp(e, "f16x3")) return (int)GBNF_MATH_F16X3;
    if (e && !strcmp(e, "bf16x6")) return (int)GBNF_MATH_BF16X6;
    return (int)GBNF_MATH_DEFAULT;
  }();
  return mode;
}

int gbnf_flow_create(const gbnf_flow_desc* desc, gbnf_flow** out) {
  return gbnf_flow_create_mode(desc, default_math_mode(), out);
}

// Shape / consistency rules of a component descriptor; reads the descriptor's HOST fields and the permutation indices,
// never the parameter arrays themselves (so the trainer can validate a descriptor whose parameters live on the device).
struct DescInfo {
  NetDims ref{};
  int act_a = 0, act_b = 0, ht = 0, ksl = 0, ot = 0, ks1 = 0;
};

static int validate_desc(const gbnf_flow_desc* desc, DescInfo* info) {
  if (desc == nullptr) return fail(GBNF_ERR_INVALID, "gbnf_flow_create: desc is null");
  const int d = desc->d, K = desc->n_steps;
  if (d < 2 || d > ZSLOTS) return fail(GBNF_ERR_UNSUPPORTED, "d=%d outside the supported range [2,%d]", d, ZSLOTS);
  if (K < 1) return fail(GBNF_ERR_INVALID, "n_steps must be >= 1");
  if (desc->kind != GBNF_KIND_GLOW && desc->kind != GBNF_KIND_REALNVP)
    return fail(GBNF_ERR_INVALID, "unknown flow kind %d", desc->kind);
  const bool glow = desc->kind == GBNF_KIND_GLOW;
  if (glow && desc->glow_steps == nullptr) return fail(GBNF_ERR_INVALID, "glow_steps is null");
  if (!glow && desc->realnvp_steps == nullptr) return fail(GBNF_ERR_INVALID, "realnvp_steps is null");
  const bool additive = glow && desc->coupling == GBNF_COUPLING_ADDITIVE;
  if (glow && desc->coupling != GBNF_COUPLING_AFFINE && desc->coupling != GBNF_COUPLING_ADDITIVE)
    return fail(GBNF_ERR_INVALID, "unknown coupling %d", desc->coupling);
  const int d1 = d / 2, d2 = d - d1;
  // ---- validate every step, collect the common network geometry
  NetDims ref{};
  int act_a = 0, act_b = 0;
  int max_out_entries = 0;
  int max_in = 0;
  for (int s = 0; s < K; ++s) {
    NetDims a{}, b{};
    int rc;
    char what[64];
    if (glow) {
      const gbnf_glow_step& st = desc->glow_steps[s];
      if (!st.actnorm_bias || !st.actnorm_logs || !st.perm_indices)
        return fail(GBNF_ERR_INVALID, "glow step %d: null actnorm/permutation pointer", s);
      std::vector<char> seen(d, 0);
      for (int j = 0; j < d; ++j) {
        const int64_t v = st.perm_indices[j];
        if (v < 0 || v >= d || seen[v]) return fail(GBNF_ERR_INVALID, "glow step %d: perm_indices is not a permutation", s);
        seen[v] = 1;
      }
      snprintf(what, sizeof(what), "glow step %d block", s);
      rc = check_net(st.block, d1, additive ? d2 : 2 * d2, &a, what);
      if (rc) return rc;
      b = a;
    } else {
      const gbnf_realnvp_step& st = desc->realnvp_steps[s];
      const int in_f = st.flipped ? d2 : d1, out_f = st.flipped ? d1 : d2;
      if (st.has_batch_norm && (!st.bn_log_gamma || !st.bn_beta || !st.bn_running_mean || !st.bn_running_var))
        return fail(GBNF_ERR_INVALID, "realnvp step %d: null batch-norm pointer", s);
      snprintf(what, sizeof(what), "realnvp step %d t_net", s);
      rc = check_net(st.t_net, in_f, out_f, &a, what);
      if (rc) return rc;
      snprintf(what, sizeof(what), "realnvp step %d s_net", s);
      rc = check_net(st.s_net, in_f, out_f, &b, what);
      if (rc) return rc;
      if (a.hidden != b.hidden || a.depth != b.depth || a.residual != b.residual)
        return fail(GBNF_ERR_UNSUPPORTED, "realnvp step %d: t_net and s_net differ in width/depth/architecture", s);
    }
    if (s == 0) {
      ref = a; act_a = a.act; act_b = b.act;
    } else if (a.hidden != ref.hidden || a.depth != ref.depth || a.residual != ref.residual) {
      return fail(GBNF_ERR_UNSUPPORTED, "step %d: coupling-network width/depth/architecture differs from step 0", s);
    } else if (a.act != act_a || b.act != act_b) {
      // the reference's `--coupling_network random` draws tanh / relu per step (glow.py:295-296) or per net
      // (realnvp.py:59-60): the kernels then pick the activation per step and net from the step header
      act_a = act_b = GBNF_ACT_PER_STEP;
    }
    if (a.in_f > 4 * KS1MAX)
      return fail(GBNF_ERR_UNSUPPORTED, "coupling-net input width %d > %d", a.in_f, 4 * KS1MAX);
    if (a.in_f > max_in) max_in = a.in_f;
    const int entries = (glow && !additive) ? ceil_div(a.out_f / 2, 8) * 2 : ceil_div(a.out_f, 16) * 4;
    if (entries > max_out_entries) max_out_entries = entries;
  }
  const int h = ref.hidden, depth = ref.depth;
  if (ref.residual ? depth > 4 : depth > 2)
    return fail(GBNF_ERR_UNSUPPORTED, "coupling_network_depth=%d > 2 has no compiled variant", ref.residual ? depth / 2 : depth);
  const int ksh = ceil_div(h, 4);
  const int ht = ceil_div(ksh, 4), ksl = ksh - 4 * (ht - 1);
  // out tiles: glow affine -> pairs (shift_j, raw_j): 8 pairs per 16-row tile; else 16 rows per tile
  const int max_out = glow ? (additive ? d2 : 2 * d2) : d2;  // realnvp: max(d1,d2) = d2
  const int ot = ceil_div(max_out, 16);
  if (max_out_entries > NENT || ot > 4)
    return fail(GBNF_ERR_UNSUPPORTED, "coupled half of %d features exceeds the per-lane table (%d)", max_out, NENT);

  info->ref = ref; info->act_a = act_a; info->act_b = act_b;
  info->ht = ht; info->ksl = ksl; info->ot = ot; info->ks1 = ceil_div(max_in, 4);
  return GBNF_OK;
}

int gbnf_flow_validate(const gbnf_flow_desc* desc) {
  DescInfo info;
  return validate_desc(desc, &info);
}

int gbnf_flow_create_mode(const gbnf_flow_desc* desc, int32_t math_mode, gbnf_flow** out) {
  return gbnf_flow_create_ex(desc, math_mode, 0, out);
}

// ---- host-side blob of one component for one kernel variant ---------------------------------------------------
struct VariantChoice {
  bool hx3 = false;
  int np = 0;                         // split kernels: pieces per operand (2 = f16x3, 3 = bf16x6)
  int ht = 0, ksl = 0, ks1 = 0, ot = 0;
  gbnf::LaunchFn launch_nt[3] = {nullptr, nullptr, nullptr};
  const char* name_nt[3] = {nullptr, nullptr, nullptr};
};

struct PackedBlob {
  std::vector<uint32_t> words;
  double macs = 0, padded = 0;
  bool in_range = true;               // f16x3: every (pre-scaled) weight fits the fp16 range
};

// the cheapest compiled split-kernel variant (key.ksl = -3 f16x3 / -6 bf16x6) that covers hidden width h and `ot` output tiles
static bool choose_hx3(int kind, int h, int ot, int depth, int act_a, int act_b, int ksl_key, VariantChoice* vc, int train = 0) {
  const int ht_b = ceil_div(h, 16);
  long best = -1;
  for (const Variant& v : variants()) {
    const VariantKey& k = v.key;
    if (k.ksl != ksl_key || k.ks1 != train || k.kind != kind || k.lmid != depth || k.act_a != act_a || k.act_b != act_b) continue;
    if (k.ht < ht_b || k.ot < ot) continue;
    const Variant* v1 = find_variant(VariantKey{k.kind, k.ht, ksl_key, train, k.ot, 1, depth, k.act_a, k.act_b});
    const Variant* v2 = find_variant(VariantKey{k.kind, k.ht, ksl_key, train, k.ot, 2, depth, k.act_a, k.act_b});
    if (!v1 || (!v2 && !train)) continue;      // (the TRAIN variants exist for 16-sample waves only)
    const long cost = (long)k.ht * (k.ht + 1) / 2 * 2 + k.ht * k.ot;
    if (best < 0 || cost < best) {
      best = cost;
      vc->hx3 = true; vc->np = ksl_key == -3 ? 2 : 3;
      vc->ht = k.ht; vc->ksl = ksl_key; vc->ks1 = 0; vc->ot = k.ot;
      vc->launch_nt[1] = v1->fn; vc->name_nt[1] = v1->name;
      vc->launch_nt[2] = v2 ? v2->fn : nullptr; vc->name_nt[2] = v2 ? v2->name : nullptr;
    }
  }
  return best >= 0;
}

static bool choose_f32(int kind, int ht, int ksl, int ks1, int ot, int depth, int lmid_key, int act_a, int act_b,
                       VariantChoice* vc) {
  long best_cost = -1;
  for (const Variant& v : variants()) {
    const VariantKey& k = v.key;
    if (k.ksl < 0) continue;
    if (k.kind != kind || k.lmid != lmid_key || k.act_a != act_a || k.act_b != act_b) continue;
    if (k.ot < ot || k.ks1 < ks1) continue;
    // a variant processes hidden k-steps [0, 4(k.ht-1)+k.ksl); ours are [0, 4(ht-1)+ksl); extra ones
    // multiply zero padding, so any superset is exact (just slower)
    const bool covers_h = (k.ht > ht) || (k.ht == ht && k.ksl >= ksl);
    if (!covers_h) continue;
    // NT=1 must exist; a geometry built for 16-sample waves only (`nt1` in variants.list) has no NT=2 entry
    const Variant* v1 = find_variant(VariantKey{k.kind, k.ht, k.ksl, k.ks1, k.ot, 1, k.lmid, k.act_a, k.act_b});
    const Variant* v2 = find_variant(VariantKey{k.kind, k.ht, k.ksl, k.ks1, k.ot, 2, k.lmid, k.act_a, k.act_b});
    if (!v1) continue;
    const long cost = (long)(4 * (k.ht - 1) + k.ksl) * (k.ht * 16L * depth + k.ot * 16L) + k.ht * 16L * 4 * k.ks1;
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      vc->hx3 = false; vc->np = 0;
      vc->ht = k.ht; vc->ksl = k.ksl; vc->ks1 = k.ks1; vc->ot = k.ot;
      vc->launch_nt[1] = v1->fn; vc->name_nt[1] = v1->name;
      vc->launch_nt[2] = v2 ? v2->fn : nullptr; vc->name_nt[2] = v2 ? v2->name : nullptr;
    }
  }
  return best_cost >= 0;
}

// Pads, tiles and folds the slot maps of `desc` for the kernel variant `vc` (host memory only).
static void pack_component(const gbnf_flow_desc* desc, const DescInfo& info, const VariantChoice& vc, PackedBlob* pb) {
  const int d = desc->d, K = desc->n_steps;
  const bool glow = desc->kind == GBNF_KIND_GLOW;
  const bool additive = glow && desc->coupling == GBNF_COUPLING_ADDITIVE;
  const int d1 = d / 2, d2 = d - d1;
  const int h = info.ref.hidden, depth = info.ref.depth;
  const bool hx3 = vc.hx3;
  const int HT = vc.ht, OT = vc.ot, KS1V = vc.ks1;
  const int nnets = glow ? 1 : 2;
  const size_t NW = hx3 ? (size_t)Hx3Layout(HT, OT, vc.np, depth).NET_WORDS : net_words(HT, KS1V, OT, depth);
  const size_t step_words = SMALL_WORDS + nnets * NW;
  const size_t total_words = step_words * K + 64;
  std::vector<uint32_t>& blob = pb->words;
  blob.assign(total_words, 0u);

  // slot map: sigma[j] = LDS slot of logical feature j at the current step
  std::vector<int> sigma(d), prev(d);
  for (int j = 0; j < d; ++j) sigma[j] = j;

  auto put_f = [&](size_t off, float v) { std::memcpy(&blob[off], &v, 4); };
  auto put_i = [&](size_t off, int v) { std::memcpy(&blob[off], &v, 4); };

  double macs = 0, padded = 0;
  for (int s = 0; s < K; ++s) {
    const size_t sb = step_words * s;
    prev = sigma;
    int in_f, out_f;
    // per logical (post-permutation) feature j: slot + norm params p0..p3
    std::vector<float> P0(d, 0.f), P1(d, 1.f), P2(d, 1.f), P3(d, 0.f);
    std::vector<int> in_feat, out_feat;  // logical (new order) feature ids feeding the net / being coupled
    float ld_const = 0.f;
    if (glow) {
      const gbnf_glow_step& st = desc->glow_steps[s];
      // z'[j] = actnorm(z)[perm[j]]
      float sum_logs = 0.f;
      for (int m = 0; m < d; ++m) sum_logs += st.actnorm_logs[m];   // torch.sum(logs), sequential f32
      ld_const = sum_logs;
      for (int j = 0; j < d; ++j) {
        const int m = (int)st.perm_indices[j];
        sigma[j] = prev[m];
        P0[j] = st.actnorm_bias[m];
        P1[j] = expf(st.actnorm_logs[m]);
        P2[j] = expf(-st.actnorm_logs[m]);      // for the inverse flow (ActNorm reverse multiplies by exp(-logs))
      }
      in_f = d1; out_f = d2;
      for (int j = 0; j < d1; ++j) in_feat.push_back(j);
      for (int j = 0; j < d2; ++j) out_feat.push_back(d1 + j);
    } else {
      const gbnf_realnvp_step& st = desc->realnvp_steps[s];
      // BN acts on the OLD logical order; new order = cat(z1, z2) with z1 = upper half when flipped
      std::vector<float> q0(d, 0.f), q1(d, 1.f), q2(d, 1.f), q3(d, 0.f);
      if (st.has_batch_norm) {
        float acc = 0.f;
        for (int m = 0; m < d; ++m) {
          const float ve = st.bn_running_var[m] + st.bn_eps;
          q0[m] = st.bn_running_mean[m];
          q1[m] = sqrtf(ve);
          q2[m] = expf(st.bn_log_gamma[m]);
          q3[m] = st.bn_beta[m];
          acc += st.bn_log_gamma[m] - 0.5f * logf(ve);      // models/layers.py:357-358
        }
        ld_const = acc;
      }
      in_f = st.flipped ? d2 : d1;
      out_f = st.flipped ? d1 : d2;
      for (int j = 0; j < d; ++j) {
        int m;  // old logical index of new logical feature j
        if (st.flipped) m = (j < d2) ? d1 + j : j - d2;
        else m = j;
        sigma[j] = prev[m];
        P0[j] = q0[m]; P1[j] = q1[m]; P2[j] = q2[m]; P3[j] = q3[m];
      }
      for (int j = 0; j < in_f; ++j) in_feat.push_back(j);
      for (int j = 0; j < out_f; ++j) out_feat.push_back(in_f + j);
    }
    put_i(sb + 0, ceil_div(in_f, 4));
    put_f(sb + 1, ld_const);
    {   // activation of the step's net(s): 1 = relu (read by the per-step-activation kernel variants only)
      const gbnf_net& na = glow ? desc->glow_steps[s].block : desc->realnvp_steps[s].t_net;
      const gbnf_net& nb = glow ? desc->glow_steps[s].block : desc->realnvp_steps[s].s_net;
      put_i(sb + 2, na.activation == GBNF_ACT_RELU ? 1 : 0);
      put_i(sb + 3, nb.activation == GBNF_ACT_RELU ? 1 : 0);
    }
    // in tables [g][e]: f32 kernel k = 4e + g (k-step e, lane group g); split kernels k = 8g + e
    for (int gg = 0; gg < 4; ++gg)
      for (int e = 0; e < NENT; ++e) {
        const int k = hx3 ? 8 * gg + e : 4 * e + gg;
        const size_t o = sb + SMALL_HDR + gg * NENT + e;
        if (k < in_f) {
          const int j = in_feat[k];
          put_i(o, sigma[j]); put_f(o + 32, P0[j]); put_f(o + 64, P1[j]); put_f(o + 96, P2[j]); put_f(o + 128, P3[j]);
        } else {
          put_i(o, -1); put_f(o + 32, 0.f); put_f(o + 64, 1.f); put_f(o + 96, 1.f); put_f(o + 128, 0.f);
        }
      }
    // out tables [g][e]: affine pairs j = 8o + 2g + pp (e = 2o + pp); plain j = 16o + 4g + r (e = 4o + r)
    const bool paired = glow && !additive;
    for (int gg = 0; gg < 4; ++gg)
      for (int e = 0; e < NENT; ++e) {
        int jj;
        if (paired) jj = 8 * (e >> 1) + 2 * gg + (e & 1);
        else jj = 16 * (e >> 2) + 4 * gg + (e & 3);
        const size_t o = sb + SMALL_HDR + 160 + gg * NENT + e;
        if (jj < out_f) {
          const int j = out_feat[jj];
          put_i(o, sigma[j]); put_f(o + 32, P0[j]); put_f(o + 64, P1[j]); put_f(o + 96, P2[j]); put_f(o + 128, P3[j]);
        } else {
          put_i(o, -1); put_f(o + 32, 0.f); put_f(o + 64, 1.f); put_f(o + 96, 1.f); put_f(o + 128, 0.f);
        }
      }
    const int net_out = paired ? 2 * out_f : out_f;
    if (hx3) {
      if (glow) {
        pb->in_range &= pack_net_hx3(blob, sb + SMALL_WORDS, desc->glow_steps[s].block, HT, OT, vc.np, depth, in_f, h, net_out);
      } else {
        pb->in_range &= pack_net_hx3(blob, sb + SMALL_WORDS, desc->realnvp_steps[s].t_net, HT, OT, vc.np, depth, in_f, h, net_out);
        pb->in_range &= pack_net_hx3(blob, sb + SMALL_WORDS + NW, desc->realnvp_steps[s].s_net, HT, OT, vc.np, depth, in_f, h, net_out);
      }
    } else if (glow) {
      pack_net(blob, sb + SMALL_WORDS, desc->glow_steps[s].block, HT, KS1V, OT, depth, in_f, h, net_out);
    } else {
      pack_net(blob, sb + SMALL_WORDS, desc->realnvp_steps[s].t_net, HT, KS1V, OT, depth, in_f, h, net_out);
      pack_net(blob, sb + SMALL_WORDS + NW, desc->realnvp_steps[s].s_net, HT, KS1V, OT, depth, in_f, h, net_out);
    }
    macs += (double)nnets * ((double)in_f * h + (double)depth * h * h + (double)h * net_out);
    if (hx3) {   // executed narrow MACs / products per f32 product, k padded to 32
      const double hc = (HT + 1) / 2;
      padded += (double)nnets * (16.0 * HT * 32 + 16.0 * HT * 32 * hc + 16.0 * OT * 32 * hc);
    } else {
      const double kh = 4.0 * (HT - 1) + vc.ksl;  // live hidden k-steps in the variant
      padded += (double)nnets * (16.0 * HT * 4 * KS1V + depth * 16.0 * HT * 4 * kh + 16.0 * OT * 4 * kh);
    }
  }
  for (int j = 0; j < d; ++j) put_i(step_words * K + j, sigma[j]);
  pb->macs = macs; pb->padded = padded;
}

static hipError_t upload_blob(const std::vector<uint32_t>& words, uint32_t** blob_dev, const uint32_t*** table_dev) {
  hipError_t e = hipMalloc((void**)blob_dev, words.size() * 4);
  if (e == hipSuccess) e = hipMemcpy(*blob_dev, words.data(), words.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)table_dev, sizeof(uint32_t*));
  if (e == hipSuccess) e = hipMemcpy(*table_dev, blob_dev, sizeof(uint32_t*), hipMemcpyHostToDevice);
  return e;
}

static void free_flow(gbnf_flow* f) {
  if (!f) return;
  if (f->blob_dev) (void)hipFree(f->blob_dev);
  if (f->self_table_dev) (void)hipFree(f->self_table_dev);
  if (f->blob2_dev) (void)hipFree(f->blob2_dev);
  if (f->self_table2_dev) (void)hipFree(f->self_table2_dev);
  gbnf::free_guard(f->guard);
  delete f;
}

}  // extern "C"  (the probe needs launch_flow, defined below in namespace gbnf)

namespace gbnf {
static int probe_split_modes(gbnf_flow* f, float* rel_err);
}

extern "C" {

int gbnf_flow_create_ex(const gbnf_flow_desc* desc, int32_t math_mode, int32_t flags, gbnf_flow** out) {
  if (out == nullptr) return fail(GBNF_ERR_INVALID, "gbnf_flow_create: out is null");
  *out = nullptr;
  if (flags & ~GBNF_CREATE_PER_STEP_ACTIVATION) return fail(GBNF_ERR_INVALID, "gbnf_flow_create_ex: unknown flags 0x%x", flags);
  DescInfo info;
  {
    const int rc = validate_desc(desc, &info);
    if (rc) return rc;
  }
  if (flags & GBNF_CREATE_PER_STEP_ACTIVATION) info.act_a = info.act_b = GBNF_ACT_PER_STEP;
  {   // an activation pair nobody compiled a kernel for (`random` can draw a tanh shift net with a relu scale net for
      // every step): the per-step variants cover it
    bool compiled = false;
    for (const Variant& v : variants())
      compiled = compiled || (v.key.kind == desc->kind && v.key.act_a == info.act_a && v.key.act_b == info.act_b);
    if (!compiled) info.act_a = info.act_b = GBNF_ACT_PER_STEP;
  }
  const int d = desc->d, K = desc->n_steps;
  const bool glow = desc->kind == GBNF_KIND_GLOW;
  const bool additive = glow && desc->coupling == GBNF_COUPLING_ADDITIVE;
  const NetDims ref = info.ref;
  const int act_a = info.act_a, act_b = info.act_b;
  const int h = ref.hidden, depth = ref.depth;
  const int ht = info.ht, ksl = info.ksl, ot = info.ot;

  if (math_mode != GBNF_MATH_F32 && math_mode != GBNF_MATH_F16X3 && math_mode != GBNF_MATH_BF16X6 && math_mode != GBNF_MATH_DEFAULT)
    return fail(GBNF_ERR_INVALID, "unknown math mode %d", math_mode);
  // what the split kernels take: TanhNet / ReLUNet of depth 0, 1 or 2, and ResidualNets of ONE block (the reference's default
  // coupling_network_depth = 1: two hidden -> hidden layers; round 3 -- kernel key act = GBNF_ACT_RESIDUAL_RELU, depth 2)
  const bool split_shape = ref.residual ? (depth == 2 || depth == 4) : depth <= 2;      // (round 5: two-block ResidualNets too)
  const int act_a_split = ref.residual ? GBNF_ACT_RESIDUAL_RELU : act_a, act_b_split = ref.residual ? GBNF_ACT_RESIDUAL_RELU : act_b;
  if (!split_shape && (math_mode == GBNF_MATH_F16X3 || math_mode == GBNF_MATH_BF16X6))
    return fail(GBNF_ERR_UNSUPPORTED, "the split kernels (f16x3 / bf16x6) support TanhNet / ReLUNet of coupling_network_depth <= 2 and ResidualNets of one or two blocks only (got %s%d)",
                ref.residual ? "a ResidualNet, hidden layers " : "", depth);

  // ---- pick compiled variants (exact geometry first, then the cheapest zero-padded superset)
  VariantChoice fast, safe, exact;
  const bool want_split = split_shape && math_mode != GBNF_MATH_F32;
  const bool have_fast = want_split && math_mode != GBNF_MATH_BF16X6 && choose_hx3(desc->kind, h, ot, depth, act_a_split, act_b_split, -3, &fast);
  const bool have_safe = want_split && choose_hx3(desc->kind, h, ot, depth, act_a_split, act_b_split, -6, &safe);
  const bool retry_per_step = act_a != GBNF_ACT_PER_STEP && !ref.residual;   // the per-step-activation variants are generic supersets
  if (math_mode == GBNF_MATH_F16X3 && !have_fast) {
    if (retry_per_step) return gbnf_flow_create_ex(desc, math_mode, flags | GBNF_CREATE_PER_STEP_ACTIVATION, out);
    return fail(GBNF_ERR_UNSUPPORTED, "no compiled f16x3 kernel variant for kind=%d hidden=%d out_tiles=%d depth=%d act=(%d,%d); "
                "add it to csrc/variants.list", desc->kind, h, ot, depth, act_a, act_b);
  }
  if (math_mode == GBNF_MATH_BF16X6 && !have_safe) {
    if (retry_per_step) return gbnf_flow_create_ex(desc, math_mode, flags | GBNF_CREATE_PER_STEP_ACTIVATION, out);
    return fail(GBNF_ERR_UNSUPPORTED, "no compiled bf16x6 kernel variant for kind=%d hidden=%d out_tiles=%d depth=%d act=(%d,%d); "
                "add it to csrc/variants.list", desc->kind, h, ot, depth, act_a, act_b);
  }
  const bool use_split = have_fast || (have_safe && math_mode != GBNF_MATH_F16X3);
  if (!use_split) {
    const int lmid_key = ref.residual ? 10 + depth / 2 : depth;      // the LMID field of the variant key (gbnf_flow_kernel.hip.h)
    if (!choose_f32(desc->kind, ht, ksl, info.ks1, ot, depth, lmid_key, act_a, act_b, &exact)) {
      if (retry_per_step) return gbnf_flow_create_ex(desc, math_mode, flags | GBNF_CREATE_PER_STEP_ACTIVATION, out);
      return fail(GBNF_ERR_UNSUPPORTED,
                  "no compiled kernel variant for kind=%d hidden=%d (tiles=%d,last k-steps=%d) in k-steps=%d out_tiles=%d "
                  "depth=%d act=(%d,%d); add it to csrc/variants.list", desc->kind, h, ht, ksl, info.ks1, ot, depth, act_a,
                  act_b);
    }
  }

  gbnf_flow* f = new gbnf_flow();
  f->requested_mode = math_mode;
  f->kind = desc->kind; f->d = d; f->n_steps = K; f->additive = additive ? 1 : 0;
  f->hidden = h; f->depth = depth; f->act_a = act_a; f->act_b = act_b; f->residual = ref.residual;
  f->ht = ht; f->ksl = ksl; f->ot = ot; f->ks1 = info.ks1;

  auto install_primary = [&](const VariantChoice& vc, const PackedBlob& pb, int mode) -> hipError_t {
    f->math_mode = mode;
    f->var_ht = vc.ht; f->var_ksl = vc.ksl; f->var_ks1 = vc.ks1; f->var_ot = vc.ot;
    for (int nt = 1; nt <= 2; ++nt) { f->launch_nt[nt] = vc.launch_nt[nt]; f->name_nt[nt] = vc.name_nt[nt]; }
    f->macs = pb.macs; f->padded_macs = pb.padded; f->blob_words = pb.words.size();
    if (mode == GBNF_MATH_F16X3 && vc.hx3 && depth == 1 && !ref.residual) {
      // the latency form reads the SAME blob: a variant of exactly the packed geometry and activation key
      for (int nt = 1; nt <= 4; ++nt) {
        const Variant* v = find_variant(VariantKey{desc->kind, vc.ht, -13, 0, vc.ot, nt, 1, act_a_split, act_b_split});
        f->launch_coop_nt[nt] = v ? v->fn : nullptr;
        f->name_coop_nt[nt] = v ? v->name : nullptr;
      }
    }
    return upload_blob(pb.words, &f->blob_dev, &f->self_table_dev);
  };
  auto install_secondary = [&](const VariantChoice& vc, const PackedBlob& pb) -> hipError_t {
    f->var2_ht = vc.ht; f->var2_ot = vc.ot;
    for (int nt = 1; nt <= 2; ++nt) { f->launch2_nt[nt] = vc.launch_nt[nt]; f->name2_nt[nt] = vc.name_nt[nt]; }
    f->padded_macs2 = pb.padded; f->blob2_words = pb.words.size();
    return upload_blob(pb.words, &f->blob2_dev, &f->self_table2_dev);
  };

  hipError_t e = hipSuccess;
  if (!use_split) {
    PackedBlob pb;
    pack_component(desc, info, exact, &pb);
    e = install_primary(exact, pb, GBNF_MATH_F32);
  } else {
    PackedBlob pf, ps;
    if (have_fast) pack_component(desc, info, fast, &pf);
    if (have_safe) pack_component(desc, info, safe, &ps);
    if (have_fast && !pf.in_range) {       // a (pre-scaled) weight beyond +-65504: the f16x3 packing cannot represent this model
      if (math_mode == GBNF_MATH_F16X3 || !have_safe) {
        free_flow(f);
        return fail(GBNF_ERR_UNSUPPORTED, "a coupling-network weight exceeds the fp16 range (|w| > 65504): evaluate this model "
                    "with GBNF_MATH_BF16X6 or GBNF_MATH_F32");
      }
      e = install_primary(safe, ps, GBNF_MATH_BF16X6);
    } else if (have_fast) {
      e = install_primary(fast, pf, GBNF_MATH_F16X3);
      if (e == hipSuccess && have_safe) e = install_secondary(safe, ps);
    } else {
      e = install_primary(safe, ps, GBNF_MATH_BF16X6);
    }
  }
  if (e != hipSuccess) {
    free_flow(f);
    return fail(GBNF_ERR_HIP, "uploading packed parameters failed: %s", hipGetErrorString(e));
  }
  // ---- DEFAULT: the probe decides between the fast and the f32-faithful split mode
  if (math_mode == GBNF_MATH_DEFAULT && f->math_mode == GBNF_MATH_F16X3 && f->blob2_dev != nullptr) {
    float err = 0.0f;
    const int rc = probe_split_modes(f, &err);
    if (rc) { free_flow(f); return rc; }
    f->probe_rel_err = err;
    if (!(err <= PROBE_MAX_REL_ERR)) {      // (a NaN difference fails the test too): run this component on its bf16x6 packing
      std::swap(f->blob_dev, f->blob2_dev);
      std::swap(f->self_table_dev, f->self_table2_dev);
      std::swap(f->blob_words, f->blob2_words);
      for (int nt = 1; nt <= 2; ++nt) { f->launch_nt[nt] = f->launch2_nt[nt]; f->name_nt[nt] = f->name2_nt[nt]; }
      f->padded_macs = f->padded_macs2;
      f->var_ht = f->var2_ht; f->var_ot = f->var2_ot; f->var_ksl = -6;
      f->math_mode = GBNF_MATH_BF16X6;
      for (int nt = 1; nt <= 4; ++nt) { f->launch_coop_nt[nt] = nullptr; f->name_coop_nt[nt] = nullptr; }
      // the f16x3 packing is of no further use
      (void)hipFree(f->blob2_dev); (void)hipFree(f->self_table2_dev);
      f->blob2_dev = nullptr; f->self_table2_dev = nullptr; f->blob2_words = 0;
      for (int nt = 1; nt <= 2; ++nt) { f->launch2_nt[nt] = nullptr; f->name2_nt[nt] = nullptr; }
    }
  }
  // ---- a DEFAULT handle that stays on f16x3 keeps being checked against its bf16x6 packing on the caller's data
  if (math_mode == GBNF_MATH_DEFAULT && f->math_mode == GBNF_MATH_F16X3 && f->blob2_dev != nullptr) {
    e = make_guard(1, &f->guard);
    if (e != hipSuccess) {
      free_flow(f);
      return fail(GBNF_ERR_HIP, "allocating the numerics guard failed: %s", hipGetErrorString(e));
    }
  }
  *out = f;
  return GBNF_OK;
}

int gbnf_flow_destroy(gbnf_flow* flow) {
  free_flow(flow);
  return GBNF_OK;
}

int gbnf_flow_info(const gbnf_flow* flow, gbnf_kernel_info* info) {
  if (!flow || !info) return fail(GBNF_ERR_INVALID, "gbnf_flow_info: null argument");
  info->hidden_tiles = flow->var_ht;
  info->out_tiles = flow->var_ot;
  info->samples_per_wave = 32;
  info->math_mode = flow->math_mode;
  info->n_steps = flow->n_steps;
  info->macs_per_sample = flow->macs;
  info->padded_macs_per_sample = flow->padded_macs;
  info->packed_bytes = (int64_t)(flow->blob_words + flow->blob2_words) * 4;
  info->probe_rel_err = flow->probe_rel_err;
  return GBNF_OK;
}

}  // extern "C"

namespace gbnf {

// samples per wave: 32 (NT=2) once there is enough work to give every SIMD of the chip a wave that way (256 CUs x 4
// SIMDs: the split kernels then run 4-wave workgroups, one or two per CU), otherwise 16 (NT=1) to expose more waves.
static int pick_nt(int64_t n, int n_comp) {
  const int forced = tuning().force_nt.load(std::memory_order_relaxed);     // tuning / test knob: 1 or 2
  if (forced == 1 || forced == 2) return forced;
  const int64_t waves32 = ((n + 31) / 32) * n_comp;
  return waves32 >= tuning().nt2_min_waves.load(std::memory_order_relaxed) ? 2 : 1;
}

// The latency form (flow_kernel_coop: the waves of a workgroup share ONE sample tile) pays off while every workgroup gets a CU to
// itself: one component's weights then stream through that CU's L2 port once per tile, whatever the tile's size -- that stream
// (~35 B/clk per CU, 1.5 MB per MINIBOONE component) is what a tile costs.  0 = the throughput kernel; forms 1 / 2 / 3 = 16-sample
// tiles on 4 waves / 32-sample tiles on 4 waves / 32-sample tiles on 8 waves: the smallest tile that still fills at most
// coop_max_wgs workgroups (more CUs at work); of the 32-sample forms the eight-wave one where it is built (csrc/variants.list builds
// it for the geometries it wins on: Glow from 14 hidden tiles on -- the same bytes per workgroup, twice the waves under them).
static int pick_coop(const gbnf_flow* f, int64_t n, int n_comp, int n_batches) {
  const int mode = tuning().coop.load(std::memory_order_relaxed);
  if (mode == 0) return 0;
  if (mode >= 1 && mode <= 4) return f->launch_coop_nt[mode] ? mode : 0;
  const int64_t max_wgs = tuning().coop_max_wgs.load(std::memory_order_relaxed);
  const int64_t wg16 = ((n + 15) / 16) * n_comp * n_batches, wg32 = ((n + 31) / 32) * n_comp * n_batches;
  if (wg16 <= max_wgs && f->launch_coop_nt[1]) return 1;
  if (wg32 <= max_wgs) return f->launch_coop_nt[3] ? 3 : (f->launch_coop_nt[2] ? 2 : 0);
  return 0;
}

#if defined(GBNF_STAMPS) || defined(GBNF_TIMELINE) || defined(GBNF_CLOCK)
static unsigned long long* g_stamp_buf = nullptr;
#endif

static bool repair_enabled() { return tuning().repair.load(std::memory_order_relaxed) != 0; }   // 0: diagnostic, times the bare f16x3 launch

static unsigned long long next_serial() {      // launch serial numbers: process-wide, never 0 (the memset value of the marks), never reused
  static std::atomic<unsigned long long> launch_serial{1};
  return launch_serial.fetch_add(1, std::memory_order_relaxed);
}

// The device side of the numerics guard: largest relative difference between the f16x3 (a) and bf16x6 (b) log-likelihoods of the
// check rows; rows the f16x3 launch marked as out of range (NaN) belong to the repair protocol and are skipped.
__global__ void __launch_bounds__(256) guard_compare_kernel(const float* __restrict__ a, const float* __restrict__ b, int n_comp,
                                                            int rows, float tol, unsigned* __restrict__ flag,
                                                            GuardStatus* __restrict__ st) {
  __shared__ float red[4];
  float worst = 0.0f;
  for (int idx = threadIdx.x; idx < n_comp * rows; idx += blockDim.x) {
    const int c = idx / rows, r = idx - c * rows;
    const float va = a[c * GUARD_ROWS + r], vb = b[c * GUARD_ROWS + r];
    if (va == vb || va != va) continue;                       // equal (infinities included) | marked by the f16x3 launch
    float rel;
    if (vb != vb || isinf(va) || isinf(vb)) rel = INFINITY;   // only one of them finite
    else rel = fabsf(va - vb) / fmaxf(fabsf(vb), 1.0f);
    worst = fmaxf(worst, rel);
  }
  for (int off = 32; off > 0; off >>= 1) worst = fmaxf(worst, __shfl_xor(worst, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = worst;
  __syncthreads();
  if (threadIdx.x == 0) {
    worst = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (!(worst <= tol)) {
      flag[0] = 1u;
      st->demoted = 1u;
    }
    st->worst_rel_err = fmaxf(st->worst_rel_err, worst);
    st->checks = st->checks + 1ull;
    __threadfence_system();
  }
}

// One launch of the flow kernel of `f` (all components of `table` share its variant).  `table2` (or null): the bf16x6
// packings of the same components -- an f16x3 launch is followed by the repair launch over the same grid.
static int launch_flow(const gbnf_flow* f, const uint32_t* const* table, const uint32_t* const* table2, bool use_second,
                       int c_begin, int n_comp, const float* x,
                       int64_t n, float* z, float* ldj, float* ll, const float* base, hipStream_t stream,
                       int64_t out_stride = -1, const float* const* xs = nullptr, int n_batches = 1,
                       int inverse = 0, Guard* guard = nullptr) {
  if (n == 0 || n_comp == 0 || n_batches == 0) return GBNF_OK;
  if (!use_second && guard != nullptr && f->math_mode == GBNF_MATH_F16X3 && table2 != nullptr &&
      *(volatile unsigned*)&guard->status_host->demoted != 0u) {
    // a check on the caller's data failed earlier: this handle runs on its bf16x6 packing from now on
    use_second = true;
    table = table2;
    table2 = nullptr;
  }
  const int mode = use_second ? GBNF_MATH_BF16X6 : f->math_mode;
  const LaunchFn* launch = use_second ? f->launch2_nt : f->launch_nt;
  const char* const* names = use_second ? f->name2_nt : f->name_nt;
  int nt = pick_nt(n * n_batches, n_comp);
  if (launch[nt] == nullptr) nt = 1;                // geometry compiled for 16-sample waves only
  // a call of a few sample tiles: the latency form (forward, f16x3 only; its repair launch is the bf16x6 kernel's as ever)
  const int coop_nt = (mode == GBNF_MATH_F16X3 && !use_second && !inverse) ? pick_coop(f, n, n_comp, n_batches) : 0;
  // (Launch geometry, measured in round 5 and NOT shipped: a group whose last round of 32-sample waves is a quarter full -- one
  //  rank of eight at the driver's --steps 20: 20 batches x 4096 rows x 1 component = 2560 waves = 1.25 rounds of 2048 -- split
  //  by whole batches into the full rounds + the rest as a second launch of 16-sample waves over every CU: 118.9 + 48.8 us
  //  against 171.2 us for the one launch, and the second launch's own repair launch takes the difference back:
  //  profiles/r5_emulated_rank_steps20_*_kernel_stats.csv.  The hardware's own tail -- 128 workgroups, one per CU, each a lone
  //  wave per SIMD -- already runs at 0.44 of a round.)
  const int64_t tiles = (n + 16 * nt - 1) / (16 * nt);
  // f32 kernel: one wave (= block) per tile; the split kernels size their own grid
  const int64_t grid = tiles * n_comp * n_batches;
  if (grid > 0x7fffffffLL) return fail(GBNF_ERR_UNSUPPORTED, "batch too large for one launch (%lld tiles)", (long long)grid);
  FlowLaunch p{};
  p.blobs = table; p.z_out = z; p.ldj_out = ldj; p.ll_out = ll;
  p.n_batches = n_batches;
  p.inverse = inverse;
  for (int b = 0; b < n_batches; ++b) p.xs[b] = xs ? xs[b] : x;
  p.base_mean = base; p.base_std = base ? base + f->d : nullptr;
  p.n = n; p.out_stride = out_stride < 0 ? n * n_batches : out_stride; p.d = f->d; p.n_steps = f->n_steps; p.c_begin = c_begin; p.n_comp = n_comp;
  p.n_tiles = (int32_t)tiles; p.additive = f->additive;
#if defined(GBNF_STAMPS) || defined(GBNF_TIMELINE) || defined(GBNF_CLOCK)
  p.dbg = g_stamp_buf;
#endif
  p.sat = reinterpret_cast<unsigned long long*>(saturation_counter());
  p.seq = next_serial();
  hipError_t e = coop_nt ? f->launch_coop_nt[coop_nt](p, (unsigned)grid, stream) : launch[nt](p, (unsigned)grid, stream);
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "launch of %s failed: %s", coop_nt ? f->name_coop_nt[coop_nt] : names[nt], hipGetErrorString(e));
  const bool split_pair = mode == GBNF_MATH_F16X3 && table2 != nullptr && f->launch2_nt[nt] != nullptr;
  if (split_pair && repair_enabled()) {
    FlowLaunch r = p;
    r.blobs = table2;
    r.repair = 1;                      // (same seq: the launch it repairs; the mark is read from r.sat)
    r.guard = guard ? guard->flag_dev : nullptr;
    e = f->launch2_nt[nt](r, (unsigned)grid, stream);
    if (e != hipSuccess) return fail(GBNF_ERR_HIP, "repair launch of %s failed: %s", f->name2_nt[nt], hipGetErrorString(e));
  }
  // ---- numerics guard: this handle's first launch and every check_every-th re-check the f16x3 choice on the caller's rows
  if (split_pair && guard != nullptr && !inverse && f->launch_nt[1] != nullptr && f->launch2_nt[1] != nullptr && n_comp <= guard->n_comp) {
    // (a launch that is being captured into a HIP graph is not checked: the check's launches would be baked into every replay,
    //  and the host-side `demoted` switch is not consulted on replay -- the capture keeps the probe's verdict and the
    //  flag-driven full re-evaluation that any earlier, un-captured check has armed)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream != nullptr && hipStreamIsCapturing(stream, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
    const long long k = cap == hipStreamCaptureStatusNone ? guard->launches.fetch_add(1, std::memory_order_relaxed) : -1;
    const int every = tuning().check_every.load(std::memory_order_relaxed);
    bool check_now = k >= 0 && every >= 0 && (k == 0 || (every > 0 && k % every == 0));
    if (check_now) {
      // one check in flight per guard: its rows sit in scratch_dev until the compare launch has read them
      std::lock_guard<std::mutex> lk(guard->check_mutex);
      if (guard->check_pending && guard->check_stream != stream && hipEventQuery(guard->check_done) == hipErrorNotReady) {
        check_now = false;
        guard->launches.store(0, std::memory_order_relaxed);      // (the next launch takes the check)
      } else {
        guard->check_pending = true;
        guard->check_stream = stream;
      }
    }
    if (check_now) {
      const int rows = (int)(n < GUARD_ROWS ? n : GUARD_ROWS);
      FlowLaunch q = p;
      q.n = rows; q.n_batches = 1; q.out_stride = GUARD_ROWS; q.n_tiles = (rows + 15) / 16;
      q.z_out = nullptr; q.ldj_out = nullptr; q.repair = 0; q.guard = nullptr;
      q.sat = nullptr;                 // the check rows were counted (and marked) by the launch itself: not twice
      const unsigned mini_grid = (unsigned)(q.n_tiles * n_comp);
      q.ll_out = guard->scratch_dev;
      q.seq = next_serial();
      e = f->launch_nt[1](q, mini_grid, stream);
      if (e == hipSuccess) {
        q.blobs = table2;
        q.ll_out = guard->scratch_dev + (size_t)guard->n_comp * GUARD_ROWS;
        q.seq = next_serial();
        e = f->launch2_nt[1](q, mini_grid, stream);
      }
      if (e != hipSuccess) return fail(GBNF_ERR_HIP, "numerics-guard launch failed: %s", hipGetErrorString(e));
      hipLaunchKernelGGL(guard_compare_kernel, dim3(1), dim3(256), 0, stream, (const float*)guard->scratch_dev,
                         (const float*)(guard->scratch_dev + (size_t)guard->n_comp * GUARD_ROWS), n_comp, rows,
                         1e-9f * (float)tuning().check_tolerance_e9.load(std::memory_order_relaxed), guard->flag_dev,
                         guard->status_dev);
      e = hipGetLastError();
      if (e != hipSuccess) return fail(GBNF_ERR_HIP, "numerics-guard compare launch failed: %s", hipGetErrorString(e));
      (void)hipEventRecord(guard->check_done, stream);
      // the launch that was just checked: re-evaluated in full on bf16x6 if (and only if) the check failed
      FlowLaunch r = p;
      r.blobs = table2;
      r.repair = 2;
      r.guard = guard->flag_dev;
      e = f->launch2_nt[nt](r, (unsigned)grid, stream);
      if (e != hipSuccess) return fail(GBNF_ERR_HIP, "numerics-guard re-evaluation launch failed: %s", hipGetErrorString(e));
    }
  }
  return GBNF_OK;
}

// DEFAULT math mode: the same probe batch (rows ~ N(0,1), and N(0, 2^2) for the second half: what z-scored data looks
// like, tails included) through the f16x3 and the bf16x6 packing of a new component; the largest relative
// log-likelihood difference decides which one the handle runs on.  Synchronises (create does anyway).
constexpr int PROBE_ROWS = 128;
static int probe_split_modes(gbnf_flow* f, float* rel_err) {
  const int d = f->d;
  std::vector<float> x((size_t)PROBE_ROWS * d);
  uint64_t st = 0x9E3779B97F4A7C15ull;
  auto next_u = [&]() {                        // splitmix64 -> (0,1)
    st += 0x9E3779B97F4A7C15ull;
    uint64_t zz = st;
    zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ull;
    zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBull;
    zz ^= zz >> 31;
    return ((double)(zz >> 11) + 0.5) / 9007199254740992.0;
  };
  for (int r = 0; r < PROBE_ROWS; ++r)
    for (int j = 0; j < d; ++j) {
      const double u1 = next_u(), u2 = next_u();
      const double g0 = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
      x[(size_t)r * d + j] = (float)(g0 * (r < PROBE_ROWS / 2 ? 1.0 : 2.0));
    }
  float *x_dev = nullptr, *ll_dev = nullptr;
  GBNF_HIP(hipMalloc((void**)&x_dev, x.size() * 4));
  hipError_t e = hipMalloc((void**)&ll_dev, 2 * PROBE_ROWS * 4);
  if (e == hipSuccess) e = hipMemcpy(x_dev, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  int rc = GBNF_OK;
  if (e == hipSuccess) {
    rc = launch_flow(f, f->self_table_dev, nullptr, false, 0, 1, x_dev, PROBE_ROWS, nullptr, nullptr, ll_dev, nullptr, nullptr);
    if (!rc) rc = launch_flow(f, f->self_table2_dev, nullptr, true, 0, 1, x_dev, PROBE_ROWS, nullptr, nullptr, ll_dev + PROBE_ROWS, nullptr, nullptr);
  }
  std::vector<float> ll(2 * PROBE_ROWS, 0.0f);
  if (e == hipSuccess && !rc) e = hipMemcpy(ll.data(), ll_dev, ll.size() * 4, hipMemcpyDeviceToHost);   // (synchronises)
  (void)hipFree(x_dev);
  (void)hipFree(ll_dev);
  if (rc) return rc;
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "probing the split modes failed: %s", hipGetErrorString(e));
  float worst = 0.0f;
  for (int r = 0; r < PROBE_ROWS; ++r) {
    const float a = ll[r], b = ll[PROBE_ROWS + r];
    if (a == b) continue;                                   // (equal infinities included)
    if (!(std::isfinite(a) && std::isfinite(b))) {
      if (a != a && b != b) continue;                       // both NaN: the model explodes on this row either way
      worst = INFINITY;
      break;
    }
    const float rel = std::fabs(a - b) / std::fmax(std::fabs(b), 1.0f);
    if (rel > worst) worst = rel;
  }
  *rel_err = worst;
  return GBNF_OK;
}

// G_0 = ll_0; G_c = LSE2(log(1 - r_c) + G_{c-1}, log r_c + ll_c), r_c = rho_c / sum(rho[0..c])
// (density_experiment.py:567-571).  One thread per sample; rho prefix sums recomputed per
// thread (C is tiny, the loads are wave-uniform).  torch.logsumexp semantics for the 2-way LSE.
__global__ void __launch_bounds__(256) mixture_lse_kernel(const float* __restrict__ ll, int64_t stride,
                                                          const float* __restrict__ rho, int n_comp, int64_t n,
                                                          float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  float G = ll[idx];
  float rsum = rho[0];
  for (int c = 1; c < n_comp; ++c) {
    rsum += rho[c];
    const float r = rho[c] / rsum;
    const float a = logf(1.0f - r) + G;
    const float b = logf(r) + ll[(int64_t)c * stride + idx];
    float m = fmaxf(a, b);
    if (isinf(m)) m = 0.0f;
    G = m + logf(expf(a - m) + expf(b - m));
  }
  out[idx] = G;
}

}  // namespace gbnf

// ---- boosting sample weights (density_experiment.py:624-640): how much each training sample matters to the NEXT
// component, from the mixture log-density G of the fixed components:
//   w = softmax(-G);  w = w^beta;  if max(w) > 0.1: w = clamp(w, 0.01, 0.1);  if sum(w) != 1: w /= sum(w)
// One workgroup walks the batch four times (max, sum of exp, max/sum of w, final scale): n*4 bytes read per pass,
// launch-latency sized; fixed-order tree reductions => bit-reproducible.
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(v, off);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k) r = is_max ? fmaxf(r, red[k]) : r + red[k];
  return r;
}

__global__ void __launch_bounds__(1024) boosting_weights_kernel(const float* __restrict__ G, int64_t n, float beta,
                                                                float* __restrict__ w) {
  __shared__ float red[16];
  float m = -INFINITY;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, -G[i]);
  m = block_reduce(m, true, red);
  float s = 0.0f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += expf(-G[i] - m);
  s = block_reduce(s, false, red);
  float wmax = 0.0f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    float v = expf(-G[i] - m) / s;
    if (beta != 1.0f) v = powf(v, beta);
    w[i] = v;
    wmax = fmaxf(wmax, v);
  }
  wmax = block_reduce(wmax, true, red);
  const bool clamp = wmax > 0.1f;
  float tot = 0.0f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    float v = w[i];
    if (clamp) v = fmaxf(fminf(v, 0.1f), 0.01f);
    w[i] = v;
    tot += v;
  }
  tot = block_reduce(tot, false, red);
  if (tot != 1.0f)
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) w[i] = w[i] / tot;
}

// ---- ActNorm data-dependent initialisation (models/layers.py:473-486): per-feature statistics of a batch.
//   bias = -mean_0(z);  var = mean_0((z + bias)^2);  logs = log(scale / (sqrt(var) + 1e-6))
// HBM-bound column reduction, two passes like the reference (centred second moment).  A wave reads one row
// segment (d <= 64 consecutive floats) per load; the 4 waves of a block take rows r, r+1, r+2, r+3; block
// partials go to a workspace and are combined in a fixed order (bit-reproducible, no atomics).
constexpr int AN_MAX_BLOCKS = 256;

__global__ void __launch_bounds__(256) actnorm_partial_kernel(const float* __restrict__ z, int64_t n, int d,
                                                              const float* __restrict__ prev_partial, int pass,
                                                              float* __restrict__ partial) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, w = threadIdx.x >> 6;
  float shift = 0.0f;
  if (pass == 2 && c < d) {            // bias = -mean from the first pass's partials (fixed order)
    float s = 0.0f;
    for (int b = 0; b < (int)gridDim.x; ++b) s += prev_partial[b * 64 + c];
    shift = -(s / (float)n);
  }
  float acc = 0.0f;
  if (c < d) {
    for (int64_t r = (int64_t)blockIdx.x * 4 + w; r < n; r += (int64_t)gridDim.x * 4) {
      const float v = z[r * d + c] + shift;
      acc += (pass == 2) ? v * v : v;
    }
  }
  red[w][c] = acc;
  __syncthreads();
  if (w == 0) partial[blockIdx.x * 64 + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

__global__ void __launch_bounds__(64) actnorm_finalize_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                              int nb, int64_t n, int d, float scale,
                                                              float* __restrict__ bias, float* __restrict__ logs) {
  const int c = threadIdx.x;
  if (c >= d) return;
  float s1 = 0.0f, s2 = 0.0f;
  for (int b = 0; b < nb; ++b) {
    s1 += p1[b * 64 + c];
    s2 += p2[b * 64 + c];
  }
  const float mean = s1 / (float)n, var = s2 / (float)n;
  bias[c] = -mean;
  logs[c] = logf(scale / (sqrtf(var) + 1e-6f));
}

static float* g_actnorm_ws[gbnf::MAX_DEVICES] = {};   // per device: 2 x AN_MAX_BLOCKS x 64 floats, allocated on first use, never freed
static std::mutex g_actnorm_mu;

#if defined(GBNF_STAMPS) || defined(GBNF_TIMELINE) || defined(GBNF_CLOCK)
// diagnostic builds only (not part of include/gbnf.h): device buffer of 8 u64 per block
extern "C" int gbnf_debug_set_stamp_buffer(void* dev) { gbnf::g_stamp_buf = (unsigned long long*)dev; return 0; }
#endif

extern "C" {

int gbnf_flow_forward(const gbnf_flow* flow, const float* x, int64_t n, float* z, float* ldj, float* ll,
                      void* stream) {
  if (!flow) return fail(GBNF_ERR_INVALID, "gbnf_flow_forward: flow is null");
  if (n < 0) return fail(GBNF_ERR_INVALID, "gbnf_flow_forward: n < 0");
  if (n > 0 && !x) return fail(GBNF_ERR_INVALID, "gbnf_flow_forward: x is null");
  return launch_flow(flow, flow->self_table_dev, flow->self_table2_dev, false, 0, 1, x, n, z, ldj, ll, nullptr, (hipStream_t)stream, -1,
                     nullptr, 1, 0, flow->guard);
}

int gbnf_flow_inverse(const gbnf_flow* flow, const float* z, int64_t n, float* x, float* ldj, void* stream) {
  if (!flow) return fail(GBNF_ERR_INVALID, "gbnf_flow_inverse: flow is null");
  if (n < 0) return fail(GBNF_ERR_INVALID, "gbnf_flow_inverse: n < 0");
  if (n > 0 && (!z || !x)) return fail(GBNF_ERR_INVALID, "gbnf_flow_inverse: z / x is null");
  // every math mode (round 3: the split kernels run backwards too; an f16x3 launch is followed by its bf16x6 repair pass
  // like a forward one; the numerics guard of a DEFAULT handle keeps its verdict from the forward direction)
  return launch_flow(flow, flow->self_table_dev, flow->self_table2_dev, false, 0, 1, z, n, x, ldj, nullptr, nullptr, (hipStream_t)stream, -1,
                     nullptr, 1, 1, flow->guard);
}

int gbnf_mixture_create(gbnf_flow* const* flows, int32_t n_flows, gbnf_mixture** out) {
  if (!out) return fail(GBNF_ERR_INVALID, "gbnf_mixture_create: out is null");
  *out = nullptr;
  if (!flows || n_flows < 1) return fail(GBNF_ERR_INVALID, "gbnf_mixture_create: need >= 1 flow");
  const gbnf_flow* f0 = flows[0];
  // One launch = one kernel variant.  Components created in DEFAULT mode may have come out of their probes in different
  // split modes: the mixture then runs all of them as bf16x6 (an f16x3 handle carries that packing too).
  bool all_same = true, any_safe = false, all_have_safe = true;
  for (int c = 0; c < n_flows; ++c) {
    const gbnf_flow* f = flows[c];
    if (!f) return fail(GBNF_ERR_INVALID, "gbnf_mixture_create: flow %d is null", c);
    if (f->kind != f0->kind || f->d != f0->d || f->n_steps != f0->n_steps || f->additive != f0->additive ||
        f->hidden != f0->hidden || f->depth != f0->depth || f->act_a != f0->act_a || f->act_b != f0->act_b || f->residual != f0->residual)
      return fail(GBNF_ERR_INVALID, "gbnf_mixture_create: flow %d has a different architecture than flow 0", c);
    all_same = all_same && f->math_mode == f0->math_mode;
    any_safe = any_safe || f->math_mode == GBNF_MATH_BF16X6;
    all_have_safe = all_have_safe && (f->math_mode == GBNF_MATH_BF16X6 || (f->math_mode == GBNF_MATH_F16X3 && f->blob2_dev != nullptr));
  }
  const bool promote = !all_same && any_safe && all_have_safe;      // mixed f16x3 / bf16x6 -> everything on bf16x6
  if (!all_same && !promote)
    return fail(GBNF_ERR_INVALID, "gbnf_mixture_create: the flows were created in different math modes (%d vs %d)",
                f0->math_mode, GBNF_MATH_BF16X6);
  std::vector<const uint32_t*> table(n_flows), table2(n_flows, nullptr);
  bool have2 = !promote && f0->math_mode == GBNF_MATH_F16X3;
  const gbnf_flow* lead = f0;            // the flow whose launch table the mixture uses
  for (int c = 0; c < n_flows; ++c) {
    const gbnf_flow* f = flows[c];
    if (promote) {
      table[c] = f->math_mode == GBNF_MATH_BF16X6 ? f->blob_dev : f->blob2_dev;
      if (f->math_mode == GBNF_MATH_BF16X6) lead = f;
    } else {
      table[c] = f->blob_dev;
      table2[c] = f->blob2_dev;
      have2 = have2 && f->blob2_dev != nullptr;
    }
    const int vht = (promote && f->math_mode == GBNF_MATH_F16X3) ? f->var2_ht : f->var_ht;
    const int vot = (promote && f->math_mode == GBNF_MATH_F16X3) ? f->var2_ot : f->var_ot;
    const int lht = (promote && f0->math_mode == GBNF_MATH_F16X3) ? f0->var2_ht : f0->var_ht;
    const int lot = (promote && f0->math_mode == GBNF_MATH_F16X3) ? f0->var2_ot : f0->var_ot;
    if (vht != lht || vot != lot || (!promote && (f->var_ksl != f0->var_ksl || f->var_ks1 != f0->var_ks1)))
      return fail(GBNF_ERR_INVALID, "gbnf_mixture_create: flow %d runs on a different kernel variant than flow 0", c);
  }
  gbnf_mixture* m = new gbnf_mixture();
  m->flows.assign(flows, flows + n_flows);
  // the launch table comes from flows[0]: when the mixture is promoted and flows[0] is an f16x3 handle, its second
  // (bf16x6) launch table is used (launch_flow's use_second)
  m->use_blob2 = promote && f0->math_mode == GBNF_MATH_F16X3;
  m->math_mode = promote ? GBNF_MATH_BF16X6 : f0->math_mode;
  (void)lead;
  hipError_t e = hipMalloc((void**)&m->table_dev, sizeof(uint32_t*) * n_flows);
  if (e == hipSuccess) e = hipMemcpy(m->table_dev, table.data(), sizeof(uint32_t*) * n_flows, hipMemcpyHostToDevice);
  if (e == hipSuccess && have2) {
    e = hipMalloc((void**)&m->table2_dev, sizeof(uint32_t*) * n_flows);
    if (e == hipSuccess) e = hipMemcpy(m->table2_dev, table2.data(), sizeof(uint32_t*) * n_flows, hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && have2 && m->math_mode == GBNF_MATH_F16X3) {
    bool all_default = true;
    for (int c = 0; c < n_flows; ++c) all_default = all_default && flows[c]->requested_mode == GBNF_MATH_DEFAULT;
    if (all_default) e = make_guard(n_flows, &m->guard);
  }
  if (e != hipSuccess) {
    if (m->table_dev) (void)hipFree(m->table_dev);
    if (m->table2_dev) (void)hipFree(m->table2_dev);
    free_guard(m->guard);
    delete m;
    return fail(GBNF_ERR_HIP, "gbnf_mixture_create: %s", hipGetErrorString(e));
  }
  *out = m;
  return GBNF_OK;
}

int gbnf_mixture_destroy(gbnf_mixture* mix) {
  if (!mix) return GBNF_OK;
  if (mix->table_dev) (void)hipFree(mix->table_dev);
  if (mix->table2_dev) (void)hipFree(mix->table2_dev);
  if (mix->base_dev) (void)hipFree(mix->base_dev);
  free_guard(mix->guard);
  delete mix;
  return GBNF_OK;
}

int gbnf_mixture_set_base(gbnf_mixture* mix, const float* mean, const float* std) {
  if (!mix) return fail(GBNF_ERR_INVALID, "gbnf_mixture_set_base: mix is null");
  if ((mean == nullptr) != (std == nullptr))
    return fail(GBNF_ERR_INVALID, "gbnf_mixture_set_base: mean and std must both be given or both be null");
  GBNF_HIP(hipDeviceSynchronize());
  if (mix->base_dev) { (void)hipFree(mix->base_dev); mix->base_dev = nullptr; }
  if (mean == nullptr) return GBNF_OK;
  const int d = mix->flows[0]->d;
  for (int j = 0; j < d; ++j)
    if (!(std[j] > 0.0f)) return fail(GBNF_ERR_INVALID, "gbnf_mixture_set_base: std[%d] must be > 0", j);
  GBNF_HIP(hipMalloc((void**)&mix->base_dev, sizeof(float) * 2 * d));
  GBNF_HIP(hipMemcpy(mix->base_dev, mean, sizeof(float) * d, hipMemcpyHostToDevice));
  GBNF_HIP(hipMemcpy(mix->base_dev + d, std, sizeof(float) * d, hipMemcpyHostToDevice));
  return GBNF_OK;
}

int gbnf_mixture_component_log_prob(const gbnf_mixture* mix, const float* x, int64_t n, int32_t c_begin,
                                    int32_t c_end, float* ll, void* stream) {
  return gbnf_mixture_component_log_prob_strided(mix, x, n, c_begin, c_end, ll, n, stream);
}

int gbnf_mixture_component_log_prob_strided(const gbnf_mixture* mix, const float* x, int64_t n, int32_t c_begin,
                                            int32_t c_end, float* ll, int64_t ll_row_stride, void* stream) {
  const float* xs[1] = {x};
  return gbnf_mixture_component_log_prob_multi(mix, xs, 1, n, c_begin, c_end, ll, ll_row_stride, stream);
}

int gbnf_mixture_component_log_prob_multi(const gbnf_mixture* mix, const float* const* xs, int32_t n_batches, int64_t n,
                                          int32_t c_begin, int32_t c_end, float* ll, int64_t ll_row_stride,
                                          void* stream) {
  if (!mix) return fail(GBNF_ERR_INVALID, "gbnf_mixture_component_log_prob: mix is null");
  if (n_batches < 1 || n_batches > MAX_BATCHES)
    return fail(GBNF_ERR_INVALID, "n_batches=%d outside [1,%d]", n_batches, MAX_BATCHES);
  if (!xs) return fail(GBNF_ERR_INVALID, "xs is null");
  if (ll_row_stride < n * n_batches)
    return fail(GBNF_ERR_INVALID, "gbnf_mixture_component_log_prob: ll_row_stride < n_batches * n");
  const float* x = xs[0];
  for (int b = 0; b < n_batches; ++b)
    if (n > 0 && !xs[b]) return fail(GBNF_ERR_INVALID, "xs[%d] is null", b);
  const int C = (int)mix->flows.size();
  if (c_begin < 0 || c_end > C || c_begin > c_end)
    return fail(GBNF_ERR_INVALID, "component range [%d,%d) outside [0,%d)", c_begin, c_end, C);
  if (n < 0) return fail(GBNF_ERR_INVALID, "n < 0");
  if (n > 0 && c_end > c_begin && (!x || !ll)) return fail(GBNF_ERR_INVALID, "x / ll is null");
  return launch_flow(mix->flows[0], mix->table_dev, mix->table2_dev, mix->use_blob2, c_begin, c_end - c_begin, x, n, nullptr, nullptr, ll,
                     mix->base_dev, (hipStream_t)stream, ll_row_stride, xs, n_batches, 0, mix->guard);
}

int gbnf_mixture_component_forward(const gbnf_mixture* mix, const float* x, int64_t n, int32_t c_begin, int32_t c_end,
                                   float* z, float* ldj, float* ll, void* stream) {
  if (!mix) return fail(GBNF_ERR_INVALID, "gbnf_mixture_component_forward: mix is null");
  const int C = (int)mix->flows.size();
  if (c_begin < 0 || c_end > C || c_begin > c_end)
    return fail(GBNF_ERR_INVALID, "component range [%d,%d) outside [0,%d)", c_begin, c_end, C);
  if (n < 0) return fail(GBNF_ERR_INVALID, "n < 0");
  if (n > 0 && c_end > c_begin && (!x || (!z && !ldj && !ll))) return fail(GBNF_ERR_INVALID, "x is null or no output was asked for");
  return launch_flow(mix->flows[0], mix->table_dev, mix->table2_dev, mix->use_blob2, c_begin, c_end - c_begin, x, n, z, ldj, ll,
                     mix->base_dev, (hipStream_t)stream, n, nullptr, 1, 0, mix->guard);
}

int gbnf_mixture_lse(const float* ll, int64_t ll_row_stride, const float* rho_dev, int32_t n_components, int64_t n,
                     float* out, void* stream) {
  if (n_components < 1) return fail(GBNF_ERR_INVALID, "gbnf_mixture_lse: n_components < 1");
  if (n < 0) return fail(GBNF_ERR_INVALID, "gbnf_mixture_lse: n < 0");
  if (n == 0) return GBNF_OK;
  if (!ll || !rho_dev || !out) return fail(GBNF_ERR_INVALID, "gbnf_mixture_lse: null pointer");
  if (ll_row_stride < n) return fail(GBNF_ERR_INVALID, "gbnf_mixture_lse: row stride < n");
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(mixture_lse_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ll,
                     ll_row_stride, rho_dev, n_components, n, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "mixture_lse launch failed: %s", hipGetErrorString(e));
  return GBNF_OK;
}

int gbnf_mixture_log_prob(const gbnf_mixture* mix, const float* x, int64_t n, int32_t n_used, const float* rho_dev,
                          float* ll_workspace, float* out, void* stream) {
  if (!mix) return fail(GBNF_ERR_INVALID, "gbnf_mixture_log_prob: mix is null");
  if (n_used < 1 || n_used > (int)mix->flows.size())
    return fail(GBNF_ERR_INVALID, "n_used=%d outside [1,%d]", n_used, (int)mix->flows.size());
  int rc = gbnf_mixture_component_log_prob(mix, x, n, 0, n_used, ll_workspace, stream);
  if (rc) return rc;
  return gbnf_mixture_lse(ll_workspace, n, rho_dev, n_used, n, out, stream);
}

static void fill_numerics(const Guard* g, int mode, gbnf_numerics_status* out) {
  out->math_mode = mode; out->demoted = 0; out->checks = 0; out->worst_rel_err = 0.0f;
  out->tolerance = 1e-9f * (float)tuning().check_tolerance_e9.load(std::memory_order_relaxed);
  if (g == nullptr) return;
  const volatile GuardStatus* st = g->status_host;
  out->demoted = st->demoted != 0u ? 1 : 0;
  out->checks = (int64_t)st->checks;
  out->worst_rel_err = st->worst_rel_err;
  if (out->demoted && mode == GBNF_MATH_F16X3) out->math_mode = GBNF_MATH_BF16X6;
}

int gbnf_flow_numerics(const gbnf_flow* flow, gbnf_numerics_status* out) {
  if (!flow || !out) return fail(GBNF_ERR_INVALID, "gbnf_flow_numerics: null argument");
  fill_numerics(flow->guard, flow->math_mode, out);
  return GBNF_OK;
}

int gbnf_mixture_numerics(const gbnf_mixture* mix, gbnf_numerics_status* out) {
  if (!mix || !out) return fail(GBNF_ERR_INVALID, "gbnf_mixture_numerics: null argument");
  fill_numerics(mix->guard, mix->math_mode, out);
  return GBNF_OK;
}

int gbnf_tuning_set(const char* key, int32_t value) {
  std::atomic<int>* slot = tuning_slot(key);
  if (!slot) return fail(GBNF_ERR_INVALID, "gbnf_tuning_set: unknown key '%s'", key ? key : "(null)");
  slot->store(value, std::memory_order_relaxed);
  return GBNF_OK;
}

int gbnf_tuning_get(const char* key, int32_t* value) {
  std::atomic<int>* slot = tuning_slot(key);
  if (!slot || !value) return fail(GBNF_ERR_INVALID, "gbnf_tuning_get: unknown key '%s' or null value", key ? key : "(null)");
  *value = slot->load(std::memory_order_relaxed);
  return GBNF_OK;
}

int gbnf_actnorm_init(const float* z, int64_t n, int32_t d, float scale, float* bias_out, float* logs_out,
                      void* stream) {
  if (!z || !bias_out || !logs_out) return fail(GBNF_ERR_INVALID, "gbnf_actnorm_init: null pointer");
  if (n < 1) return fail(GBNF_ERR_INVALID, "gbnf_actnorm_init: needs at least one row");
  if (d < 1 || d > ZSLOTS) return fail(GBNF_ERR_UNSUPPORTED, "gbnf_actnorm_init: d=%d outside [1,%d]", d, ZSLOTS);
  float* ws = nullptr;
  {
    int dev = 0;
    GBNF_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= gbnf::MAX_DEVICES) return fail(GBNF_ERR_UNSUPPORTED, "gbnf_actnorm_init: device index %d", dev);
    std::lock_guard<std::mutex> lk(g_actnorm_mu);
    if (!g_actnorm_ws[dev]) GBNF_HIP(hipMalloc((void**)&g_actnorm_ws[dev], sizeof(float) * 2 * AN_MAX_BLOCKS * 64));
    ws = g_actnorm_ws[dev];
  }
  int nb = (int)((n + 63) / 64);
  if (nb > AN_MAX_BLOCKS) nb = AN_MAX_BLOCKS;
  float* p1 = ws;
  float* p2 = ws + AN_MAX_BLOCKS * 64;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(actnorm_partial_kernel, dim3(nb), dim3(256), 0, s, z, n, d, (const float*)nullptr, 1, p1);
  hipLaunchKernelGGL(actnorm_partial_kernel, dim3(nb), dim3(256), 0, s, z, n, d, (const float*)p1, 2, p2);
  hipLaunchKernelGGL(actnorm_finalize_kernel, dim3(1), dim3(64), 0, s, (const float*)p1, (const float*)p2, nb, n, d,
                     scale, bias_out, logs_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_actnorm_init launch failed: %s", hipGetErrorString(e));
  return GBNF_OK;
}

int gbnf_boosting_weights(const float* G, int64_t n, float beta, float* w_out, void* stream) {
  if (!G || !w_out) return fail(GBNF_ERR_INVALID, "gbnf_boosting_weights: null pointer");
  if (n < 1) return fail(GBNF_ERR_INVALID, "gbnf_boosting_weights: needs at least one sample");
  hipLaunchKernelGGL(boosting_weights_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, G, n, beta, w_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_boosting_weights launch failed: %s", hipGetErrorString(e));
  return GBNF_OK;
}

}  // extern "C"

// =====================================================================================================================
// Live blobs: the packed f16x3 blob of a component whose parameters live in device tensors (the training path's
// forward sweep on flow_kernel_hx3<..., TRAIN = 1>; gbnf_internal.h).
// =====================================================================================================================
namespace gbnf {

struct LiveLayer { const float* W; const float* b; int rows, cols; };
// one weight tile of the blob: NP fragments at word `dst`; kind 0: layer 0 (a = tile t), 1: hidden (a = out tile u, b = chunk c),
// 2: output layer (a = chunk c, b = out tile o); rows / cols: the layer's live extent (beyond it the tile is zero).
// The backward blob's TRANSPOSED tiles (A = W^T: tile row index = a column of W): kind 3: W3^T (a = hidden tile t, b = chunk c of
// net outputs), 4: W2^T (a = out tile u, b = chunk c), 5: W1^T (a = chunk c of hidden units, b = tile o of net inputs)
struct LiveTile { uint32_t dst; uint16_t lid; uint16_t kind; uint16_t a, b; float scale; int rows, cols; };
struct LiveBias { uint32_t dst; int lid; int row; int fold; float scale; };       // scale * (b[row] + (fold ? sum_k W[row][k] : 0)), row < 0: 0
struct LiveEntry { uint32_t dst; int m; int step; };                               // a live table entry: p0..p3 at dst + 32 / 64 / 96 / 128
struct LiveNorm { const float *na, *nb, *mean, *var; float eps; int has_norm; uint32_t ld_dst; };

struct LiveBlob {
  int kind = 0, d = 0, K = 0, additive = 0, nnets = 1;
  uint32_t* blob_dev = nullptr;
  const uint32_t** table_dev = nullptr;
  size_t blob_words = 0;
  LiveLayer* layers_dev = nullptr;
  LiveTile* tiles_dev = nullptr;
  LiveBias* bias_dev = nullptr;
  LiveEntry* entries_dev = nullptr;
  LiveNorm* norms_dev = nullptr;
  int n_tiles = 0, n_bias = 0, n_entries = 0;
  LaunchFn launch_nt[3] = {nullptr, nullptr, nullptr};
  const char* name_nt[3] = {nullptr, nullptr, nullptr};
  // the backward sweep (bwd_kernel_hx3): transposed-weight blob, its tile records, per-entry parameter indices
  uint32_t* blobB_dev = nullptr;
  const uint32_t** tableB_dev = nullptr;
  LiveTile* tilesB_dev = nullptr;
  int n_tilesB = 0;
  int32_t* bwd_tab_dev = nullptr;
  int64_t* bwd_goff_dev = nullptr;
  LaunchFn launch_bwd = nullptr;
  const char* name_bwd = nullptr;
  int bwd_waves = 4;                    // waves per workgroup of the backward launch (bwd_hx3_waves)
  int ht = 0;                           // hidden tiles of the kernel variant
  int depth = 1;                        // coupling_network_depth: hidden -> hidden layers per net
  bool tilesB_packed = false;           // the last re-pack included the transposed tiles
};

// the device twin of put_tile(): lane (i,g) computes its 8 values of one tile and writes 4 words per piece
// (sat: the device's range counter -- a weight beyond the fp16 range, +-65504 after the tanh pre-scale, cannot be split: hi rounds to inf, the
//  residual to -inf, a ReLU of the NaN they produce is 0.  No trained model has such weights; a diverged one is COUNTED here like every other
//  operand that leaves the range (gbnf_saturation_count), not passed over in silence)
__device__ __forceinline__ void live_tile(const LiveTile T, const LiveLayer* __restrict__ layers, uint32_t* __restrict__ blob, int lane,
                                          unsigned* __restrict__ sat) {
  const LiveLayer L = layers[T.lid];
  const int i = lane & 15, gg = lane >> 4;
  bool big = false;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t w[2] = {0u, 0u};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int j = 2 * q + e;
      int row, col;                          // element W[row][col] of the layer
      const int kslot_b = 16 * (2 * T.b + (j >> 2)) + 4 * gg + (j & 3), kslot_a = 16 * (2 * T.a + (j >> 2)) + 4 * gg + (j & 3);
      if (T.kind == 0) { row = 16 * T.a + i; col = 8 * gg + j; }
      else if (T.kind == 1) { row = 16 * T.a + i; col = kslot_b; }
      else if (T.kind == 2) { row = 16 * T.b + i; col = kslot_a; }
      else if (T.kind == 3 || T.kind == 4) { row = kslot_b; col = 16 * T.a + i; }      // contraction over W's rows
      else { row = kslot_a; col = 16 * T.b + i; }
      float r = (row < T.rows && col < T.cols) ? T.scale * L.W[(size_t)row * L.cols + col] : 0.0f;
      big = big || !(__builtin_fabsf(r) <= 65504.0f);
#pragma unroll
      for (int k = 0; k < 2; ++k) {       // hi = f16(r), mid = f16(r - hi): both round to nearest, like the host packer
        const _Float16 h = static_cast<_Float16>(r);
        w[k] |= (uint32_t)__builtin_bit_cast(unsigned short, h) << (16 * e);
        r -= static_cast<float>(h);
      }
    }
    blob[T.dst + (size_t)lane * 4 + q] = w[0];
    blob[T.dst + 256 + (size_t)lane * 4 + q] = w[1];
  }
  if (sat != nullptr && __any(big) && lane == 0) atomicAdd(sat, 1u);
}
__global__ void __launch_bounds__(64) live_tiles_kernel(const LiveTile* __restrict__ tiles, const LiveLayer* __restrict__ layers,
                                                        uint32_t* __restrict__ blob, unsigned* __restrict__ sat) {
  live_tile(tiles[blockIdx.x], layers, blob, (int)threadIdx.x, sat);
}

// biases (folded row sums in double, like the host packer), live table entries, per-step log-det constants
__device__ __forceinline__ void live_small(int block, const LiveBias* __restrict__ bias, int n_bias, const LiveEntry* __restrict__ ent,
                                           int n_ent, const LiveNorm* __restrict__ norms, int K, int d, int glow,
                                           const LiveLayer* __restrict__ layers, uint32_t* __restrict__ blob) {
  auto put = [&](uint32_t off, float v) { blob[off] = __builtin_bit_cast(uint32_t, v); };
  // biases: 16 lanes per word (the folded row sum of a tanh layer runs over up to 512 weights: one thread per row took 50 us)
  const int n_bias_blocks = (n_bias + 15) / 16;
  if (block < n_bias_blocks) {
    const int w = block * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    if (w < n_bias) {                       // (uniform per group of 16 lanes)
      const LiveBias B = bias[w];
      double acc = 0.0;
      if (B.row >= 0) {
        const LiveLayer L = layers[B.lid];
        if (B.fold)
          for (int k = sub; k < L.cols; k += 16) acc += (double)L.W[(size_t)B.row * L.cols + k];
        if (sub == 0) acc += (double)L.b[B.row];
      }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
      if (sub == 0) put(B.dst, B.row >= 0 ? B.scale * (float)acc : 0.0f);
    }
    return;
  }
  const int t = n_bias + (block - n_bias_blocks) * blockDim.x + threadIdx.x;
  if (t < n_bias + n_ent) {
    const LiveEntry E = ent[t - n_bias];
    const LiveNorm N = norms[E.step];
    float p0 = 0.f, p1 = 1.f, p2 = 1.f, p3 = 0.f;
    if (glow) {
      p0 = N.na[E.m]; p1 = expf(N.nb[E.m]); p2 = expf(-N.nb[E.m]);
    } else if (N.has_norm) {
      p0 = N.mean[E.m]; p1 = sqrtf(N.var[E.m] + N.eps); p2 = expf(N.na[E.m]); p3 = N.nb[E.m];
    }
    put(E.dst + 32, p0); put(E.dst + 64, p1); put(E.dst + 96, p2); put(E.dst + 128, p3);
  } else if (t < n_bias + n_ent + K) {
    const LiveNorm N = norms[t - n_bias - n_ent];
    float acc = 0.0f;
    if (glow) {
      for (int m = 0; m < d; ++m) acc += N.nb[m];                       // torch.sum(logs)
    } else if (N.has_norm) {
      for (int m = 0; m < d; ++m) acc += N.na[m] - 0.5f * logf(N.var[m] + N.eps);    // models/layers.py:357-358
    }
    put(N.ld_dst, acc);
  }
}
// ONE launch re-derives everything a training step's kernels read from the live parameters: the forward blob's tiles, the
// transposed tiles of the backward sweep (when one will follow: the trace contract), the biases / tables / log-det constants.
// (Three launches of ~6 us each were 8 % of the N = 4096 training step.)  256 threads: four tiles per block, then the small work.
__global__ void __launch_bounds__(256) live_pack_kernel(const LiveTile* __restrict__ tiles, int n_tiles, const LiveTile* __restrict__ tilesB,
                                                        int n_tilesB, uint32_t* __restrict__ blobB, const LiveBias* __restrict__ bias,
                                                        int n_bias, const LiveEntry* __restrict__ ent, int n_ent,
                                                        const LiveNorm* __restrict__ norms, int K, int d, int glow,
                                                        const LiveLayer* __restrict__ layers, uint32_t* __restrict__ blob,
                                                        unsigned* __restrict__ sat) {
  const int tile_blocks = (n_tiles + n_tilesB + 3) / 4;
  if ((int)blockIdx.x < tile_blocks) {
    const int t = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    if (t < n_tiles) live_tile(tiles[t], layers, blob, lane, sat);
    else if (t < n_tiles + n_tilesB) live_tile(tilesB[t - n_tiles], layers, blobB, lane, nullptr);      // (the same weights, transposed)
    return;
  }
  live_small((int)blockIdx.x - tile_blocks, bias, n_bias, ent, n_ent, norms, K, d, glow, layers, blob);
}

// The table entries and the log-det constant of ONE BatchNorm step re-derived from BATCH statistics (train() mode of the reference,
// models/layers.py:338-346: batch mean and unbiased variance of the step's input), behind bn_stats_kernel and in front of the launch
// of the step range that starts with it.
__global__ void __launch_bounds__(256) live_norm_step_kernel(const LiveEntry* __restrict__ ent, int n_ent, const LiveNorm* __restrict__ norms,
                                                             int step, int d, const float* __restrict__ bmean, const float* __restrict__ bvar,
                                                             uint32_t* __restrict__ blob) {
  const LiveNorm N = norms[step];
  if (!N.has_norm) return;
  auto put = [&](uint32_t off, float v) { blob[off] = __builtin_bit_cast(uint32_t, v); };
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_ent) {
    const LiveEntry E = ent[t];
    if (E.step == step) {
      put(E.dst + 32, bmean[E.m]);
      put(E.dst + 64, sqrtf(bvar[E.m] + N.eps));
    }
  } else if (t == n_ent) {
    float acc = 0.0f;
    for (int m = 0; m < d; ++m) acc += N.na[m] - 0.5f * logf(bvar[m] + N.eps);    // models/layers.py:357-358 on the batch variance
    put(N.ld_dst, acc);
  }
}

void live_blob_destroy(LiveBlob* lb) {
  if (!lb) return;
  if (lb->blob_dev) (void)hipFree(lb->blob_dev);
  if (lb->table_dev) (void)hipFree(lb->table_dev);
  if (lb->layers_dev) (void)hipFree(lb->layers_dev);
  if (lb->tiles_dev) (void)hipFree(lb->tiles_dev);
  if (lb->bias_dev) (void)hipFree(lb->bias_dev);
  if (lb->entries_dev) (void)hipFree(lb->entries_dev);
  if (lb->norms_dev) (void)hipFree(lb->norms_dev);
  if (lb->blobB_dev) (void)hipFree(lb->blobB_dev);
  if (lb->tableB_dev) (void)hipFree(lb->tableB_dev);
  if (lb->tilesB_dev) (void)hipFree(lb->tilesB_dev);
  if (lb->bwd_tab_dev) (void)hipFree(lb->bwd_tab_dev);
  if (lb->bwd_goff_dev) (void)hipFree(lb->bwd_goff_dev);
  delete lb;
}

template <typename T>
static hipError_t upload_vec(const std::vector<T>& v, T** dev) {
  *dev = nullptr;
  if (v.empty()) return hipSuccess;
  hipError_t e = hipMalloc((void**)dev, sizeof(T) * v.size());
  if (e == hipSuccess) e = hipMemcpy(*dev, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice);
  return e;
}

// The TRAIN kernel variant of a flow: the exact padded width if one is compiled, else the cheapest zero-padded superset (the
// per-step-activation variants are generic supersets).  GBNF_OK and *vc, or UNSUPPORTED.
static int live_choose(const gbnf_flow_desc* desc, DescInfo& info, VariantChoice* vc) {
  const int h = info.ref.hidden, depth = info.ref.depth;
  if (info.ref.residual ? (depth != 2 && depth != 4) : depth > 2)
    return fail(GBNF_ERR_UNSUPPORTED, "live blob: TanhNet / ReLUNet of coupling_network_depth 0, 1 or 2 and ResidualNets of one or two blocks only");
  if (info.ref.residual) info.act_a = info.act_b = GBNF_ACT_RESIDUAL_RELU;      // the kernels' key of a ResidualNet (depth 2 = one block)
  {   // an activation pair nobody compiled a kernel for runs on the per-step variants (as gbnf_flow_create_ex does)
    bool compiled = false;
    for (const Variant& v : variants())
      compiled = compiled || (v.key.ks1 == 1 && v.key.kind == desc->kind && v.key.act_a == info.act_a && v.key.act_b == info.act_b);
    // (a ResidualNet never takes the per-step-activation variants: they have no skip connection -- RES is a compile-time
    //  property of the act-2 kernels.  Glow + ResidualNet has no TRAIN variant, so such a descriptor keeps the per-step trainer:
    //  ADVICE r5)
    if (!compiled && info.ref.residual)
      return fail(GBNF_ERR_UNSUPPORTED, "live blob: no TRAIN kernel variant with a skip connection for kind=%d (ResidualNets train on the chained kernels for RealNVP only)", desc->kind);
    if (!compiled) info.act_a = info.act_b = GBNF_ACT_PER_STEP;
  }
  if (!choose_hx3(desc->kind, h, info.ot, depth, info.act_a, info.act_b, -3, vc, /*train=*/1)) {
    if (info.act_a != GBNF_ACT_PER_STEP && !info.ref.residual) {
      info.act_a = info.act_b = GBNF_ACT_PER_STEP;
      if (!choose_hx3(desc->kind, h, info.ot, depth, info.act_a, info.act_b, -3, vc, 1))
        return fail(GBNF_ERR_UNSUPPORTED, "live blob: no TRAIN kernel variant for kind=%d hidden=%d out_tiles=%d", desc->kind, h, info.ot);
    } else {
      return fail(GBNF_ERR_UNSUPPORTED, "live blob: no TRAIN kernel variant for kind=%d hidden=%d out_tiles=%d", desc->kind, h, info.ot);
    }
  }
  return GBNF_OK;
}

// Hidden rows (16 x hidden tiles) of the TRAIN variant a trainer of this flow would run, 0 if none: the trainer sizes its
// operand workspace by it (round 5: a width without a variant of its own trains on the next wider one, its extra rows are zeros)
int live_blob_train_rows(const gbnf_flow_desc* desc) {
  DescInfo info;
  if (validate_desc(desc, &info)) return 0;
  VariantChoice vc;
  if (live_choose(desc, info, &vc)) return 0;
  return 16 * vc.ht;
}

int live_blob_create(const gbnf_flow_desc* desc, const int64_t* norm_grad_offsets, LiveBlob** out) {
  *out = nullptr;
  DescInfo info;
  int rc = validate_desc(desc, &info);
  if (rc) return rc;
  const int d = desc->d, K = desc->n_steps;
  const bool glow = desc->kind == GBNF_KIND_GLOW;
  const bool additive = glow && desc->coupling == GBNF_COUPLING_ADDITIVE;
  const int h = info.ref.hidden, depth = info.ref.depth;
  VariantChoice vc;
  rc = live_choose(desc, info, &vc);
  if (rc) return rc;
  // ---- the value-independent words: pack a copy of the descriptor whose parameter arrays are host zeros
  static const std::vector<float> zeros((size_t)512 * 512 + 64, 0.0f);
  std::vector<gbnf_glow_step> gsteps;
  std::vector<gbnf_realnvp_step> rsteps;
  std::vector<std::vector<gbnf_linear>> lin_store;
  auto zero_net = [&](const gbnf_net& n) {
    lin_store.emplace_back(n.layers, n.layers + n.n_layers);
    for (gbnf_linear& l : lin_store.back()) { l.weight = zeros.data(); l.bias = zeros.data(); }
    gbnf_net z = n;
    z.layers = lin_store.back().data();
    return z;
  };
  lin_store.reserve((size_t)2 * K);
  gbnf_flow_desc dummy = *desc;
  if (glow) {
    gsteps.assign(desc->glow_steps, desc->glow_steps + K);
    for (gbnf_glow_step& g : gsteps) { g.actnorm_bias = zeros.data(); g.actnorm_logs = zeros.data(); g.block = zero_net(g.block); }
    dummy.glow_steps = gsteps.data();
  } else {
    rsteps.assign(desc->realnvp_steps, desc->realnvp_steps + K);
    for (gbnf_realnvp_step& r : rsteps) {
      if (r.has_batch_norm) { r.bn_log_gamma = r.bn_beta = r.bn_running_mean = r.bn_running_var = zeros.data(); }
      r.t_net = zero_net(r.t_net);
      r.s_net = zero_net(r.s_net);
    }
    dummy.realnvp_steps = rsteps.data();
  }
  PackedBlob pb;
  pack_component(&dummy, info, vc, &pb);

  // ---- where every value-dependent word comes from (the loops of pack_component / pack_net_hx3, recording instead of computing)
  const int HT = vc.ht, OT = vc.ot, NP = 2, nnets = glow ? 1 : 2;
  const Hx3Layout L(HT, OT, NP, depth);
  const size_t NW = (size_t)L.NET_WORDS, step_words = SMALL_WORDS + nnets * NW, TW = (size_t)NP * 256;
  const int d1 = d / 2, d2 = d - d1;
  std::vector<LiveLayer> layers;
  std::vector<LiveTile> tiles;
  std::vector<LiveBias> biases;
  std::vector<LiveEntry> entries;
  std::vector<LiveNorm> norms(K);
  const BwdLayout LB(HT, OT, depth);
  const size_t NWB = (size_t)LB.NET_WORDS, step_words_b = nnets * NWB;
  std::vector<LiveTile> tilesB;
  std::vector<int32_t> bwd_tab((size_t)K * 2 * 4 * NENT, -1);
  // (the stage order of BwdLayout: the output layer's transpose, the hidden -> hidden layers' from the last to the first, layer 0's)
  auto add_net_bwd = [&](int lid0, size_t base, int in_f, int out_f) {       // lid0: the net's first LiveLayer (layers lid0 .. lid0 + depth + 1)
    int s = 0;
    for (int i0 = 0; i0 < LB.N_L0; ++i0, ++s)
      for (int n = 0; n < LB.nf[s] / NP; ++n) {
        const int t = i0 * LB.ROWS0 + n / LB.K0, c = n % LB.K0;
        tilesB.push_back(LiveTile{(uint32_t)(base + LB.off[s] + (size_t)n * TW), (uint16_t)(lid0 + depth + 1), 3, (uint16_t)t, (uint16_t)c, 1.0f, out_f, h});
      }
    if (depth >= 1) {
      for (int j = depth; j >= 2; --j)
        for (int u = 0; u < HT; ++u, ++s)
          for (int c = 0; c < LB.HC; ++c)
            tilesB.push_back(LiveTile{(uint32_t)(base + LB.off[s] + (size_t)c * TW), (uint16_t)(lid0 + j), 4, (uint16_t)u, (uint16_t)c, 1.0f, h, h});
      for (int u = 0; u < HT; ++u, ++s) {
        for (int c = 0; c < LB.HC; ++c)
          tilesB.push_back(LiveTile{(uint32_t)(base + LB.off[s] + (size_t)c * TW), (uint16_t)(lid0 + 1), 4, (uint16_t)u, (uint16_t)c, 1.0f, h, h});
        if (u % 2 == 0 && u >= 2)
          for (int o = 0; o < 2; ++o)
            tilesB.push_back(LiveTile{(uint32_t)(base + LB.off[s] + (size_t)(LB.HC + o) * TW), (uint16_t)lid0, 5, (uint16_t)((u - 2) / 2), (uint16_t)o, 1.0f, h, in_f});
      }
      for (int o = 0; o < 2; ++o)
        tilesB.push_back(LiveTile{(uint32_t)(base + LB.off[s] + (size_t)o * TW), (uint16_t)lid0, 5, (uint16_t)(LB.HC - 1), (uint16_t)o, 1.0f, h, in_f});
    } else {
      for (int k = 0; k < LB.N_IN; ++k, ++s) {
        const int c0 = k * LB.CGI, cnt = std::min(LB.CGI, LB.HC - c0);
        for (int cc = 0; cc < cnt; ++cc)
          for (int o = 0; o < 2; ++o)
            tilesB.push_back(LiveTile{(uint32_t)(base + LB.off[s] + (size_t)(cc * 2 + o) * TW), (uint16_t)lid0, 5, (uint16_t)(c0 + cc), (uint16_t)o, 1.0f, h, in_f});
      }
    }
  };
  // (the loops of pack_net_hx3, every depth: layer 0, `depth` hidden -> hidden layers, the output layer)
  auto add_net = [&](const gbnf_net& net, size_t base, int in_f, int out_f) {
    const bool tanh_net = net.activation == GBNF_ACT_TANH;
    const float T = tanh_net ? 2.8853900817779268f : 1.0f, R = tanh_net ? -2.0f : 1.0f;
    int lid[6];
    for (int l = 0; l < depth + 2; ++l) {
      lid[l] = (int)layers.size();
      layers.push_back(LiveLayer{net.layers[l].weight, net.layers[l].bias, net.layers[l].out_features, net.layers[l].in_features});
    }
    const int lout = lid[depth + 1];
    for (int t = 0; t < HT; ++t)
      for (int k = 0; k < 16; ++k) {
        const int u = 16 * t + k;
        biases.push_back(LiveBias{(uint32_t)(base + (size_t)t * 16 + k), lid[0], u < h ? u : -1, 0, T});
        for (int j = 1; j <= depth; ++j)
          biases.push_back(LiveBias{(uint32_t)(base + (size_t)(j * HT + t) * 16 + k), lid[j], u < h ? u : -1, tanh_net ? 1 : 0, T});
      }
    for (int o = 0; o < OT; ++o)
      for (int k = 0; k < 16; ++k) {
        const int r = 16 * o + k;
        biases.push_back(LiveBias{(uint32_t)(base + (size_t)((depth + 1) * HT + o) * 16 + k), lout, r < out_f ? r : -1, tanh_net ? 1 : 0, 1.0f});
      }
    int s = 0;
    for (int i0 = 0; i0 < L.N_L0; ++i0, ++s)
      for (int tl = 0; tl < L.nf[s] / NP; ++tl)
        tiles.push_back(LiveTile{(uint32_t)(base + L.off[s] + (size_t)tl * TW), (uint16_t)lid[0], 0, (uint16_t)(i0 * L.TL0 + tl), 0, T, h, in_f});
    for (int jl = 1; jl <= depth; ++jl)
      for (int u = 0; u < HT; ++u, ++s) {
        for (int c = 0; c < L.HC; ++c)
          tiles.push_back(LiveTile{(uint32_t)(base + L.off[s] + (size_t)c * TW), (uint16_t)lid[jl], 1, (uint16_t)u, (uint16_t)c, T * R, h, h});
        if (jl == depth && u % 2 == 0 && u >= 2)
          for (int o = 0; o < OT; ++o)
            tiles.push_back(LiveTile{(uint32_t)(base + L.off[s] + (size_t)(L.HC + o) * TW), (uint16_t)lout, 2, (uint16_t)((u - 2) / 2), (uint16_t)o, R, out_f, h});
      }
    if (depth >= 1) {
      for (int o = 0; o < OT; ++o)
        tiles.push_back(LiveTile{(uint32_t)(base + L.off[s] + (size_t)o * TW), (uint16_t)lout, 2, (uint16_t)(L.HC - 1), (uint16_t)o, R, out_f, h});
    } else {
      for (int k = 0; k < L.N_OUT; ++k, ++s) {
        const int c0 = k * L.CG, cnt = std::min(L.CG, L.HC - c0);
        for (int cc = 0; cc < cnt; ++cc)
          for (int o = 0; o < OT; ++o)
            tiles.push_back(LiveTile{(uint32_t)(base + L.off[s] + (size_t)(cc * OT + o) * TW), (uint16_t)lout, 2, (uint16_t)(c0 + cc), (uint16_t)o, R, out_f, h});
      }
    }
    return lid[0];
  };
  for (int s = 0; s < K; ++s) {
    const size_t sb = step_words * s;
    int in_f, out_f;
    std::vector<int> src(d);             // src[j] = index into the step's parameter vectors of new logical feature j
    LiveNorm& N = norms[s];
    N = LiveNorm{nullptr, nullptr, nullptr, nullptr, 0.0f, 0, (uint32_t)(sb + 1)};
    if (glow) {
      const gbnf_glow_step& st = desc->glow_steps[s];
      N.na = st.actnorm_bias; N.nb = st.actnorm_logs; N.has_norm = 1;
      for (int j = 0; j < d; ++j) src[j] = (int)st.perm_indices[j];
      in_f = d1; out_f = d2;
    } else {
      const gbnf_realnvp_step& st = desc->realnvp_steps[s];
      N.has_norm = st.has_batch_norm ? 1 : 0;
      N.na = st.bn_log_gamma; N.nb = st.bn_beta; N.mean = st.bn_running_mean; N.var = st.bn_running_var; N.eps = st.bn_eps;
      in_f = st.flipped ? d2 : d1; out_f = st.flipped ? d1 : d2;
      for (int j = 0; j < d; ++j) src[j] = st.flipped ? ((j < d2) ? d1 + j : j - d2) : j;
    }
    const bool paired = glow && !additive;
    if (glow || N.has_norm) {            // (a RealNVP step without BatchNorm keeps the identity constants of the template)
      for (int gg = 0; gg < 4; ++gg)
        for (int e = 0; e < NENT; ++e) {
          const int k = 8 * gg + e;
          if (k < in_f) {
            entries.push_back(LiveEntry{(uint32_t)(sb + SMALL_HDR + gg * NENT + e), src[k], s});
            bwd_tab[((size_t)s * 2 + 0) * 4 * NENT + gg * NENT + e] = src[k];
          }
          const int jj = paired ? 8 * (e >> 1) + 2 * gg + (e & 1) : 16 * (e >> 2) + 4 * gg + (e & 3);
          if (jj < out_f) {
            entries.push_back(LiveEntry{(uint32_t)(sb + SMALL_HDR + 160 + gg * NENT + e), src[in_f + jj], s});
            bwd_tab[((size_t)s * 2 + 1) * 4 * NENT + gg * NENT + e] = src[in_f + jj];
          }
        }
    }
    const int net_out = paired ? 2 * out_f : out_f;
    const size_t sbb = step_words_b * s;
    if (glow) {
      const int l0 = add_net(desc->glow_steps[s].block, sb + SMALL_WORDS, in_f, net_out);
      add_net_bwd(l0, sbb, in_f, net_out);
    } else {
      const int l0 = add_net(desc->realnvp_steps[s].t_net, sb + SMALL_WORDS, in_f, net_out);
      const int l1 = add_net(desc->realnvp_steps[s].s_net, sb + SMALL_WORDS + NW, in_f, net_out);
      add_net_bwd(l0, sbb, in_f, net_out);
      add_net_bwd(l1, sbb + NWB, in_f, net_out);
    }
  }

  LiveBlob* lb = new LiveBlob();
  lb->kind = desc->kind; lb->d = d; lb->K = K; lb->additive = additive ? 1 : 0; lb->nnets = nnets;
  lb->blob_words = pb.words.size();
  lb->ht = vc.ht;
  lb->depth = depth;
  lb->n_tiles = (int)tiles.size(); lb->n_bias = (int)biases.size(); lb->n_entries = (int)entries.size();
  for (int nt = 1; nt <= 2; ++nt) { lb->launch_nt[nt] = vc.launch_nt[nt]; lb->name_nt[nt] = vc.name_nt[nt]; }
  hipError_t e = upload_blob(pb.words, &lb->blob_dev, &lb->table_dev);
  if (e == hipSuccess) e = upload_vec(layers, &lb->layers_dev);
  if (e == hipSuccess) e = upload_vec(tiles, &lb->tiles_dev);
  if (e == hipSuccess) e = upload_vec(biases, &lb->bias_dev);
  if (e == hipSuccess) e = upload_vec(entries, &lb->entries_dev);
  if (e == hipSuccess) e = upload_vec(norms, &lb->norms_dev);
  // ---- the backward sweep, where a variant of bwd_kernel_hx3 exists for this geometry (else the caller keeps its own)
  if (e == hipSuccess && norm_grad_offsets != nullptr) {
    const Variant* vb = find_variant(VariantKey{desc->kind, vc.ht, -3, 2, vc.ot, 1, depth, info.act_a, info.act_b});
    // (the backward kernel keeps every step's tables in LDS: flows of more than LDS_TABLE_STEPS steps, or whose tables + stage
    //  slots + per-wave state do not fit a CU, keep the round-1 backward kernels -- the forward sweep has a global-table form)
    if (vb != nullptr && (K > LDS_TABLE_STEPS || bwd_hx3_lds_bytes(K, bwd_hx3_waves(K, LB.STAGE_FRAGS, d), LB.STAGE_FRAGS, d) > 160 * 1024))
      vb = nullptr;
    if (vb != nullptr) {
      const std::vector<uint32_t> zerosB(step_words_b * K + 64, 0u);      // (padding words stay zero; every tile is rewritten per call)
      std::vector<int64_t> goff(norm_grad_offsets, norm_grad_offsets + 2 * K);
      lb->n_tilesB = (int)tilesB.size();
      e = upload_blob(zerosB, &lb->blobB_dev, &lb->tableB_dev);
      if (e == hipSuccess) e = upload_vec(tilesB, &lb->tilesB_dev);
      if (e == hipSuccess) e = upload_vec(bwd_tab, &lb->bwd_tab_dev);
      if (e == hipSuccess) e = upload_vec(goff, &lb->bwd_goff_dev);
      lb->launch_bwd = vb->fn;
      lb->name_bwd = vb->name;
      lb->bwd_waves = bwd_hx3_waves(K, LB.STAGE_FRAGS, d);
    }
  }
  if (e != hipSuccess) {
    live_blob_destroy(lb);
    return fail(GBNF_ERR_HIP, "live blob: %s", hipGetErrorString(e));
  }
  *out = lb;
  return GBNF_OK;
}

// grads[g_na / g_nb of step k + j] += sum over workgroups of partials[.][k][which][j], in a fixed order
__global__ void __launch_bounds__(1024) bwd_param_reduce_kernel(const float* __restrict__ partials, int n_wg, int K, int d,
                                                                const int64_t* __restrict__ goff, float* __restrict__ grads) {
  __shared__ float red[16][64];
  const int kw = blockIdx.x, j = threadIdx.x & 63, part = threadIdx.x >> 6;       // kw = step * 2 + which
  const float* src = partials + kw * 64 + j;
  const int64_t stride = (int64_t)K * 128;
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  int b = part;
  for (; b + 48 < n_wg; b += 64) {
    a0 += src[(int64_t)b * stride];
    a1 += src[(int64_t)(b + 16) * stride];
    a2 += src[(int64_t)(b + 32) * stride];
    a3 += src[(int64_t)(b + 48) * stride];
  }
  for (; b < n_wg; b += 16) a0 += src[(int64_t)b * stride];
  red[part][j] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (part == 0 && j < d) {
    float v = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += red[q][j];
    grads[goff[kw] + j] += v;
  }
}

bool live_blob_has_backward(const LiveBlob* lb) { return lb != nullptr && lb->launch_bwd != nullptr; }
int live_blob_hidden_rows(const LiveBlob* lb) { return lb ? 16 * lb->ht : 0; }

// g_z / g_ldj (either may be null) -> g_x (or null), the ActNorm / BatchNorm entries of `grads`, and the gradient-side
// operands of the weight gradients in `acts` (the workspace the forward sweep filled); gmax: gmax_kernel's result
int live_blob_backward(LiveBlob* lb, int64_t n, const float* trace, float* acts, int64_t np, int ip, int hp, int op,
                       const float* g_z, const float* g_ldj, float* g_x, float* grads, const unsigned* gmax, void* stream,
                       LiveReduce* reduce_out, const LiveRange* range) {
  hipStream_t s = (hipStream_t)stream;
  // the transposed tiles were packed by the forward call that wrote the trace (the parameters are unchanged since: the trace
  // contract of include/gbnf.h); a trainer that has not run one yet packs them here
  if (!lb->tilesB_packed)
    hipLaunchKernelGGL(live_tiles_kernel, dim3((unsigned)lb->n_tilesB), dim3(64), 0, s, (const LiveTile*)lb->tilesB_dev,
                       (const LiveLayer*)lb->layers_dev, lb->blobB_dev, (unsigned*)nullptr);
  FlowLaunch p{};
  p.blobs = lb->table_dev; p.blobs_bwd = lb->tableB_dev;
  p.n = n; p.d = lb->d; p.n_steps = lb->K; p.n_comp = 1; p.n_batches = 1; p.additive = lb->additive;
  p.sat = reinterpret_cast<unsigned long long*>(training_saturation_counter());
  p.acts_out = acts; p.np = np; p.tr_ip = ip; p.tr_hp = hp; p.tr_op = op; p.net_rows = ip + 2 * (lb->depth + 1) * hp + 2 * op;
  p.bwd_tab = lb->bwd_tab_dev; p.bwd_goff = lb->bwd_goff_dev; p.trace_in = trace;
  p.g_z = g_z; p.g_ldj = g_ldj; p.g_x = g_x; p.grads = grads; p.gmax = gmax;
#if defined(GBNF_STAMPS)
  p.dbg = g_stamp_buf;
#endif
  // the workgroups' parameter-gradient sums go to the slack rows behind the last operand region (their contents are of no
  // consequence to wgrad_kernel): 4-wave workgroups at most => np / 64 * K * 128 <= 24 np floats of the 320 np there
  p.partials = acts + (int64_t)lb->K * lb->nnets * p.net_rows * np;
  if (range != nullptr) {
    p.k_begin = range->k_begin; p.k_end = range->k_end; p.state_in = range->state_in; p.state_out = range->state_out;
  }
  hipError_t e = lb->launch_bwd(p, 0u, s);
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "launch of %s failed: %s", lb->name_bwd, hipGetErrorString(e));
  const int n_wg = (int)((np / 16 + lb->bwd_waves - 1) / lb->bwd_waves);
  if (reduce_out != nullptr) {
    *reduce_out = LiveReduce{p.partials, n_wg, lb->K, lb->d, lb->bwd_goff_dev, 0u};
    return GBNF_OK;
  }
  hipLaunchKernelGGL(bwd_param_reduce_kernel, dim3((unsigned)(2 * lb->K)), dim3(1024), 0, s, (const float*)p.partials, n_wg, lb->K, lb->d,
                     (const int64_t*)lb->bwd_goff_dev, grads);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "bwd_param_reduce launch failed: %s", hipGetErrorString(e));
  return GBNF_OK;
}

static void live_blob_repack(LiveBlob* lb, hipStream_t s, bool with_backward = false) {
  const int n_tilesB = (with_backward && lb->launch_bwd != nullptr) ? lb->n_tilesB : 0;
  const int small_blocks = (lb->n_bias + 15) / 16 + (lb->n_entries + lb->K + 255) / 256;
  const int tile_blocks = (lb->n_tiles + n_tilesB + 3) / 4;
  hipLaunchKernelGGL(live_pack_kernel, dim3((unsigned)(tile_blocks + small_blocks)), dim3(256), 0, s, (const LiveTile*)lb->tiles_dev,
                     lb->n_tiles, (const LiveTile*)lb->tilesB_dev, n_tilesB, lb->blobB_dev, (const LiveBias*)lb->bias_dev, lb->n_bias,
                     (const LiveEntry*)lb->entries_dev, lb->n_entries, (const LiveNorm*)lb->norms_dev, lb->K, lb->d,
                     lb->kind == GBNF_KIND_GLOW ? 1 : 0, (const LiveLayer*)lb->layers_dev, lb->blob_dev, training_saturation_counter());
  lb->tilesB_packed = n_tilesB > 0;
}

int live_blob_words(const LiveBlob* lb, uint32_t* out_host, int64_t* n_words) {
  if (n_words) *n_words = (int64_t)lb->blob_words;
  if (out_host) {
    live_blob_repack(const_cast<LiveBlob*>(lb), nullptr);
    GBNF_HIP(hipDeviceSynchronize());
    GBNF_HIP(hipMemcpy(out_host, lb->blob_dev, lb->blob_words * 4, hipMemcpyDeviceToHost));
  }
  return GBNF_OK;
}

int live_blob_forward(LiveBlob* lb, const float* x, int64_t n, float* z, float* ldj, float* trace, float* acts, int64_t np,
                      int ip, int hp, int op, void* stream, const LiveRange* range) {
  hipStream_t s = (hipStream_t)stream;
  if (range == nullptr || range->repack)
    live_blob_repack(lb, s, /*with_backward=*/true);      // (a traced forward call: the backward call follows on the same parameters)
  if (range != nullptr && range->bmean != nullptr)
    hipLaunchKernelGGL(live_norm_step_kernel, dim3((unsigned)((lb->n_entries + 1 + 255) / 256)), dim3(256), 0, s,
                       (const LiveEntry*)lb->entries_dev, lb->n_entries, (const LiveNorm*)lb->norms_dev, range->k_begin, lb->d,
                       range->bmean, range->bvar, lb->blob_dev);
  // 32-sample waves once that still gives every SIMD of the chip a wave, else 16-sample ones (as pick_nt does)
  int nt = pick_nt(n, 1);
  if (lb->launch_nt[nt] == nullptr) nt = 1;
  FlowLaunch p{};
  p.blobs = lb->table_dev; p.z_out = z; p.ldj_out = ldj; p.ll_out = nullptr;
  p.n_batches = 1; p.xs[0] = x;
  p.n = n; p.out_stride = n; p.d = lb->d; p.n_steps = lb->K; p.c_begin = 0; p.n_comp = 1;
  p.n_tiles = (int32_t)((n + 16 * nt - 1) / (16 * nt)); p.additive = lb->additive;
  p.sat = reinterpret_cast<unsigned long long*>(training_saturation_counter());      // (TRAIN instantiations touch the counter word only, never the marks)
  p.seq = next_serial();
  p.trace_out = trace; p.acts_out = acts; p.np = np; p.tr_ip = ip; p.tr_hp = hp; p.tr_op = op; p.net_rows = ip + 2 * (lb->depth + 1) * hp + 2 * op;
  if (range != nullptr) {
    p.k_begin = range->k_begin; p.k_end = range->k_end; p.state_in = range->state_in; p.state_out = range->state_out;
    p.ldj_accumulate = range->ldj_accumulate;
  }
#if defined(GBNF_STAMPS)
  p.dbg = g_stamp_buf;
#endif
  const hipError_t e = lb->launch_nt[nt](p, 0u, s);
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "launch of %s failed: %s", lb->name_nt[nt], hipGetErrorString(e));
  return GBNF_OK;
}

}  // namespace gbnf

// (tests) the words of a packed evaluation handle, and of a trainer's live blob after a device re-pack: tests/test_hip_train.py
// compares them bit for bit (the device packer must reproduce the host packer)
extern "C" int gbnf_debug_flow_blob(const gbnf_flow* flow, uint32_t* out_host, int64_t* n_words) {
  if (!flow) return gbnf::fail(GBNF_ERR_INVALID, "gbnf_debug_flow_blob: flow is null");
  if (n_words) *n_words = (int64_t)flow->blob_words;
  if (out_host) GBNF_HIP(hipMemcpy(out_host, flow->blob_dev, flow->blob_words * 4, hipMemcpyDeviceToHost));
  return GBNF_OK;
}
