/*
 * gbnf.h -- C ABI of libgbnf_hip.so: the MI355X (gfx950) boosted normalizing-flow
 * density evaluator.
 *
 * The reference project has NO native / FFI layer: its boundary for this path is the
 * Python call  BoostedFlow.forward(x=, components=c)  ->  self.flows[c](x)
 * (models/boosted_flow.py:220-228) and the mixture arithmetic its callers write inline
 * (density_experiment.py:561-573).  Each entry point below states which reference
 * code it replaces.  INTEGRATION.md shows the ctypes binding a reference maintainer
 * would add.
 *
 * Conventions
 *  - plain C, no torch types.  `x`, `z`, `ldj`, `ll`, `rho_dev`, `out` are DEVICE pointers
 *    to contiguous float32 owned by the caller (PyTorch tensors' data_ptr()); descriptor
 *    pointers (`gbnf_*_desc`, `gbnf_linear.weight`, ...) are HOST pointers that only need
 *    to stay valid for the duration of the create call (the handle owns packed copies).
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Kernels are
 *    enqueued on it; no call synchronises the device except create/destroy/set_base.
 *  - every function returns GBNF_OK (0) or a negative gbnf_status; it never throws and
 *    never exits.  gbnf_last_error() gives the message for the calling thread.
 *  - handles are immutable after creation: concurrent forward calls on different streams
 *    are safe; create/destroy must not race with in-flight work on the same handle.
 */
#ifndef GBNF_H_
#define GBNF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GBNF_ABI_VERSION 4

typedef enum gbnf_status {
  GBNF_OK = 0,
  GBNF_ERR_INVALID = -1,      /* bad argument / inconsistent descriptor            */
  GBNF_ERR_UNSUPPORTED = -2,  /* shape or option with no compiled kernel variant    */
  GBNF_ERR_HIP = -3,          /* a HIP runtime call failed                          */
  GBNF_ERR_NO_DEVICE = -4     /* no gfx950 device visible                           */
} gbnf_status;

enum { GBNF_KIND_GLOW = 0, GBNF_KIND_REALNVP = 1 };
/* TANH / RELU: TanhNet / ReLUNet (models/layers.py:208-243), n_layers = coupling_network_depth + 2 Linear layers.
 * RESIDUAL_RELU: ResidualNet (models/layers.py:246-301) of B = coupling_network_depth blocks, n_layers = 2 + 2 B Linear
 * layers in the order initial_layer, (blocks[b].linear_layers[0], blocks[b].linear_layers[1]) for b < B, final_layer:
 *     t = initial(x);  t += lin1_b(relu(lin0_b(relu(t)))) for every block;  out = final(t)
 * One block (the reference's default depth): the split kernels, every hidden width, both directions and training; two blocks: the
 * split kernels and the register-chained training kernels to hidden width 256 (round 5), the exact-f32 kernel and the per-step
 * training kernels beyond (B <= 2). */
enum { GBNF_ACT_TANH = 0, GBNF_ACT_RELU = 1, GBNF_ACT_RESIDUAL_RELU = 2 };
enum { GBNF_COUPLING_AFFINE = 0, GBNF_COUPLING_ADDITIVE = 1 };
/* How the coupling-network matrix products are evaluated:
 *   F32     exact f32 MFMA (v_mfma_f32_16x16x4_f32), bitwise an ordered fmaf chain (depth 0 / 1 / 2, ResidualNets of <= 2 blocks; hidden <= 512);
 *   F16X3   each f32 operand split into two fp16 pieces (22 significand bits); a.b = a_mid.b_hi + a_hi.b_mid +
 *           a_hi.b_hi on the f16 matrix pipe with f32 accumulation (TanhNet / ReLUNet, coupling_network_depth 0 / 1 / 2, hidden <= 512;
 *           ResidualNets of one block -- the reference's default depth -- at every width and of two blocks to 256, round 5): the fast
 *           path, ~1e-7 relative in the log-likelihood on well-conditioned models.  An operand beyond the fp16 range
 *           (|v| > 65504) cannot be represented: the kernel marks such samples and a bf16x6 repair pass behind every
 *           f16x3 launch re-evaluates them (same stream, no host synchronisation), so results are range-safe;
 *   BF16X6  three bf16 pieces per operand (>= 24 bits, f32 range), six products: f32-faithful for every finite input;
 *           ~2x the matrix work of F16X3 (same shapes as F16X3);
 *   DEFAULT per component: both split packings are built and a probe batch (128 rows ~ N(0,1) / N(0,4)) is evaluated
 *           on both at creation; F16X3 if they agree to 2.5e-6 relative in the log-likelihood, else BF16X6 (ill-conditioned
 *           models: e.g. un-normalised ReLU RealNVPs); F32 where no split kernel applies (two-block ResidualNets wider than 256).
 *           Env GBNF_MATH=f32|f16x3|bf16x6 overrides DEFAULT in gbnf_flow_create. */
enum { GBNF_MATH_DEFAULT = -1, GBNF_MATH_F32 = 0, GBNF_MATH_F16X3 = 1, GBNF_MATH_BF16X6 = 2 };

/* nn.Linear: y = x W^T + b, W row-major (out_features, in_features).
 * TanhNet / ReLUNet layers, models/layers.py:208-243. */
typedef struct gbnf_linear {
  const float* weight;
  const float* bias;
  int32_t out_features;
  int32_t in_features;
} gbnf_linear;

/* Coupling network: Linear -> [act, Linear] x depth -> act, Linear  (n_layers = depth + 2). */
typedef struct gbnf_net {
  int32_t activation; /* GBNF_ACT_* */
  int32_t n_layers;
  const gbnf_linear* layers;
} gbnf_net;

/* One tabular Glow FlowStep: ActNorm1d -> Permute1d -> coupling.  models/glow.py:317-342. */
typedef struct gbnf_glow_step {
  const float* actnorm_bias;   /* (d,)  _ActNorm.bias,  models/layers.py:467 */
  const float* actnorm_logs;   /* (d,)  _ActNorm.logs,  models/layers.py:468 */
  const int64_t* perm_indices; /* (d,)  PermuteNd.indices (NOT in state_dict), models/layers.py:636-651 */
  gbnf_net block;              /* in d/2 -> out 2*(d-d/2) (affine) or d-d/2 (additive) */
} gbnf_glow_step;

/* One RealNVP step: [BatchNorm eval] -> split/swap -> t,s nets -> affine.
 * models/transformations.py:560-579, models/layers.py:337-358. */
typedef struct gbnf_realnvp_step {
  int32_t flipped;             /* (k + flip_init) % 2, models/realnvp.py:38,118 */
  int32_t has_batch_norm;
  const float* bn_log_gamma;   /* (d,) each; ignored when !has_batch_norm */
  const float* bn_beta;
  const float* bn_running_mean;
  const float* bn_running_var;
  float bn_eps;
  gbnf_net t_net;              /* shift net */
  gbnf_net s_net;              /* log-scale net */
} gbnf_realnvp_step;

/* One boosted component = BoostedFlow.flows[c]  (models/boosted_flow.py:42-50). */
typedef struct gbnf_flow_desc {
  int32_t kind;      /* GBNF_KIND_* */
  int32_t d;         /* features (z_size) */
  int32_t n_steps;   /* K = num_flows */
  int32_t coupling;  /* GBNF_COUPLING_* (glow only) */
  const gbnf_glow_step* glow_steps;       /* n_steps entries when kind == GLOW */
  const gbnf_realnvp_step* realnvp_steps; /* n_steps entries when kind == REALNVP */
} gbnf_flow_desc;

typedef struct gbnf_flow gbnf_flow;       /* one component, packed on the device */
typedef struct gbnf_mixture gbnf_mixture; /* C same-architecture components      */

/* What a handle's kernel variant looks like (for roofline accounting in bench.py). */
typedef struct gbnf_kernel_info {
  int32_t hidden_tiles;        /* 16-unit MFMA tiles per hidden layer            */
  int32_t out_tiles;
  int32_t samples_per_wave;    /* 16 or 32                                        */
  int32_t n_steps;
  double macs_per_sample;        /* algorithmic multiply-adds per sample per component */
  double padded_macs_per_sample; /* what the MFMA tiles actually execute          */
  int64_t packed_bytes;          /* device bytes of packed parameters per component */
  int32_t math_mode;             /* GBNF_MATH_F32, _F16X3 or _BF16X6 actually used */
  float probe_rel_err;           /* DEFAULT mode: f16x3 vs bf16x6 on the creation-time probe batch (max relative
                                    log-likelihood difference; inf = a probe row left the fp16 range); -1 = no probe ran */
} gbnf_kernel_info;

int gbnf_version(void);
const char* gbnf_last_error(void);

/* The split-f16 kernels (evaluation, training) represent an f32 operand by two fp16 pieces: beyond +-65504 it cannot be
 * stored (the normalised input of a coupling net, a ReLU activation; in training also scaled gradients).  The EVALUATION
 * kernels repair such samples themselves (bf16x6 pass, see GBNF_MATH_F16X3); the training kernels saturate (and a trainer's
 * re-pack counts a WEIGHT beyond the range: it cannot be split at all).  This
 * returns how many waves ran into that since the last reset -- 0 for z-scored data on a trained flow -- and optionally
 * resets the counter.  One counter per device: this reports (and resets) the CURRENT device's, after a
 * hipDeviceSynchronize() (every stream of that device). */
int gbnf_saturation_count(int64_t* count, int32_t reset);
/* The part of that count that came from TRAINING launches (gbnf_trainer_forward / _backward: traced and untraced sweeps, the weight
 * re-pack): these saturate and are NOT repaired, so a non-zero value means steps with wrong gradients -- while the rest of
 * gbnf_saturation_count() (evaluation launches) was re-evaluated in the same call.  Same device, same synchronisation; `reset` clears
 * this part only.  (The drop-in module reads it when it leaves training mode: BoostedFlow.train / .eval.) */
int gbnf_training_saturation_count(int64_t* count, int32_t reset);

/* Numerics guard of a GBNF_MATH_DEFAULT handle that runs on f16x3 (round 3; what the creation-time probe cannot know is
 * the caller's data).  The library itself re-checks the choice on the data it is given: on the FIRST launch of a handle
 * and then on every `check_every`-th (tuning key below, default 256) it evaluates up to 256 rows of the launch's first
 * batch on f16x3 and on bf16x6 and compares the log-likelihoods ON THE DEVICE, in stream order behind the launch; if they
 * differ by more than `tolerance` (2.5e-6 relative, a quarter of the 1e-5 parity bar) the handle's guard word is set and
 *   - the bf16x6 pass that follows every f16x3 launch (the repair pass for out-of-range samples) re-evaluates the WHOLE
 *     work list from then on, starting with the launch that failed the check: no result of a failed mode reaches the caller;
 *   - the host sees the flag (pinned memory, no synchronisation) on a later call and launches bf16x6 directly.
 * Every entry point that evaluates a flow (gbnf_flow_forward, gbnf_mixture_component_log_prob*, gbnf_mixture_log_prob)
 * is covered, so C-ABI callers, sharded.GroupPipeline and bench.py get it like BoostedFlow does.  Handles created with
 * an explicit GBNF_MATH_F16X3 keep the caller's choice (range repair only).  These calls never synchronise: `checks`
 * and `worst_rel_err` are those of the checks that have COMPLETED on the device. */
typedef struct gbnf_numerics_status {
  int32_t math_mode;       /* GBNF_MATH_* the NEXT launch of this handle will run in                       */
  int32_t demoted;         /* 1 = a check failed: the handle left f16x3 for bf16x6                          */
  int64_t checks;          /* device-side checks completed so far                                           */
  float worst_rel_err;     /* largest relative log-likelihood difference f16x3 vs bf16x6 seen by a check    */
  float tolerance;
} gbnf_numerics_status;
int gbnf_flow_numerics(const gbnf_flow* flow, gbnf_numerics_status* out);
int gbnf_mixture_numerics(const gbnf_mixture* mix, gbnf_numerics_status* out);

/* Launch-policy knobs (process-wide; tests, soak runs and tuning -- the defaults are what is measured and shipped):
 *   "force_nt"      0 = automatic | 1 | 2 : samples per wave = 16 x NT            (env GBNF_FORCE_NT at first use)
 *   "wg_pairs"      -1 = automatic (two 4-wave workgroups per CU whenever they fit -- small batches too, round 5 --, except LONG launches of
 *                   geometries with heavy weight stages, which run as one 8-wave workgroup per CU: +1.2 % on the headline, round 6) |
 *                   0 = never (always 8-wave workgroups) | 1 = pairs whenever they fit                (env GBNF_NO_WG_PAIRS=1 -> 0)
 *   "coop"          the LATENCY form of the f16x3 flow kernel (csrc/gbnf_flow_kernel_coop.hip.h, round 6: the waves of a workgroup share
 *                   ONE sample tile; one log_prob call at the reference's 512 / 1024-row batches 48 -> 29 / 37 us): -1 = automatic (taken
 *                   while every workgroup gets a CU to itself) | 0 = never | 1 / 2 / 3 = always form 1 / 2 / 3 (16-sample tiles on 4 waves,
 *                   32 on 4, 32 on 8) where the geometry has it.  Forward direction, depth-1 TanhNet / ReLUNet geometries listed as
 *                   `coop` lines in csrc/variants.list.  Results of the forms agree to 2e-6 (another summation order of the output layer):
 *                   a lone batch and the same batch inside a longer launch may differ in the last bits.
 *   "coop_max_wgs"  the latency form is taken while the call's sample tiles x components fill at most this many workgroups (default 256)
 *   "repair"        1 | 0 : the bf16x6 pass behind f16x3 launches                 (env GBNF_NO_REPAIR=1 -> 0)
 *   "nt2_min_waves" 32-sample waves from this many waves on (default 1024)        (env GBNF_NT2_MIN_WAVES)
 *   "check_every"   numerics guard: a check on launch 0 and every this many launches (default 256; 0 = first launch only;
 *                   -1 = never)
 *   "check_tolerance_e9"  numerics guard: tolerance of a check in units of 1e-9 relative (default 2500 = 2.5e-6; 0 makes
 *                   every check fail: how the tests exercise the demotion path)
 * Returns GBNF_ERR_INVALID for an unknown key. */
int gbnf_tuning_set(const char* key, int32_t value);
int gbnf_tuning_get(const char* key, int32_t* value);

/* Replaces: constructing flows[c] + .to(device)  (models/boosted_flow.py:42-50).
 * Packs (pads, tiles, folds slot maps) and uploads the parameters.
 * LIMITS (enforced: GBNF_ERR_UNSUPPORTED with the reason in gbnf_last_error; nothing falls back to a non-HIP path):
 *   features                1 <= d <= 64                 (a sample tile's features are LDS slots of one wave; the reference's five
 *                                                         tabular datasets have d = 6, 8, 21, 43, 63)
 *   coupling-net input      <= 32 features               (d / 2, or d - d / 2 for a flipped RealNVP step: one k = 32 MFMA chunk)
 *   hidden width            1 <= h <= 512                (TanhNet / ReLUNet at coupling_network_depth 0, 1, 2 and ResidualNets of one or two
 *                                                         blocks: all on the split kernels at every width -- two-block ResidualNets above
 *                                                         256 since round 6, evaluation; their TRAINING keeps the per-step kernels)
 *   coupling_network_depth  0, 1, 2; ResidualNet blocks 1, 2 (RealNVP only, as in the reference)
 *   flow steps              any K (per-step tables are staged in LDS up to K = 24 where they fit, read from the blob beyond; the chained
 *                                                         training sweeps take K <= 24)
 *   activations             tanh / relu, also drawn per step or per net (`--coupling_network random`) */
int gbnf_flow_create(const gbnf_flow_desc* desc, gbnf_flow** out);
/* Same with an explicit GBNF_MATH_* mode (gbnf_flow_create uses GBNF_MATH_DEFAULT, or env GBNF_MATH=f32|f16x3|bf16x6). */
int gbnf_flow_create_mode(const gbnf_flow_desc* desc, int32_t math_mode, gbnf_flow** out);
/* ... and creation flags.  A component whose coupling nets do not all use the same activation (the reference's
 * `--coupling_network random` draws TanhNet / ReLUNet per step, models/glow.py:295-296, or per net,
 * models/realnvp.py:59-60) runs on kernel variants that read the activation per step from the packed blob; that choice
 * is automatic.  GBNF_CREATE_PER_STEP_ACTIVATION asks for those variants although this component is uniform, so that it
 * can share a mixture (one launch, gbnf_mixture_create wants one kernel for all components) with components that are not. */
enum { GBNF_CREATE_PER_STEP_ACTIVATION = 1 };
int gbnf_flow_create_ex(const gbnf_flow_desc* desc, int32_t math_mode, int32_t flags, gbnf_flow** out);
int gbnf_flow_destroy(gbnf_flow* flow);
int gbnf_flow_info(const gbnf_flow* flow, gbnf_kernel_info* info);

/* Replaces: z, _, _, ldj, _ = self.flows[c](x)   (models/boosted_flow.py:220-222 ->
 * Glow.encode models/glow.py:92-110 / RealNVPFlow.encode models/realnvp.py:115-127).
 * x (n,d) -> z (n,d), ldj (n,), ll (n,) = log N(z;0,I) + ldj  (utils/distributions.py:44-60,
 * density_experiment.py:565).  Any of z / ldj / ll may be NULL to skip that output. */
int gbnf_flow_forward(const gbnf_flow* flow, const float* x, int64_t n,
                      float* z, float* ldj, float* ll, void* stream);

/* The flow backwards, z (n,d) -> x (n,d) and log|det dx/dz| (n,) (= -ldj of the forward pass at x): what
 * BoostedFlow.decode / Glow.decode / FlowStep.decode (models/boosted_flow.py:209-218, models/glow.py:112-123, 344-366)
 * and RealNVPFlow.decode (models/realnvp.py:97-113) are meant to do.  In the reference this direction is dead or wrong
 * on tabular data, so parity is defined by inverse(forward(x)) == x (and the one case the reference can
 * decode, additive Glow, fixture g9).  Any math mode (the split kernels run backwards since round 3; ResidualNets: f32). */
int gbnf_flow_inverse(const gbnf_flow* flow, const float* z, int64_t n, float* x, float* ldj, void* stream);

/* Replaces: the nn.ModuleList of components (models/boosted_flow.py:42).  All flows must
 * share one architecture (they do: every component is built from the same args) and one math mode -- except that
 * DEFAULT-created components which came out of their probes as a mix of F16X3 and BF16X6 are accepted: the mixture
 * then runs every component on its bf16x6 packing.
 * The mixture does NOT take ownership of the flows; they must outlive it. */
int gbnf_mixture_create(gbnf_flow* const* flows, int32_t n_flows, gbnf_mixture** out);
int gbnf_mixture_destroy(gbnf_mixture* mix);

/* Optional base density N(mean_j, std_j) per feature instead of N(0,1): the toy driver's
 * model.base_dist (models/generative_flow.py:22-23,38-42; toy_experiment.py:424).
 * HOST pointers (d,), or NULL/NULL to restore the standard normal. */
int gbnf_mixture_set_base(gbnf_mixture* mix, const float* mean, const float* std);

/* Replaces: the loop  for c in range(...): model(x=x, components=c); log_normal_standard(z)+ldj
 * (density_experiment.py:562-565) for components [c_begin, c_end) in ONE launch.
 * ll is (c_end - c_begin, n) row-major. */
int gbnf_mixture_component_log_prob(const gbnf_mixture* mix, const float* x, int64_t n,
                                    int32_t c_begin, int32_t c_end, float* ll, void* stream);
/* Same, writing row c of ll at ll + (c - c_begin) * ll_row_stride (>= n): lets several batches share one
 * (components, batches*n) table so that ONE all-gather / ONE recursion launch serves a group of batches. */
int gbnf_mixture_component_log_prob_strided(const gbnf_mixture* mix, const float* x, int64_t n,
                                            int32_t c_begin, int32_t c_end, float* ll, int64_t ll_row_stride,
                                            void* stream);
/* Same for a GROUP of up to 32 independent batches in ONE launch: xs is a HOST array of n_batches DEVICE pointers to
 * (n,d) inputs; batch b's log-densities go to columns [b*n, (b+1)*n) of the (c_end-c_begin, >= n_batches*n) table.
 * (A rank that holds few components does not fill the GPU with one batch; serving several batches per launch does,
 * and one all-gather + one recursion launch then cover the whole group.) */
int gbnf_mixture_component_log_prob_multi(const gbnf_mixture* mix, const float* const* xs, int32_t n_batches,
                                          int64_t n, int32_t c_begin, int32_t c_end, float* ll,
                                          int64_t ll_row_stride, void* stream);

/* Replaces: the calls  z, _, _, ldj, _ = model(x=x, components=c)  of the reference's evaluate loop
 * (density_experiment.py:562-563; models/boosted_flow.py:220-228) for ALL components [c_begin, c_end) of one batch in ONE
 * launch: z is (c_end - c_begin, n, d), ldj and ll are (c_end - c_begin, n) row-major; any of the three may be NULL (not all).
 * The host mirror answers the loop's per-component calls from this table (BoostedFlow.encode in eval mode). */
int gbnf_mixture_component_forward(const gbnf_mixture* mix, const float* x, int64_t n, int32_t c_begin, int32_t c_end,
                                   float* z, float* ldj, float* ll, void* stream);

/* Replaces: the recursive prefix-normalised 2-way logsumexp (density_experiment.py:567-571):
 *   G_0 = ll_0;  r_c = rho_c / sum(rho[0..c]);  G_c = LSE(log(1-r_c)+G_{c-1}, log(r_c)+ll_c).
 * ll is (n_components, n) with row stride `ll_row_stride` floats; rho_dev is the DEVICE
 * `rho` buffer (models/boosted_flow.py:32-39), at least n_components long. */
int gbnf_mixture_lse(const float* ll, int64_t ll_row_stride, const float* rho_dev,
                     int32_t n_components, int64_t n, float* out, void* stream);

/* The measured path: component_log_prob for components [0, n_used) + mixture_lse.
 * ll_workspace is (n_used, n) floats of caller-owned device scratch (also an output). */
int gbnf_mixture_log_prob(const gbnf_mixture* mix, const float* x, int64_t n, int32_t n_used,
                          const float* rho_dev, float* ll_workspace, float* out, void* stream);

/* ---- the component-sharded group inside the library (multi-GPU, round 4) -------------------------------------------------
 * Replaces: the serial component loop + recursion of density_experiment.py:561-573 when the C components are sharded over W
 * GPUs (BASELINE.json north star: one component block per GPU, an RCCL all-gather of log p_c(x), then the mixture recursion).
 * RCCL is bound directly (librccl.so is loaded at first use: no torch.distributed on the data path):
 *   gbnf_comm_unique_id   rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks through the caller's
 *                         own rendezvous (the host mirror uses the torch.distributed store);
 *   gbnf_comm_create      every rank of the group, collectively (ncclCommInitRank); world = 1 is allowed.
 * gbnf_mixture_group_log_prob: ONE call = flow launch of this rank's components (the mixture holds exactly its block, in global
 * component order rank by rank) over n_batches batches -> ncclAllGather of its (C/W, n_batches n) table into ll_full
 * (C, n_batches n), component order -> recursion -> G (n_batches n), all on `stream`.  comm NULL: one rank, no exchange (ll_full
 * unused).  All buffers are caller-owned device memory.
 * gbnf_group_graph_*: the same sequence captured ONCE into a HIP graph (bound to these very buffers; an un-captured warm-up call
 * runs first) and replayed with one hipGraphLaunch per group.  A capture that the runtime or RCCL refuses is reported as an
 * error: the caller keeps gbnf_mixture_group_log_prob. */
typedef struct gbnf_comm gbnf_comm;
typedef struct gbnf_group_graph gbnf_group_graph;
int gbnf_comm_unique_id(uint8_t* id128);
int gbnf_comm_create(const uint8_t* id128, int32_t rank, int32_t world, gbnf_comm** out);
int gbnf_comm_destroy(gbnf_comm* comm);
int gbnf_comm_info(const gbnf_comm* comm, int32_t* rank, int32_t* world);
int gbnf_mixture_group_log_prob(const gbnf_mixture* mix, gbnf_comm* comm, const float* const* xs, int32_t n_batches, int64_t n,
                                int32_t n_components, const float* rho_dev, float* ll_local, float* ll_full, float* G,
                                void* stream);
int gbnf_group_graph_create(const gbnf_mixture* mix, gbnf_comm* comm, const float* const* xs, int32_t n_batches, int64_t n,
                            int32_t n_components, const float* rho_dev, float* ll_local, float* ll_full, float* G,
                            gbnf_group_graph** out);
int gbnf_group_graph_launch(gbnf_group_graph* graph, void* stream);
int gbnf_group_graph_destroy(gbnf_group_graph* graph);

/* Replaces: _ActNorm.initialize_parameters (models/layers.py:473-486), the data-dependent initialisation the
 * reference runs on the first training batches (density_experiment.py:346-356):
 *   bias = -mean_0(z);  logs = log(scale / (sqrt(mean_0((z + bias)^2)) + 1e-6)).
 * z is the (n,d) DEVICE input of the ActNorm layer (the output of the preceding flow steps); bias_out / logs_out
 * are (d,) DEVICE buffers.  The first call allocates a small internal workspace. */
int gbnf_actnorm_init(const float* z, int64_t n, int32_t d, float scale, float* bias_out, float* logs_out,
                      void* stream);

/* Replaces: the boosting sample weights of compute_kl_pq_loss (density_experiment.py:624-640), computed from the
 * mixture log-density G (n,) of the fixed components:
 *   w = softmax(-G) (utils/utilities.py:12-14);  w = w^beta;  if max(w) > 0.1: w = clamp(w, 0.01, 0.1);
 *   if sum(w) != 1: w /= sum(w).
 * G and w_out are DEVICE buffers of n floats.  (The multinomial resampling that follows stays with the caller's RNG.) */
int gbnf_boosting_weights(const float* G, int64_t n, float beta, float* w_out, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Training path: what loss.backward() / optimizer.step() of density_experiment.py:366-374 do
 * to ONE component -- the forward on the LIVE parameters and its backward pass.
 *
 * A trainer binds the DEVICE addresses of the caller's parameter tensors: the descriptor has the same shape as for
 * gbnf_flow_create, but every float pointer (weights, biases, ActNorm / BatchNorm arrays) is a DEVICE pointer to the
 * tensor in its reference layout (nn.Linear.weight (out,in) row-major, ...) and stays owned by the caller;
 * perm_indices stay HOST pointers.  Nothing is packed or copied, so parameters updated in place by an optimiser are
 * seen by the next call; re-create the trainer only when a tensor is re-allocated.  RealNVP BatchNorm is differentiated
 * in its running-statistics (eval) form, models/layers.py:347-358, unless gbnf_trainer_set_batch_stats(1).
 * --------------------------------------------------------------------------------------------------------------- */
typedef struct gbnf_trainer gbnf_trainer;

/* The descriptor checks gbnf_flow_create applies (reads sizes and perm_indices, never the parameter arrays). */
int gbnf_flow_validate(const gbnf_flow_desc* desc);

int gbnf_trainer_create(const gbnf_flow_desc* desc_device_params, gbnf_trainer** out);
int gbnf_trainer_destroy(gbnf_trainer* trainer);

/* z, ldj = flows[c](x) on the live parameters (same semantics and outputs as gbnf_flow_forward).  `trace` (optional,
 * gbnf_trainer_trace_floats(n) floats of DEVICE memory) receives every step's normalised state; handing it to
 * gbnf_trainer_backward saves that call the forward sweep and the re-splitting of the weights (valid only while the
 * parameters are unchanged and no other forward call of this trainer ran on changed parameters in between).
 * Since round 3 the trace buffer of a flow that runs on the register-chained kernels (TanhNet / ReLUNet coupling nets of
 * coupling_network_depth 0, 1, 2 and RealNVP ResidualNets of one or two blocks -- all but depth 1 since round 5 -- of a compiled width)
 * is also the OPERAND WORKSPACE of the step: behind the states the forward call stores the coupling nets' inputs, hidden
 * activations and outputs (gbnf_trainer_trace_floats accounts for it: K * nets * (ip + 2 L hp + 2 op) rows of
 * n-rounded-up-to-32 floats, L = hidden layers per net -- 20 KB per sample for MINIBOONE, K = 5, depth 1), and
 * gbnf_trainer_backward WRITES the gradient-side operands of the weight gradients into the same buffer (its `trace`
 * argument is const for the states only).  One trace buffer therefore serves one forward + one backward call. */
int gbnf_trainer_trace_floats(const gbnf_trainer* trainer, int64_t n, int64_t* n_floats);
int gbnf_trainer_forward(const gbnf_trainer* trainer, const float* x, int64_t n, float* z, float* ldj, float* trace,
                         void* stream);

/* RealNVP BatchNorm in the reference's train() form (models/layers.py:338-346): normalise with the BATCH mean and the
 * unbiased batch variance of every step's input, differentiate through them.  bind: DEVICE (d,) buffers that receive a
 * step's batch mean / variance at every forward call (the reference keeps them as BatchNorm.batch_mean / batch_var; the
 * running-statistics update with momentum stays with the caller).  set(1) switches forward / backward to one launch per
 * step (the statistics need the whole batch); it requires the trace buffer in both calls and n >= 2. */
int gbnf_trainer_bind_batch_stats(gbnf_trainer* trainer, int32_t step, float* mean_dev, float* var_dev);
int gbnf_trainer_set_batch_stats(gbnf_trainer* trainer, int32_t on);

/* Size of the flat parameter-gradient buffer, in floats.  Layout, step by step in descriptor order:
 *   glow:    [actnorm bias (d)] [actnorm logs (d)]  then per Linear of the block:   [weight (out*in)] [bias (out)]
 *   realnvp: [bn log_gamma (d)] [bn beta (d)]  (left zero without batch norm)  then t_net's Linears, then s_net's. */
int gbnf_trainer_grad_floats(const gbnf_trainer* trainer, int64_t* n_floats);
/* Bytes of caller-owned DEVICE scratch one backward call over n samples needs. */
int gbnf_trainer_workspace_bytes(const gbnf_trainer* trainer, int64_t n, int64_t* bytes);

/* Backward of (z, ldj) = flows[c](x): given g_z = dL/dz (n,d) and g_ldj = dL/dldj (n,) (either may be NULL = zero),
 * ACCUMULATES dL/dparameters into `grads` (the caller zeroes it; layout above) and writes dL/dx to g_x (n,d) unless
 * NULL.  `trace` is the forward call's trace buffer or NULL.  With it, a flow on the register-chained kernels (above)
 * recomputes nothing: the backward kernel chains W3^T -> W2^T -> W1^T on the saved activations, ActNorm / BatchNorm
 * gradients are summed per workgroup and added in a fixed order (bit-identical from run to run; the weight gradients still
 * use float atomics across sample blocks).  Other flows recompute the coupling nets' activations from the trace, and with
 * NULL the whole forward sweep from x.
 * Batch-statistics mode (gbnf_trainer_set_batch_stats): the BatchNorm entries of `grads` (log_gamma, beta of every step) MUST
 * be zero on entry -- the correction for the statistics' dependence on every sample reads THIS call's two batch sums from
 * them; accumulate several calls by adding up separately zeroed buffers (what the Python mirror does: a fresh buffer per
 * call).  That mode needs n >= 2 (unbiased variance) and K <= 32. */
int gbnf_trainer_backward(const gbnf_trainer* trainer, const float* x, int64_t n, const float* trace, const float* g_z,
                          const float* g_ldj, float* g_x, float* grads, void* workspace, int64_t workspace_bytes,
                          void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Image components (BASELINE.json configs[3]): density evaluation of one multi-scale image
 * Glow, models/glow.py:92-110 with the image branches of FlowNet / FlowStep (:192-252, :317-342).  All arrays are HOST
 * pointers read at creation (the handle is immutable, like gbnf_flow).
 * --------------------------------------------------------------------------------------------------------------- */
/* Conv2d (+ its ActNorm2d) or Conv2dZeros, models/layers.py:577-630: stride 1, 'same' zero padding. */
typedef struct gbnf_conv {
  const float* weight;        /* (out, in, k, k) = nn.Conv2d.weight                                      */
  const float* bias;          /* (out,) or NULL (a Conv2d followed by ActNorm2d has none)                 */
  const float* actnorm_bias;  /* (out,) ActNorm2d.bias after the convolution, or NULL                     */
  const float* actnorm_logs;  /* (out,)                                                                   */
  const float* logs;          /* (out,) Conv2dZeros.logs (output scale exp(3 logs)), or NULL              */
  int32_t out_channels, in_channels, kernel_size;
} gbnf_conv;

/* One image FlowStep: ActNorm2d -> InvertibleConv1x1 | Permute2d -> ConvNet coupling. */
typedef struct gbnf_image_step {
  const float* actnorm_bias;    /* (C,) */
  const float* actnorm_logs;    /* (C,) */
  const float* perm_weight;     /* (C,C) the matrix InvertibleConv1x1.get_weight returns (models/layers.py:751-776,
                                   either parameterisation), or NULL for a channel permutation */
  const int64_t* perm_indices;  /* (C,) Permute2d.indices when perm_weight is NULL */
  int32_t n_convs;              /* coupling_network_depth + 2: 3x3 (+ActNorm2d), 1x1 (+ActNorm2d) x depth, Conv2dZeros 3x3 */
  const gbnf_conv* convs;
} gbnf_image_step;

typedef struct gbnf_image_level {   /* SqueezeLayer(2), n_steps FlowSteps, Split2d (all levels but the last) */
  int32_t n_steps;
  const gbnf_image_step* steps;
  const gbnf_conv* split_prior;     /* Split2d.conv (Conv2dZeros C/2 -> C), NULL on the last level */
} gbnf_image_level;

typedef struct gbnf_image_flow_desc {
  int32_t channels, height, width;  /* input_size, e.g. 3, 32, 32 */
  int32_t n_levels;                 /* L = num_blocks */
  int32_t coupling;                 /* GBNF_COUPLING_* */
  int32_t hidden;                   /* h_size (width of the coupling ConvNets) */
  float bounds;                     /* Glow.bounds, models/glow.py:50 (0.9) */
  const gbnf_image_level* levels;
  const gbnf_conv* learn_top;       /* Glow.learn_top_fn (Conv2dZeros 2Cz -> 2Cz) or NULL: zero-mean unit-variance prior */
} gbnf_image_flow_desc;

typedef struct gbnf_image_flow gbnf_image_flow;

/* LIMITS of the image path (enforced with GBNF_ERR_UNSUPPORTED): inputs of at most 32 x 32 pixels with even sides at every squeeze,
 * 1 to 3 levels -- the reference's CIFAR-10 / SVHN (3 x 32 x 32), MNIST / Omniglot / Caltech (1 x 28 x 28) and Frey faces (1 x 28 x 20)
 * loaders, utils/load_data.py:389-529.  The kernels work on 16- and 8-wide square maps; a smaller map (a 14 x 14 first level, the
 * 4 x 4 map of a third level) lives in the top-left corner of that storage, zero outside (x, z, eps and noise at the boundary
 * always have the map's own size);
 * <= 64 channels per level; coupling ConvNets of hidden width <= 512 with 2 .. 5 convolutions (coupling_network_depth 0 .. 3);
 * the split-f16 kernels serve depth 1 at every hidden width (above 256 the fused kernel works in two halves of the hidden
 * channels), depth 0 and 2 to hidden width 256 (round 6), and <= 24 input channels of the first 3 x 3 (the 48-channel third level of a
 * 3 x 32 x 32 input); everything else runs on the exact-f32 convolution kernels.  Not built: y-conditioning, learned dequantisation flows, image training. */
int gbnf_image_flow_create(const gbnf_image_flow_desc* desc, gbnf_image_flow** out);
/* ... with an explicit GBNF_MATH_* mode: DEFAULT (what gbnf_image_flow_create does: split-f16 coupling nets if the create-time
 * probe passes), F32 (exact-f32 convolutions everywhere, no probe), F16X3. */
int gbnf_image_flow_create_mode(const gbnf_image_flow_desc* desc, int32_t math_mode, gbnf_image_flow** out);
int gbnf_image_flow_destroy(gbnf_image_flow* flow);
/* Shape of z (per image) and the algorithmic multiply-adds per image; any pointer may be NULL. */
int gbnf_image_flow_info(const gbnf_image_flow* flow, int32_t* z_channels, int32_t* z_height, int32_t* z_width,
                         double* macs_per_image);
int gbnf_image_flow_workspace_bytes(const gbnf_image_flow* flow, int64_t n, int64_t* bytes);
/* Replaces: z, z_mu, z_var, logdet, _ = self.flows[c](x) for image input (models/glow.py:92-110) and
 * ll = log_normal_diag(z, z_mu, z_var) + logdet (image_experiment.py:227).  x (n,C,H,W) in [0,1]; noise (n,C,H,W) is the
 * U(0,1) dequantisation noise of models/glow.py:135 (NULL = none); z (n,Cz,Hz,Wz) and ll (n,) may be NULL; ldj (n,) is
 * required (it is also the accumulator).  z_mu / z_var are per-channel constants: gbnf_image_flow_prior. */
int gbnf_image_flow_forward(const gbnf_image_flow* flow, const float* x, const float* noise, int64_t n, float* z,
                            float* ldj, float* ll, void* workspace, int64_t workspace_bytes, void* stream);
/* The top prior per channel: mean (Cz,) then log-variance (Cz,) into a HOST buffer of 2 Cz floats. */
int gbnf_image_flow_prior(const gbnf_image_flow* flow, float* mean_logvar_host);
/* State of the image path's numerics protocol (the tabular one: gbnf_flow_numerics).  The coupling nets run on
 * split-f16 MFMA (GBNF_MATH_F16X3) when the create-time probe -- 4 synthetic images through both arithmetic paths --
 * agrees with the exact-f32 kernels to `tolerance` (worst_rel_err = what it measured), else the handle runs on exact f32
 * (math_mode GBNF_MATH_F32, demoted = 1).  On split-f16 every operand is range-watched, and what the watch finds is dealt
 * with IN THE SAME CALL, on the same stream, for every image concerned (round 5: no capacity, no "from the next call on"):
 *   - range: an image that met a value beyond +-65504 is counted (gbnf_saturation_count) and marked; ONE repair launch behind
 *     the split-f16 pass (it returns at once when no image of the call is marked) walks every marked image through the
 *     exact-f32 sequence and overwrites its z / ldj / ll (x on the way back): the caller sees what a GBNF_MATH_F32 handle
 *     returns, never NaN and never a clamped value.  Slow per image (one workgroup each), rare by construction.
 *   - precision on the caller's data: on a handle's first forward launch and every `check_every`-th after it (gbnf_tuning_set;
 *     counted on the device, so HIP-graph replays are covered) up to 2 unmarked images are evaluated on exact f32 as well and
 *     compared with the split-f16 result on the device (`check_tolerance_e9`).  A failed check makes the repair launch of that
 *     very call -- and of every later one -- re-evaluate ALL images, and raises a pinned word: later (non-captured) calls run
 *     the exact-f32 kernels directly (math_mode reads GBNF_MATH_F32, demoted = 1).
 * Environment GBNF_IMAGE_REPAIR=0: nothing behind the split-f16 pass (timing only: out-of-range images keep clamped values).
 * `checks` = calls that marked an image so far; gbnf_image_flow_repair_counts has the rest.  Never synchronises. */
int gbnf_image_flow_numerics(const gbnf_image_flow* flow, gbnf_numerics_status* out);
/* Counters of that protocol (pinned host words the device updates; no synchronisation; any pointer may be NULL): calls that
 * marked an image, images re-evaluated on exact f32, on-data checks completed / failed, the worst relative log-likelihood
 * difference a check has seen. */
int gbnf_image_flow_repair_counts(const gbnf_image_flow* flow, int64_t* marked_calls, int64_t* repaired_images,
                                  int64_t* data_checks, int64_t* failed_checks, float* worst_check_rel_err);
/* Replaces (one layer at a time): _ActNorm.initialize_parameters for ActNorm2d (models/layers.py:473-486, 548-557), the
 * data-dependent initialisation the first training-mode forward of an image Glow performs layer by layer.  ActNorm2d number
 * `index` of the component in module order -- per FlowStep its own ActNorm2d, then the one behind each Conv2d of its coupling
 * net (models/glow.py:283-308, models/layers.py:577-606) -- sees a tensor (n, channels, H, W); this call evaluates the component
 * on (x, noise) up to that tensor with the exact-f32 kernels and writes mean_dev[c] = mean over (n, H, W) and var_dev[c] =
 * mean((t - mean)^2) (device, `channels` floats each; *channels is set).  The caller sets bias = -mean,
 * logs = log(scale / (sqrt(var) + 1e-6)), re-creates the handle and moves to the next index: the reference's sequence.  Valid for
 * a layer whose bias and logs are still zero (an un-initialised ActNorm2d is the identity).  index past the last layer:
 * GBNF_ERR_INVALID.  workspace: gbnf_image_flow_workspace_bytes(flow, n). */
int gbnf_image_flow_actnorm_stats(const gbnf_image_flow* flow, const float* x, const float* noise, int64_t n, int32_t index,
                                  float* mean_dev, float* var_dev, int32_t* channels, void* workspace, int64_t workspace_bytes,
                                  void* stream);
/* The z -> x direction.  Replaces: x = self.flows[c].decode(z, None, temperature) for image input (models/glow.py:112-123):
 * FlowNet.decode (models/glow.py:254-260) = per level, last first: Split2d reverse (models/layers.py:695-699: the dropped
 * half is re-drawn as Normal(mean, exp(log-var) * temperature) with (mean, log-var) = conv(z1)), the FlowSteps backwards
 * (FlowStep.decode, models/glow.py:344-366: coupling^-1, torch.inverse of the 1x1 weight or indices_inverse, ActNorm2d
 * reverse), unsqueeze2d (utils/utilities.py:121-135); then to_logits(reverse=True) (models/glow.py:151-158).
 * z (n,Cz,Hz,Wz); x (n,C,H,W).  eps: the standard-normal draws behind Split2d's samples (the caller's RNG, so that
 * sampling is reproducible and testable): level 0 first, each level a contiguous (n, C_l/2, H_l, W_l) array,
 * gbnf_image_flow_eps_floats() floats per image in all (= C*H*W - Cz*Hz*Wz); may be NULL for a one-level flow.
 * The coupling networks run on the fused split-f16 kernel like the forward where the handle does; an image whose hidden
 * activation leaves the fp16 range is re-evaluated on the exact-f32 sequence by the repair launch behind the pass, in this
 * call (see gbnf_image_flow_numerics).  A handle created with GBNF_MATH_F32, demoted by the probe or by an on-data check of
 * the forward direction runs the exact-f32 convolution kernels.  workspace as for gbnf_image_flow_forward. */
int gbnf_image_flow_eps_floats(const gbnf_image_flow* flow, int64_t* per_image);
int gbnf_image_flow_inverse(const gbnf_image_flow* flow, const float* z, const float* eps, float temperature, int64_t n,
                            float* x, void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GBNF_H_ */
