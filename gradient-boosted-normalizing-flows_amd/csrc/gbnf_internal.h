// gbnf_internal.h -- shared between the translation units of libgbnf_hip.so (not part of the C ABI).
#pragma once
#include <cstdint>
#include "../../include/gbnf.h"

namespace gbnf {

// Records the message behind gbnf_last_error() (thread-local) and returns `code`.
int fail(int code, const char* fmt, ...);

// The CURRENT device's counter of waves that stored a split-f16 operand beyond +-65504 (it saturates there); null if it
// could not be allocated.  Call it at launch time, with the launch's device current.  Read and reset by
// gbnf_saturation_count().
unsigned* saturation_counter();
// The same device's counter of the TRAINING kernels (word [1] of the same allocation): the traced forward sweep, the backward sweep, the
// per-step training kernels and the trainer's weight re-pack count here -- they saturate and have no repair pass, so THIS count means wrong
// gradients, while word [0] (evaluation: every marked sample is re-evaluated in the same call) is a data-quality alarm.  gbnf_saturation_count()
// reports their sum, gbnf_training_saturation_count() this word alone.
unsigned* training_saturation_counter();

// Kernel-variant key only (not a descriptor value): the activation differs between the steps / nets of a component
// (`--coupling_network random` in the reference); the kernel reads it per step and net from the step header.
constexpr int GBNF_ACT_PER_STEP = 3;

// ---- the training path's forward sweep on the evaluation kernels (gbnf_api.hip; used by gbnf_train.hip) -------------
// A "live blob": the packed hx3 (f16x3) parameter blob of ONE component whose parameters live in device tensors that an
// optimiser updates in place.  `live_blob_create` packs everything that does not depend on parameter VALUES once on the
// host (slot maps, activation flags, layout) and records where every value-dependent word comes from;
// `live_blob_forward` re-derives those words on the device (one gather + split kernel, ~10 us) and launches the TRAIN
// instantiation of flow_kernel_hx3: x -> z, ldj + the trace and operand saves the backward pass needs.
struct LiveBlob;
// desc: a component descriptor whose parameter pointers are DEVICE pointers (perm_indices: host).  GBNF_ERR_UNSUPPORTED
// (and *out = nullptr) when no TRAIN kernel variant covers the geometry: the caller keeps its own forward kernel.
// norm_grad_offsets: [K][2] float offsets of every step's two normalisation-parameter gradients in the caller's flat gradient
// buffer (null: no backward sweep wanted)
int live_blob_create(const gbnf_flow_desc* desc, const int64_t* norm_grad_offsets, LiveBlob** out);
// Hidden rows (16 x hidden tiles) of the TRAIN variant a trainer of this flow would run, 0 if none
int live_blob_train_rows(const gbnf_flow_desc* desc);
bool live_blob_has_backward(const LiveBlob* lb);
int live_blob_hidden_rows(const LiveBlob* lb);      // 16 x the hidden tiles of the kernel variant behind it
// The backward kernel leaves the ActNorm / BatchNorm parameter gradients as per-workgroup partial sums; adding them up (in a
// fixed order) is a reduction of n_wg rows per (step, parameter array).  With `reduce_out` the caller takes that over (the
// trainer folds it into wgrad_kernel's launch as extra blocks: one launch less per step); with null it is launched here.
struct LiveReduce {
  const float* partials;      // [n_wg][K][2][64]
  int n_wg, K, d;
  const int64_t* goff;        // [K][2] float offsets into the flat gradient buffer
  unsigned skip_steps;        // bit k: step k's two sums are NOT added (they were, by an earlier launch: batch-statistics BatchNorm)
};
// One step RANGE of a training sweep (round 4: BatchNorm on batch statistics; FlowLaunch::k_begin ..).  Forward: state_in / state_out =
// the state parked in slot layout [d][np] (null: x rows / the flow's z, ldj outputs); bmean / bvar: device (d,) batch statistics
// of step k_begin's BatchNorm, or null (running statistics: what the re-pack derived); repack: re-derive the blob from the live
// parameters first (the first range of a sweep).  Backward: state_in / state_out = the scaled gradient state behind step
// k_end - 1 / in front of step k_begin (null: g_z rows / g_x rows).
struct LiveRange {
  int k_begin, k_end;
  const float* state_in;
  float* state_out;
  int ldj_accumulate;
  bool repack;
  const float* bmean;
  const float* bvar;
};
int live_blob_backward(LiveBlob* lb, int64_t n, const float* trace, float* acts, int64_t np, int ip, int hp, int op,
                       const float* g_z, const float* g_ldj, float* g_x, float* grads, const unsigned* gmax, void* stream,
                       LiveReduce* reduce_out = nullptr, const LiveRange* range = nullptr);
void live_blob_destroy(LiveBlob* lb);
// trace: [K][d][np] normalised states (slot layout); acts: the operand workspace (FlowLaunch::acts_out); np: padded rows
int live_blob_forward(LiveBlob* lb, const float* x, int64_t n, float* z, float* ldj, float* trace, float* acts, int64_t np,
                      int ip, int hp, int op, void* stream, const LiveRange* range = nullptr);
// (tests) the blob as the device packer left it / size in words
int live_blob_words(const LiveBlob* lb, uint32_t* out_host, int64_t* n_words);

}  // namespace gbnf
