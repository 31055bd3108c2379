// gbnf_comm.hip -- the multi-GPU group of the component-sharded mixture, entirely inside the library (round 4):
//
//   flow launch of this rank's components over S batches  ->  ncclAllGather of the (C/W, S n) table  ->  recursion launch
//
// on ONE stream, optionally captured once into a HIP graph and replayed per group.  The reference evaluates the components in a
// serial loop on one device (density_experiment.py:562-571); BASELINE.json's north star shards them one per GPU with an RCCL
// all-gather of log p_c(x) before the mixture log-sum-exp.  Rounds 1-3 drove the exchange from Python through
// torch.distributed: two ctypes calls + one collective call + stream bookkeeping per group, ~50 us of host time and ~45 us from
// the end of the kernel to the end of the recursion -- as long as the per-rank kernel itself at the driver's `--steps 20`
// (profiles/r3_emulated_steps20_timeline.txt).  Here RCCL is bound directly (dlopen of librccl.so: ncclGetUniqueId,
// ncclCommInitRank, ncclAllGather; the unique id travels through the caller's rendezvous, e.g. the torch.distributed store) and a
// group costs the host ONE call -- or one hipGraphLaunch.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/gbnf.h"
#include "gbnf_internal.h"

namespace {

// the few RCCL entry points used, resolved at first use (the library stays loadable on a box without RCCL: the single-GPU path
// never touches this file)
struct ncclUniqueId_ { char internal[128]; };
typedef void* ncclComm_t_;
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(ncclUniqueId_*) = nullptr;
  int (*CommInitRank)(ncclComm_t_*, int, ncclUniqueId_, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t_, hipStream_t) = nullptr;
  int (*CommDestroy)(ncclComm_t_) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
    r.GetUniqueId = (int (*)(ncclUniqueId_*))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(ncclComm_t_*, int, ncclUniqueId_, int))dlsym(r.lib, "ncclCommInitRank");
    r.AllGather = (int (*)(const void*, void*, size_t, int, ncclComm_t_, hipStream_t))dlsym(r.lib, "ncclAllGather");
    r.CommDestroy = (int (*)(ncclComm_t_))dlsym(r.lib, "ncclCommDestroy");
    r.GetErrorString = (const char* (*)(int))dlsym(r.lib, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy;
  });
  return r;
}
const char* rccl_error(int code) {
  Rccl& r = rccl();
  return r.GetErrorString ? r.GetErrorString(code) : "?";
}
constexpr int NCCL_FLOAT32 = 7;      // ncclFloat32 (rccl.h)

}  // namespace

struct gbnf_comm {
  ncclComm_t_ comm = nullptr;
  int rank = 0, world = 1;
};

struct gbnf_group_graph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

using gbnf::fail;

extern "C" {

int gbnf_comm_unique_id(uint8_t* id128) {
  if (!id128) return fail(GBNF_ERR_INVALID, "gbnf_comm_unique_id: null buffer");
  Rccl& r = rccl();
  if (!r.ok) return fail(GBNF_ERR_UNSUPPORTED, "gbnf_comm_unique_id: librccl.so could not be loaded");
  ncclUniqueId_ id;
  const int rc = r.GetUniqueId(&id);
  if (rc != 0) return fail(GBNF_ERR_HIP, "ncclGetUniqueId: %s", rccl_error(rc));
  std::memcpy(id128, id.internal, 128);
  return GBNF_OK;
}

int gbnf_comm_create(const uint8_t* id128, int32_t rank, int32_t world, gbnf_comm** out) {
  if (!out) return fail(GBNF_ERR_INVALID, "gbnf_comm_create: out is null");
  *out = nullptr;
  if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(GBNF_ERR_INVALID, "gbnf_comm_create: bad rank %d / world %d", rank, world);
  Rccl& r = rccl();
  if (!r.ok) return fail(GBNF_ERR_UNSUPPORTED, "gbnf_comm_create: librccl.so could not be loaded");
  ncclUniqueId_ id;
  std::memcpy(id.internal, id128, 128);
  auto* c = new gbnf_comm();
  c->rank = rank; c->world = world;
  const int rc = r.CommInitRank(&c->comm, world, id, rank);        // collective: every rank of the group calls it
  if (rc != 0) {
    delete c;
    return fail(GBNF_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, rccl_error(rc));
  }
  *out = c;
  return GBNF_OK;
}

int gbnf_comm_destroy(gbnf_comm* c) {
  if (!c) return GBNF_OK;
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  delete c;
  return GBNF_OK;
}

int gbnf_comm_info(const gbnf_comm* c, int32_t* rank, int32_t* world) {
  if (!c) return fail(GBNF_ERR_INVALID, "gbnf_comm_info: comm is null");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return GBNF_OK;
}

int gbnf_mixture_group_log_prob(const gbnf_mixture* mix, gbnf_comm* comm, const float* const* xs, int32_t n_batches, int64_t n,
                                int32_t n_components, const float* rho_dev, float* ll_local, float* ll_full, float* G, void* stream) {
  if (!mix || !xs || !rho_dev || !ll_local || !G) return fail(GBNF_ERR_INVALID, "gbnf_mixture_group_log_prob: null argument");
  const int world = comm ? comm->world : 1;
  if (n_components < 1 || n_components % world) return fail(GBNF_ERR_INVALID, "%d components do not split over %d ranks", n_components, world);
  const int c_local = n_components / world;
  if (comm && !ll_full) return fail(GBNF_ERR_INVALID, "gbnf_mixture_group_log_prob: ll_full is null");
  const int64_t cols = (int64_t)n_batches * n;
  // this rank's components are ITS mixture's components [0, c_local) (the caller builds the mixture from its own block)
  int rc = gbnf_mixture_component_log_prob_multi(mix, xs, n_batches, n, 0, c_local, ll_local, cols, stream);
  if (rc) return rc;
  const float* table = ll_local;
  if (comm) {
    // rank r's (c_local, cols) block lands at rows [r c_local, (r + 1) c_local): the table in component order on every rank
    const int nrc = rccl().AllGather(ll_local, ll_full, (size_t)c_local * (size_t)cols, NCCL_FLOAT32, comm->comm, (hipStream_t)stream);
    if (nrc != 0) return fail(GBNF_ERR_HIP, "ncclAllGather: %s", rccl_error(nrc));
    table = ll_full;
  }
  return gbnf_mixture_lse(table, cols, rho_dev, n_components, cols, G, stream);
}

int gbnf_group_graph_create(const gbnf_mixture* mix, gbnf_comm* comm, const float* const* xs, int32_t n_batches, int64_t n,
                            int32_t n_components, const float* rho_dev, float* ll_local, float* ll_full, float* G,
                            gbnf_group_graph** out) {
  if (!out) return fail(GBNF_ERR_INVALID, "gbnf_group_graph_create: out is null");
  *out = nullptr;
  hipStream_t cap = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&cap, hipStreamNonBlocking);
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "gbnf_group_graph_create: %s", hipGetErrorString(e));
  // once outside the capture: first-use work of the launch path and of the communicator (lazy allocations, the numerics guard's
  // first-launch check) must not be baked into the graph -- or fail inside it
  int rc = gbnf_mixture_group_log_prob(mix, comm, xs, n_batches, n, n_components, rho_dev, ll_local, ll_full, G, cap);
  if (rc == GBNF_OK && hipStreamSynchronize(cap) != hipSuccess) rc = fail(GBNF_ERR_HIP, "gbnf_group_graph_create: warm-up failed");
  auto* g = new gbnf_group_graph();
  if (rc == GBNF_OK) {
    e = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) rc = fail(GBNF_ERR_HIP, "hipStreamBeginCapture: %s", hipGetErrorString(e));
  }
  if (rc == GBNF_OK) {
    rc = gbnf_mixture_group_log_prob(mix, comm, xs, n_batches, n, n_components, rho_dev, ll_local, ll_full, G, cap);
    e = hipStreamEndCapture(cap, &g->graph);            // (always: leaves the stream out of capture mode)
    if (rc == GBNF_OK && e != hipSuccess) rc = fail(GBNF_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
  }
  if (rc == GBNF_OK) {
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) rc = fail(GBNF_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
  }
  (void)hipStreamDestroy(cap);
  if (rc != GBNF_OK) {
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return rc;
  }
  *out = g;
  return GBNF_OK;
}

int gbnf_group_graph_launch(gbnf_group_graph* g, void* stream) {
  if (!g || !g->exec) return fail(GBNF_ERR_INVALID, "gbnf_group_graph_launch: graph is null");
  const hipError_t e = hipGraphLaunch(g->exec, (hipStream_t)stream);
  if (e != hipSuccess) return fail(GBNF_ERR_HIP, "hipGraphLaunch: %s", hipGetErrorString(e));
  return GBNF_OK;
}

int gbnf_group_graph_destroy(gbnf_group_graph* g) {
  if (!g) return GBNF_OK;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
  return GBNF_OK;
}

}  // extern "C"
