// agz_comm: RCCL communicators over the devices of agz_ctx handles (one host thread + ctx per GPU; games are
// sharded with no data-path collective, the only exchange is the example gather before dual.Train and the gradient
// all-reduce of the data-parallel step — agogo.go:118-133, dualnet/meta.go:16-54).
//
// xGMI is point-to-point (7 links x ~153 GB/s per GPU): the gather is issued as one grouped set of n broadcasts (every rank
// the root of its own rows), which RCCL runs over all links concurrently instead of a 7-step ring of padded blocks.
#include "comm.hpp"

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>

#include <mutex>

namespace agz {
const Rccl* rccl() {
  static Rccl table{};
  static bool ok = false;
  static std::once_flag once;
  static std::string err;
  std::call_once(once, [] {
    // AGZ_RCCL_LIB=<path>: the RCCL build to bind (a site-specific build; tests/fake_rccl's process-per-rank double on one-GPU
    // boxes).  Otherwise a copy already mapped into the process (e.g. PyTorch's) is reused: two RCCL instances on one device
    // fight over IPC handles.
    void* h = nullptr;
    if (const char* path = getenv("AGZ_RCCL_LIB")) {
      h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
      if (!h) { const char* de = dlerror(); err = std::string("AGZ_RCCL_LIB=") + path + ": " + (de ? de : "?"); return; }   // (dlerror() clears on read: once)
    }
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) { const char* de = dlerror(); err = std::string("librccl not found: ") + (de ? de : "?"); return; }
    bool all = true;
    auto get = [&](const char* name) { void* p = dlsym(h, name); if (!p) { all = false; err = std::string("librccl lacks ") + name; } return p; };
    table.GetUniqueId = (decltype(table.GetUniqueId))get("ncclGetUniqueId");
    table.CommInitRank = (decltype(table.CommInitRank))get("ncclCommInitRank");
    table.CommInitAll = (decltype(table.CommInitAll))get("ncclCommInitAll");
    table.CommDestroy = (decltype(table.CommDestroy))get("ncclCommDestroy");
    table.AllGather = (decltype(table.AllGather))get("ncclAllGather");
    table.AllReduce = (decltype(table.AllReduce))get("ncclAllReduce");
    table.Broadcast = (decltype(table.Broadcast))get("ncclBroadcast");
    table.GroupStart = (decltype(table.GroupStart))get("ncclGroupStart");
    table.GroupEnd = (decltype(table.GroupEnd))get("ncclGroupEnd");
    table.GetErrorString = (decltype(table.GetErrorString))get("ncclGetErrorString");
    ok = all;
  });
  if (!ok) { set_error("agz_comm: %s (RCCL is required for multi-GPU exchange; there is no fallback)", err.c_str()); return nullptr; }
  return &table;
}
}  // namespace agz

using namespace agz;

extern "C" {

int agz_comm_unique_id(void* id128) {
  AGZ_REQUIRE(id128, AGZ_E_INVALID, "agz_comm_unique_id: NULL argument");
  const Rccl* R = rccl();
  if (!R) return AGZ_E_UNSUPPORTED;
  static_assert(sizeof(ncclUniqueId) == 128, "AGZ_COMM_ID_BYTES");
  ncclUniqueId id;
  AGZ_NCCL_TRY(R->GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return AGZ_OK;
}

int agz_comm_init_rank(agz_ctx* ctx, int n_ranks, int rank, const void* id128, agz_comm** out) {
  AGZ_REQUIRE(ctx && id128 && out, AGZ_E_INVALID, "agz_comm_init_rank: NULL argument");
  AGZ_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, AGZ_E_INVALID, "agz_comm_init_rank: rank %d of %d", rank, n_ranks);
  const Rccl* R = rccl();
  if (!R) return AGZ_E_UNSUPPORTED;
  AGZ_HIP_TRY(hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  agz_comm* c = new agz_comm();
  c->ctx = ctx; c->rank = rank; c->size = n_ranks;
  ncclResult_t r = R->CommInitRank(&c->comm, n_ranks, id, rank);
  if (r != ncclSuccess) { set_error("agz_comm_init_rank: ncclCommInitRank -> %s", R->GetErrorString(r)); delete c; return AGZ_E_HIP; }
  if (hipMalloc(&c->d_words, (size_t)(n_ranks + 3) * 8) != hipSuccess) {
    set_error("agz_comm_init_rank: out of device memory for the exchange words");
    R->CommDestroy(c->comm); delete c; return AGZ_E_NOMEM;
  }
  *out = c;
  return AGZ_OK;
}

int agz_comm_init_all(agz_ctx* const* ctxs, int n, agz_comm** comms) {
  AGZ_REQUIRE(ctxs && comms && n >= 1 && n <= 64, AGZ_E_INVALID, "agz_comm_init_all: bad argument");
  const Rccl* R = rccl();
  if (!R) return AGZ_E_UNSUPPORTED;
  std::vector<int> dev(n);
  for (int i = 0; i < n; i++) {
    AGZ_REQUIRE(ctxs[i], AGZ_E_INVALID, "agz_comm_init_all: ctxs[%d] is NULL", i);
    dev[i] = ctxs[i]->device;
    for (int j = 0; j < i; j++) AGZ_REQUIRE(dev[j] != dev[i], AGZ_E_INVALID, "agz_comm_init_all: device %d appears twice (one rank per GPU)", dev[i]);
  }
  std::vector<ncclComm_t> cs(n, nullptr);
  AGZ_NCCL_TRY(R->CommInitAll(cs.data(), n, dev.data()));
  for (int i = 0; i < n; i++) comms[i] = nullptr;
  int rc = AGZ_OK;
  for (int i = 0; i < n; i++) {
    agz_comm* c = new agz_comm();
    c->ctx = ctxs[i]; c->comm = cs[i]; c->rank = i; c->size = n;
    comms[i] = c;
    if (hipSetDevice(dev[i]) != hipSuccess || hipMalloc(&c->d_words, (size_t)(n + 3) * 8) != hipSuccess) rc = AGZ_E_NOMEM;
  }
  if (rc != AGZ_OK) {
    set_error("agz_comm_init_all: out of device memory for the exchange words");
    for (int i = 0; i < n; i++) { agz_comm_destroy(comms[i]); comms[i] = nullptr; }
  }
  return rc;
}

void agz_comm_destroy(agz_comm* c) {
  if (!c) return;
  const Rccl* R = rccl();
  if (R && c->comm) { hipSetDevice(c->ctx->device); hipStreamSynchronize(c->ctx->stream); R->CommDestroy(c->comm); }
  if (c->d_words) { hipSetDevice(c->ctx->device); hipFree(c->d_words); }
  if (c->ar_stream) { hipSetDevice(c->ctx->device); hipStreamSynchronize(c->ar_stream); hipStreamDestroy(c->ar_stream); hipEventDestroy(c->ev_ready); hipEventDestroy(c->ev_done); if (c->h_status) hipHostFree(c->h_status); }
  delete c;
}

int agz_comm_rank(const agz_comm* c) { return c ? c->rank : -1; }
int agz_comm_size(const agz_comm* c) { return c ? c->size : 0; }

int agz_trainer_allreduce(agz_comm* c, agz_trainer* t) {
  AGZ_REQUIRE(c && t, AGZ_E_INVALID, "agz_trainer_allreduce: NULL argument");
  // the gradients are produced on the trainer's ctx stream and reduced on the communicator's: they must be the same queue
  AGZ_REQUIRE(agz_trainer_ctx(t) == c->ctx, AGZ_E_INVALID, "agz_trainer_allreduce: the communicator and the trainer belong to different contexts");
  const Rccl* R = rccl();
  if (!R) return AGZ_E_UNSUPPORTED;
  AGZ_HIP_TRY(hipSetDevice(c->ctx->device));
  float* g = nullptr;
  size_t n = 0;
  int r = agz_trainer_grads_dev(t, &g, &n);
  if (r != AGZ_OK) return r;
  // ONE collective per step: all learnables' gradients live in one flat buffer (train.hip); in place, on the ctx stream
  AGZ_NCCL_TRY(R->AllReduce(g, g, n, ncclFloat32, ncclSum, c->comm, c->ctx->stream));
  return AGZ_OK;
}

// The data-parallel step with the reduction UNDER the backward pass.  The flat gradient buffer of the G19 trainer is 7.7 GB (98 % the
// reference's batch-shaped gamma / beta, dual.go:105-132): as one call after the backward (agz_trainer_allreduce) that is ~40 ms of
// xGMI time on eight GPUs next to a 49 ms compute step.  Here every slice — the heads, then layer L .. 0 as the backward pass
// finishes them — is handed to RCCL on the communicator's own queue the moment its last writer is enqueued: the first slices move
// while the rest of the backward runs; only layer 0's 0.19 GB is left exposed.  Slices tile the buffer exactly and a sum over ranks
// does not depend on how the buffer is cut: the result is that of agz_trainer_allreduce (bit for bit wherever RCCL's own result is).
static int dp_begin(agz_comm* c, agz_trainer* t, const Rccl* R) {
  if (!c->ar_stream) {
    AGZ_HIP_TRY(hipStreamCreateWithFlags(&c->ar_stream, hipStreamNonBlocking));
    AGZ_HIP_TRY(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
    AGZ_HIP_TRY(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    AGZ_HIP_TRY(hipHostMalloc((void**)&c->h_status, 16, hipHostMallocDefault));
  }
  float* g = nullptr;
  size_t n_all = 0;
  int r = agz_trainer_grads_dev(t, &g, &n_all);
  if (r != AGZ_OK) return r;
  agz_trainer_slices(t, c->dp_slices);
  c->dp_issued = 0;
  agz_trainer_set_slice_hook(t, [c, g, n_all, R](size_t off, size_t n, hipStream_t ready) -> int {
    AGZ_REQUIRE(off + n <= n_all, AGZ_E_STATE, "data-parallel step: slice [%zu, %zu) outside the gradient buffer", off, off + n);
    AGZ_REQUIRE(c->dp_issued < c->dp_slices.size() && c->dp_slices[c->dp_issued] == std::make_pair(off, n), AGZ_E_STATE,
                "data-parallel step: slice %zu is not the one the slice table announces", c->dp_issued);
    if (c->debug_fail_slice == (int)c->dp_issued) { c->debug_fail_slice = -1; set_error("injected failure before slice %zu (agz_comm_debug_fail_slice)", c->dp_issued); return AGZ_E_STATE; }
    AGZ_HIP_TRY(hipEventRecord(c->ev_ready, ready));
    AGZ_HIP_TRY(hipStreamWaitEvent(c->ar_stream, c->ev_ready, 0));
    AGZ_NCCL_TRY(R->AllReduce(g + off, g + off, n, ncclFloat32, ncclSum, c->comm, c->ar_stream));
    c->dp_issued++;
    return AGZ_OK;
  });
  return AGZ_OK;
}
// End of the step on every rank, whatever happened on this one: a rank that failed part-way (an allocation, a launch, a
// collective) still ENTERS the slices it has not issued — its peers are inside them or about to be, and a collective one rank never
// enters is a hang — and then all ranks exchange one status word (the number of ranks that failed).  A non-zero word makes the call fail on EVERY
// rank (AGZ_E_PEER where this rank itself was fine): the gradients of such a step are undefined everywhere and must not be applied.
// A failure of RCCL itself (a collective that returns an error here) is fatal for the process group: tear the communicator down.
static int dp_end(agz_comm* c, agz_trainer* t, const Rccl* R, int r) {
  agz_trainer_set_slice_hook(t, nullptr);
  float* g = nullptr;
  size_t n_all = 0;
  const bool have_g = agz_trainer_grads_dev(t, &g, &n_all) == AGZ_OK;
  std::string first_error = r != AGZ_OK ? agz_last_error() : "";
  if (have_g)
    for (; c->dp_issued < c->dp_slices.size(); c->dp_issued++) {   // only ever non-empty after a local failure
      const auto& sl = c->dp_slices[c->dp_issued];
      if (R->AllReduce(g + sl.first, g + sl.first, sl.second, ncclFloat32, ncclSum, c->comm, c->ar_stream) != ncclSuccess && r == AGZ_OK) r = AGZ_E_HIP;
    }
  c->h_status[0] = r != AGZ_OK ? 1ull : 0ull;
  c->h_status[1] = 0;
  unsigned long long* dw = c->d_words + c->size + 2;   // the last exchange word of the communicator (allocated with it)
  bool ok = hipMemcpyAsync(dw, &c->h_status[0], 8, hipMemcpyHostToDevice, c->ar_stream) == hipSuccess &&
            R->AllReduce(dw, dw, 1, ncclUint64, ncclSum, c->comm, c->ar_stream) == ncclSuccess &&
            hipMemcpyAsync(&c->h_status[1], dw, 8, hipMemcpyDeviceToHost, c->ar_stream) == hipSuccess &&
            hipStreamSynchronize(c->ar_stream) == hipSuccess;
  // the step's stream carries everything again: whatever comes next (agz_trainer_apply) sees the summed gradients
  ok = ok && hipEventRecord(c->ev_done, c->ar_stream) == hipSuccess && hipStreamWaitEvent(c->ctx->stream, c->ev_done, 0) == hipSuccess;
  if (r != AGZ_OK) { set_error("data-parallel step failed on this rank (its peers are told; gradients undefined): %s", first_error.c_str()); return r; }
  if (!ok) { set_error("data-parallel step: the status exchange / joining the reduction queue failed (fatal for the process group)"); return AGZ_E_HIP; }
  if (c->h_status[1] != 0) { set_error("data-parallel step: another rank failed in this step; the gradients are undefined and must not be applied"); return AGZ_E_PEER; }
  return AGZ_OK;
}

int agz_comm_debug_fail_slice(agz_comm* c, int k) {
  AGZ_REQUIRE(c, AGZ_E_INVALID, "agz_comm_debug_fail_slice: NULL communicator");
  c->debug_fail_slice = k;
  return AGZ_OK;
}

int agz_trainer_forward_backward_allreduce(agz_comm* c, agz_trainer* t, const float* planes, const float* pi, const float* v, float* cost) {
  AGZ_REQUIRE(c && t && planes && pi && v, AGZ_E_INVALID, "agz_trainer_forward_backward_allreduce: NULL argument");
  AGZ_REQUIRE(agz_trainer_ctx(t) == c->ctx, AGZ_E_INVALID, "agz_trainer_forward_backward_allreduce: the communicator and the trainer belong to different contexts");
  const Rccl* R = rccl();
  if (!R) return AGZ_E_UNSUPPORTED;
  AGZ_HIP_TRY(hipSetDevice(c->ctx->device));
  int r = dp_begin(c, t, R);
  if (r != AGZ_OK) return r;
  return dp_end(c, t, R, agz_trainer_forward_backward(t, planes, pi, v, cost));
}

int agz_trainer_forward_backward_allreduce_dev(agz_comm* c, agz_trainer* t, const float* planes_dev, const float* pi_dev, const float* v_dev, float* cost) {
  AGZ_REQUIRE(c && t && planes_dev && pi_dev && v_dev, AGZ_E_INVALID, "agz_trainer_forward_backward_allreduce_dev: NULL argument");
  AGZ_REQUIRE(agz_trainer_ctx(t) == c->ctx, AGZ_E_INVALID, "agz_trainer_forward_backward_allreduce_dev: the communicator and the trainer belong to different contexts");
  const Rccl* R = rccl();
  if (!R) return AGZ_E_UNSUPPORTED;
  AGZ_HIP_TRY(hipSetDevice(c->ctx->device));
  int r = dp_begin(c, t, R);
  if (r != AGZ_OK) return r;
  return dp_end(c, t, R, agz_trainer_forward_backward_dev(t, planes_dev, pi_dev, v_dev, cost));
}

}  // extern "C"
