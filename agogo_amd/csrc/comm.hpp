// RCCL inside libagz: the one exchange step of the path — the per-GPU example buffers gathered before
// dual.Train (agogo.go:118-133) and the gradient all-reduce of the data-parallel training step.
//
// librccl is resolved at run time (dlopen + dlsym) the first time a communicator is made: libagz keeps loading where RCCL is
// absent and never clashes with another RCCL copy the host process may already hold (PyTorch ships its own); a missing
// library is a loud AGZ_E_UNSUPPORTED, never a silent single-GPU fallback.
#pragma once
#include <rccl/rccl.h>

#include <functional>
#include <utility>
#include <vector>

#include "common.hpp"

namespace agz {
struct Rccl {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*GroupStart)();
  ncclResult_t (*GroupEnd)();
  const char* (*GetErrorString)(ncclResult_t);
};
// nullptr (with agz_last_error set) when librccl cannot be loaded
const Rccl* rccl();
}  // namespace agz

// (train.hip) the context a trainer was created on; the per-slice hook of a data-parallel step (agz_trainer::on_slice)
agz_ctx* agz_trainer_ctx(const agz_trainer* t);
void agz_trainer_set_slice_hook(agz_trainer* t, std::function<int(size_t off, size_t n, hipStream_t ready)> f);
void agz_trainer_slices(const agz_trainer* t, std::vector<std::pair<size_t, size_t>>& out);   // (offset, count) in issue order

struct agz_comm {
  agz_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, size = 1;
  // size + 3 device words allocated with the communicator: the count exchange and the allocation agreement of
  // agz_examples_allgather never allocate, so no rank can skip a collective its peers enter (a hang) for want of memory
  unsigned long long* d_words = nullptr;
  // the gradient slices of a data-parallel step are reduced on their own queue while the backward pass goes on (comm.hip)
  hipStream_t ar_stream = nullptr;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  // one data-parallel step: the slice table of the trainer, how many of its collectives this rank has entered, and the status word
  // every rank exchanges at the end of the step (dp_end) so that a failure on ONE rank is an error on ALL of them
  std::vector<std::pair<size_t, size_t>> dp_slices;
  size_t dp_issued = 0;
  unsigned long long* h_status = nullptr;   // pinned
  int debug_fail_slice = -1;                // agz_comm_debug_fail_slice (agz_debug.h)
};

#define AGZ_NCCL_TRY(expr)                                                                                   \
  do {                                                                                                       \
    ncclResult_t _r = (expr);                                                                                \
    if (_r != ncclSuccess) {                                                                                 \
      agz::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, agz::rccl() ? agz::rccl()->GetErrorString(_r) : "rccl error"); \
      return AGZ_E_HIP;                                                                                      \
    }                                                                                                        \
  } while (0)
