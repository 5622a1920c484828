// TEST INFRASTRUCTURE — a stand-in for librccl that lets the N > 1 branches of libagz's exchange step (agz_examples_allgather,
// agz_trainer_allreduce: agogo_amd/csrc/examples.hip, comm.hip) execute on a box with ONE GPU.
//
// Ranks are PROCESSES that all use the same device.  A collective is: drain the caller's stream, copy the send buffer to a POSIX
// shared-memory object of this rank, barrier, read the peers' objects (rank order) into the receive buffer, barrier, unlink.
// Everything is synchronous and deterministic (sums in rank order); the semantics are NCCL's for exactly the ten entry points
// comm.hip binds (same names, same signatures, in-place and root-copies-send-to-recv behaviour included).  Selected through
// AGZ_RCCL_LIB=<path to this .so> (the dlopen path libagz honours for any RCCL build).  Never part of the product.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {
struct Ctl {
  std::atomic<int> arrive;
  std::atomic<int> gen;
  std::atomic<int> joined;
};
struct FakeComm {
  int n = 1, rank = 0, device = 0;
  std::string name;
  Ctl* ctl = nullptr;
  unsigned long long seq = 0;
};
const double kTimeoutS = 120.0;

bool barrier(FakeComm* c) {
  if (c->n == 1) return true;
  const int g = c->ctl->gen.load();
  if (c->ctl->arrive.fetch_add(1) + 1 == c->n) {
    c->ctl->arrive.store(0);
    c->ctl->gen.fetch_add(1);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (c->ctl->gen.load() == g) {
    sched_yield();
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutS) return false;
  }
  return true;
}
size_t type_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}
std::string obj_name(FakeComm* c, int rank, unsigned long long seq) {
  char b[200];
  snprintf(b, sizeof b, "/%s_r%d_%llu", c->name.c_str(), rank, seq);
  return b;
}
// publish `bytes` of device memory as this rank's object of operation `seq`
void* publish(FakeComm* c, const void* dev, size_t bytes, unsigned long long seq) {
  if (!bytes) return nullptr;
  const std::string nm = obj_name(c, c->rank, seq);
  int fd = shm_open(nm.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0600);
  if (fd < 0) return nullptr;
  if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); return nullptr; }
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return nullptr;
  if (hipMemcpy(p, dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) { munmap(p, bytes); return nullptr; }
  return p;
}
const void* peer(FakeComm* c, int rank, size_t bytes, unsigned long long seq) {
  const std::string nm = obj_name(c, rank, seq);
  int fd = shm_open(nm.c_str(), O_RDONLY, 0600);
  if (fd < 0) return nullptr;
  void* p = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
  close(fd);
  return p == MAP_FAILED ? nullptr : p;
}
void retire(FakeComm* c, void* p, size_t bytes, unsigned long long seq) {
  if (!p) return;
  munmap(p, bytes);
  shm_unlink(obj_name(c, c->rank, seq).c_str());
}
FakeComm* F(ncclComm_t c) { return reinterpret_cast<FakeComm*>(c); }
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof(*id));
  static std::atomic<int> k{0};
  snprintf(id->internal, sizeof(id->internal), "agzfake_%d_%d_%lld", (int)getpid(), k.fetch_add(1),
           (long long)std::chrono::steady_clock::now().time_since_epoch().count());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  id.internal[sizeof(id.internal) - 1] = 0;
  if (!id.internal[0]) return ncclInvalidArgument;
  FakeComm* c = new FakeComm();
  c->n = nranks; c->rank = rank; c->name = id.internal;
  hipGetDevice(&c->device);
  const std::string nm = "/" + c->name + "_ctl";
  int fd = shm_open(nm.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0) { delete c; return ncclSystemError; }
  if (ftruncate(fd, sizeof(Ctl)) != 0) { close(fd); delete c; return ncclSystemError; }
  c->ctl = (Ctl*)mmap(nullptr, sizeof(Ctl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);   // a fresh object is zero-filled
  close(fd);
  if (c->ctl == MAP_FAILED) { delete c; return ncclSystemError; }
  c->ctl->joined.fetch_add(1);
  if (!barrier(c)) { delete c; return ncclSystemError; }                                   // everybody has joined
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
  // one process driving several devices: not what the double is for (ranks = processes); a single rank is the trivial communicator
  if (!comms || ndev != 1) return ncclInvalidUsage;
  FakeComm* c = new FakeComm();
  c->device = devlist ? devlist[0] : 0;
  c->name = "single";
  comms[0] = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  FakeComm* c = F(comm);
  if (!c) return ncclSuccess;
  if (c->ctl) {
    if (c->ctl->joined.fetch_sub(1) == 1) shm_unlink(("/" + c->name + "_ctl").c_str());
    munmap(c->ctl, sizeof(Ctl));
  }
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }   // every rank issues the group's collectives in the same order: run them as they come
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t s) {
  FakeComm* c = F(comm);
  const size_t bytes = count * type_size(type);
  if (!c || !type_size(type)) return ncclInvalidArgument;
  hipSetDevice(c->device);
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  const unsigned long long seq = ++c->seq;
  void* mine = publish(c, send, bytes, seq);
  if (bytes && !mine) return ncclSystemError;
  if (!barrier(c)) return ncclSystemError;
  ncclResult_t rc = ncclSuccess;
  for (int r = 0; r < c->n && bytes; r++) {
    const void* p = r == c->rank ? mine : peer(c, r, bytes, seq);
    if (!p || hipMemcpy((char*)recv + (size_t)r * bytes, p, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclSystemError;
    if (p && r != c->rank) munmap(const_cast<void*>(p), bytes);
  }
  if (!barrier(c)) rc = ncclSystemError;
  retire(c, mine, bytes, seq);
  return rc;
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm, hipStream_t s) {
  FakeComm* c = F(comm);
  const size_t bytes = count * type_size(type);
  if (!c || !type_size(type) || root < 0 || root >= c->n) return ncclInvalidArgument;
  hipSetDevice(c->device);
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  const unsigned long long seq = ++c->seq;
  void* mine = c->rank == root ? publish(c, send, bytes, seq) : nullptr;
  if (c->rank == root && bytes && !mine) return ncclSystemError;
  if (!barrier(c)) return ncclSystemError;
  ncclResult_t rc = ncclSuccess;
  if (bytes) {
    if (c->rank == root) {
      if (send != recv && hipMemcpy(recv, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) rc = ncclSystemError;
    } else {
      const void* p = peer(c, root, bytes, seq);
      if (!p || hipMemcpy(recv, p, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclSystemError;
      if (p) munmap(const_cast<void*>(p), bytes);
    }
  }
  if (!barrier(c)) rc = ncclSystemError;
  if (c->rank == root) retire(c, mine, bytes, seq);
  return rc;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t s) {
  FakeComm* c = F(comm);
  const size_t bytes = count * type_size(type);
  if (!c || op != ncclSum || !(type == ncclFloat32 || type == ncclInt32 || type == ncclUint64)) return ncclInvalidArgument;
  hipSetDevice(c->device);
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  const unsigned long long seq = ++c->seq;
  void* mine = publish(c, send, bytes, seq);
  if (bytes && !mine) return ncclSystemError;
  if (!barrier(c)) return ncclSystemError;
  ncclResult_t rc = ncclSuccess;
  if (bytes) {
    std::vector<char> acc(bytes, 0);
    for (int r = 0; r < c->n; r++) {            // rank order: deterministic, the same bits on every rank
      const void* p = r == c->rank ? mine : peer(c, r, bytes, seq);
      if (!p) { rc = ncclSystemError; break; }
      if (type == ncclFloat32) { float* a = (float*)acc.data(); const float* b = (const float*)p; for (size_t i = 0; i < count; i++) a[i] = r == 0 ? b[i] : a[i] + b[i]; }
      else if (type == ncclInt32) { int* a = (int*)acc.data(); const int* b = (const int*)p; for (size_t i = 0; i < count; i++) a[i] += b[i]; }
      else { unsigned long long* a = (unsigned long long*)acc.data(); const unsigned long long* b = (const unsigned long long*)p; for (size_t i = 0; i < count; i++) a[i] += b[i]; }
      if (r != c->rank) munmap(const_cast<void*>(p), bytes);
    }
    if (rc == ncclSuccess && hipMemcpy(recv, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclSystemError;
  }
  if (!barrier(c)) rc = ncclSystemError;
  retire(c, mine, bytes, seq);
  return rc;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake rccl: HIP error";
    case ncclSystemError: return "fake rccl: system error (shared memory / barrier timeout)";
    case ncclInvalidArgument: return "fake rccl: invalid argument";
    case ncclInvalidUsage: return "fake rccl: invalid usage";
    default: return "fake rccl: error";
  }
}

}  // extern "C"
