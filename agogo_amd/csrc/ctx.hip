// agz_ctx: one HIP device + stream, thread-local error string, kernel-class event timers.
#include "common.hpp"

namespace agz {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }
}  // namespace agz

void agz_ctx::prof_begin(int klass) {
  agz::ProfClass& p = prof[klass];
  if (p.used == p.pairs.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    p.pairs.push_back({a, b});
  }
  hipEventRecord(p.pairs[p.used].first, stream);
  prof_open++;
}
void agz_ctx::prof_end(int klass) {
  agz::ProfClass& p = prof[klass];
  if (p.used >= p.pairs.size()) return;
  hipEventRecord(p.pairs[p.used].second, stream);
  p.used++;
  if (prof_open > 0) prof_open--;
  if (p.used >= 4096 && prof_open == 0) prof_collect();  // bound the number of live events (never while an outer scope is open)
}
int agz_ctx::prof_collect() {
  AGZ_HIP_TRY(hipStreamSynchronize(stream));
  for (int k = 0; k < AGZ_PROF_NCLASS; k++) {
    agz::ProfClass& p = prof[k];
    for (size_t i = 0; i < p.used; i++) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, p.pairs[i].first, p.pairs[i].second) == hipSuccess) {
        p.total_ms += ms;
        p.launches++;
      }
    }
    p.used = 0;
  }
  return AGZ_OK;
}

extern "C" {

const char* agz_last_error(void) { return agz::get_error(); }
const char* agz_version(void) { return "libagz 0.1 (gfx950, HIP, fp32 MFMA conv tower, device MCTS)"; }

int agz_ctx_create(int device, agz_ctx** out) {
  AGZ_REQUIRE(out != nullptr, AGZ_E_INVALID, "agz_ctx_create: out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    agz::set_error("agz_ctx_create: no HIP device available (%s) — libagz has no CPU fallback",
                   e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return AGZ_E_HIP;
  }
  AGZ_REQUIRE(device >= 0 && device < n, AGZ_E_INVALID, "agz_ctx_create: device %d out of range [0,%d)", device, n);
  AGZ_HIP_TRY(hipSetDevice(device));
  agz_ctx* c = new agz_ctx();
  c->device = device;
  hipDeviceProp_t prop;
  AGZ_HIP_TRY(hipGetDeviceProperties(&prop, device));
  c->num_cus = prop.multiProcessorCount;
  AGZ_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  *out = c;
  return AGZ_OK;
}
int agz_host_alloc(agz_ctx* ctx, size_t bytes, void** out) {
  AGZ_REQUIRE(ctx && out && bytes > 0, AGZ_E_INVALID, "agz_host_alloc: bad argument");
  AGZ_HIP_TRY(hipSetDevice(ctx->device));
  void* p = nullptr;
  hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
  if (e != hipSuccess) { agz::set_error("agz_host_alloc: %zu bytes of pinned host memory: %s", bytes, hipGetErrorString(e)); return AGZ_E_NOMEM; }
  *out = p;
  return AGZ_OK;
}

int agz_host_free(agz_ctx* ctx, void* p) {
  AGZ_REQUIRE(ctx, AGZ_E_INVALID, "agz_host_free: ctx is NULL");
  if (!p) return AGZ_OK;
  AGZ_HIP_TRY(hipSetDevice(ctx->device));
  AGZ_HIP_TRY(hipHostFree(p));
  return AGZ_OK;
}

void agz_ctx_destroy(agz_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  for (auto& p : c->prof)
    for (auto& pr : p.pairs) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
  if (c->stream2) hipStreamDestroy(c->stream2);
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  if (c->ev_join) hipEventDestroy(c->ev_join);
  hipStreamDestroy(c->stream);
  delete c;
}
int agz_ctx_sync(agz_ctx* c) {
  AGZ_REQUIRE(c, AGZ_E_INVALID, "ctx is NULL");
  AGZ_HIP_TRY(hipStreamSynchronize(c->stream));
  return AGZ_OK;
}
void* agz_ctx_stream(agz_ctx* c) { return c ? (void*)c->stream : nullptr; }
int agz_ctx_prof_enable(agz_ctx* c, int enable) {
  AGZ_REQUIRE(c, AGZ_E_INVALID, "ctx is NULL");
  if (enable) {
    AGZ_HIP_TRY(hipStreamSynchronize(c->stream));
    for (auto& p : c->prof) { p.used = 0; p.launches = 0; p.total_ms = 0; }
    // enable == 1: every class; otherwise bit k+1 selects class k (each recorded launch costs two hipEventRecord calls on the
    // launching thread: bench.py times only the dominant class inside its timed region)
    c->prof_mask = enable == 1 ? ~0u : ((unsigned)enable >> 1);
    c->prof_on = true;
  } else {
    c->prof_on = false;
    return c->prof_collect();
  }
  return AGZ_OK;
}
int agz_ctx_prof_set_stride(agz_ctx* c, int klass, int stride) {
  AGZ_REQUIRE(c && klass >= 0 && klass < AGZ_PROF_NCLASS && stride >= 1, AGZ_E_INVALID, "agz_ctx_prof_set_stride: bad argument");
  c->prof_stride[klass] = stride;
  c->prof_seen[klass] = 0;
  return AGZ_OK;
}
int agz_ctx_prof_read(agz_ctx* c, int klass, int64_t* launches, double* total_ms) {
  AGZ_REQUIRE(c && klass >= 0 && klass < AGZ_PROF_NCLASS, AGZ_E_INVALID, "bad prof class");
  int r = c->prof_collect();
  if (r != AGZ_OK) return r;
  if (launches) *launches = c->prof[klass].launches;
  if (total_ms) *total_ms = c->prof[klass].total_ms;
  return AGZ_OK;
}

}  // extern "C"
