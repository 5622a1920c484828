// libagz internals: error plumbing, context, kernel-class timers.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/agz_debug.h"

namespace agz {

void set_error(const char* fmt, ...);
const char* get_error();

#define AGZ_HIP_TRY(expr)                                                                      \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      agz::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));     \
      return AGZ_E_HIP;                                                                        \
    }                                                                                          \
  } while (0)

#define AGZ_REQUIRE(cond, code, ...)  \
  do {                                \
    if (!(cond)) {                    \
      agz::set_error(__VA_ARGS__);    \
      return (code);                  \
    }                                 \
  } while (0)

struct ProfClass {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs;  // recorded this session
  size_t used = 0;
  int64_t launches = 0;
  double total_ms = 0;
};

}  // namespace agz

struct agz_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;          // second queue: the other half of a batch (agz_net two-stream tower), created on first use
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool prof_on = false;
  unsigned prof_mask = ~0u;   // classes that record events while prof_on (agz_ctx_prof_enable)
  int prof_stride[AGZ_PROF_NCLASS];   // every prof_stride[k]-th launch of class k is bracketed (agz_ctx_prof_set_stride; default 1)
  int64_t prof_seen[AGZ_PROF_NCLASS] = {};
  agz_ctx() { for (auto& v : prof_stride) v = 1; }
  agz::ProfClass prof[AGZ_PROF_NCLASS];
  int prof_open = 0;   // scopes begun and not yet ended (classes nest: a layer scope around its kernels' scopes)
  int num_cus = 256;
  // per-DEVICE function attributes set through this context (hipFuncSetAttribute applies to the current device only): bit 0 = the
  // fused out->in kernel's 80 KB of dynamic LDS is available, bit 1 = the attempt was made (conv_wino_h2c.hpp)
  unsigned func_attr_state = 0;

  // record a start event for a kernel class (no-op unless profiling)
  void prof_begin(int klass);
  void prof_end(int klass);
  int prof_collect();  // sync + fold recorded pairs into totals
};

namespace agz {
struct ProfScope {
  agz_ctx* c;
  int k;
  bool on;
  ProfScope(agz_ctx* c_, int k_) : c(c_), k(k_), on(c_->prof_on && ((c_->prof_mask >> k_) & 1u) && (c_->prof_seen[k_]++ % c_->prof_stride[k_]) == 0) { if (on) c->prof_begin(k); }
  ~ProfScope() { if (on) c->prof_end(k); }
};

// a scope that only records when the launch really goes to the ctx stream (the timers' events are recorded there)
struct ProfScopeOn {
  agz_ctx* c;
  int k;
  bool on;
  ProfScopeOn(agz_ctx* c_, int k_, bool enable)
      : c(c_), k(k_), on(enable && c_->prof_on && ((c_->prof_mask >> k_) & 1u) && (c_->prof_seen[k_]++ % c_->prof_stride[k_]) == 0) { if (on) c->prof_begin(k); }
  ~ProfScopeOn() { if (on) c->prof_end(k); }
};

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// SplitMix64: the build's RNG (same algorithm the oracle states; Go's math/rand is not reproducible here)
struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed = 0) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  int32_t int31() { return (int32_t)(next() >> 33); }
  double float64() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};
}  // namespace agz
