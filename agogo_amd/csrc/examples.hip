// agz_examples: the example plumbing between self-play and dual.Train, device resident.
//   RotateBoard        encoding_helper.go:80-107
//   Augmenter          datatypes.go:38-39 applied at arena.go:115-120 (rotation augmenter: e, rot e, rot^2 e, rot^3 e)
//   shuffleExamples    agogo.go:251-257, maxExamples cut agogo.go:118-121, prepareExamples agogo.go:211-249
// The 27 KB/example payload (19x19) never leaves HBM: the host only handles 4-byte row indices (the Fisher-Yates
// permutation is inherently sequential), the device does the gathers.  All kernels are pure data movement:
// bound by HBM, algorithmic bytes = 2 x payload (one read + one write per row).
#include <algorithm>
#include <numeric>

#include "comm.hpp"

namespace agz {

// dst[r][:] = src[idx[r]][:]   (idx == nullptr: identity)
__global__ void k_gather_rows(const float* __restrict__ src, const int32_t* __restrict__ idx, float* __restrict__ dst,
                              int row_len, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i / row_len;
    int c = (int)(i - r * row_len);
    size_t s = idx ? (size_t)idx[r] : r;
    dst[i] = src[s * row_len + c];
  }
}

// RotateBoard composed q times as an index map: one application is new[i][j] = old[j][m-1-i]
// (encoding_helper.go:92-103: it[i][j] <- it[j][mi1] <- it[mi1][mj1] <- it[mj1][i] <- it[i][j]).
__device__ __forceinline__ int rot_src(int i, int j, int m, int q) {
  switch (q & 3) {
    case 0: return i * m + j;
    case 1: return j * m + (m - 1 - i);
    case 2: return (m - 1 - i) * m + (m - 1 - j);
    default: return (m - 1 - j) * m + i;
  }
}

// out row reps*e+q = rot^(q0+q)(in row e): `planes` rows are nplanes boards of m*m followed by `tail` untouched floats
// (Board: nplanes = F, tail = 0; Policy: nplanes = 1, tail = pass entry; Value: nplanes = 0, tail = 1).
__global__ void k_augment_rot(const float* __restrict__ in, float* __restrict__ out, int nplanes, int m, int tail, int reps,
                              int q0, size_t total_out) {
  const int mm = m * m, row = nplanes * mm + tail;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_out; i += (size_t)gridDim.x * blockDim.x) {
    size_t ro = i / row;
    int c = (int)(i - ro * row);
    size_t e = ro / reps;
    int q = q0 + (int)(ro - e * reps);
    int sc = c;
    if (c < nplanes * mm) {
      int pl = c / mm, p = c - pl * mm;
      int pi = p / m, pj = p - pi * m;
      sc = pl * mm + rot_src(pi, pj, m, q);
    }
    out[i] = in[e * row + sc];
  }
}

static inline unsigned grid_for(size_t total) { return (unsigned)std::min<size_t>((total + 255) / 256, 65536u * 4u); }

}  // namespace agz

using namespace agz;

struct agz_examples {
  agz_ctx* ctx = nullptr;
  int F = 0, H = 0, W = 0, A1 = 0;
  size_t xs = 0;
  // raw store (append order = the reference's `ex = append(ex, a.SelfPlay()...)` order)
  float *planes = nullptr, *policy = nullptr, *value = nullptr;
  size_t n = 0, cap = 0;
  // prepared tensors
  float *Xs = nullptr, *Pi = nullptr, *V = nullptr;
  size_t n_prep = 0, prep_cap = 0;
  int batches = 0;
  int32_t* d_idx = nullptr;
  size_t idx_cap = 0;

  int reserve(size_t want) {
    if (want <= cap) return AGZ_OK;
    size_t nc = std::max(want, cap + cap / 2);
    float *p = nullptr, *q = nullptr, *v = nullptr;
    if (hipMalloc(&p, nc * xs * 4) != hipSuccess || hipMalloc(&q, nc * A1 * 4) != hipSuccess || hipMalloc(&v, nc * 4) != hipSuccess) {
      hipFree(p); hipFree(q); hipFree(v);
      agz::set_error("agz_examples: out of device memory growing the store to %zu examples", nc);
      return AGZ_E_HIP;
    }
    if (n) {
      AGZ_HIP_TRY(hipMemcpyAsync(p, planes, n * xs * 4, hipMemcpyDeviceToDevice, ctx->stream));
      AGZ_HIP_TRY(hipMemcpyAsync(q, policy, n * A1 * 4, hipMemcpyDeviceToDevice, ctx->stream));
      AGZ_HIP_TRY(hipMemcpyAsync(v, value, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
      AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    hipFree(planes); hipFree(policy); hipFree(value);
    planes = p; policy = q; value = v; cap = nc;
    return AGZ_OK;
  }
  int upload_idx(const std::vector<int32_t>& idx, size_t count) {
    if (count > idx_cap) {
      AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
      hipFree(d_idx); d_idx = nullptr; idx_cap = 0;
      AGZ_HIP_TRY(hipMalloc(&d_idx, count * 4));
      idx_cap = count;
    }
    AGZ_HIP_TRY(hipMemcpyAsync(d_idx, idx.data(), count * 4, hipMemcpyHostToDevice, ctx->stream));
    AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));  // idx is a caller-owned temporary
    return AGZ_OK;
  }
  // append `cnt` rows gathered from (p,q,v) in `order` (nullptr: as they are)
  int append(const float* p, const float* q, const float* v, size_t cnt, const std::vector<int32_t>* order) {
    if (!cnt) return AGZ_OK;
    int r = reserve(n + cnt);
    if (r != AGZ_OK) return r;
    const int32_t* di = nullptr;
    if (order) { r = upload_idx(*order, cnt); if (r != AGZ_OK) return r; di = d_idx; }
    hipStream_t s = ctx->stream;
    hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(cnt * xs)), dim3(256), 0, s, p, di, planes + n * xs, (int)xs, cnt * xs);
    hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(cnt * A1)), dim3(256), 0, s, q, di, policy + n * A1, A1, cnt * (size_t)A1);
    hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(cnt)), dim3(256), 0, s, v, di, value + n, 1, cnt);
    AGZ_HIP_TRY(hipGetLastError());
    n += cnt;
    return AGZ_OK;
  }
};

extern "C" {

int agz_examples_create(agz_ctx* ctx, int Features, int Height, int Width, int PolicyLen, agz_examples** out) {
  AGZ_REQUIRE(ctx && out, AGZ_E_INVALID, "agz_examples_create: NULL argument");
  AGZ_REQUIRE(Features >= 1 && Height >= 1 && Width >= 1 && PolicyLen >= 1, AGZ_E_INVALID, "agz_examples_create: bad shape");
  agz_examples* e = new agz_examples();
  e->ctx = ctx; e->F = Features; e->H = Height; e->W = Width; e->A1 = PolicyLen;
  e->xs = (size_t)Features * Height * Width;
  *out = e;
  return AGZ_OK;
}

void agz_examples_destroy(agz_examples* e) {
  if (!e) return;
  hipSetDevice(e->ctx->device);
  hipStreamSynchronize(e->ctx->stream);
  hipFree(e->planes); hipFree(e->policy); hipFree(e->value);
  hipFree(e->Xs); hipFree(e->Pi); hipFree(e->V); hipFree(e->d_idx);
  delete e;
}

int agz_examples_count(const agz_examples* e, int64_t* n) {
  AGZ_REQUIRE(e && n, AGZ_E_INVALID, "NULL argument");
  *n = (int64_t)e->n;
  return AGZ_OK;
}

int agz_examples_clear(agz_examples* e) {
  AGZ_REQUIRE(e, AGZ_E_INVALID, "NULL argument");
  e->n = 0; e->n_prep = 0; e->batches = 0;
  return AGZ_OK;
}

// `ex = append(ex, a.SelfPlay()...)` for every game of a batched arena (agogo.go:110-114).  The arena records in
// completion order of the per-game workgroups; the reference's order is episode after episode, each in ply order, i.e.
// a stable sort by game index — done here on the 4-byte game indices, the rows are gathered device to device.
int agz_examples_append_arena(agz_examples* e, agz_arena* arena) {
  AGZ_REQUIRE(e && arena, AGZ_E_INVALID, "NULL argument");
  AGZ_HIP_TRY(hipSetDevice(e->ctx->device));
  float *p = nullptr, *q = nullptr, *v = nullptr;
  int cnt = 0;
  int r = agz_arena_examples_dev(arena, &p, &q, &v, &cnt);
  if (r != AGZ_OK) return r;
  if (cnt == 0) return AGZ_OK;
  std::vector<int32_t> game(cnt);
  r = agz_arena_get_examples(arena, nullptr, nullptr, nullptr, game.data(), cnt, &cnt);
  if (r != AGZ_OK) return r;
  // only examples of FINISHED games carry a training target: SelfPlay returns after the game has ended and been labelled
  // (arena.go:140-155); rows of games still running (agz_arena_selfplay stopped at its target, agz_arena_play(n_moves > 0))
  // hold the raw mover colour and stay in the arena until their game ends (agz_arena_clear_examples would discard them with
  // their games' earlier plies: do not clear an arena in mid-game)
  const uint8_t* lab_dev = nullptr;
  r = agz_arena_examples_labelled_dev(arena, &lab_dev);
  if (r != AGZ_OK) return r;
  std::vector<uint8_t> lab(cnt);
  AGZ_HIP_TRY(hipMemcpy(lab.data(), lab_dev, (size_t)cnt, hipMemcpyDeviceToHost));
  std::vector<int32_t> order;
  order.reserve(cnt);
  for (int i = 0; i < cnt; i++) if (lab[i]) order.push_back(i);
  if (order.empty()) return AGZ_OK;
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return game[a] < game[b]; });
  r = e->append(p, q, v, order.size(), &order);
  if (r != AGZ_OK) return r;
  // TAKE semantics: what was appended leaves the arena (a second append does not duplicate it), the rows of games still in
  // flight stay — compacted to the front with their per-game chains re-linked — and are labelled and handed over once their game ends
  return agz_arena_drop_labelled_examples(arena);
}

int agz_examples_append_dev(agz_examples* e, const float* planes_dev, const float* policy_dev, const float* value_dev, int64_t n) {
  AGZ_REQUIRE(e && n >= 0 && (n == 0 || (planes_dev && policy_dev && value_dev)), AGZ_E_INVALID, "agz_examples_append_dev: bad argument");
  AGZ_HIP_TRY(hipSetDevice(e->ctx->device));
  return e->append(planes_dev, policy_dev, value_dev, (size_t)n, nullptr);
}

int agz_examples_append_host(agz_examples* e, const float* planes, const float* policy, const float* value, int64_t n) {
  AGZ_REQUIRE(e && n >= 0 && (n == 0 || (planes && policy && value)), AGZ_E_INVALID, "agz_examples_append_host: bad argument");
  if (n == 0) return AGZ_OK;
  AGZ_HIP_TRY(hipSetDevice(e->ctx->device));
  int r = e->reserve(e->n + (size_t)n);
  if (r != AGZ_OK) return r;
  hipStream_t s = e->ctx->stream;
  AGZ_HIP_TRY(hipMemcpyAsync(e->planes + e->n * e->xs, planes, (size_t)n * e->xs * 4, hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipMemcpyAsync(e->policy + e->n * e->A1, policy, (size_t)n * e->A1 * 4, hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipMemcpyAsync(e->value + e->n, value, (size_t)n * 4, hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipStreamSynchronize(s));
  e->n += (size_t)n;
  return AGZ_OK;
}

// The exchange step of the path (agogo.go:118-133 runs on the union of all self-play examples): every rank
// contributes the rows its store holds; afterwards every rank's store holds the rows of rank 0, 1, ... in that order.
// Counts are exchanged first (one 8-byte all-gather), then n broadcasts — rank r the root of its own rows, received straight
// into their final position in a new store — are issued as ONE group: no padding, no staging copy, all xGMI links busy at once.
int agz_examples_allgather(agz_comm* c, agz_examples* e) {
  AGZ_REQUIRE(c && e, AGZ_E_INVALID, "agz_examples_allgather: NULL argument");
  AGZ_REQUIRE(c->ctx == e->ctx, AGZ_E_INVALID, "agz_examples_allgather: the communicator and the example set belong to different contexts");
  const agz::Rccl* R = agz::rccl();
  if (!R) return AGZ_E_UNSUPPORTED;
  AGZ_HIP_TRY(hipSetDevice(e->ctx->device));
  hipStream_t s = e->ctx->stream;
  const int n = c->size;
  unsigned long long* d_cnt = c->d_words;            // [n] gathered counts + [1] mine (allocated with the communicator)
  AGZ_REQUIRE(d_cnt, AGZ_E_STATE, "agz_examples_allgather: communicator without exchange words");
  std::vector<unsigned long long> cnt(n, 0);
  const unsigned long long mine = (unsigned long long)e->n;
  hipError_t he = hipMemcpyAsync(d_cnt + n, &mine, 8, hipMemcpyHostToDevice, s);
  ncclResult_t nr = he == hipSuccess ? R->AllGather(d_cnt + n, d_cnt, 1, ncclUint64, c->comm, s) : ncclSuccess;
  if (he == hipSuccess && nr == ncclSuccess) he = hipMemcpyAsync(cnt.data(), d_cnt, (size_t)n * 8, hipMemcpyDeviceToHost, s);
  if (he == hipSuccess && nr == ncclSuccess) he = hipStreamSynchronize(s);
  AGZ_REQUIRE(nr == ncclSuccess, AGZ_E_HIP, "agz_examples_allgather: count exchange -> %s", R->GetErrorString(nr));
  AGZ_HIP_TRY(he);
  size_t total = 0;
  for (int r = 0; r < n; r++) total += (size_t)cnt[r];
  if (n == 1 || total == 0) return AGZ_OK;          // one rank: the store already is the union
  float *p = nullptr, *q = nullptr, *v = nullptr;
  const bool got = hipMalloc(&p, total * e->xs * 4) == hipSuccess && hipMalloc(&q, total * e->A1 * 4) == hipSuccess && hipMalloc(&v, total * 4) == hipSuccess;
  // every rank must enter the broadcast group or none: the ranks agree on "everybody has its receive store" with a one-word sum
  {
    // (the two words live in the communicator: nothing is allocated here, and the all-reduce is entered whatever the copy of this
    //  rank's word returned — a rank that skipped it while its peers entered would hang them; a failed copy counts as a failure)
    unsigned long long* d_fail = c->d_words + n + 1;
    unsigned long long fails = got ? 0 : 1, all_fails = 1;
    he = hipMemcpyAsync(d_fail, &fails, 8, hipMemcpyHostToDevice, s);
    if (he != hipSuccess) { (void)hipMemsetAsync(d_fail, 0xff, 8, s); }
    nr = R->AllReduce(d_fail, d_fail + 1, 1, ncclUint64, ncclSum, c->comm, s);
    if (he == hipSuccess && nr == ncclSuccess) he = hipMemcpyAsync(&all_fails, d_fail + 1, 8, hipMemcpyDeviceToHost, s);
    if (he == hipSuccess && nr == ncclSuccess) he = hipStreamSynchronize(s);
    if (nr != ncclSuccess || he != hipSuccess || all_fails != 0) {
      hipFree(p); hipFree(q); hipFree(v);
      AGZ_REQUIRE(nr == ncclSuccess, AGZ_E_HIP, "agz_examples_allgather: allocation agreement -> %s", R->GetErrorString(nr));
      AGZ_HIP_TRY(he);
      agz::set_error("agz_examples_allgather: out of device memory for %zu gathered examples on %llu rank(s); no rank gathered", total, all_fails);
      return AGZ_E_NOMEM;
    }
  }
  nr = R->GroupStart();
  size_t off = 0;
  for (int r = 0; r < n && nr == ncclSuccess; r++) {
    const size_t k = (size_t)cnt[r];
    if (k) {
      const bool root = r == c->rank;
      nr = R->Broadcast(root ? e->planes : p + off * e->xs, p + off * e->xs, k * e->xs, ncclFloat32, r, c->comm, s);
      if (nr == ncclSuccess) nr = R->Broadcast(root ? e->policy : q + off * e->A1, q + off * e->A1, k * (size_t)e->A1, ncclFloat32, r, c->comm, s);
      if (nr == ncclSuccess) nr = R->Broadcast(root ? e->value : v + off, v + off, k, ncclFloat32, r, c->comm, s);
    }
    off += k;
  }
  ncclResult_t ge = R->GroupEnd();
  if (nr == ncclSuccess) nr = ge;
  he = hipStreamSynchronize(s);
  if (nr != ncclSuccess || he != hipSuccess) {
    hipFree(p); hipFree(q); hipFree(v);
    AGZ_REQUIRE(nr == ncclSuccess, AGZ_E_HIP, "agz_examples_allgather: broadcast group -> %s", R->GetErrorString(nr));
    AGZ_HIP_TRY(he);
  }
  hipFree(e->planes); hipFree(e->policy); hipFree(e->value);
  e->planes = p; e->policy = q; e->value = v; e->n = total; e->cap = total;
  return AGZ_OK;
}

int agz_examples_raw_dev(agz_examples* e, float** planes, float** policy, float** value) {
  AGZ_REQUIRE(e, AGZ_E_INVALID, "NULL argument");
  AGZ_HIP_TRY(hipSetDevice(e->ctx->device));
  AGZ_HIP_TRY(hipStreamSynchronize(e->ctx->stream));
  if (planes) *planes = e->planes;
  if (policy) *policy = e->policy;
  if (value) *value = e->value;
  return AGZ_OK;
}

int agz_examples_get(agz_examples* e, float* planes, float* policy, float* value, int64_t cap, int64_t* n) {
  AGZ_REQUIRE(e && n, AGZ_E_INVALID, "NULL argument");
  AGZ_HIP_TRY(hipSetDevice(e->ctx->device));
  *n = (int64_t)e->n;
  size_t k = (size_t)std::min<int64_t>(cap, (int64_t)e->n);
  AGZ_HIP_TRY(hipStreamSynchronize(e->ctx->stream));
  if (k == 0) return AGZ_OK;
  if (planes) AGZ_HIP_TRY(hipMemcpy(planes, e->planes, k * e->xs * 4, hipMemcpyDeviceToHost));
  if (policy) AGZ_HIP_TRY(hipMemcpy(policy, e->policy, k * e->A1 * 4, hipMemcpyDeviceToHost));
  if (value) AGZ_HIP_TRY(hipMemcpy(value, e->value, k * 4, hipMemcpyDeviceToHost));
  return AGZ_OK;
}

// The rotation Augmenter over the whole set: every example e becomes [e, rot e, rot^2 e, rot^3 e] in place of e
// (same multiset and order as applying it per example at record time, arena.go:115-120).  RotateBoard's error for
// non-square boards (encoding_helper.go:81-83) is AGZ_E_INVALID.
int agz_examples_augment_rotate(agz_examples* e) {
  AGZ_REQUIRE(e, AGZ_E_INVALID, "NULL argument");
  AGZ_REQUIRE(e->H == e->W, AGZ_E_INVALID, "Cannot handle m %d, n %d. This function only takes square boards", e->H, e->W);
  AGZ_REQUIRE(e->A1 >= e->H * e->W, AGZ_E_INVALID, "agz_examples_augment_rotate: policy shorter than the board");
  if (e->n == 0) return AGZ_OK;
  AGZ_HIP_TRY(hipSetDevice(e->ctx->device));
  const size_t n4 = e->n * 4;
  float *p = nullptr, *q = nullptr, *v = nullptr;
  if (hipMalloc(&p, n4 * e->xs * 4) != hipSuccess || hipMalloc(&q, n4 * e->A1 * 4) != hipSuccess || hipMalloc(&v, n4 * 4) != hipSuccess) {
    hipFree(p); hipFree(q); hipFree(v);
    agz::set_error("agz_examples_augment_rotate: out of device memory for %zu examples", n4);
    return AGZ_E_HIP;
  }
  hipStream_t s = e->ctx->stream;
  const int m = e->H;
  hipLaunchKernelGGL(k_augment_rot, dim3(grid_for(n4 * e->xs)), dim3(256), 0, s, e->planes, p, e->F, m, 0, 4, 0, n4 * e->xs);
  hipLaunchKernelGGL(k_augment_rot, dim3(grid_for(n4 * e->A1)), dim3(256), 0, s, e->policy, q, 1, m, e->A1 - m * m, 4, 0, n4 * (size_t)e->A1);
  hipLaunchKernelGGL(k_augment_rot, dim3(grid_for(n4)), dim3(256), 0, s, e->value, v, 0, m, 1, 4, 0, n4);
  AGZ_HIP_TRY(hipGetLastError());
  AGZ_HIP_TRY(hipStreamSynchronize(s));
  hipFree(e->planes); hipFree(e->policy); hipFree(e->value);
  e->planes = p; e->policy = q; e->value = v; e->n = n4; e->cap = n4;
  return AGZ_OK;
}

// agogo.go:118-121 (maxExamples cut) + prepareExamples (agogo.go:211-249): shuffle, batches = n / BatchSize, keep
// batches*BatchSize rows as Xs [rows,F,H,W], Policies [rows,PolicyLen], Values [rows] — device tensors.
int agz_examples_prepare(agz_examples* e, int BatchSize, int maxExamples, uint64_t seed, int* batches) {
  AGZ_REQUIRE(e && batches && BatchSize >= 1, AGZ_E_INVALID, "agz_examples_prepare: bad argument");
  AGZ_HIP_TRY(hipSetDevice(e->ctx->device));
  size_t n = e->n;
  std::vector<int32_t> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  SplitMix64 rng(seed);
  auto shuffle = [&](size_t cnt) {  // shuffleExamples agogo.go:251-257
    for (size_t i = 0; i < cnt; i++) { size_t j = (size_t)(rng.next() % (uint64_t)(i + 1)); std::swap(idx[i], idx[j]); }
  };
  if (maxExamples > 0 && n > (size_t)maxExamples) { shuffle(n); n = (size_t)maxExamples; idx.resize(n); }
  shuffle(n);
  int nb = (int)(n / (size_t)BatchSize);
  size_t total = (size_t)nb * BatchSize;
  e->batches = nb; e->n_prep = total;
  *batches = nb;
  if (total == 0) return AGZ_OK;  // the caller reports "batches is nil" (agogo.go:123-125)
  if (total > e->prep_cap) {
    AGZ_HIP_TRY(hipStreamSynchronize(e->ctx->stream));
    hipFree(e->Xs); hipFree(e->Pi); hipFree(e->V);
    e->Xs = e->Pi = e->V = nullptr; e->prep_cap = 0;
    AGZ_HIP_TRY(hipMalloc(&e->Xs, total * e->xs * 4));
    AGZ_HIP_TRY(hipMalloc(&e->Pi, total * e->A1 * 4));
    AGZ_HIP_TRY(hipMalloc(&e->V, total * 4));
    e->prep_cap = total;
  }
  int r = e->upload_idx(idx, total);
  if (r != AGZ_OK) return r;
  hipStream_t s = e->ctx->stream;
  hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(total * e->xs)), dim3(256), 0, s, e->planes, e->d_idx, e->Xs, (int)e->xs, total * e->xs);
  hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(total * e->A1)), dim3(256), 0, s, e->policy, e->d_idx, e->Pi, e->A1, total * (size_t)e->A1);
  hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(total)), dim3(256), 0, s, e->value, e->d_idx, e->V, 1, total);
  AGZ_HIP_TRY(hipGetLastError());
  return AGZ_OK;
}

int agz_examples_tensors_dev(agz_examples* e, float** Xs, float** Policies, float** Values, int64_t* rows, int* batches) {
  AGZ_REQUIRE(e, AGZ_E_INVALID, "NULL argument");
  if (Xs) *Xs = e->Xs;
  if (Policies) *Policies = e->Pi;
  if (Values) *Values = e->V;
  if (rows) *rows = (int64_t)e->n_prep;
  if (batches) *batches = e->batches;
  return AGZ_OK;
}

int agz_examples_get_tensors(agz_examples* e, float* Xs, float* Policies, float* Values) {
  AGZ_REQUIRE(e, AGZ_E_INVALID, "NULL argument");
  AGZ_HIP_TRY(hipSetDevice(e->ctx->device));
  AGZ_HIP_TRY(hipStreamSynchronize(e->ctx->stream));
  if (e->n_prep == 0) return AGZ_OK;
  if (Xs) AGZ_HIP_TRY(hipMemcpy(Xs, e->Xs, e->n_prep * e->xs * 4, hipMemcpyDeviceToHost));
  if (Policies) AGZ_HIP_TRY(hipMemcpy(Policies, e->Pi, e->n_prep * e->A1 * 4, hipMemcpyDeviceToHost));
  if (Values) AGZ_HIP_TRY(hipMemcpy(Values, e->V, e->n_prep * 4, hipMemcpyDeviceToHost));
  return AGZ_OK;
}

// RotateBoard (encoding_helper.go:80-107) for `count` boards of m x n floats, host buffers in and out.
int agz_rotate_boards(agz_ctx* ctx, const float* boards, int count, int m, int n, float* out) {
  AGZ_REQUIRE(ctx && boards && out && count >= 0 && m >= 1, AGZ_E_INVALID, "agz_rotate_boards: bad argument");
  AGZ_REQUIRE(m == n, AGZ_E_INVALID, "Cannot handle m %d, n %d. This function only takes square boards", m, n);
  if (count == 0) return AGZ_OK;
  AGZ_HIP_TRY(hipSetDevice(ctx->device));
  size_t tot = (size_t)count * m * m;
  float *a = nullptr, *b = nullptr;
  AGZ_HIP_TRY(hipMalloc(&a, tot * 4));
  AGZ_HIP_TRY(hipMalloc(&b, tot * 4));
  AGZ_HIP_TRY(hipMemcpyAsync(a, boards, tot * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_augment_rot, dim3(grid_for(tot)), dim3(256), 0, ctx->stream, a, b, 1, m, 0, 1, 1, tot);
  hipError_t le = hipGetLastError();
  hipError_t ce = le == hipSuccess ? hipMemcpyAsync(out, b, tot * 4, hipMemcpyDeviceToHost, ctx->stream) : le;
  hipError_t se = hipStreamSynchronize(ctx->stream);
  hipFree(a); hipFree(b);
  AGZ_HIP_TRY(ce);
  AGZ_HIP_TRY(se);
  return AGZ_OK;
}

}  // extern "C"
