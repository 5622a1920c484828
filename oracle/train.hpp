// ORACLE — TEST INFRASTRUCTURE ONLY (see game.hpp header).
// CPU restatement of dual.Train (dualnet/meta.go:16-54): the training graph of Dual.fwd + Dual.bwd
// (dualnet/dual.go:50-132), the loss of ermahagerdmonards.go:106-147, vanilla SGD (meta.go:17-20: lr 0.1, no
// momentum, no weight decay — Config.L2 is never read).
//
// PARITY UNPINNED (like dualnet.hpp): the op semantics live in un-vendored gorgonia.  Restated here as published:
//   * BatchNorm in TRAINING mode: per-channel batch statistics over (B,H,W), biased variance, eps 1e-5
//     (ermahagerdmonards.go:54); nil scale/bias make gorgonia create gamma/beta with the FULL shape of x
//     [B,C,H,W]: every batch row owns its gamma/beta; likewise FC biases are [B,units]
//     (ermahagerdmonards.go:82).  Inference uses row 0 only (App. B b5).
//   * "xent" = -mean(Pi*logits + (1-Pi)*(1-logits)) over all B*A entries, on the LOGITS — linear in the logits,
//     so dL/dlogits = (1 - 2*Pi)/(B*A)  (ermahagerdmonards.go:106-147).
//   * value cost = mean((o - V)^2) on the PRE-tanh output o (dual.go:116-118).
// Templated on the scalar type so the same code runs in double for the finite-difference gradient check.
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "dualnet.hpp"

namespace oracle {

template <typename T>
struct TrainNet {
  DualConfig conf;
  int B, K, F, H, W, HW, A, FC, L;
  // full-shape learnables in Model() order (include/agz.h): conv filter [O,I,k,k]; gamma/beta [B,C,H,W]; fc w [in,units]; fc b [B,units]
  std::vector<std::vector<T>> P, G;  // parameters and gradients
  std::vector<std::string> names;
  std::vector<int> kinds;            // 0 conv, 1 bn gamma/beta, 2 fc w, 3 fc b
  std::vector<std::vector<int>> shapes;

  explicit TrainNet(const DualConfig& c) : conf(c) {
    B = c.BatchSize; K = c.K; F = c.Features; H = c.Height; W = c.Width; HW = H * W; A = c.ActionSpace; FC = c.FC; L = c.SharedLayers;
    auto add = [&](const std::string& nm, int kind, std::vector<int> shp) {
      size_t n = 1; for (int d : shp) n *= (size_t)d;
      names.push_back(nm); kinds.push_back(kind); shapes.push_back(shp);
      P.emplace_back(n, T(0)); G.emplace_back(n, T(0));
    };
    auto conv = [&](const std::string& nm, int o, int i, int k) { add("Filter" + nm, 0, {o, i, k, k}); };
    auto bn = [&](const std::string& nm, int C) { add(nm + "_gamma", 1, {B, C, H, W}); add(nm + "_beta", 1, {B, C, H, W}); };
    auto fc = [&](const std::string& nm, int in, int u) { add(nm + "_w", 2, {in, u}); add(nm + "_b", 3, {B, u}); };
    conv("Init", K, F, 3); bn("Init", K);
    for (int i = 0; i < L; i++) {
      std::string s = std::to_string(i);
      conv("Layer1 of Shared Layer " + s, K, K, 3); bn("L1_" + s, K);
      conv("Layer2 of Shared Layer " + s, K, K, 3); bn("L2_" + s, K);
    }
    conv("PolicyHead", 2, K, 1); bn("PolicyHead", 2); fc("Policy", 2 * HW, A);
    conv("ValueHead", 1, K, 1); bn("ValueHead", 1); fc("Value", HW, FC); fc("ValueOutput", FC, 1);
  }

  void InitRandom(uint64_t seed) {  // same Glorot recipe as dualnet.hpp, over the FULL shapes
    SplitMix64 r(seed);
    for (size_t pi = 0; pi < P.size(); pi++) {
      const std::vector<int>& s = shapes[pi];
      double field = 1; for (size_t i = 2; i < s.size(); i++) field *= s[i];
      double stdev = std::sqrt(2.0 / ((double)(s[0] + s[1]) * field));
      std::vector<T>& v = P[pi];
      if (kinds[pi] == 0) { double lim = stdev * std::sqrt(3.0); for (T& x : v) x = (T)(float)((r.float64() * 2.0 - 1.0) * lim); }
      else if (kinds[pi] == 1 || kinds[pi] == 2) {
        for (size_t i = 0; i < v.size(); i += 2) {
          double u1 = 1.0 - r.float64(), u2 = r.float64();
          double rad = std::sqrt(-2.0 * std::log(u1)), th = 6.283185307179586476925 * u2;
          v[i] = (T)(float)(rad * std::cos(th) * stdev);
          if (i + 1 < v.size()) v[i + 1] = (T)(float)(rad * std::sin(th) * stdev);
        }
      } else for (T& x : v) x = T(0);
    }
  }

  // ---- layers (NCHW, batch-major) ------------------------------------------------------------------------------
  struct BNCache { std::vector<T> xhat, inv; };  // xhat [B,C,HW], inv-std [C]
  void conv_fwd(const std::vector<T>& x, int Cin, const std::vector<T>& w, int Cout, int k, std::vector<T>* y) const {
    int pad = (k - 1) / 2;
    y->assign((size_t)B * Cout * HW, T(0));
    for (int b = 0; b < B; b++) for (int o = 0; o < Cout; o++) for (int h = 0; h < H; h++) for (int x0 = 0; x0 < W; x0++) {
      T s = 0;
      for (int ky = 0; ky < k; ky++) { int ih = h + ky - pad; if (ih < 0 || ih >= H) continue;
        for (int kx = 0; kx < k; kx++) { int iw = x0 + kx - pad; if (iw < 0 || iw >= W) continue;
          for (int c = 0; c < Cin; c++) s += x[((size_t)b * Cin + c) * HW + ih * W + iw] * w[(((size_t)o * Cin + c) * k + ky) * k + kx]; } }
      (*y)[((size_t)b * Cout + o) * HW + h * W + x0] = s;
    }
  }
  void conv_bwd(const std::vector<T>& x, int Cin, const std::vector<T>& w, int Cout, int k, const std::vector<T>& dy,
                std::vector<T>* dx, std::vector<T>* dw) const {
    int pad = (k - 1) / 2;
    if (dx) dx->assign((size_t)B * Cin * HW, T(0));
    dw->assign(w.size(), T(0));
    for (int b = 0; b < B; b++) for (int o = 0; o < Cout; o++) for (int h = 0; h < H; h++) for (int x0 = 0; x0 < W; x0++) {
      T g = dy[((size_t)b * Cout + o) * HW + h * W + x0];
      if (g == T(0)) continue;
      for (int ky = 0; ky < k; ky++) { int ih = h + ky - pad; if (ih < 0 || ih >= H) continue;
        for (int kx = 0; kx < k; kx++) { int iw = x0 + kx - pad; if (iw < 0 || iw >= W) continue;
          for (int c = 0; c < Cin; c++) {
            size_t xi = ((size_t)b * Cin + c) * HW + ih * W + iw, wi = (((size_t)o * Cin + c) * k + ky) * k + kx;
            (*dw)[wi] += g * x[xi];
            if (dx) (*dx)[xi] += g * w[wi];
          } } }
    }
  }
  // y = relu(gamma * xhat + beta), batch statistics
  void bn_relu_fwd(const std::vector<T>& z, int C, const std::vector<T>& gamma, const std::vector<T>& beta, BNCache* cache,
                   std::vector<T>* y) const {
    size_t m = (size_t)B * HW;
    cache->xhat.assign(z.size(), T(0)); cache->inv.assign(C, T(0));
    y->assign(z.size(), T(0));
    for (int c = 0; c < C; c++) {
      T mean = 0; for (int b = 0; b < B; b++) for (int p = 0; p < HW; p++) mean += z[((size_t)b * C + c) * HW + p];
      mean /= (T)m;
      T var = 0; for (int b = 0; b < B; b++) for (int p = 0; p < HW; p++) { T d = z[((size_t)b * C + c) * HW + p] - mean; var += d * d; }
      var /= (T)m;
      T inv = T(1) / std::sqrt(var + (T)conf.bn_eps);
      cache->inv[c] = inv;
      for (int b = 0; b < B; b++) for (int p = 0; p < HW; p++) {
        size_t i = ((size_t)b * C + c) * HW + p;
        T xh = (z[i] - mean) * inv;
        cache->xhat[i] = xh;
        T v = gamma[i] * xh + beta[i];
        (*y)[i] = v > T(0) ? v : T(0);
      }
    }
  }
  // dy is d/d(relu output); y is the relu output (mask)
  void bn_relu_bwd(const std::vector<T>& dy_in, const std::vector<T>& y, int C, const std::vector<T>& gamma, const BNCache& cache,
                   std::vector<T>* dz, std::vector<T>* dgamma, std::vector<T>* dbeta) const {
    size_t m = (size_t)B * HW;
    dz->assign(y.size(), T(0)); dgamma->assign(y.size(), T(0)); dbeta->assign(y.size(), T(0));
    for (int c = 0; c < C; c++) {
      T s1 = 0, s2 = 0;
      for (int b = 0; b < B; b++) for (int p = 0; p < HW; p++) {
        size_t i = ((size_t)b * C + c) * HW + p;
        T g = y[i] > T(0) ? dy_in[i] : T(0);
        (*dgamma)[i] = g * cache.xhat[i];
        (*dbeta)[i] = g;
        T dxh = g * gamma[i];
        (*dz)[i] = dxh;  // temporarily d/dxhat
        s1 += dxh; s2 += dxh * cache.xhat[i];
      }
      for (int b = 0; b < B; b++) for (int p = 0; p < HW; p++) {
        size_t i = ((size_t)b * C + c) * HW + p;
        (*dz)[i] = cache.inv[c] * ((*dz)[i] - s1 / (T)m - cache.xhat[i] * s2 / (T)m);
      }
    }
  }

  // One forward+backward on a batch: planes [B,F,H,W], Pi [B,A], V [B].  Fills G; returns the cost.
  T ForwardBackward(const T* planes, const T* Pi, const T* V, bool backward = true) {
    size_t pi = 0;
    std::vector<T> x0(planes, planes + (size_t)B * F * HW);
    struct Blk { std::vector<T> x_in, za, zb, ya, yb, out; BNCache ca, cb; size_t pa, pb; };
    // init layer
    std::vector<T> z_init, a_init; BNCache c_init; size_t p_init = pi;
    conv_fwd(x0, F, P[pi], K, 3, &z_init); bn_relu_fwd(z_init, K, P[pi + 1], P[pi + 2], &c_init, &a_init); pi += 3;
    std::vector<Blk> blk(L);
    const std::vector<T>* cur = &a_init;
    for (int l = 0; l < L; l++) {
      Blk& k = blk[l]; k.x_in = *cur; k.pa = pi; k.pb = pi + 3;
      conv_fwd(k.x_in, K, P[pi], K, 3, &k.za); bn_relu_fwd(k.za, K, P[pi + 1], P[pi + 2], &k.ca, &k.ya); pi += 3;
      conv_fwd(k.x_in, K, P[pi], K, 3, &k.zb); bn_relu_fwd(k.zb, K, P[pi + 1], P[pi + 2], &k.cb, &k.yb); pi += 3;
      k.out.resize(k.ya.size());
      for (size_t i = 0; i < k.out.size(); i++) { T s = k.ya[i] + k.yb[i]; k.out[i] = s > T(0) ? s : T(0); }
      cur = &k.out;
    }
    const std::vector<T>& xs = *cur;
    // policy head
    size_t p_pol = pi; std::vector<T> zp, yp; BNCache cp;
    conv_fwd(xs, K, P[pi], 2, 1, &zp); bn_relu_fwd(zp, 2, P[pi + 1], P[pi + 2], &cp, &yp); pi += 3;
    size_t p_pfc = pi; pi += 2;
    std::vector<T> logits((size_t)B * A);
    for (int b = 0; b < B; b++) for (int j = 0; j < A; j++) {
      T s = 0; for (int i = 0; i < 2 * HW; i++) s += yp[(size_t)b * 2 * HW + i] * P[p_pfc][(size_t)i * A + j];
      logits[(size_t)b * A + j] = s + P[p_pfc + 1][(size_t)b * A + j];
    }
    // value head
    size_t p_val = pi; std::vector<T> zv, yv; BNCache cv;
    conv_fwd(xs, K, P[pi], 1, 1, &zv); bn_relu_fwd(zv, 1, P[pi + 1], P[pi + 2], &cv, &yv); pi += 3;
    size_t p_v1 = pi; pi += 2; size_t p_v2 = pi; pi += 2;
    std::vector<T> hpre((size_t)B * FC), hid((size_t)B * FC), o(B);
    for (int b = 0; b < B; b++) {
      for (int j = 0; j < FC; j++) {
        T s = 0; for (int i = 0; i < HW; i++) s += yv[(size_t)b * HW + i] * P[p_v1][(size_t)i * FC + j];
        s += P[p_v1 + 1][(size_t)b * FC + j];
        hpre[(size_t)b * FC + j] = s; hid[(size_t)b * FC + j] = s > T(0) ? s : T(0);
      }
      T s = 0; for (int j = 0; j < FC; j++) s += hid[(size_t)b * FC + j] * P[p_v2][j];
      o[b] = s + P[p_v2 + 1][b];
    }
    // costs (dual.go:113-121)
    T pcost = 0;
    for (size_t i = 0; i < logits.size(); i++) pcost += -(Pi[i] * logits[i] + (T(1) - Pi[i]) * (T(1) - logits[i]));
    pcost /= (T)logits.size();
    T vcost = 0; for (int b = 0; b < B; b++) { T d = o[b] - V[b]; vcost += d * d; } vcost /= (T)B;
    T cost = pcost + vcost;
    if (!backward) return cost;
    for (auto& g : G) std::fill(g.begin(), g.end(), T(0));
    // ---- backward
    std::vector<T> dxs(xs.size(), T(0));
    {  // value head
      std::vector<T> dyv((size_t)B * HW, T(0));
      for (int b = 0; b < B; b++) {
        T dob = T(2) * (o[b] - V[b]) / (T)B;
        G[p_v2 + 1][b] = dob;
        for (int j = 0; j < FC; j++) {
          G[p_v2][j] += dob * hid[(size_t)b * FC + j];
          T dh = hpre[(size_t)b * FC + j] > T(0) ? dob * P[p_v2][j] : T(0);
          G[p_v1 + 1][(size_t)b * FC + j] = dh;
          if (dh != T(0)) for (int i = 0; i < HW; i++) { G[p_v1][(size_t)i * FC + j] += dh * yv[(size_t)b * HW + i]; dyv[(size_t)b * HW + i] += dh * P[p_v1][(size_t)i * FC + j]; }
        }
      }
      std::vector<T> dzv, dx;
      bn_relu_bwd(dyv, yv, 1, P[p_val + 1], cv, &dzv, &G[p_val + 1], &G[p_val + 2]);
      conv_bwd(xs, K, P[p_val], 1, 1, dzv, &dx, &G[p_val]);
      for (size_t i = 0; i < dxs.size(); i++) dxs[i] += dx[i];
    }
    {  // policy head
      std::vector<T> dyp((size_t)B * 2 * HW, T(0));
      T sc = T(1) / (T)logits.size();
      for (int b = 0; b < B; b++) for (int j = 0; j < A; j++) {
        T dl = (T(1) - T(2) * Pi[(size_t)b * A + j]) * sc;
        G[p_pfc + 1][(size_t)b * A + j] = dl;
        for (int i = 0; i < 2 * HW; i++) { G[p_pfc][(size_t)i * A + j] += dl * yp[(size_t)b * 2 * HW + i]; dyp[(size_t)b * 2 * HW + i] += dl * P[p_pfc][(size_t)i * A + j]; }
      }
      std::vector<T> dzp, dx;
      bn_relu_bwd(dyp, yp, 2, P[p_pol + 1], cp, &dzp, &G[p_pol + 1], &G[p_pol + 2]);
      conv_bwd(xs, K, P[p_pol], 2, 1, dzp, &dx, &G[p_pol]);
      for (size_t i = 0; i < dxs.size(); i++) dxs[i] += dx[i];
    }
    std::vector<T> dcur = dxs;
    for (int l = L - 1; l >= 0; l--) {
      Blk& k = blk[l];
      std::vector<T> dsum(dcur.size());
      for (size_t i = 0; i < dsum.size(); i++) dsum[i] = k.out[i] > T(0) ? dcur[i] : T(0);
      std::vector<T> dza, dzb, dxa, dxb;
      bn_relu_bwd(dsum, k.ya, K, P[k.pa + 1], k.ca, &dza, &G[k.pa + 1], &G[k.pa + 2]);
      bn_relu_bwd(dsum, k.yb, K, P[k.pb + 1], k.cb, &dzb, &G[k.pb + 1], &G[k.pb + 2]);
      conv_bwd(k.x_in, K, P[k.pa], K, 3, dza, &dxa, &G[k.pa]);
      conv_bwd(k.x_in, K, P[k.pb], K, 3, dzb, &dxb, &G[k.pb]);
      dcur.resize(dxa.size());
      for (size_t i = 0; i < dcur.size(); i++) dcur[i] = dxa[i] + dxb[i];
    }
    {
      std::vector<T> dz;
      bn_relu_bwd(dcur, a_init, K, P[p_init + 1], c_init, &dz, &G[p_init + 1], &G[p_init + 2]);
      conv_bwd(x0, F, P[p_init], K, 3, dz, nullptr, &G[p_init]);
    }
    return cost;
  }
  void Step(T lr) {  // VanillaSolver (meta.go:20,40): w -= lr * grad
    for (size_t i = 0; i < P.size(); i++) for (size_t j = 0; j < P[i].size(); j++) P[i][j] -= lr * G[i][j];
  }
};

}  // namespace oracle
