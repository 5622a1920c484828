// ORACLE — TEST INFRASTRUCTURE ONLY (see game.hpp header).
// CPU restatement of the forward (inference) path of gorgonia/agogo's dualnet package:
//   topology   dualnet/dual.go:50-103, layer helpers dualnet/ermahagerdmonards.go:33-104,
//   call shape dualnet/meta.go:125-190 (every board evaluated with the row-0 parameters).
//
// PARITY UNPINNED for the NN arithmetic: the op semantics (Conv2d, BatchNorm, SoftMax, Glorot) live in
// gorgonia.org/gorgonia v0.9.17-0.20210124090702-531c6df2c434 / gorgonia.org/tensor v0.9.18, which are
// not vendored under /root/reference, and the reference's own tests assert no NN numbers
// (dualnet/dual_test.go:17-109).  What is restated here is the PUBLISHED algorithm of those ops:
//   Conv2d      = cross-correlation, stride 1, pad (k-1)/2, no bias     (ermahagerdmonards.go:33-46)
//   BatchNorm   = (x-mean)/sqrt(var+eps)*gamma+beta, eps 1e-5; inference statistics selectable (bn_mode)
//   SoftMax     = exp(x-max)/sum over the last axis;  Tanh, ReLU elementwise.
// The only reference-pinned piece is round() (dualnet/config_test.go:5-17).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "game.hpp"

namespace oracle {

// dualnet/config.go:4-16 (+ bn_mode/bn_eps: see include/agz.h)
struct DualConfig {
  int K = 0, SharedLayers = 0, FC = 0;
  int BatchSize = 0, Width = 0, Height = 0, Features = 0, ActionSpace = 0;
  int bn_mode = 0;
  float bn_eps = 1e-5f;
  bool IsValid() const {  // config.go:33-42
    return K >= 1 && ActionSpace >= 3 && SharedLayers >= 0 && FC > 1 && BatchSize >= 1 && Features > 0;
  }
};

inline int dual_round(int a) {  // config.go:44-58
  int n = a - 1;
  n |= n >> 1; n |= n >> 2; n |= n >> 4; n |= n >> 8; n |= n >> 16;
  n++;
  int lt = n / 2;
  if ((a - lt) < (n - a)) return lt;
  return n;
}
inline DualConfig DefaultConf(int m, int n, int actionSpace) {  // config.go:18-31
  DualConfig c;
  int k = dual_round((m * n) / 3);
  c.K = k; c.SharedLayers = m; c.FC = 2 * k; c.BatchSize = 256; c.Width = n; c.Height = m;
  c.Features = 18; c.ActionSpace = actionSpace;
  return c;
}

struct Param {
  std::string name;
  std::vector<float> v;
  // glorot fan shape: dims of the reference tensor (batch-shaped for BN / bias)
  std::vector<int> ref_shape;
  int kind;  // 0 conv filter (GlorotU), 1 BN gamma/beta (GlorotN), 2 FC weight (GlorotN), 3 FC bias (zeros)
};

struct BNStats { std::vector<float> mean, var; };

struct Dual {
  DualConfig conf;
  std::vector<Param> params;   // Model() order, see include/agz.h
  std::vector<BNStats> bn;     // Dual.ops order: Init, (L1,L2) x SharedLayers, policy, value
  int HW() const { return conf.Width * conf.Height; }

  explicit Dual(const DualConfig& c) : conf(c) {
    int K = c.K, F = c.Features, H = c.Height, W = c.Width, hw = H * W, B = c.BatchSize;
    auto conv = [&](const std::string& nm, int out, int in, int k) {
      params.push_back(Param{"Filter" + nm, std::vector<float>((size_t)out * in * k * k, 0.f), {out, in, k, k}, 0});
    };
    auto bnp = [&](const std::string& nm, int C) {
      params.push_back(Param{nm + "_gamma", std::vector<float>((size_t)C * hw, 0.f), {B, C, H, W}, 1});
      params.push_back(Param{nm + "_beta", std::vector<float>((size_t)C * hw, 0.f), {B, C, H, W}, 1});
      bn.push_back(BNStats{std::vector<float>(C, 0.f), std::vector<float>(C, 1.f)});
    };
    auto fc = [&](const std::string& nm, int in, int units) {
      params.push_back(Param{nm + "_w", std::vector<float>((size_t)in * units, 0.f), {in, units}, 2});
      params.push_back(Param{nm + "_b", std::vector<float>((size_t)units, 0.f), {B, units}, 3});
    };
    conv("Init", K, F, 3); bnp("Init", K);
    for (int i = 0; i < c.SharedLayers; i++) {
      std::string s = std::to_string(i);
      conv("Layer1 of Shared Layer " + s, K, K, 3); bnp("L1_" + s, K);
      conv("Layer2 of Shared Layer " + s, K, K, 3); bnp("L2_" + s, K);
    }
    conv("PolicyHead", 2, K, 1); bnp("PolicyHead", 2);
    fc("Policy", 2 * hw, c.ActionSpace);
    conv("ValueHead", 1, K, 1); bnp("ValueHead", 1);
    fc("Value", hw, c.FC);
    fc("ValueOutput", c.FC, 1);
  }

  // Glorot et al. as gorgonia implements it [UPSTREAM-RECALLED]: fan = (s0 + s1) * prod(s[2:]),
  // stdev = sqrt(2/fan); GlorotU: U(-gain*stdev*sqrt3, +...), GlorotN: N(0, (gain*stdev)^2).
  void InitRandom(uint64_t seed) {
    SplitMix64 r(seed);
    for (Param& p : params) {
      double field = 1;
      for (size_t i = 2; i < p.ref_shape.size(); i++) field *= p.ref_shape[i];
      double fan = (double)(p.ref_shape[0] + p.ref_shape[1]) * field;
      double stdev = std::sqrt(2.0 / fan);
      if (p.kind == 0) {
        double lim = stdev * std::sqrt(3.0);
        for (float& x : p.v) x = (float)((r.float64() * 2.0 - 1.0) * lim);
      } else if (p.kind == 1 || p.kind == 2) {
        for (size_t i = 0; i < p.v.size(); i += 2) {  // Box-Muller
          double u1 = 1.0 - r.float64(), u2 = r.float64();
          double rad = std::sqrt(-2.0 * std::log(u1)), th = 6.283185307179586476925 * u2;
          p.v[i] = (float)(rad * std::cos(th) * stdev);
          if (i + 1 < p.v.size()) p.v[i + 1] = (float)(rad * std::sin(th) * stdev);
        }
      } else {
        for (float& x : p.v) x = 0.f;
      }
    }
  }

  // x: [C,H,W] -> y [Cout,H,W]; cross-correlation, zero pad (k-1)/2
  void conv2d(const std::vector<float>& x, int Cin, const std::vector<float>& w, int Cout, int k,
              std::vector<float>* y) const {
    int H = conf.Height, W = conf.Width, hw = H * W, pad = (k - 1) / 2;
    // accumulate in HWC-with-Cout-inner order so the inner loop vectorises; sum order = (ky,kx,cin)
    std::vector<float> acc((size_t)hw * Cout, 0.f);
    std::vector<float> wt((size_t)k * k * Cin * Cout);  // [ky][kx][cin][cout]
    for (int o = 0; o < Cout; o++)
      for (int c = 0; c < Cin; c++)
        for (int t = 0; t < k * k; t++) wt[((size_t)t * Cin + c) * Cout + o] = w[((size_t)o * Cin + c) * k * k + t];
    // blocks of PB pixels share each weight row (the sum order per output stays (ky,kx,cin))
    constexpr int PB = 8;
    std::vector<float> blk((size_t)Cout * PB);
    for (int p0 = 0; p0 < hw; p0 += PB) {
      int np = hw - p0 < PB ? hw - p0 : PB;
      std::fill(blk.begin(), blk.end(), 0.f);
      for (int ky = 0; ky < k; ky++)
        for (int kx = 0; kx < k; kx++) {
          const float* wp = &wt[(size_t)(ky * k + kx) * Cin * Cout];
          int off[PB];
          bool any = false;
          for (int q = 0; q < PB; q++) {
            off[q] = -1;
            if (q >= np) continue;
            int p = p0 + q, h = p / W, x0 = p % W;
            int ih = h + ky - pad, iw = x0 + kx - pad;
            if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
            off[q] = ih * W + iw;
            any = true;
          }
          if (!any) continue;
          for (int c = 0; c < Cin; c++) {
            float xv[PB];
            for (int q = 0; q < PB; q++) xv[q] = off[q] >= 0 ? x[(size_t)c * hw + off[q]] : 0.f;
            const float* wr = wp + (size_t)c * Cout;
            for (int o = 0; o < Cout; o++) {
              float wv = wr[o];
              float* bo = &blk[(size_t)o * PB];
              for (int q = 0; q < PB; q++) bo[q] += xv[q] * wv;
            }
          }
        }
      for (int q = 0; q < np; q++)
        for (int o = 0; o < Cout; o++) acc[(size_t)(p0 + q) * Cout + o] = blk[(size_t)o * PB + q];
    }
    y->assign((size_t)Cout * hw, 0.f);
    for (int p = 0; p < hw; p++)
      for (int o = 0; o < Cout; o++) (*y)[(size_t)o * hw + p] = acc[(size_t)p * Cout + o];
  }
  // BatchNorm inference + ReLU in place on [C,H,W]
  void bn_relu(std::vector<float>* x, int C, const Param& gamma, const Param& beta, const BNStats& st) const {
    int hw = HW();
    for (int c = 0; c < C; c++) {
      float mean = 0.f, inv = 1.f;
      if (conf.bn_mode == 0) { mean = 0.f; inv = 1.0f / std::sqrt(0.0f + conf.bn_eps); }
      else if (conf.bn_mode == 1) { mean = st.mean[c]; inv = 1.0f / std::sqrt(st.var[c] + conf.bn_eps); }
      for (int p = 0; p < hw; p++) {
        size_t i = (size_t)c * hw + p;
        float norm = ((*x)[i] - mean) * inv;
        float y = norm * gamma.v[i] + beta.v[i];
        (*x)[i] = y > 0.f ? y : 0.f;
      }
    }
  }
  // One board: planes [F,H,W] -> policy [ActionSpace] (softmax), value (tanh).  dual.go:50-103.
  void Infer(const float* planes, std::vector<float>* policy, float* value) const {
    int K = conf.K, F = conf.Features, hw = HW(), A = conf.ActionSpace, FCn = conf.FC;
    size_t pi = 0;
    int bi = 0;
    std::vector<float> x(planes, planes + (size_t)F * hw), y, a, b;
    conv2d(x, F, params[pi].v, K, 3, &y); bn_relu(&y, K, params[pi + 1], params[pi + 2], bn[bi]); pi += 3; bi++;
    x.swap(y);
    for (int l = 0; l < conf.SharedLayers; l++) {  // ermahagerdmonards.go:67-73: both branches read the same input
      conv2d(x, K, params[pi].v, K, 3, &a); bn_relu(&a, K, params[pi + 1], params[pi + 2], bn[bi]); pi += 3; bi++;
      conv2d(x, K, params[pi].v, K, 3, &b); bn_relu(&b, K, params[pi + 1], params[pi + 2], bn[bi]); pi += 3; bi++;
      for (size_t i = 0; i < a.size(); i++) { float s = a[i] + b[i]; x[i] = s > 0.f ? s : 0.f; }
    }
    // policy head (dual.go:72-82)
    std::vector<float> p;
    conv2d(x, K, params[pi].v, 2, 1, &p); bn_relu(&p, 2, params[pi + 1], params[pi + 2], bn[bi]); pi += 3; bi++;
    const std::vector<float>& Wp = params[pi].v; const std::vector<float>& bp = params[pi + 1].v; pi += 2;
    std::vector<float> logits(A);
    for (int j = 0; j < A; j++) {
      float s = 0.f;
      for (int i = 0; i < 2 * hw; i++) s += p[i] * Wp[(size_t)i * A + j];
      logits[j] = s + bp[j];
    }
    float mx = logits[0];
    for (int j = 1; j < A; j++) mx = logits[j] > mx ? logits[j] : mx;
    float sum = 0.f;
    policy->resize(A);
    for (int j = 0; j < A; j++) { (*policy)[j] = std::exp(logits[j] - mx); sum += (*policy)[j]; }
    for (int j = 0; j < A; j++) (*policy)[j] /= sum;
    // value head (dual.go:85-97)
    std::vector<float> v;
    conv2d(x, K, params[pi].v, 1, 1, &v); bn_relu(&v, 1, params[pi + 1], params[pi + 2], bn[bi]); pi += 3; bi++;
    const std::vector<float>& W1 = params[pi].v; const std::vector<float>& b1 = params[pi + 1].v; pi += 2;
    const std::vector<float>& W2 = params[pi].v; const std::vector<float>& b2 = params[pi + 1].v; pi += 2;
    std::vector<float> hdn(FCn);
    for (int j = 0; j < FCn; j++) {
      float s = 0.f;
      for (int i = 0; i < hw; i++) s += v[i] * W1[(size_t)i * FCn + j];
      s += b1[j];
      hdn[j] = s > 0.f ? s : 0.f;
    }
    float o = 0.f;
    for (int j = 0; j < FCn; j++) o += hdn[j] * W2[j];
    o += b2[0];
    *value = std::tanh(o);
  }
  // SURVEY App. D
  double FlopsPerEval() const {
    double K = conf.K, F = conf.Features, hw = HW(), A = conf.ActionSpace, L = conf.SharedLayers, FCn = conf.FC;
    return 2 * F * K * 9 * hw + L * 2 * (2 * K * K * 9 * hw) + (2 * K * 2 * hw + 2 * 2 * hw * A) +
           (2 * K * hw + 2 * hw * FCn + 2 * FCn);
  }
};

}  // namespace oracle
