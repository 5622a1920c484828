// ORACLE (test infrastructure only — see oracle/README or DESIGN.md §6; never linked into the product).
// CPU restatement of the example plumbing between self-play and dual.Train:
//   RotateBoard        encoding_helper.go:80-107  (in-place ring-by-ring 4-cycle swaps on an m x m float board)
//   Augmenter          datatypes.go:38-39, applied per recorded example at arena.go:115-120; the reference ships no
//                      Augmenter — the rotation augmenter restated here is [e, rot(e), rot^2(e), rot^3(e)] with
//                      RotateBoard applied to every plane of Board and to the m*n board part of Policy (pass kept)
//   shuffleExamples    agogo.go:251-257  (for i: j = r.Intn(i+1); swap) — RNG: SplitMix64 (declared deviation q3)
//   maxExamples cut    agogo.go:118-121
//   prepareExamples    agogo.go:211-249  (shuffle, batches = n / BatchSize, keep batches*BatchSize rows, tensorise)
#pragma once
#include <cstdint>
#include <stdexcept>
#include <utility>
#include <vector>

#include "game.hpp"

namespace oracle {

// encoding_helper.go:80-107.  Returns false (the reference returns an error) when m != n.
inline bool RotateBoard(const float* board, int m, int n, float* out) {
  if (m != n) return false;
  for (int i = 0; i < m * n; i++) out[i] = board[i];
  auto at = [&](int r, int c) -> float& { return out[r * n + c]; };
  for (int i = 0; i < m / 2; i++) {
    int mi1 = m - i - 1;
    for (int j = i; j < mi1; j++) {
      int mj1 = m - j - 1;
      float tmp = at(i, j);
      at(i, j) = at(j, mi1);        // right to top
      at(j, mi1) = at(mi1, mj1);    // bottom to right
      at(mi1, mj1) = at(mj1, i);    // left to bottom
      at(mj1, i) = tmp;             // tmp is left
    }
  }
  return true;
}

struct ExampleSet {
  int F, m, n, A1;  // A1 = policy length (board moves + pass, arena.go:107 / agogo.go:243)
  std::vector<std::vector<float>> board, policy;
  std::vector<float> value;
  size_t size() const { return value.size(); }
  void push(const float* b, const float* p, float v) {
    board.emplace_back(b, b + (size_t)F * m * n);
    policy.emplace_back(p, p + A1);
    value.push_back(v);
  }
  void swap_rows(size_t i, size_t j) {
    std::swap(board[i], board[j]); std::swap(policy[i], policy[j]); std::swap(value[i], value[j]);
  }
  // rotation Augmenter applied to every example in order (arena.go:115-120: examples = append(examples, aug(ex)...))
  bool AugmentRotate() {
    if (m != n) return false;
    ExampleSet o{F, m, n, A1, {}, {}, {}};
    for (size_t e = 0; e < size(); e++) {
      std::vector<float> b = board[e], p = policy[e];
      o.push(b.data(), p.data(), value[e]);
      for (int r = 0; r < 3; r++) {
        std::vector<float> nb(b.size()), np(p);
        for (int c = 0; c < F; c++) RotateBoard(&b[(size_t)c * m * n], m, n, &nb[(size_t)c * m * n]);
        RotateBoard(p.data(), m, n, np.data());  // entries >= m*n (pass) stay
        o.push(nb.data(), np.data(), value[e]);
        b.swap(nb); p.swap(np);
      }
    }
    *this = std::move(o);
    return true;
  }
  void Shuffle(SplitMix64& r) {  // agogo.go:251-257
    for (size_t i = 0; i < size(); i++) { size_t j = (size_t)(r.next() % (uint64_t)(i + 1)); swap_rows(i, j); }
  }
  // agogo.go:118-121 then prepareExamples agogo.go:211-249.  Returns batches; Xs/Pi/V hold batches*BatchSize rows.
  int Prepare(int BatchSize, int maxExamples, uint64_t seed, std::vector<float>* Xs, std::vector<float>* Pi, std::vector<float>* V) {
    SplitMix64 r(seed);
    if (maxExamples > 0 && (int)size() > maxExamples) {
      Shuffle(r);
      board.resize(maxExamples); policy.resize(maxExamples); value.resize(maxExamples);
    }
    Shuffle(r);
    int batches = (int)(size() / (size_t)BatchSize);
    size_t total = (size_t)batches * BatchSize;
    Xs->clear(); Pi->clear(); V->clear();
    for (size_t i = 0; i < total; i++) {
      Xs->insert(Xs->end(), board[i].begin(), board[i].end());
      Pi->insert(Pi->end(), policy[i].begin(), policy[i].end());
      V->push_back(value[i]);
    }
    return batches;
  }
};

}  // namespace oracle
