// ORACLE — TEST INFRASTRUCTURE ONLY (see game.hpp header).
// Sequential, deterministic restatement of gorgonia/agogo's mcts package.
//
// Declared substitutions, everything else follows the cited lines:
//   q1  Timeout -> exactly Budget pipeline iterations (search.go:132-133 is wall-clock bound).
//   q2  one searchState, no goroutines: virtual loss is never observable and is omitted.
//   q3  RNG = SplitMix64 with an explicit seed (Go math/rand is not reproducible here).
//   q4  sort.Sort -> stable sort (score desc, insertion order) for byScore and fancySort.
// Arithmetic is float32 in the reference's operation order; build with -ffp-contract=off.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <limits>
#include <map>
#include <vector>

#include "game.hpp"

namespace oracle {

// mcts/mcts.go:15-18
struct Inferencer {
  virtual ~Inferencer() {}
  virtual void Infer(const State& state, std::vector<float>* policy, float* value) = 0;
};

// mcts/mcts.go:30-38
enum PassPreference { DontPreferPass = 0, PreferPass = 1, DontResign = 2 };

// mcts/tree.go:15-29
struct MCTSConfig {
  float PUCT = 1.0f;
  int M = 0, N = 0;
  int RandomCount = 0;
  int32_t Budget = 10000;
  uint32_t RandomMinVisits = 0;
  float RandomTemperature = 0;
  bool DumbPass = true;
  float ResignPercentage = 0;
  int PassPreference = DontPreferPass;
  // BUILD EXTENSION (not in mcts.Config): simulations of one tree run in rounds of `Parallel` lanes whose leaves are
  // evaluated as one batch — the deterministic, lane-ordered form of the reference's NumCPU goroutines sharing a tree
  // with a stored virtual loss (search.go:112-131, node.go:248-260).  1 = the sequential search (declared semantics).
  int Parallel = 1;
  bool IsValid() const { return PUCT > 0 && PUCT <= 1; }  // tree.go:42-44
};

typedef int naughty;  // mcts/naughty.go
static const naughty nilNode = -1;

enum Status : uint32_t { Invalid = 0, Active = 1, Pruned = 2 };  // node.go:12-18

// mcts/node.go:32-49
struct Node {
  int32_t move = 0;
  uint32_t visits = 0;
  uint32_t status = 0;
  float blackScores = 0;
  float minPSARatioChildren = 2.0f;  // defaultMinPsaRatio, mcts.go:28
  float score = 0;
  float value = 0;
  float virtualLoss = 0;  // node.go:41, set to virtualLoss1 = 3.0 by addVirtualLoss, 0 by undoVirtualLoss (mcts.go:27)
  naughty id = 0;
};

struct MCTS {
  MCTSConfig conf;
  Inferencer* nn;
  SplitMix64 rand;
  std::vector<Node> nodes;
  std::vector<std::vector<naughty>> children;
  std::vector<naughty> freelist, freeables;
  // searchState (search.go:53-63)
  StatePtr current, prev;
  naughty root = nilNode;
  int depth = 0, maxDepth = 0;
  int32_t playouts = 0, nc = 0;
  int32_t lastIter = 0;       // iterations of the last Search (iter, search.go:120)
  int64_t nnEvals = 0;
  std::map<std::pair<uint32_t, Single>, float> cachedPolicies;  // tree.go:75

  MCTS(StatePtr game, const MCTSConfig& c, Inferencer* nn_, uint64_t seed)  // tree.go:80-103
      : conf(c), nn(nn_), rand(seed), current(game) {
    maxDepth = conf.M * conf.N;
  }

  static const int MAXTREESIZE = 25000000;  // search.go:23

  Node& N(naughty n) { return nodes.at(n); }

  // tree.go:145-167
  naughty alloc() {
    if (freelist.empty()) {
      Node nd;
      nd.id = (naughty)nodes.size();
      nd.minPSARatioChildren = 2.0f;
      nodes.push_back(nd);
      children.emplace_back();
      return (naughty)nodes.size() - 1;
    }
    naughty i = freelist.back();
    freelist.pop_back();
    return i;
  }
  // tree.go:106-117
  naughty New(Single move, float score, float value) {
    naughty n = alloc();
    Node& nd = N(n);
    nd.move = move;
    nd.visits = 1;
    nd.status = Active;
    nd.score = score;
    nd.value = value;
    return n;
  }
  // tree.go:174-180 + node.go:298-307
  void free_(naughty n) {
    children.at(n).clear();
    freelist.push_back(n);
    Node& nd = N(n);
    nd.move = -1; nd.visits = 0; nd.status = 0; nd.blackScores = 0;
    nd.minPSARatioChildren = 2.0f; nd.score = 0; nd.value = 0; nd.virtualLoss = 0;
  }
  void SetGame(StatePtr g) { current = g; }  // tree.go:120-124
  int Nodes() const { return (int)nodes.size(); }

  // node.go
  bool HasChildren(const Node& n) const { return n.minPSARatioChildren <= 1; }                 // :129
  bool IsExpandable(const Node& n, float r) const { return r < n.minPSARatioChildren; }        // :132
  static float Evaluate(const Node& n, Player player) {                                        // :147-159
    float bs = n.blackScores;
    if (player == White) bs += n.virtualLoss;  // node.go:150-152 (0 whenever observable in the sequential search, q2)
    float score = bs / (float)n.visits;
    if (player == White) score = 1 - score;
    return score;
  }
  static float NNEvaluate(const Node& n, Player player) { return player == White ? 1.0f - n.value : n.value; }  // :162-167
  void Update(Node& n, float score) { n.visits += 1; n.blackScores = n.blackScores + score; }  // :70-76,263-270

  // node.go:170-237
  naughty Select(naughty parent, Player of) {
    Node& n = N(parent);
    uint32_t parentVisits = 0;
    const std::vector<naughty>& kids = children.at(n.id);
    for (naughty kid : kids) {
      const Node& child = N(kid);
      if (child.status != Invalid) parentVisits += child.visits;
    }
    naughty best = nilNode;
    float bestValue = -std::numeric_limits<float>::infinity();
    float fpu = NNEvaluate(n, of);
    float numerator = std::sqrt((float)parentVisits);
    for (naughty kid : kids) {
      const Node& child = N(kid);
      if (child.status != Active) continue;
      float qsa = fpu;
      uint32_t visits = child.visits;
      if (visits > 0) qsa = Evaluate(child, of);
      float psa = child.score;
      float denominator = 1.0f + (float)visits;
      float lastTerm = numerator / denominator;
      float puct = conf.PUCT * psa * lastTerm;
      float usa = qsa + puct;
      if (usa > bestValue) { bestValue = usa; best = kid; }
    }
    if (best == nilNode) throw std::runtime_error("Cannot return nil");
    return best;
  }
  naughty findChild(naughty parent, Single move) {  // node.go:285-296
    for (naughty kid : children.at(parent)) if (N(kid).move == move) return kid;
    return nilNode;
  }
  int countChildren(naughty n) {  // node.go:272-283
    int r = 0;
    for (naughty kid : children.at(n)) {
      if (N(kid).status == Active) r += countChildren(kid);
      r++;
    }
    return r;
  }

  float minPsaRatio() const {  // search.go:81-90
    float ratio = (float)nc / (float)MAXTREESIZE;
    if (ratio > 0.95f) return 0.01f;
    if (ratio > 0.5f) return 0.001f;
    return 0;
  }
  static float combinedScore(const State& s) {  // utils.go:62-67
    float w = s.Score(White), b = s.Score(Black), komi = s.AdditionalScore();
    return b - w - komi;
  }

  struct pair { Single Coord; float Score; };  // utils.go:50-53

  // search.go:259-339
  bool expandAndSimulate(naughty parent, State& state, float minPsaRatio_, float* value_out) {
    *value_out = 0;
    if (!IsExpandable(N(parent), minPsaRatio_)) return false;
    if (state.Passes() >= 2) return false;
    std::vector<float> policy;
    float value;
    nn->Infer(state, &policy, &value);
    nnEvals++;
    float passProb = policy.at(policy.size() - 1);
    Player player = state.ToMove();
    if (player == White) value = 1 - value;
    std::vector<pair> nodelist;
    float legalSum = 0;
    int A = current->ActionSpace();
    for (int i = 0; i < A; i++) {
      if (state.Check(PlayerMove{player, i})) {
        nodelist.push_back(pair{i, policy.at(i)});
        legalSum += policy[i];
      }
    }
    if (state.Check(PlayerMove{player, Pass})) {
      nodelist.push_back(pair{Pass, passProb});
      legalSum += passProb;
    }
    if (legalSum > 1e-45f /* math32.SmallestNonzeroFloat32 */) {
      for (auto& p : nodelist) p.Score /= legalSum;
    } else {
      float prob = 1 / (float)nodelist.size();
      for (auto& p : nodelist) p.Score = prob;
    }
    *value_out = value;
    if (nodelist.empty()) return true;
    std::stable_sort(nodelist.begin(), nodelist.end(), [](const pair& a, const pair& b) { return a.Score > b.Score; });  // q4
    float maxPsa = nodelist[0].Score;
    float oldMinPsa = maxPsa * N(parent).minPSARatioChildren;
    float newMinPsa = maxPsa * minPsaRatio_;
    bool skippedChildren = false;
    for (const pair& p : nodelist) {
      if (p.Score < newMinPsa) {
        skippedChildren = true;
      } else if (p.Score < oldMinPsa) {
        if (findChild(parent, p.Coord) == nilNode) {
          naughty nn_ = New(p.Coord, p.Score, value);
          children.at(parent).push_back(nn_);
        }
      }
    }
    N(parent).minPSARatioChildren = skippedChildren ? minPsaRatio_ : 0.0f;
    return true;
  }

  // search.go:209-257. Result: NaN-tagged null (search.go:39-51) -> here (ok=false).
  bool pipeline(StatePtr cur, naughty start, float* result) {
    depth++;
    if (depth > maxDepth) { depth--; return false; }
    Player player = cur->ToMove();
    bool have = false;
    float ret = 0;
    bool isExpandable = IsExpandable(N(start), 0);
    if (isExpandable && cur->Passes() >= 2) {
      ret = combinedScore(*cur);
      have = true;
    } else if (isExpandable && nc < MAXTREESIZE) {
      bool hadChildren = HasChildren(N(start));
      float value;
      bool ok = expandAndSimulate(start, *cur, minPsaRatio(), &value);
      if (!hadChildren && ok) { ret = value; have = true; }
    }
    if (HasChildren(N(start)) && !have) {
      naughty next = Select(start, player);
      Single move = N(next).move;
      PlayerMove pm{player, move};
      if (cur->Check(pm)) {
        cur = cur->Apply(pm);
        have = pipeline(cur, next, &ret);
      }
    }
    if (have) Update(N(start), ret);
    depth--;
    *result = ret;
    return have;
  }

  // One round of `lanes` simulations on this tree (MCTSConfig::Parallel).  Phase 1: the lanes descend one after the
  // other; every node entered gets addVirtualLoss (a STORE of 3.0, node.go:248-253) which stays until that lane's own
  // backup — later lanes of the round see it through Evaluate (only White's evaluation includes it, node.go:150-152,
  // reproduced as is).  A lane stops at the first expandable node (the leaf), at a two-pass terminal, or at the depth
  // cap.  Phase 2, in lane order: the leaf is expanded with its network evaluation (one batch on the device) unless an
  // earlier lane of the round already expanded the very same node — then this lane is the reference's second goroutine
  // arriving at a node whose expansion is in flight: hadChildren was false for it too, its own Infer of the same state
  // returns the same value, findChild stops it adding the children twice (search.go:229-234,318-323) — the value is
  // backed up again; then Update along the path and undoVirtualLoss.  Returns the number of non-null results.
  int parallelRound(int lanes) {
    struct Lane { std::vector<naughty> path; StatePtr state; int kind = 0; float result = 0; };  // kind 0 null, 1 expand, 2 terminal
    std::vector<Lane> L(lanes);
    for (int l = 0; l < lanes; l++) {
      Lane& ln = L[l];
      StatePtr cur = current->Clone();
      naughty node = root;
      int dep = 0;
      while (true) {
        dep++;
        if (dep > maxDepth) { ln.kind = 0; break; }           // search.go:211-215 (before addVirtualLoss)
        ln.path.push_back(node);
        N(node).virtualLoss = 3.0f;
        Player player = cur->ToMove();
        bool isExpandable = IsExpandable(N(node), 0);
        if (isExpandable && cur->Passes() >= 2) { ln.kind = 2; ln.result = combinedScore(*cur); break; }
        if (isExpandable && nc < MAXTREESIZE) { ln.kind = 1; ln.state = cur; break; }
        if (!HasChildren(N(node))) { ln.kind = 0; break; }
        naughty next = Select(node, player);
        PlayerMove pm{player, N(next).move};
        if (!cur->Check(pm)) { ln.kind = 0; break; }
        cur = cur->Apply(pm);
        node = next;
      }
    }
    int nonnull = 0;
    std::map<naughty, float> expanded;   // leaf -> value, this round
    for (int l = 0; l < lanes; l++) {
      Lane& ln = L[l];
      bool have = false;
      float ret = 0;
      if (ln.kind == 2) { have = true; ret = ln.result; }
      else if (ln.kind == 1) {
        naughty leaf = ln.path.back();
        auto it = expanded.find(leaf);
        if (it != expanded.end()) { have = true; ret = it->second; }
        else {
          float value;
          bool ok = expandAndSimulate(leaf, *ln.state, minPsaRatio(), &value);
          if (ok) { have = true; ret = value; expanded[leaf] = value; }
        }
      }
      for (size_t j = ln.path.size(); j-- > 0;) {
        Node& nd = N(ln.path[j]);
        if (have) Update(nd, ret);
        nd.virtualLoss = 0.0f;
      }
      if (have) nonnull++;
    }
    return nonnull;
  }

  // tree.go:183-209
  void cleanChildren(naughty r) {
    for (naughty kid : children.at(r)) {
      N(kid).status = Invalid;
      freeables.push_back(kid);
      cleanChildren(kid);
    }
    children.at(r).clear();
  }
  void cleanup(naughty oldRoot, naughty newRoot) {
    for (naughty kid : children.at(oldRoot)) {
      if (kid != newRoot) {
        N(kid).status = Invalid;
        freeables.push_back(kid);
        cleanChildren(kid);
      }
    }
    children.at(oldRoot).assign(1, newRoot);
  }

  // search.go:424-469
  bool newRootState() {
    if (root == nilNode || !prev) return false;
    int d = current->MoveNumber() - prev->MoveNumber();
    if (d < 0) return false;
    StatePtr tmp = current->Clone();
    for (int i = 0; i < d; i++) tmp->UndoLastMove();
    if (!tmp->Eq(prev.get())) return false;
    for (int i = 0; i < d; i++) {
      tmp->Fwd();
      PlayerMove move = tmp->LastMove();
      naughty oldRoot = root;
      naughty newRoot = findChild(oldRoot, move.move);
      if (newRoot == nilNode) return false;
      root = newRoot;
      cleanup(oldRoot, newRoot);
      prev = prev->Apply(move);
    }
    if (current->MoveNumber() != prev->MoveNumber()) return false;
    if (!current->Eq(prev.get())) return false;
    return true;
  }
  // search.go:473-500
  void updateRoot() {
    freeables.clear();
    Player player = current->ToMove();
    if (!newRootState() || root == nilNode) {
      if (current->Check(PlayerMove{player, Pass})) {
        root = New(Pass, 0, 0);
      } else {
        for (int i = 0; i < current->ActionSpace(); i++) {
          if (current->Check(PlayerMove{player, i})) { root = New(i, 0, 0); break; }
        }
      }
    }
    prev = nullptr;
    nc = (int32_t)countChildren(root);
    if (children.at(root).empty()) N(root).minPSARatioChildren = 2.0f;
  }
  // search.go:392-408
  void prepareRoot(Player player, State& state) {
    bool hadChildren = !children.at(root).empty();
    bool expandable = IsExpandable(N(root), 0);
    float value = 0;
    if (expandable) expandAndSimulate(root, state, minPsaRatio(), &value);
    if (!hadChildren) Update(N(root), value);
  }

  // utils.go:18-47 with a stable sort (q4)
  void fancySort(std::vector<naughty>& l, Player underEval) {
    std::stable_sort(l.begin(), l.end(), [&](naughty a, naughty b) {
      const Node& li = N(a);
      const Node& lj = N(b);
      if (li.visits != lj.visits) return li.visits > lj.visits;
      if (li.visits == 0) return li.score > lj.score;
      return Evaluate(li, underEval) > Evaluate(lj, underEval);
    });
  }
  // tree.go:212-247
  void randomizeChildren(naughty of) {
    float accum = 0, norm = 0;
    std::vector<float> accumVector;
    std::vector<naughty>& kids = children.at(of);
    for (naughty kid : kids) {
      uint32_t visits = N(kid).visits;
      if (norm == 0) {
        norm = (float)visits;
        if (visits <= conf.RandomMinVisits) return;
      }
      if (visits > conf.RandomMinVisits) {
        accum += (float)std::pow((double)((float)visits / norm), (double)(1 / conf.RandomTemperature));   // math32.Pow = float32(math.Pow(float64, float64))
        accumVector.push_back(accum);
      }
    }
    float rnd = rand.float32() * accum;
    int index = 0;
    for (size_t i = 0; i < accumVector.size(); i++)
      if (rnd < accumVector[i]) { index = (int)i; break; }
    if (index == 0) return;
    for (int i = 0; i < (int)kids.size() - index; i++) std::swap(kids[i], kids[i + index]);
  }
  // search.go:530-563
  naughty noPass(naughty of, State& state, Player player) {
    for (naughty kid : children.at(of)) {
      Single move = N(kid).move;
      bool ok = state.Check(PlayerMove{player, move});
      if (move != Pass && ok) return kid;
    }
    return nilNode;
  }
  void noPassBestMove(Single* bestMove, float* bestScore, naughty of, State& state, Player player) {
    naughty np = noPass(of, state, player);
    if (np >= 0) {
      *bestMove = N(np).move;
      *bestScore = 1;
      if (N(np).visits != 0) *bestScore = Evaluate(N(np), player);
    }
  }
  bool shouldResign(float bestScore, Player) {  // search.go:502-528
    if (conf.PassPreference == DontResign) return false;
    if (conf.ResignPercentage == 0) return false;
    int squares = conf.M * conf.N;
    int threshold = squares / 4;
    if (current->MoveNumber() <= threshold) return false;
    float resignThreshold = conf.ResignPercentage < 0 ? 0.1f : conf.ResignPercentage;
    if (bestScore > resignThreshold) return false;
    return true;
  }
  // search.go:341-390
  Single bestMove() {
    Player player = current->ToMove();
    int moveNum = current->MoveNumber();
    std::vector<naughty>& kids = children.at(root);
    fancySort(kids, player);
    if (moveNum < conf.RandomCount) randomizeChildren(root);
    if (kids.empty()) return Pass;
    const Node& first = N(kids[0]);
    Single best = first.move;
    float bestScore = Evaluate(first, player);
    const Node& rootN = N(root);
    if (conf.PassPreference == DontPreferPass && best == Pass) {
      noPassBestMove(&best, &bestScore, root, *current, player);
    } else if (!conf.DumbPass && best == Pass) {
      float score = rootN.score;
      if ((score > 0 && player == White) || (score < 0 && player == Black))
        noPassBestMove(&best, &bestScore, root, *current, player);
    } else if (!conf.DumbPass && current->LastMove().move == Pass) {
      float score = rootN.score;
      if ((score > 0 && player == White) || (score < 0 && player == Black)) {
      } else {
        best = Pass;
      }
    }
    if (best == Pass && shouldResign(bestScore, player)) best = Resign;
    return best;
  }

  static int argmax(const std::vector<float>& a) {  // utils.go:84-94
    int r = 0;
    float mx = -std::numeric_limits<float>::infinity();
    for (size_t i = 0; i < a.size(); i++) if (a[i] > mx) { mx = a[i]; r = (int)i; }
    return r;
  }

  // search.go:92-164
  Single Search(Player player) {
    updateRoot();
    current->SetToMove(player);
    uint32_t boardHash = current->Hash();
    for (naughty f : freeables) free_(f);
    prepareRoot(player, *current);
    int32_t iter = 0;
    // q1: exactly Budget iterations of doSearch's body (search.go:170-181)
    if (conf.Parallel > 1) {
      while (iter < conf.Budget) {
        int lanes = std::min<int32_t>(conf.Parallel, conf.Budget - iter);
        playouts += parallelRound(lanes);
        iter += lanes;
      }
    } else
    for (; iter < conf.Budget; iter++) {
      StatePtr cur = current->Clone();
      float res;
      depth = 0;
      if (pipeline(cur, root, &res)) playouts++;
    }
    lastIter = iter;
    if (!HasChildren(N(root))) {  // search.go:141-149
      std::vector<float> policy; float v;
      nn->Infer(*current, &policy, &v);
      nnEvals++;
      int moveID = argmax(policy);
      if (moveID > current->ActionSpace()) return Pass;
      return moveID;
    }
    Single ret = bestMove();
    prev = current->Clone();
    cachedPolicies[{boardHash, ret}] += 1;
    return ret;
  }

  // tree.go:128-142
  std::vector<float> Policies(const State& g) {
    uint32_t hash = g.Hash();
    float sum = 0;
    int n = g.ActionSpace() + 1;
    std::vector<float> ret(n);
    for (int i = 0; i < n; i++) {
      auto it = cachedPolicies.find({hash, (Single)i});
      float prob = it == cachedPolicies.end() ? 0.0f : it->second;
      ret[i] = prob;
      sum += prob;
    }
    for (auto& r : ret) r /= sum;
    return ret;
  }
};

}  // namespace oracle
