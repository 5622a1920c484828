// ORACLE — TEST INFRASTRUCTURE ONLY (see game.hpp header).
// Restatement of agogo's Agent / Arena / encoders / dummy inferer:
//   arena.go:80-179, agent.go:60-80, dummy.go:5-25, encoding_helper.go:10-78, cmd/tictactoe/main.go:26-47,
//   mcts/example_test.go:38-72,158-166 (the scripted fake networks).
#pragma once
#include <cmath>
#include <functional>

#include "dualnet.hpp"
#include "mcts.hpp"

namespace oracle {

// encoding_helper.go:10-26
inline void EncodeTwoPlayerBoard(const std::vector<Colour>& a, float* out) {
  for (size_t i = 0; i < a.size(); i++) out[i] = a[i] == Black ? 1.f : (a[i] == White ? -1.f : 0.f);
}
// cmd/tictactoe/main.go:26-47 : F = 2
inline std::vector<float> EncodeTwoPlane(const State& s) {
  const std::vector<Colour>& b = s.Board();
  std::vector<float> r(2 * b.size(), 0.f);
  EncodeTwoPlayerBoard(b, r.data());
  for (size_t i = 0; i < b.size(); i++) if (r[i] == 0.f) r[i] = 0.001f;
  Player next = s.ToMove();
  float pl = next == Black ? 1.f : (next == White ? -1.f : 0.f);
  for (size_t i = 0; i < b.size(); i++) r[b.size() + i] = pl;
  return r;
}
// encoding_helper.go:29-68 : F = 18; the current board is never encoded, slot 7 of each half stays zero
inline std::vector<float> WQEncoder(const State& a) {
  const int lookback = 8, features = 2 * lookback + 2;
  int size = (int)a.Board().size();
  std::vector<float> r((size_t)size * features, 0.f);
  Player next = a.ToMove();
  float encodedPlayer = 1.f;
  int blackStart, whiteStart, nextStart;
  if (next == Black) { blackStart = 0; whiteStart = lookback * size; nextStart = 2 * lookback * size; }
  else { blackStart = lookback * size; whiteStart = 0; nextStart = (2 * lookback + 1) * size; encodedPlayer = -1.f; }
  int current = a.MoveNumber() - 1;
  for (int i = 1; i < lookback; i++) {
    int h = current - i;
    if (h > 0 && h < current) {
      const std::vector<Colour>& past = a.Historical(h);
      EncodeTwoPlayerBoard(past, &r[blackStart]);
      EncodeTwoPlayerBoard(past, &r[whiteStart]);
      for (int j = 0; j < size; j++) r[whiteStart + j] *= -1.f;  // vecf32.Scale(retVal, -1)
    }
    blackStart += size; whiteStart += size;
  }
  for (int i = nextStart; i < nextStart + size; i++) r[i] = encodedPlayer;
  return r;
}
typedef std::function<std::vector<float>(const State&)> GameEncoder;  // datatypes.go:28

// ---- inferencers ---------------------------------------------------------------------------------
inline uint32_t mix32(uint32_t x) {  // murmur3 finaliser
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
// AGZ_INF_HASH: synthetic deterministic network (build-defined; exact in float on CPU and GPU).
struct HashNN : Inferencer {
  int policy_len;
  explicit HashNN(int n) : policy_len(n) {}
  static uint32_t position_hash(const State& s) {
    const std::vector<Colour>& b = s.Board();
    uint32_t h = 0;
    for (size_t i = 0; i < b.size(); i++) h += mix32((uint32_t)i * 4u + (uint32_t)b[i] + 1u);
    h += mix32(0xABCD0000u + (uint32_t)s.ToMove());
    return h;
  }
  void Infer(const State& s, std::vector<float>* policy, float* value) override {
    uint32_t h = position_hash(s);
    policy->resize(policy_len);
    for (int i = 0; i < policy_len; i++)
      (*policy)[i] = (float)((mix32(h + (uint32_t)i * 0x9E3779B9u) >> 8) + 1u) * (1.0f / 16777216.0f);
    *value = (float)(mix32(h ^ 0xDEADBEEFu) >> 8) * (1.0f / 16777216.0f);
  }
};
// dummy.go:5-25
struct DummyInferer : Inferencer {
  int outputSize;
  Player currentPlayer;
  DummyInferer(int n, Player p) : outputSize(n), currentPlayer(p) {}
  void Infer(const State&, std::vector<float>* policy, float* value) override {
    *value = currentPlayer == 1 ? 1.f : (currentPlayer == 2 ? -1.f : 0.f);
    policy->assign(outputSize, 1 / (float)outputSize);
  }
};
// mcts/example_test.go:38-72 (value = 8 / 9 is Go integer-constant division = 0)
struct ScriptNN : Inferencer {
  void Infer(const State& s, std::vector<float>* policy, float* value) override {
    policy->assign(10, 0.f);
    *value = 0.f;
    switch (s.MoveNumber()) {
      case 0: (*policy)[4] = 0.9f; *value = 0.5f; break;
      case 1: (*policy)[0] = 0.1f; *value = 0.5f; break;
      case 2: (*policy)[2] = 0.9f; *value = 0.f; break;
      case 3: (*policy)[6] = 0.1f; *value = 0.f; break;
      case 4: (*policy)[3] = 0.9f; *value = 0.f; break;
      case 5: (*policy)[5] = 0.1f; *value = 0.5f; break;
      case 6: (*policy)[1] = 0.9f; *value = 0.f; break;
      case 7: (*policy)[7] = 0.1f; *value = 0.f; break;
      case 8: (*policy)[8] = 0.9f; *value = 0.f; break;
    }
  }
};
// mcts/example_test.go:158-166
struct UniformNN : Inferencer {
  int n;
  explicit UniformNN(int n_) : n(n_) {}
  void Infer(const State&, std::vector<float>* policy, float* value) override {
    policy->assign(n, 1 / 25.0f);
    *value = 1 / 25.0f;
  }
};
// Agent.Infer over a dual net: agent.go:60-74 + dualnet/meta.go:168-190
struct NetInferencer : Inferencer {
  const Dual* net;
  GameEncoder enc;
  NetInferencer(const Dual* d, GameEncoder e) : net(d), enc(e) {}
  void Infer(const State& s, std::vector<float>* policy, float* value) override {
    std::vector<float> planes = enc(s);
    net->Infer(planes.data(), policy, value);
  }
};
// test hook: evaluate through a C callback (e.g. the GPU net) so MCTS parity can be checked with identical NN outputs
typedef void (*infer_cb)(const float* planes, int n_planes, float* policy, int policy_len, float* value, void* user);
struct CallbackInferencer : Inferencer {
  infer_cb cb; void* user; GameEncoder enc; int policy_len;
  CallbackInferencer(infer_cb c, void* u, GameEncoder e, int pl) : cb(c), user(u), enc(e), policy_len(pl) {}
  void Infer(const State& s, std::vector<float>* policy, float* value) override {
    std::vector<float> planes = enc(s);
    policy->assign(policy_len, 0.f);
    cb(planes.data(), (int)planes.size(), policy->data(), policy_len, value, user);
  }
};

// datatypes.go:41-46
struct Example {
  std::vector<float> Board, Policy;
  float Value;
};

// agent.go:14-31
struct Agent {
  Inferencer* nn = nullptr;
  std::unique_ptr<MCTS> mcts;
  Player player = None;
  float Wins = 0, Loss = 0, Draw = 0;
  Single Search(StatePtr g) {  // agent.go:77-80
    mcts->SetGame(g);
    return mcts->Search(player);
  }
};

inline bool validPolicies(const std::vector<float>& p) {  // arena.go:241-251
  for (float v : p) if (std::isinf(v) || std::isnan(v)) return false;
  return true;
}

// arena.go:20-70
struct Arena {
  SplitMix64 r;
  StatePtr game;
  Agent A, B;
  Agent* currentPlayer = nullptr;
  MCTSConfig conf;
  GameEncoder enc;
  uint64_t seed;
  int max_moves;
  // per-game record kept for tests
  std::vector<Single> moves;
  std::vector<Example> examples;
  Player winner = None;
  bool ended = false;
  int passCount = 0;
  bool started = false;

  Arena(StatePtr g, const MCTSConfig& c, GameEncoder e, uint64_t seed_, int max_moves_)
      : r(seed_), game(g), conf(c), enc(e), seed(seed_), max_moves(max_moves_) {}

  void newTrees() {  // arena.go:49,55,175-176 (mcts.New)
    A.mcts.reset(new MCTS(game, conf, A.nn, seed * 2 + 1));
    B.mcts.reset(new MCTS(game, conf, B.nn, seed * 2 + 2));
  }
  void switchPlayer() { currentPlayer = currentPlayer == &A ? &B : &A; }  // arena.go:226-233

  // arena.go:81-92. a_is_black < 0 draws it (a.r.Intn(2) == 0 -> A is Black).
  void Begin(int a_is_black) {
    bool ab = a_is_black < 0 ? ((r.next() >> 63) == 0) : (a_is_black != 0);
    if (ab) { A.player = Black; B.player = White; currentPlayer = &A; }
    else { A.player = White; B.player = Black; currentPlayer = &B; }
    game->SetToMove(currentPlayer->player);
    newTrees();
    moves.clear(); examples.clear(); winner = None; ended = false; passCount = 0; started = true;
    ended = game->Ended(&winner);
  }
  // one iteration of the loop body arena.go:96-138; returns false when the game is over
  // external != nullptr: the move comes from outside (the opponent of a tournament game: the caller of Agent.Search
  // applies it to its game.State, agent.go:76-81) — no search, no example; returns false and changes nothing if illegal.
  bool Step(bool record, const Single* external = nullptr) {
    if (ended) return false;
    if (external && *external != Resign && !game->Check(PlayerMove{currentPlayer->player, *external})) return false;
    Single best = external ? *external : currentPlayer->Search(game);
    if (best == Pass) passCount++; else passCount = 0;
    if (record && !external) {
      Example ex;
      ex.Board = enc(*game);
      ex.Policy = currentPlayer->mcts->Policies(*game);
      ex.Value = (float)currentPlayer->player;
      if (validPolicies(ex.Policy)) examples.push_back(ex);
    }
    moves.push_back(best);
    bool resigned = best == Resign;
    if (!resigned) game = game->Apply(PlayerMove{currentPlayer->player, best});
    Player mover = currentPlayer->player;
    switchPlayer();
    if (resigned) {  // build-defined: the reference would index board[-2] (mnk.go:129)
      ended = true; winner = Opponent(mover);
    } else if (passCount >= 2) {
      // arena.go:135 breaks WITHOUT re-evaluating Ended(): winner stays None — except wq, whose Game is a
      // completion: two passes end the game and the area score decides.
      ended = true;
      if (game->Kind() == 3) game->Ended(&winner); else winner = None;
    } else if (max_moves > 0 && (int)moves.size() >= max_moves) {
      ended = true;  // build-defined safety cap (no ko rule in the reference)
      if (game->Kind() == 3) { Player w; game->Ended(&w); float wb = game->Score(Black), ww = game->Score(White) + game->AdditionalScore(); winner = wb > ww ? Black : (ww > wb ? White : None); }
      else winner = None;
    } else {
      ended = game->Ended(&winner);
    }
    if (ended) Finish();
    return !ended;
  }
  void Finish() {  // arena.go:146-171
    for (Example& ex : examples) {
      if (winner == None) ex.Value = 0;
      else if (ex.Value == (float)winner) ex.Value = 1;
      else ex.Value = -1;
    }
    if (winner == None) { A.Draw++; B.Draw++; }
    else if (winner == A.player) { A.Wins++; B.Loss++; }
    else if (winner == B.player) { B.Wins++; A.Loss++; }
  }
};

}  // namespace oracle
