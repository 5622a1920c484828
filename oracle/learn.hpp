// ORACLE (test infrastructure only — DESIGN.md section 6; never linked into the product).
// CPU restatement of the AZ.Learn composition, SURVEY 8(f)-2:
//   AZ.Learn            agogo.go:100-172   epochs of { setupSelfPlay, episodes x SelfPlay, maxExamples cut, prepareExamples,
//                                           dual.Train(B), B.SwitchToInference, resetStats, arenaGames x Play(false), gating, update, newB }
//   setupSelfPlay       agogo.go:75-90     SwitchToInference for A and B; epoch 0 && useDummy: both agents take the dummyInferer
//   gating              agogo.go:155-165   B.Wins / (B.Wins + A.Wins) > updateThreshold (float32; 0/0 = NaN compares false) -> A.NN = B.NN
//   Statistics.update   statistics.go:27-38 per network (keyed by its identity) the A-side's Wins / Loss / Draw of every epoch it was A
//   newB                arena.go:205-224   a freshly initialised network for B every epoch (the Clone branch is commented out upstream)
//   dual.Train          dualnet/meta.go:16-54   iterations x batches, cost of the last batch, shuffleBatch after every iteration
//   SwitchToInference   agent.go:42-57, dualnet/meta.go:125-162: every learnable copied with Go `copy` (min length): row 0 of the
//                       batch-shaped ones
// on top of the oracle's Arena (one game), ExampleSet and TrainNet.  Seeds, the per-game colour draw and the order of the games
// follow the build's declared RNG substitution (agogo_amd/host/agogo.hpp states the same numbers): arena of epoch
// e with G games = oracle Arena(seed + 1000 e (+ 500 for the evaluation games) + g) for game g, colours by the SplitMix64
// finaliser of the arena's second reset (its creation draws once, Play draws again).
// Test hook (not in the reference): the inferencers of the evaluation games (and of self-play after epoch 0) can be replaced by
// the synthetic ones (hash / uniform / dummy), so that a comparison with the device path cannot be flipped by fp32 rounding in a
// network evaluation; inferencer kind 0 plays the networks as the reference does.
#pragma once
#include <map>
#include <memory>

#include "arena.hpp"
#include "examples.hpp"
#include "train.hpp"

namespace oracle {

struct LearnConfig {
  int kind = 0, m = 3, n = 3, k = 3;    // game (make_game's kinds: AGZ_GAME_*)
  double komi = 0;
  int enc = 0;                          // AGZ_ENC_*
  DualConfig nn;
  MCTSConfig mc;
  float UpdateThreshold = 0.52f;
  int MaxExamples = 0;
  bool AugmentRotate = false;
  int sp_inf[2] = {0, 0};               // self-play inferencers when the dummy is not in use (0 = the networks)
  int eval_inf[2] = {0, 0};             // evaluation-game inferencers (0 = the networks)
  int max_moves = 0;
  uint64_t seed = 1337;
};

struct EpochLog {
  int epoch = 0;
  size_t examples = 0;
  int batches = 0;
  float cost = 0;
  float a_wins = 0, a_loss = 0, a_draw = 0, b_wins = 0, b_loss = 0, b_draw = 0;
  bool killedA = false;
  int a_id = 0;                         // identity of the network A holds after the epoch (serial number: the reference prints %p)
};

// statistics.go:10-38 (the key is the network's identity; the reference formats its pointer)
struct Statistics {
  std::vector<int> Creation;
  std::map<int, std::vector<float>> Wins, Losses, Draws;
  void update(int a_id, float wins, float loss, float draw) {
    if (!Wins.count(a_id)) Creation.push_back(a_id);
    Wins[a_id].push_back(wins); Losses[a_id].push_back(loss); Draws[a_id].push_back(draw);
  }
};

// the game factory and encoder table (AGZ_GAME_* / AGZ_ENC_*), as oracle_capi.cpp has them
inline StatePtr learn_make_game(int kind, int m, int n, int k, double komi) {
  switch (kind) {
    case 0: return std::make_shared<MNK>(m, n, k);
    case 1: return std::make_shared<C4>(m, n, k);
    case 2: return std::make_shared<Komi>(m, n, k);
    case 3: return std::make_shared<WQ>(m, 0, komi);
  }
  return nullptr;
}
inline GameEncoder learn_make_enc(int enc) {
  if (enc == 1) return [](const State& s) { return WQEncoder(s); };
  return [](const State& s) { return EncodeTwoPlane(s); };
}

struct Learner {
  LearnConfig cf;
  std::unique_ptr<TrainNet<float>> A, B;     // Agent.NN
  std::unique_ptr<Dual> infA, infB;          // the inference networks SwitchToInference fills
  int a_id = 1, b_id = 2, next_id = 3;
  bool useDummy = true;
  std::vector<EpochLog> log;
  Statistics stats;

  explicit Learner(const LearnConfig& c) : cf(c) {  // agogo.New, agogo.go:41-72: two networks, both initialised
    A.reset(new TrainNet<float>(cf.nn)); A->InitRandom(cf.seed * 3 + 1);
    B.reset(new TrainNet<float>(cf.nn)); B->InitRandom(cf.seed * 3 + 2);
    infA.reset(new Dual(cf.nn)); infB.reset(new Dual(cf.nn));
  }

  // meta.go:141-146: copy(dst, src) per learnable — the first len(dst) values = row 0 of the batch-shaped tensors
  static void SwitchToInference(const TrainNet<float>& t, Dual& d) {
    for (size_t i = 0; i < d.params.size(); i++) {
      const size_t n = std::min(d.params[i].v.size(), t.P[i].size());
      std::copy(t.P[i].begin(), t.P[i].begin() + n, d.params[i].v.begin());
    }
  }

  // dual.Train, meta.go:16-54 (learn rate 0.1, vanilla solver); shuffleBatch (meta.go:57-102) as a Fisher-Yates over the rows
  static float Train(TrainNet<float>& d, std::vector<float>& Xs, std::vector<float>& Pi, std::vector<float>& V, int batches, int iterations, uint64_t seed) {
    const size_t xs = (size_t)d.F * d.HW, ps = (size_t)d.A, n = (size_t)batches * d.B;
    SplitMix64 rng(seed);
    float cost = 0;
    std::vector<float> tmp(std::max(xs, ps));
    for (int it = 0; it < iterations; it++) {
      for (int b = 0; b < batches; b++) {
        const size_t s0 = (size_t)b * d.B;
        cost = d.ForwardBackward(&Xs[s0 * xs], &Pi[s0 * ps], &V[s0]);
        d.Step(0.1f);
      }
      for (size_t i = 0; i < n; i++) {
        const size_t j = (size_t)(rng.next() % (uint64_t)(i + 1));
        if (j == i) continue;
        std::swap_ranges(Xs.begin() + i * xs, Xs.begin() + (i + 1) * xs, Xs.begin() + j * xs);
        std::swap_ranges(Pi.begin() + i * ps, Pi.begin() + (i + 1) * ps, Pi.begin() + j * ps);
        std::swap(V[i], V[j]);
      }
    }
    return cost;
  }

  Inferencer* make_inf(int kind, Dual* net, std::vector<std::unique_ptr<Inferencer>>& keep, int action_space) {
    Inferencer* p = nullptr;
    switch (kind) {
      case 0: p = new NetInferencer(net, learn_make_enc(cf.enc)); break;
      case 1: p = new DummyInferer(action_space, None); break;   // useDummy captures Agent.Player before any colour is drawn (agent.go:105-113)
      case 3: p = new HashNN(action_space + 1); break;
      default: p = new UniformNN(25); break;
    }
    keep.emplace_back(p);
    return p;
  }

  // G games of one batched arena of the device path: game g = Arena(arena_seed + g), its colours from the arena's second reset
  struct Played { float a_wins = 0, a_loss = 0, a_draw = 0, b_wins = 0, b_loss = 0, b_draw = 0; };
  Played play_games(int G, uint64_t arena_seed, int infA_kind, int infB_kind, bool record, ExampleSet* ex) {
    Played r;
    const uint64_t reset_seed = arena_seed + 0x9E3779B97F4A7C15ull;   // (creation reset drew with arena_seed)
    for (int g = 0; g < G; g++) {
      StatePtr game = learn_make_game(cf.kind, cf.m, cf.n, cf.k, cf.komi);
      const int max_moves = cf.max_moves > 0 ? cf.max_moves : 2 * cf.m * cf.n;
      Arena ar(game, cf.mc, learn_make_enc(cf.enc), arena_seed + (uint64_t)g, max_moves);
      std::vector<std::unique_ptr<Inferencer>> keep;
      ar.A.nn = make_inf(infA_kind, infA.get(), keep, game->ActionSpace());
      ar.B.nn = make_inf(infB_kind, infB.get(), keep, game->ActionSpace());
      uint64_t z = reset_seed + 0x9E3779B97F4A7C15ull * (uint64_t)(g + 1);
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
      ar.Begin((z >> 63) == 0 ? 1 : 0);                               // a.r.Intn(2) == 0 -> A is Black (arena.go:81)
      while (ar.Step(record)) {}
      r.a_wins += ar.A.Wins; r.a_loss += ar.A.Loss; r.a_draw += ar.A.Draw;
      r.b_wins += ar.B.Wins; r.b_loss += ar.B.Loss; r.b_draw += ar.B.Draw;
      if (record && ex)
        for (const Example& e : ar.examples) ex->push(e.Board.data(), e.Policy.data(), e.Value);   // ex = append(ex, a.SelfPlay()...)
    }
    return r;
  }

  // agogo.go:100-172.  Returns false where the reference returns its "batches is nil" error.
  bool Learn(int iters, int episodes, int nniters, int arenaGames) {
    for (int epoch = 0; epoch < iters; epoch++) {
      EpochLog st;
      st.epoch = epoch;
      SwitchToInference(*A, *infA); SwitchToInference(*B, *infB);                      // setupSelfPlay
      const bool dummy = epoch == 0 && useDummy;
      StatePtr g0 = learn_make_game(cf.kind, cf.m, cf.n, cf.k, cf.komi);
      ExampleSet ex{cf.nn.Features, cf.m, cf.n, g0->ActionSpace() + 1, {}, {}, {}};
      play_games(episodes, cf.seed + 1000ull * epoch, dummy ? 1 : cf.sp_inf[0], dummy ? 1 : cf.sp_inf[1], true, &ex);
      if (cf.AugmentRotate) ex.AugmentRotate();                                        // Arena.Play's aug(ex), arena.go:115-120
      st.examples = ex.size();
      std::vector<float> Xs, Pi, V;
      st.batches = ex.Prepare(cf.nn.BatchSize, cf.MaxExamples, cf.seed + 13ull * epoch + 1, &Xs, &Pi, &V);   // agogo.go:118-122
      if (st.batches == 0) return false;                                               // agogo.go:123-125
      st.cost = Train(*B, Xs, Pi, V, st.batches, nniters, cf.seed + 17ull * epoch);    // agogo.go:133
      SwitchToInference(*B, *infB);                                                    // agogo.go:137
      // resetStats (agogo.go:139-140), then the evaluation games (agogo.go:144-148)
      Played ev = play_games(arenaGames, cf.seed + 1000ull * epoch + 500, cf.eval_inf[0], cf.eval_inf[1], false, nullptr);
      st.a_wins = ev.a_wins; st.a_loss = ev.a_loss; st.a_draw = ev.a_draw;
      st.b_wins = ev.b_wins; st.b_loss = ev.b_loss; st.b_draw = ev.b_draw;
      st.killedA = false;
      if (ev.b_wins / (ev.b_wins + ev.a_wins) > cf.UpdateThreshold) {                  // agogo.go:155 (0 / 0 = NaN: false)
        A = std::move(B); a_id = b_id;                                                 // a.A.NN = a.B.NN
        st.killedA = true;
      }
      stats.update(a_id, ev.a_wins, ev.a_loss, ev.a_draw);                             // a.update(a.A), agogo.go:166
      B.reset(new TrainNet<float>(cf.nn));                                             // newB, arena.go:205-224
      B->InitRandom(cf.seed * 3 + 100 + epoch);
      b_id = next_id++;
      st.a_id = a_id;
      log.push_back(st);
    }
    return true;
  }
};

}  // namespace oracle
