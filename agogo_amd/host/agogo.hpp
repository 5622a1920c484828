// Header-only C++ mirror of the reference's operator interface for the hot path, over the C ABI (include/agz.h).
// Same names, argument meaning and error behaviour as the Go types, so host code and tests read like the
// reference's: dual::Config / dual::Dual / dual::Inferencer (dualnet/config.go, dual.go, meta.go),
// mcts::Config (mcts/tree.go:15-29), agogo::Arena (arena.go) for a batch of games.
// Errors: the reference returns `error` or panics; here agz::Error is thrown with agz_last_error().
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/agz.h"

namespace agz {
struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
inline void check(int rc, const char* what) {
  if (rc != AGZ_OK) throw Error(std::string(what) + ": " + agz_last_error());
}
struct Ctx {
  agz_ctx* h = nullptr;
  explicit Ctx(int device = 0) { check(agz_ctx_create(device, &h), "agz_ctx_create"); }
  ~Ctx() { agz_ctx_destroy(h); }
  Ctx(const Ctx&) = delete;
  Ctx& operator=(const Ctx&) = delete;
};
}  // namespace agz

namespace dual {
// dualnet/config.go:4-16
struct Config {
  int K = 0, SharedLayers = 0, FC = 0;
  double L2 = 0;
  int BatchSize = 0, Width = 0, Height = 0, Features = 0, ActionSpace = 0;
  bool FwdOnly = false;
  bool IsValid() const { return K >= 1 && ActionSpace >= 3 && SharedLayers >= 0 && FC > 1 && BatchSize >= 1 && Features > 0; }
};
inline int round(int a) {  // config.go:44-58
  int n = a - 1;
  n |= n >> 1; n |= n >> 2; n |= n >> 4; n |= n >> 8; n |= n >> 16; n++;
  int lt = n / 2;
  return (a - lt) < (n - a) ? lt : n;
}
inline Config DefaultConf(int m, int n, int actionSpace) {  // config.go:18-31
  Config c; int k = round((m * n) / 3);
  c.K = k; c.SharedLayers = m; c.FC = 2 * k; c.BatchSize = 256; c.Width = n; c.Height = m; c.Features = 18; c.ActionSpace = actionSpace;
  return c;
}
// dual.Dual (dualnet/dual.go:16-47): parameters live on the device
struct Dual {
  Config conf;
  agz_net* h = nullptr;
  Dual(agz::Ctx& ctx, const Config& c, int bn_mode = AGZ_BN_DEGENERATE_EPS) : conf(c) {
    if (!c.IsValid()) throw agz::Error("NNConf is not valid. Unable to proceed");  // agogo.go:42-44 panics
    agz_net_conf nc{c.K, c.SharedLayers, c.FC, c.BatchSize, c.Width, c.Height, c.Features, c.ActionSpace, bn_mode, 1e-5f};
    agz::check(agz_net_create(ctx.h, &nc, &h), "dual.New");
  }
  ~Dual() { agz_net_destroy(h); }
  void Init(uint64_t seed) { agz::check(agz_net_init_random(h, seed), "Dual.Init"); agz::check(agz_net_commit(h), "Dual.Init"); }
  int NumLearnables() const { return agz_net_num_params(h); }  // len(Model())
  void Let(int i, const std::vector<float>& v) { agz::check(agz_net_set_param(h, i, v.data(), v.size()), "G.Let"); }
  void Commit() { agz::check(agz_net_commit(h), "commit"); }
};
// dual.Inferencer (dualnet/meta.go:106-194), batched
struct Inferencer {
  Dual& d;
  explicit Inferencer(Dual& dd) : d(dd) {}
  // Infer: planes [B,F,H,W] -> policy [B,ActionSpace], value [B]
  void Infer(const std::vector<float>& planes, int B, std::vector<float>* policy, std::vector<float>* value) {
    policy->resize((size_t)B * d.conf.ActionSpace);
    value->resize(B);
    agz::check(agz_net_infer(d.h, planes.data(), B, policy->data(), value->data()), "Inferencer.Infer");
  }
};
// The trainable side of dual.Dual: learnables in their FULL (batch-shaped) form + dual.Train (dualnet/meta.go:16-54)
struct Trainable {
  Config conf;
  agz_trainer* h = nullptr;
  Trainable(agz::Ctx& ctx, const Config& c) : conf(c) {
    if (!c.IsValid()) throw agz::Error("NNConf is not valid. Unable to proceed");
    agz_net_conf nc{c.K, c.SharedLayers, c.FC, c.BatchSize, c.Width, c.Height, c.Features, c.ActionSpace, AGZ_BN_DEGENERATE_EPS, 1e-5f};
    agz::check(agz_trainer_create(ctx.h, &nc, &h), "dual.New");
  }
  ~Trainable() { agz_trainer_destroy(h); }
  Trainable(const Trainable&) = delete;
  Trainable& operator=(const Trainable&) = delete;
  void Init(uint64_t seed) { agz::check(agz_trainer_init_random(h, seed), "Dual.Init"); }
  void SetComputeMode(int mode) { agz::check(agz_trainer_set_compute_mode(h, mode), "compute mode"); }  // AGZ_COMPUTE_BF16X3: faster fwd/dgrad
  // dual.Infer (meta.go:125-162): copy row 0 of every learnable into an inference net
  void SwitchToInference(Dual& inf) const { agz::check(agz_trainer_export(h, inf.h), "SwitchToInference"); }
  // AZ.Save / AZ.Load (agogo.go:175-209) of the learning side, full batch-shaped learnables
  void Save(const std::string& path) const { agz::check(agz_trainer_save(h, path.c_str()), "AZ.Save"); }
  void Load(const std::string& path) { agz::check(agz_trainer_load(h, path.c_str()), "AZ.Load"); }
};
// dual.Train(d, Xs, policies, values, batches, iterations) — shuffles the three arrays in place like the reference
inline float Train(Trainable& d, std::vector<float>& Xs, std::vector<float>& policies, std::vector<float>& values, int batches,
                   int iterations, uint64_t seed) {
  float cost = 0;
  agz::check(agz_train(d.h, Xs.data(), policies.data(), values.data(), batches, iterations, seed, &cost), "dual.Train");
  return cost;
}
// the same over device tensors (agz_examples_tensors_dev): nothing leaves HBM
inline float TrainDev(Trainable& d, const float* Xs_dev, const float* policies_dev, const float* values_dev, int batches, int iterations,
                      uint64_t seed) {
  float cost = 0;
  agz::check(agz_train_dev(d.h, Xs_dev, policies_dev, values_dev, batches, iterations, seed, &cost), "dual.Train");
  return cost;
}
}  // namespace dual

namespace mcts {
enum PassPreference { DontPreferPass = AGZ_DONT_PREFER_PASS, PreferPass = AGZ_PREFER_PASS, DontResign = AGZ_DONT_RESIGN };
// mcts/tree.go:15-29 (Timeout -> exactly Budget simulations)
struct Config {
  float PUCT = 1.0f;
  int M = 0, N = 0, RandomCount = 0;
  int32_t Budget = 10000;
  uint32_t RandomMinVisits = 0;
  float RandomTemperature = 0;
  bool DumbPass = true;
  float ResignPercentage = 0;
  int PassPreference = DontPreferPass;
  bool IsValid() const { return PUCT > 0 && PUCT <= 1; }
};
inline Config DefaultConfig(int boardSize) { Config c; c.M = c.N = boardSize; return c; }  // tree.go:31-41
}  // namespace mcts

namespace agogo {
struct Example { std::vector<float> Board, Policy; float Value; };  // datatypes.go:41-46
// n_games x Arena (arena.go:20-179)
struct Arena {
  agz_arena* h = nullptr;
  int cells, F, A;
  Arena(agz::Ctx& ctx, int kind, int m, int n, int k, float komi, int encoder, const mcts::Config& mc, int n_games, uint64_t seed)
      : cells(m * n), F(encoder == AGZ_ENC_WQ ? 18 : 2), A(kind == AGZ_GAME_C4 ? n : m * n) {
    if (!mc.IsValid()) throw agz::Error("MCTSConf is not valid. Unable to proceed");  // agogo.go:45-47
    agz_game_conf g{kind, m, n, k, komi, 0, encoder};
    agz_mcts_conf c{mc.PUCT, mc.M, mc.N, mc.RandomCount, mc.Budget, mc.RandomMinVisits, mc.RandomTemperature, mc.DumbPass ? 1 : 0,
                    mc.ResignPercentage, mc.PassPreference};
    agz::check(agz_arena_create(ctx.h, &g, &c, n_games, seed, 0, &h), "MakeArena");
  }
  ~Arena() { agz_arena_destroy(h); }
  void SetAgents(dual::Dual* a, dual::Dual* b) {  // nil -> dummyInferer (agogo.go:83-87)
    agz::check(agz_arena_set_inferencer(h, 0, a ? AGZ_INF_NET : AGZ_INF_DUMMY, a ? a->h : nullptr), "Agent A");
    agz::check(agz_arena_set_inferencer(h, 1, b ? AGZ_INF_NET : AGZ_INF_DUMMY, b ? b->h : nullptr), "Agent B");
  }
  // test hook: a synthetic inferencer kind per agent (AGZ_INF_NET keeps the network)
  void SetAgentKinds(int ka, dual::Dual* a, int kb, dual::Dual* b) {
    agz::check(agz_arena_set_inferencer(h, 0, ka, ka == AGZ_INF_NET ? a->h : nullptr), "Agent A");
    agz::check(agz_arena_set_inferencer(h, 1, kb, kb == AGZ_INF_NET ? b->h : nullptr), "Agent B");
  }
  // Tournament use (Agent.Search against an outside opponent, agent.go:76-81): Search() decides and plays one move for
  // every unfinished game; Opponent() applies the outside player's replies (State.Check'ed on the device).
  void Search(int budget) {
    agz::check(agz_arena_begin_move(h), "Agent.Search");
    agz::check(agz_arena_simulate(h, budget), "Agent.Search");
    agz::check(agz_arena_end_move(h, 0), "Agent.Search");
  }
  void SetParallel(int lanes) { agz::check(agz_arena_set_parallel(h, lanes), "SetParallel"); }  // lane-ordered tree-parallel rounds
  void Opponent(const std::vector<int32_t>& moves) { agz::check(agz_arena_apply_moves(h, moves.data()), "opponent move"); }
  // Play every game to the end, leaving the recorded examples on the device (see Examples::Append)
  void PlayOnDevice(bool record) {
    agz::check(agz_arena_reset(h, nullptr), "Arena.Play");
    agz::check(agz_arena_play(h, 0, record ? 1 : 0), "Arena.Play");
  }
  // Play every game to the end (Arena.Play, arena.go:80-179), recording examples
  std::vector<Example> Play(bool record) {
    agz::check(agz_arena_reset(h, nullptr), "Arena.Play");
    agz::check(agz_arena_play(h, 0, record ? 1 : 0), "Arena.Play");
    int n = 0;
    agz::check(agz_arena_get_examples(h, nullptr, nullptr, nullptr, nullptr, 0, &n), "examples");
    std::vector<float> P((size_t)n * F * cells), Q((size_t)n * (A + 1)), V(n);
    if (n) agz::check(agz_arena_get_examples(h, P.data(), Q.data(), V.data(), nullptr, n, &n), "examples");
    std::vector<Example> ex(n);
    for (int i = 0; i < n; i++) {
      ex[i].Board.assign(P.begin() + (size_t)i * F * cells, P.begin() + (size_t)(i + 1) * F * cells);
      ex[i].Policy.assign(Q.begin() + (size_t)i * (A + 1), Q.begin() + (size_t)(i + 1) * (A + 1));
      ex[i].Value = V[i];
    }
    return ex;
  }
};
// []Example kept on the device (include/agz.h "example sets"): Augmenter, shuffleExamples, prepareExamples
struct Examples {
  agz_examples* h = nullptr;
  Examples(agz::Ctx& ctx, int F, int H, int W, int policyLen) { agz::check(agz_examples_create(ctx.h, F, H, W, policyLen, &h), "Examples"); }
  ~Examples() { agz_examples_destroy(h); }
  Examples(const Examples&) = delete;
  Examples& operator=(const Examples&) = delete;
  size_t size() const { int64_t n = 0; agz::check(agz_examples_count(h, &n), "len(ex)"); return (size_t)n; }
  void Append(Arena& a) { agz::check(agz_examples_append_arena(h, a.h), "append(ex, SelfPlay()...)"); }
  void AugmentRotate() { agz::check(agz_examples_augment_rotate(h), "RotateBoard"); }
  int Prepare(int BatchSize, int maxExamples, uint64_t seed) {  // agogo.go:118-121 + prepareExamples
    int b = 0;
    agz::check(agz_examples_prepare(h, BatchSize, maxExamples, seed, &b), "prepareExamples");
    return b;
  }
};
// agogo.Config (datatypes.go:14-25)
struct Config {
  std::string Name;
  dual::Config NNConf;
  mcts::Config MCTSConf;
  double UpdateThreshold = 0.52;
  int MaxExamples = 0;
  int Encoder = AGZ_ENC_TWOPLANE;
  bool AugmentRotate = false;  // Augmenter (datatypes.go:24): the RotateBoard-based rotation augmenter, or none
  // build extension: the arithmetic of the inference networks (agz_net_set_compute_mode) and, through it, of the trainer: AGZ_COMPUTE_F32_MFMA
  // (default), AGZ_COMPUTE_BF16X3, AGZ_COMPUTE_WINO (training then uses BF16X3), AGZ_COMPUTE_WINO_H2 / AGZ_COMPUTE_AUTO (the measured mode;
  // training takes the trainer's AGZ_COMPUTE_WINO_H2)
  int ComputeMode = AGZ_COMPUTE_F32_MFMA;
  // test hooks (not in the reference): inferencers of the self-play games once the dummy is no longer in use, and of the evaluation
  // games — AGZ_INF_NET plays the networks as the reference does; the synthetic kinds make an epoch's games independent of fp32 rounding
  // in a network evaluation (tests/test_learn_parity_gpu.py compares the epoch log with oracle/learn.hpp)
  int SelfPlayInferencer[2] = {AGZ_INF_NET, AGZ_INF_NET};
  int EvalInferencer[2] = {AGZ_INF_NET, AGZ_INF_NET};
};
// Statistics (statistics.go:10-38): per network — keyed by its identity, the reference formats the pointer — the A side's Wins / Loss /
// Draw of every epoch in which it was A
struct Statistics {
  std::vector<int> Creation;
  std::map<int, std::vector<float>> Wins, Losses, Draws;
  void update(int a_id, float wins, float loss, float draw) {
    if (!Wins.count(a_id)) Creation.push_back(a_id);
    Wins[a_id].push_back(wins); Losses[a_id].push_back(loss); Draws[a_id].push_back(draw);
  }
};
struct GameSpec { int kind, m, n, k; float komi; };

// AZ (agogo.go:21-172): the trainer loop around the device hot path.  Self-play episodes and arena games of an epoch
// run concurrently as one batched arena each; everything else follows AZ.Learn line by line.
struct AZ {
  agz::Ctx& ctx;
  GameSpec game;
  Config conf;
  std::unique_ptr<dual::Trainable> A, B;      // Agent.NN (trainable, full shapes)
  std::unique_ptr<dual::Dual> infA, infB;     // SwitchToInference products (agent.go:42-57)
  bool useDummy = true;
  uint64_t seed;
  struct EpochStats { int epoch; size_t examples; int batches; float cost; long a_wins, b_wins, draws; bool killedA; int a_id; };
  std::vector<EpochStats> log;
  Statistics stats;
  int a_id = 1, b_id = 2, next_id = 3;         // network identities (the reference's %p of Agent.NN)

  AZ(agz::Ctx& c, GameSpec g, const Config& cf, uint64_t seed_ = 1337) : ctx(c), game(g), conf(cf), seed(seed_) {  // agogo.go:41-72
    if (!cf.NNConf.IsValid()) throw agz::Error("NNConf is not valid. Unable to proceed");
    if (!cf.MCTSConf.IsValid()) throw agz::Error("MCTSConf is not valid. Unable to proceed");
    A.reset(new dual::Trainable(ctx, cf.NNConf)); A->Init(seed * 3 + 1);
    B.reset(new dual::Trainable(ctx, cf.NNConf)); B->Init(seed * 3 + 2);
    infA.reset(new dual::Dual(ctx, cf.NNConf)); infB.reset(new dual::Dual(ctx, cf.NNConf));
  }
  // AZ.Learn (agogo.go:100-172)
  void Learn(int iters, int episodes, int nniters, int arenaGames) {
    for (int epoch = 0; epoch < iters; epoch++) {
      EpochStats st{}; st.epoch = epoch;
      A->SwitchToInference(*infA); B->SwitchToInference(*infB);           // setupSelfPlay, agogo.go:75-90
      agz::check(agz_net_set_compute_mode(infA->h, conf.ComputeMode), "compute mode");
      agz::check(agz_net_set_compute_mode(infB->h, conf.ComputeMode), "compute mode");
      if (conf.ComputeMode == AGZ_COMPUTE_BF16X3 || conf.ComputeMode == AGZ_COMPUTE_WINO) B->SetComputeMode(AGZ_COMPUTE_BF16X3);
      else if (conf.ComputeMode == AGZ_COMPUTE_WINO_H2 || conf.ComputeMode == AGZ_COMPUTE_AUTO) B->SetComputeMode(AGZ_COMPUTE_WINO_H2);
      Examples ex(ctx, conf.NNConf.Features, conf.NNConf.Height, conf.NNConf.Width, conf.NNConf.ActionSpace);
      {
        Arena sp(ctx, game.kind, game.m, game.n, game.k, game.komi, conf.Encoder, conf.MCTSConf, episodes, seed + 1000 * epoch);
        if (epoch == 0 && useDummy) sp.SetAgents(nullptr, nullptr);
        else sp.SetAgentKinds(conf.SelfPlayInferencer[0], infA.get(), conf.SelfPlayInferencer[1], infB.get());
        sp.PlayOnDevice(true);                                             // episodes x SelfPlay(), agogo.go:110-114
        ex.Append(sp);                                                     // device to device, episode after episode
      }
      if (conf.AugmentRotate) ex.AugmentRotate();                          // Arena.Play's aug(ex), arena.go:115-120
      st.examples = ex.size();
      // maxExamples cut (agogo.go:118-121) + prepareExamples (agogo.go:211-249), tensors stay in HBM
      int batches = ex.Prepare(conf.NNConf.BatchSize, conf.MaxExamples, seed + 13 * epoch + 1);
      if (batches == 0) throw agz::Error("batches is nil, probably too few examples regarding the batchsize");  // agogo.go:123-125
      st.batches = batches;
      float *Xs = nullptr, *Pi = nullptr, *V = nullptr;
      agz::check(agz_examples_tensors_dev(ex.h, &Xs, &Pi, &V, nullptr, nullptr), "prepareExamples");
      st.cost = dual::TrainDev(*B, Xs, Pi, V, batches, nniters, seed + 17 * epoch);  // agogo.go:133
      B->SwitchToInference(*infB);                                         // agogo.go:137
      {
        Arena ev(ctx, game.kind, game.m, game.n, game.k, game.komi, conf.Encoder, conf.MCTSConf, arenaGames, seed + 1000 * epoch + 500);
        ev.SetAgentKinds(conf.EvalInferencer[0], infA.get(), conf.EvalInferencer[1], infB.get());
        ev.Play(false);                                                    // agogo.go:144-148
        int64_t aw = 0, bw = 0, dr = 0;
        agz::check(agz_arena_get_results(ev.h, &aw, &bw, &dr), "results");
        st.a_wins = aw; st.b_wins = bw; st.draws = dr;
      }
      st.killedA = false;
      if (st.b_wins + st.a_wins > 0 && (float)st.b_wins / (float)(st.b_wins + st.a_wins) > (float)conf.UpdateThreshold) {  // agogo.go:155
        A = std::move(B); a_id = b_id;                                     // a.A.NN = a.B.NN
        st.killedA = true;
      }
      // a.update(a.A) (agogo.go:166, statistics.go:27-38): the A agent's wins / losses / draws of the evaluation games, under the name of
      // the network A holds NOW (B's, when A was killed)
      stats.update(a_id, (float)st.a_wins, (float)st.b_wins, (float)st.draws);
      st.a_id = a_id;
      b_id = next_id++;
      B.reset(new dual::Trainable(ctx, conf.NNConf));                      // newB: a fresh random net (arena.go:205-224)
      B->Init(seed * 3 + 100 + epoch);
      useDummy = useDummy && false;                                        // the dummy is only used in epoch 0 (agogo.go:83-87)
      log.push_back(st);
    }
  }
};
}  // namespace agogo
