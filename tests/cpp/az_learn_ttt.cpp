// Host-side smoke test of the C++ mirror (agogo_amd/host/agogo.hpp) over the C ABI: AZ.Learn on tic-tac-toe
// (BASELINE config #1: README.md:90-118 — DefaultConf(3,3,10), Features 2, K 3, SharedLayers 3), reduced counts.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../agogo_amd/host/agogo.hpp"

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 2, episodes = argc > 2 ? atoi(argv[2]) : 64, nniters = argc > 3 ? atoi(argv[3]) : 5,
      arenaGames = argc > 4 ? atoi(argv[4]) : 32, budget = argc > 5 ? atoi(argv[5]) : 40;
  // parity hooks (tests/test_learn_parity_gpu.py): threshold, seed, inferencer kinds of self-play (after the dummy epoch) and of the evaluation games
  const double threshold = argc > 6 ? atof(argv[6]) : 0.52;
  const unsigned long long seed = argc > 7 ? strtoull(argv[7], nullptr, 10) : 1337ull;
  const int spa = argc > 8 ? atoi(argv[8]) : AGZ_INF_NET, spb = argc > 9 ? atoi(argv[9]) : AGZ_INF_NET;
  const int eva = argc > 10 ? atoi(argv[10]) : AGZ_INF_NET, evb = argc > 11 ? atoi(argv[11]) : AGZ_INF_NET;
  const int batch = argc > 12 ? atoi(argv[12]) : 100;
  try {
    agz::Ctx ctx(0);
    agogo::Config conf;
    conf.Name = "Tic Tac Toe";
    conf.NNConf = dual::DefaultConf(3, 3, 10);
    conf.NNConf.BatchSize = batch; conf.NNConf.Features = 2; conf.NNConf.K = 3; conf.NNConf.SharedLayers = 3;
    conf.MCTSConf = mcts::DefaultConfig(3);
    conf.MCTSConf.Budget = budget;
    conf.UpdateThreshold = threshold;
    conf.SelfPlayInferencer[0] = spa; conf.SelfPlayInferencer[1] = spb; conf.EvalInferencer[0] = eva; conf.EvalInferencer[1] = evb;
    agogo::AZ az(ctx, agogo::GameSpec{AGZ_GAME_MNK, 3, 3, 3, 0.f}, conf, seed);
    az.Learn(iters, episodes, nniters, arenaGames);
    bool ok = (int)az.log.size() == iters;
    for (auto& e : az.log) {
      printf("epoch %d examples %zu batches %d cost %.9g A %ld B %ld draw %ld killedA %d a_id %d\n", e.epoch, e.examples, e.batches, e.cost,
             e.a_wins, e.b_wins, e.draws, (int)e.killedA, e.a_id);
      ok = ok && (int)e.examples >= batch && e.batches >= 1 && std::isfinite(e.cost) && e.a_wins + e.b_wins + e.draws == arenaGames;
    }
    for (int id : az.stats.Creation) {   // Statistics (statistics.go): per network the A side's record of every epoch it was A
      printf("stats net %d:", id);
      for (size_t j = 0; j < az.stats.Wins[id].size(); j++) printf(" %g/%g/%g", az.stats.Wins[id][j], az.stats.Losses[id][j], az.stats.Draws[id][j]);
      printf("\n");
    }
    printf("AZ_LEARN %s\n", ok ? "OK" : "FAIL");
    return ok ? 0 : 1;
  } catch (const std::exception& e) {
    printf("AZ_LEARN EXCEPTION %s\n", e.what());
    return 2;
  }
}
