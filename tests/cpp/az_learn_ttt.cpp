// Host-side smoke test of the C++ mirror (agogo_amd/host/agogo.hpp) over the C ABI: AZ.Learn on tic-tac-toe
// (BASELINE config #1: README.md:90-118 — DefaultConf(3,3,10), Features 2, K 3, SharedLayers 3), reduced counts.
#include <cmath>
#include <cstdio>

#include "../../agogo_amd/host/agogo.hpp"

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 2, episodes = argc > 2 ? atoi(argv[2]) : 64, nniters = argc > 3 ? atoi(argv[3]) : 5,
      arenaGames = argc > 4 ? atoi(argv[4]) : 32, budget = argc > 5 ? atoi(argv[5]) : 40;
  try {
    agz::Ctx ctx(0);
    agogo::Config conf;
    conf.Name = "Tic Tac Toe";
    conf.NNConf = dual::DefaultConf(3, 3, 10);
    conf.NNConf.BatchSize = 100; conf.NNConf.Features = 2; conf.NNConf.K = 3; conf.NNConf.SharedLayers = 3;
    conf.MCTSConf = mcts::DefaultConfig(3);
    conf.MCTSConf.Budget = budget;
    conf.UpdateThreshold = 0.52;
    agogo::AZ az(ctx, agogo::GameSpec{AGZ_GAME_MNK, 3, 3, 3, 0.f}, conf);
    az.Learn(iters, episodes, nniters, arenaGames);
    bool ok = (int)az.log.size() == iters;
    for (auto& e : az.log) {
      printf("epoch %d examples %zu batches %d cost %.6f A %ld B %ld draw %ld killedA %d\n", e.epoch, e.examples, e.batches, e.cost,
             e.a_wins, e.b_wins, e.draws, (int)e.killedA);
      ok = ok && e.examples >= 100 && e.batches >= 1 && std::isfinite(e.cost) && e.a_wins + e.b_wins + e.draws == arenaGames;
    }
    printf("AZ_LEARN %s\n", ok ? "OK" : "FAIL");
    return ok ? 0 : 1;
  } catch (const std::exception& e) {
    printf("AZ_LEARN EXCEPTION %s\n", e.what());
    return 2;
  }
}
