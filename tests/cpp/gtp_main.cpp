// GTP engine over libagz (agogo_amd/host/gtp.hpp): reads commands on stdin.  gtp_main [size K blocks sims lanes]
// gtp_main proto NAME VERSION : the game-less protocol (the reference's gtp.New(nil, name, version, nil)); needs no GPU.
#include <cstdio>
#include <iostream>

#include "../../agogo_amd/host/gtp.hpp"

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "proto") {
    gtp::Protocol p(argc > 2 ? argv[2] : "agz-hip", argc > 3 ? argv[3] : "r01");
    p.run(std::cin, std::cout);
    return 0;
  }
  int size = argc > 1 ? atoi(argv[1]) : 9, K = argc > 2 ? atoi(argv[2]) : 64, L = argc > 3 ? atoi(argv[3]) : 4,
      sims = argc > 4 ? atoi(argv[4]) : 64, lanes = argc > 5 ? atoi(argv[5]) : 1;
  try {
    agz::Ctx ctx(0);
    dual::Config nc = dual::DefaultConf(size, size, size * size + 1);
    nc.K = K; nc.SharedLayers = L; nc.FC = 2 * K; nc.Features = 18; nc.BatchSize = 1;
    dual::Dual net(ctx, nc, AGZ_BN_IDENTITY);
    net.Init(1337);          // random-init weights (no checkpoint in this repository); a trained net would be Load()ed here
    net.Commit();
    mcts::Config mc = mcts::DefaultConfig(size);
    mc.Budget = sims;
    gtp::Engine e(ctx, net, mc, size, lanes);
    e.run(std::cin, std::cout);
  } catch (const std::exception& ex) {
    fprintf(stderr, "gtp_main: %s\n", ex.what());
    return 2;
  }
  return 0;
}
