#!/bin/bash
# BASELINE configs[0]: the README tic-tac-toe run AZ.Learn(5, 50, 100, 100), DefaultConf(3,3,10), Budget 1000, through the
# C++ host mirror over the C ABI (tests/cpp/az_learn_ttt.cpp).  Prints the per-epoch log and the wall time.
cd "$(dirname "$0")/.."
[ -x tests/cpp/az_learn_ttt ] || make tests/cpp/az_learn_ttt
s=$(date +%s%N)
tests/cpp/az_learn_ttt 5 50 100 100 1000
e=$(date +%s%N)
echo "CONFIG0_WALL_MS $(( (e - s) / 1000000 ))"
