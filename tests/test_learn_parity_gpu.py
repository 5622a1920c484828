"""GPU: the AZ.Learn COMPOSITION (agogo.go:100-172, gating :155-165, newB arena.go:205-224, Statistics.update
statistics.go:27-38) — agogo_amd/host/agogo.hpp's AZ::Learn over the C ABI against oracle/learn.hpp, epoch log by epoch log.

Deterministic setup: epoch 0 self-plays with the reference's dummyInferer; later self-play and the evaluation
games use the synthetic inferencers (hash for A, uniform or hash for B) on both sides, so that no game can be flipped by fp32
rounding in a network evaluation — the networks are still created, trained (device trainer vs oracle trainer: cost within
tolerance), exported and swapped exactly as the loop prescribes.  Compared exactly: examples, batches, the A / B / draw table of
the evaluation games, killedA, and the identity of the network A holds after every epoch (which Statistics keys on); run under a
thresholds B passes in every epoch, in none, and in some, so both branches of the gating comparison are taken."""
import os
import re
import subprocess

import pytest

import oracle_lib as O
from agogo_amd import capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _device_log(iters, episodes, nniters, games, budget, threshold, seed, sp, ev, batch):
    exe = os.path.join(ROOT, "tests", "cpp", "az_learn_ttt")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", ROOT, "tests/cpp/az_learn_ttt"])
    out = subprocess.run([exe] + [str(v) for v in (iters, episodes, nniters, games, budget, threshold, seed, sp[0], sp[1], ev[0], ev[1], batch)],
                         capture_output=True, text=True, timeout=600)
    assert "AZ_LEARN OK" in out.stdout, out.stdout + out.stderr
    rows = []
    for l in out.stdout.splitlines():
        mt = re.match(r"epoch (\d+) examples (\d+) batches (\d+) cost (\S+) A (\d+) B (\d+) draw (\d+) killedA (\d) a_id (\d+)", l)
        if mt:
            rows.append(dict(epoch=int(mt[1]), examples=int(mt[2]), batches=int(mt[3]), cost=float(mt[4]), a_wins=int(mt[5]),
                             b_wins=int(mt[6]), draws=int(mt[7]), killedA=int(mt[8]), a_id=int(mt[9])))
    stats = {int(mt[1]): mt[2].split() for mt in (re.match(r"stats net (\d+): (.*)", l) for l in out.stdout.splitlines()) if mt}
    return rows, stats


@pytest.mark.parametrize("threshold,ev,kills_want", [
    (0.50, (capi.INF_HASH, capi.INF_HASH), "mixed"),      # 9/24 then 16/24 for B: A survives one epoch and is replaced in another
    (0.30, (capi.INF_HASH, capi.INF_HASH), "all"),
    (0.70, (capi.INF_HASH, capi.INF_HASH), "none"),
    (0.90, (capi.INF_HASH, capi.INF_UNIFORM), "all"),     # B wins every game
    (0.10, (capi.INF_UNIFORM, capi.INF_HASH), "none"),    # B wins none: 0 / 24 > 0.1 is false
], ids=["mixed", "low", "high", "b_sweeps", "a_sweeps"])
def test_az_learn_epoch_log_matches_the_oracle(threshold, ev, kills_want):
    iters, episodes, nniters, games, budget, seed, batch = 3, 48, 2, 24, 30, 4242, 32
    sp = (capi.INF_HASH, capi.INF_HASH)
    dev, stats = _device_log(iters, episodes, nniters, games, budget, threshold, seed, sp, ev, batch)
    # the oracle's composition: mnk.TicTacToe(), dual.DefaultConf(3,3,10) (FC = 8) with Features 2, K 3, 3 blocks (tests/cpp/az_learn_ttt.cpp)
    orc = O.learn_run(O.MNK, 3, 3, 3, 0.0, O.ENC_TWOPLANE, 3, 3, 8, batch, 2, 10, 1.0, budget, threshold, seed, iters, episodes, nniters, games,
                      sp_inf=sp, eval_inf=ev)
    assert len(dev) == len(orc) == iters
    kills = 0
    for d, o in zip(dev, orc):
        assert d["epoch"] == o["epoch"]
        assert d["examples"] == o["examples"], (d, o)
        assert d["batches"] == o["batches"] == d["examples"] // batch
        assert (d["a_wins"], d["b_wins"], d["draws"]) == (o["a_wins"], o["b_wins"], o["a_draw"]), (d, o)
        assert o["a_wins"] == o["b_loss"] and o["b_wins"] == o["a_loss"] and o["a_draw"] == o["b_draw"]
        assert d["a_wins"] + d["b_wins"] + d["draws"] == games
        # the gating decision and what it does to A's identity (agogo.go:155-165)
        want = (o["b_wins"] + o["a_wins"]) > 0 and o["b_wins"] / (o["b_wins"] + o["a_wins"]) > threshold
        assert bool(d["killedA"]) == bool(o["killedA"]) == want, (d, o, threshold)
        assert d["a_id"] == o["a_id"]
        kills += d["killedA"]
        # the trainers start from the same stream and see the same batches: last-batch cost within the trainer's fp32 tolerance
        assert abs(d["cost"] - o["cost"]) <= 2e-3 * max(1.0, abs(o["cost"])), (d["cost"], o["cost"])
    assert {"mixed": 0 < kills < iters, "all": kills == iters, "none": kills == 0}[kills_want], (kills, [d["killedA"] for d in dev])
    # Statistics.update (statistics.go:27-38): one entry per epoch under the network A holds after the gating
    ids = [d["a_id"] for d in dev]
    assert sorted(stats) == sorted(set(ids))
    for nid in stats:
        want = ["%d/%d/%d" % (d["a_wins"], d["b_wins"], d["draws"]) for d in dev if d["a_id"] == nid]
        assert stats[nid] == want


@pytest.mark.parametrize("threshold", [0.20, 0.83])
def test_az_learn_with_the_trained_networks_playing_the_evaluation_games(threshold):
    """the same composition with eval_inf = (NET, NET) — the freshly trained B really PLAYS A in the evaluation games
    (device: agz_trainer_export -> agz_net_commit -> arena with two networks; oracle: its own trainer's row 0 in its own inference net).
    Self-play stays on the synthetic inferencers, so examples and batches are exact; an evaluation game can in principle be flipped by
    fp32 rounding of a network output (device and oracle trainers agree to 1e-4, not bit for bit), so the A / B / draw tables are held to
    +-3 of 24 games and the thresholds sit far from any observed rate: the gating decision, killedA, A's identity and the Statistics
    keys must be identical."""
    iters, episodes, nniters, games, budget, seed, batch = 3, 48, 2, 24, 30, 4242, 32
    sp = (capi.INF_HASH, capi.INF_HASH)
    ev = (capi.INF_NET, capi.INF_NET)
    dev, stats = _device_log(iters, episodes, nniters, games, budget, threshold, seed, sp, ev, batch)
    orc = O.learn_run(O.MNK, 3, 3, 3, 0.0, O.ENC_TWOPLANE, 3, 3, 8, batch, 2, 10, 1.0, budget, threshold, seed, iters, episodes, nniters, games,
                      sp_inf=sp, eval_inf=ev)
    assert len(dev) == len(orc) == iters
    print("threshold", threshold, "device", [(d["a_wins"], d["b_wins"], d["draws"]) for d in dev], "oracle", [(o["a_wins"], o["b_wins"], o["a_draw"]) for o in orc])
    for d, o in zip(dev, orc):
        assert d["examples"] == o["examples"] and d["batches"] == o["batches"]
        assert d["a_wins"] + d["b_wins"] + d["draws"] == games == o["a_wins"] + o["b_wins"] + o["a_draw"]
        for kd, ko in (("a_wins", "a_wins"), ("b_wins", "b_wins"), ("draws", "a_draw")):
            assert abs(d[kd] - o[ko]) <= 3, (d, o)
        decided = d["a_wins"] + d["b_wins"]
        rate = d["b_wins"] / decided if decided else 0.0
        assert abs(rate - threshold) > 0.15 or decided == 0, (rate, threshold)     # far from the gate: rounding cannot decide it
        assert bool(d["killedA"]) == bool(o["killedA"]) and d["a_id"] == o["a_id"], (d, o)
        assert abs(d["cost"] - o["cost"]) <= 2e-3 * max(1.0, abs(o["cost"]))
    assert sorted(stats) == sorted({d["a_id"] for d in dev})
