"""GPU: the C++ host mirror of the reference API (agogo_amd/host/agogo.hpp: dual::Config/Dual/Trainable/Train,
mcts::Config, agogo::Arena/AZ) driving the C ABI end to end: AZ.Learn on tic-tac-toe (BASELINE config #1 shape)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_az_learn_tictactoe_cpp():
    exe = os.path.join(ROOT, "tests", "cpp", "az_learn_ttt")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", ROOT, "tests/cpp/az_learn_ttt"])
    out = subprocess.run([exe, "3", "64", "5", "32", "40"], capture_output=True, text=True, timeout=300)
    assert "AZ_LEARN OK" in out.stdout, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("epoch")]
    assert len(lines) == 3
    # epoch 0 plays with the reference's dummyInferer (agogo.go:83-87): every game still yields labelled examples
    assert int(lines[0].split()[3]) >= 64 * 5


def test_configs0_readme_run_in_full():
    """BASELINE configs[0] exactly as the README runs it: AZ.Learn(5, 50, 100, 100) on mnk.TicTacToe() with dual.DefaultConf(3, 3, 10)
    and MCTS Budget 1000, through the C++ host mirror over the C ABI (12.8 s of wall time in round 1; budget here: 60 s).  The wall
    time goes to gpurun_out/config0_wall.txt for the record."""
    import time
    exe = os.path.join(ROOT, "tests", "cpp", "az_learn_ttt")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", ROOT, "tests/cpp/az_learn_ttt"])
    t0 = time.perf_counter()
    out = subprocess.run([exe, "5", "50", "100", "100", "1000"], capture_output=True, text=True, timeout=60)
    wall = time.perf_counter() - t0
    assert "AZ_LEARN OK" in out.stdout, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("epoch")]
    assert len(lines) == 5
    assert int(lines[0].split()[3]) >= 50 * 5          # every one of the 50 episodes yields at least five labelled examples
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "config0_wall.txt"), "w") as f:
            f.write("configs[0] AZ.Learn(5,50,100,100) Budget 1000: %.2f s wall\n%s" % (wall, "\n".join(lines)))
    print("configs[0] full run: %.2f s" % wall)
