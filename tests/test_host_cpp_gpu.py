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
