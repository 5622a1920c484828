"""GPU: the GTP front end (agogo_amd/host/gtp.hpp) driven through its text protocol: reply framing of
internal/gtp/gtp.go:139-154, ids, play / genmove alternation, illegal moves, the generated moves are legal for the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def vertex_to_move(v, size):
    v = v.strip().lower()
    if v == "pass":
        return -1
    if v == "resign":
        return -2
    col = ord(v[0]) - ord("a") - (1 if v[0] > "i" else 0)
    row = int(v[1:])
    return (size - row) * size + col


def run_gtp(script, args=("5", "64", "1", "16", "1")):
    exe = os.path.join(ROOT, "tests", "cpp", "gtp_main")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", ROOT, "tests/cpp/gtp_main"])
    out = subprocess.run([exe, *args], input=script, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    return [r for r in out.stdout.split("\n\n") if r.strip()]


def test_protocol_framing_and_commands():
    replies = run_gtp("protocol_version\n7 name\nknown_command genmove\nknown_command frobnicate\n# a comment\n\n3\nboardsize 5\n"
                      "boardsize 19\n12 frobnicate\nlist_commands\nquit\n")
    assert replies[0] == "= 2"
    assert replies[1] == "=7 agz-hip"
    assert replies[2] == "= true" and replies[3] == "= false"
    assert replies[4] == "="                              # boardsize 5 accepted (the network's size)
    assert replies[5].startswith("? unacceptable size")
    assert replies[6] == '?12 Unknown command "frobnicate"'
    assert "genmove" in replies[7] and "play" in replies[7]
    assert replies[8] == "="                              # quit


@pytest.mark.parametrize("lanes", ["1", "4"])
def test_a_short_game_with_legal_generated_moves(lanes):
    size = 5
    script = "clear_board\nkomi 0.5\nplay black C3\ngenmove white\nplay black C3\nplay white A1\nplay black B2\ngenmove w\ngenmove b\nshowboard\nquit\n"
    replies = run_gtp(script, args=(str(size), "64", "1", "24", lanes))
    assert replies[0] == "=" and replies[1] == "=" and replies[2] == "="
    o = O.Arena(O.WQ, size, size, komi=0.5, enc=O.ENC_WQ, Budget=1)
    o.set_inferencer(0, O.INF_HASH)
    o.set_inferencer(1, O.INF_HASH)
    o.begin(1)
    assert o.apply_move(vertex_to_move("C3", size)) >= 0
    w1 = replies[3]
    assert w1.startswith("= ")
    assert o.apply_move(vertex_to_move(w1[2:], size)) >= 0          # the engine's reply is legal
    assert replies[4].startswith("? illegal move")                  # C3 is occupied
    assert replies[5].startswith("? it is the other colour's turn")  # white just moved
    assert replies[6] == "="                                        # black B2
    assert o.apply_move(vertex_to_move("B2", size)) >= 0
    w2, b3 = replies[7], replies[8]
    assert w2.startswith("= ") and b3.startswith("= ")
    assert o.apply_move(vertex_to_move(w2[2:], size)) >= 0
    assert o.apply_move(vertex_to_move(b3[2:], size)) >= 0
    board = replies[9]
    assert board.startswith("=") and board.count("X") + board.count("O") >= 3
