#!/usr/bin/env python3
"""Writes tests/golden/*.json.

Two kinds of fixtures (data only — inputs and expected outputs):
  1. reference_kats.json — the known-answer tables the reference's OWN tests hold for this path, transcribed
     by hand from /root/reference (file:line given per table).  The Go reference cannot be executed here
     (no Go toolchain), so these tables are what pins the oracle.
  2. oracle_mcts_example.json — root statistics of the mcts Example game (mcts/example_test.go:74-156) at
     Budget N=200 produced by the oracle; they are pinned to the reference through the Example's documented
     move list 4,0,2,6,3,5,1,7,8 / "WINNER None" and equal the SURVEY App. A validation table.
Run from the repo root:  python tests/golden/make_goldens.py
"""
import ctypes as C
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

N, B, W = 0, 1, 2
X, O, Z = B, W, N  # mnk/c4 test aliases: X = Cross = Black, O = Nought = White

kats = {
    "_source": "transcribed from gorgonia/agogo tests; colours: 0 None, 1 Black, 2 White",
    # dualnet/config_test.go:5-17
    "round": [[0, 0], [1, 1], [2, 2], [3, 4], [5, 4], [8, 8], [10, 8], [31, 32], [33, 32], [80, 64], [100, 128]],
    # game/komi/komi_test.go:10-181 (applyTests) — New(m, n, 3)
    "komi_apply": [
        {"m": 3, "n": 3, "board": [N, N, N, N, N, N, N, N, N], "player": B, "move": 4,
         "board2": [N, N, N, N, B, N, N, N, N], "taken": 0, "white": 0, "black": 0, "err": False},
        {"m": 3, "n": 3, "board": [N, W, N, W, B, W, N, N, N], "player": W, "move": 7,
         "board2": [N, W, N, W, N, W, N, W, N], "taken": 1, "white": 1, "black": 0, "err": False},
        {"m": 4, "n": 4, "board": [N, W, N, N, W, B, W, N, W, B, W, N, N, N, N, N], "player": W, "move": 13,
         "board2": [N, W, N, N, W, N, W, N, W, N, W, N, N, W, N, N], "taken": 2, "white": 2, "black": 0, "err": False},
        {"m": 4, "n": 4, "board": [N, N, N, N, N, N, N, N, N, B, B, N, B, W, W, N], "player": B, "move": 15,
         "board2": [N, N, N, N, N, N, N, N, N, B, B, N, B, N, N, B], "taken": 2, "white": 0, "black": 2, "err": False},
        {"m": 3, "n": 3, "board": [N, W, N, W, N, W, N, W, N], "player": B, "move": 4, "board2": None, "taken": 0,
         "white": 0, "black": 0, "err": True},
        {"m": 3, "n": 3, "board": [N] * 9, "player": B, "move": 15, "board2": None, "taken": 0, "white": 0, "black": 0,
         "err": True},
        {"m": 3, "n": 3, "board": [N] * 9, "player": N, "move": 15, "board2": None, "taken": 0, "white": 0, "black": 0,
         "err": True},
    ],
    # game/komi/komi_test.go:264-290 (TestKomi_Ended): White to move, game must have ended
    "komi_ended": {"m": 5, "n": 5, "k": 3, "to_move": W, "board": [
        W, B, B, N, B,
        N, B, B, B, B,
        B, B, N, W, B,
        B, N, W, N, W,
        N, W, N, B, N], "ended": True},
    # game/komi/komi_test.go:292-313 (TestCheck): White cannot play point 3 on this 3x7 board
    "komi_check": {"m": 3, "n": 7, "k": 3, "to_move": W, "board": [
        B, N, B, N, B, B, W,
        N, N, W, B, B, W, N,
        W, W, N, B, B, W, W], "player": W, "move": 3, "legal": False},
    # game/wq/wq_test.go:33-196 (applyTests): Board.Apply + Board.Score (the scores pin the reference's flood fill)
    "wq_apply": [
        {"size": 3, "board": [N, N, N, N, N, N, N, N, N], "player": B, "move": 4,
         "board2": [N, N, N, N, B, N, N, N, N], "taken": 0, "white": 0, "black": 3, "err": False},
        {"size": 3, "board": [N, W, N, W, B, W, N, N, N], "player": W, "move": 7,
         "board2": [N, W, N, W, N, W, N, W, N], "taken": 1, "white": 6, "black": 0, "err": False},
        {"size": 4, "board": [N, W, N, N, W, B, W, N, W, B, W, N, N, N, N, N], "player": W, "move": 13,
         "board2": [N, W, N, N, W, N, W, N, W, N, W, N, N, W, N, N], "taken": 2, "white": 9, "black": 0, "err": False},
        {"size": 4, "board": [N, N, N, N, N, N, N, N, N, B, B, N, B, W, W, N], "player": B, "move": 15,
         "board2": [N, N, N, N, N, N, N, N, N, B, B, N, B, N, N, B], "taken": 2, "white": 0, "black": 4, "err": False},
        {"size": 3, "board": [N, W, N, W, N, W, N, W, N], "player": B, "move": 4, "board2": None, "taken": 0, "white": 0,
         "black": 0, "err": True},
        {"size": 3, "board": [N] * 9, "player": B, "move": 15, "board2": None, "taken": 0, "white": 0, "black": 0, "err": True},
        {"size": 3, "board": [N] * 9, "player": N, "move": 15, "board2": None, "taken": 0, "white": 0, "black": 0, "err": True},
    ],
    # game/mnk/mnk_test.go:9-129
    "mnk_winner": [
        {"m": 3, "n": 3, "k": 3, "board": [X, O, X, O, X, O, O, O, X], "winner_is": X, "ended": True},
        {"m": 3, "n": 3, "k": 3, "board": [X, O, O, X, O, X, O, X, X], "winner_is": O, "ended": None},
        {"m": 7, "n": 7, "k": 5, "board": [
            Z, X, Z, Z, Z, Z, Z,
            Z, Z, X, Z, Z, Z, Z,
            Z, Z, Z, X, Z, Z, Z,
            Z, Z, Z, Z, X, Z, Z,
            Z, Z, Z, Z, Z, X, Z,
            Z, Z, Z, Z, Z, X, Z,
            Z, Z, Z, Z, Z, X, Z], "winner_is": X, "ended": True},
        {"m": 7, "n": 7, "k": 5, "board": [
            Z, Z, Z, Z, Z, Z, Z,
            Z, Z, Z, Z, Z, O, Z,
            Z, Z, Z, Z, O, Z, Z,
            Z, Z, Z, O, Z, Z, Z,
            Z, Z, O, Z, Z, Z, Z,
            Z, O, Z, Z, Z, Z, Z,
            Z, Z, Z, Z, Z, Z, Z], "winner_is": O, "ended": True},
    ],
    "mnk_ended": [
        {"board": [O, Z, X, Z, Z, X, Z, O, X], "ended": True, "winner": X},
        {"board": [O, O, O, Z, Z, X, X, O, X], "ended": True, "winner": O},
        {"board": [Z, Z, X, X, O, X, O, O, O], "ended": True, "winner": O},
        {"board": [O, Z, X, X, O, X, O, Z, O], "ended": True, "winner": O},
    ],
    # game/c4/c4_test.go:17-100: New(6, 7, 4)
    "c4_ended": [
        {"board": [X, Z, Z, Z, Z, Z, Z, O, Z, Z, Z, Z, Z, Z, O, Z, Z, Z, Z, Z, Z, X, Z, Z, Z, Z, Z, Z,
                   O, O, Z, Z, X, Z, X, X, O, Z, O, X, Z, X], "ended": False, "winner": Z},
        {"board": [X, O, X, O, X, O, X, O, O, X, O, X, O, O, X, X, X, O, X, O, X, O, X, O, X, O, X, O,
                   X, O, O, X, O, O, X, X, X, O, X, O, X, X], "ended": True, "winner": Z},
        {"board": [X, Z, Z, Z, Z, Z, Z, O, Z, Z, Z, Z, Z, Z, O, Z, Z, X, Z, Z, Z, X, Z, X, Z, Z, Z, Z,
                   O, X, Z, Z, X, Z, X, X, O, Z, O, X, Z, X], "ended": True, "winner": X},
        {"board": [X, Z, Z, Z, Z, Z, Z, O, Z, Z, Z, Z, Z, Z, O, X, Z, O, Z, Z, Z, X, Z, X, Z, Z, Z, Z,
                   O, X, Z, X, X, Z, X, X, O, Z, O, X, Z, X], "ended": True, "winner": X},
        {"board": [X, Z, Z, Z, Z, Z, Z, O, Z, Z, Z, X, Z, Z, O, Z, Z, X, X, Z, Z, X, Z, Z, Z, X, Z, Z,
                   O, X, Z, Z, X, Z, X, X, O, Z, O, O, Z, X], "ended": True, "winner": X},
        {"board": [X, Z, Z, Z, Z, Z, Z, O, Z, Z, Z, Z, Z, Z, O, Z, Z, X, Z, Z, Z, X, Z, Z, Z, X, Z, Z,
                   O, X, Z, Z, X, Z, X, O, X, X, X, X, Z, X], "ended": True, "winner": X},
    ],
    # mcts/example_test.go:38-156: scripted dummyNN on TicTacToe, ONE tree searched alternately
    "mcts_example": {"moves": [4, 0, 2, 6, 3, 5, 1, 7, 8], "winner": N,
                     "budgets_reproducing_documented_game": [25, 50, 100, 200, 400, 5000]},
    # dummy.go:10-23 closed form
    "dummy_inferer": {"policy": "1/outputSize", "value": {"1": 1.0, "2": -1.0, "0": 0.0}},
}

json.dump(kats, open(os.path.join(HERE, "reference_kats.json"), "w"), indent=1)

# --- oracle-generated root statistics for the Example game at N = 200
import numpy as np  # noqa: E402
import oracle_lib as OL  # noqa: E402

L = OL.lib()
L.orc_example_new.restype = C.c_void_p
L.orc_example_new.argtypes = [C.c_int] * 4 + [C.c_double, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64]
L.orc_example_turn.restype = C.c_int
L.orc_example_turn.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
L.orc_example_root_children.restype = C.c_int
L.orc_example_root_children.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.c_int]
h = L.orc_example_new(0, 3, 3, 3, 0.0, 1.0, 200, 2, 0, 1, 1)
e, w = C.c_int(0), C.c_int(0)
turns = []
while not e.value:
    best = L.orc_example_turn(h, C.byref(e), C.byref(w))
    mv = np.zeros(10, np.int32); vis = np.zeros(10, np.uint32); bs = np.zeros(10, np.float32)
    n = L.orc_example_root_children(h, mv.ctypes.data_as(C.POINTER(C.c_int32)), vis.ctypes.data_as(C.POINTER(C.c_uint32)),
                                    bs.ctypes.data_as(C.POINTER(C.c_float)), 10)
    turns.append({"best": int(best), "children": [[int(mv[i]), int(vis[i]), float(bs[i])] for i in range(n)]})
json.dump({"_source": "oracle, mcts Example pattern, Budget 200 (equals SURVEY App. A validation table)", "budget": 200,
           "turns": turns, "winner": w.value}, open(os.path.join(HERE, "oracle_mcts_example.json"), "w"), indent=1)
print("wrote goldens")
