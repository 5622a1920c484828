"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY (the CPU checker).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

MNK, C4, KOMI, WQ = 0, 1, 2, 3
ENC_TWOPLANE, ENC_WQ = 0, 1
INF_NET, INF_DUMMY, INF_SCRIPT, INF_HASH, INF_UNIFORM = 0, 1, 2, 3, 4
NONE, BLACK, WHITE = 0, 1, 2
PASS, RESIGN = -1, -2

INFER_CB = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float),
                       C.c_void_p)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    L = C.CDLL(path)
    vp, i32, u32, u64, f32, f64, i64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_float, C.c_double, C.c_int64
    pf, pi = C.POINTER(C.c_float), C.POINTER(C.c_int32)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_round", i32, i32)
    sig("orc_game_new", vp, i32, i32, i32, i32, f64)
    sig("orc_game_free", None, vp)
    sig("orc_game_set_board", None, vp, pi, i32)
    sig("orc_game_get_board", None, vp, pi)
    sig("orc_game_set_to_move", None, vp, i32)
    for n in ("orc_game_to_move", "orc_game_move_number", "orc_game_passes", "orc_game_action_space"):
        sig(n, i32, vp)
    sig("orc_game_hash", u32, vp)
    sig("orc_game_check", i32, vp, i32, i32)
    sig("orc_game_apply", None, vp, i32, i32)
    sig("orc_komi_apply", i32, vp, i32, i32)
    sig("orc_wq_board_apply", i32, vp, i32, i32)
    sig("orc_wq_board_score", f32, vp, i32)
    sig("orc_game_score", f32, vp, i32)
    sig("orc_game_ended", i32, vp, pi)
    sig("orc_mnk_is_winner", i32, vp, i32)
    sig("orc_game_clone", vp, vp)
    sig("orc_game_eq", i32, vp, vp)
    sig("orc_game_reset", None, vp)
    sig("orc_game_undo", None, vp)
    sig("orc_game_fwd", None, vp)
    sig("orc_game_encode", i32, vp, i32, pf, i32)
    sig("orc_net_new", vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32)
    sig("orc_net_free", None, vp)
    sig("orc_net_num_params", i32, vp)
    sig("orc_net_param_size", i64, vp, i32)
    sig("orc_net_param_name", C.c_char_p, vp, i32)
    sig("orc_net_get_param", None, vp, i32, pf)
    sig("orc_net_set_param", None, vp, i32, pf)
    sig("orc_net_set_bn_stats", None, vp, i32, pf, pf, i32)
    sig("orc_net_init_random", None, vp, u64)
    sig("orc_net_flops_per_eval", f64, vp)
    sig("orc_net_infer", None, vp, pf, i32, pf, pf)
    sig("orc_arena_new", vp, i32, i32, i32, i32, f64, i32, f32, i32, i32, i32, i32, u32, f32, i32, f32, i32, u64, i32)
    sig("orc_arena_free", None, vp)
    sig("orc_arena_set_inferencer", i32, vp, i32, i32, vp, i32, i32)
    sig("orc_arena_set_callback", i32, vp, i32, INFER_CB, vp, i32)
    sig("orc_arena_set_parallel", None, vp, i32)
    sig("orc_arena_begin", None, vp, i32)
    sig("orc_arena_step", i32, vp, i32)
    sig("orc_arena_apply_move", i32, vp, i32)
    sig("orc_arena_play", i32, vp, i32, i32)
    sig("orc_arena_history", i32, vp, pi, i32)
    sig("orc_arena_state", None, vp, pi, pi)
    sig("orc_arena_root_children", i32, vp, i32, pi, C.POINTER(C.c_uint32), pf, pf, i32)
    sig("orc_arena_tree_stats", None, vp, i32, C.POINTER(C.c_int64), pf)
    sig("orc_arena_num_examples", i32, vp)
    sig("orc_arena_get_example", None, vp, i32, pf, pf, pf)
    sig("orc_arena_example_sizes", i32, vp, pi, pi)
    sig("orc_rotate_board", i32, pf, i32, i32, pf)
    sig("orc_exset_new", vp, i32, i32, i32, i32)
    sig("orc_exset_free", None, vp)
    sig("orc_exset_push", None, vp, pf, pf, pf, i32)
    sig("orc_exset_size", i32, vp)
    sig("orc_exset_augment_rotate", i32, vp)
    sig("orc_exset_get", None, vp, pf, pf, pf)
    sig("orc_exset_prepare", i32, vp, i32, i32, u64, pf, pf, pf)
    sig("orc_learn_run", i32, i32, i32, i32, i32, f64, i32, i32, i32, i32, i32, i32, i32, f32, i32, i32, f32, i32, i32, i32, i32, i32, i32, u64,
        i32, i32, i32, i32, pf)
    sig("orc_train_new", vp, i32, i32, i32, i32, i32, i32, i32, i32, f32)
    sig("orc_train_free", None, vp)
    sig("orc_train_num_params", i32, vp)
    sig("orc_train_param_size", i64, vp, i32)
    sig("orc_train_param_name", C.c_char_p, vp, i32)
    sig("orc_train_init_random", None, vp, u64)
    sig("orc_train_get_param", None, vp, i32, pf)
    sig("orc_train_set_param", None, vp, i32, pf)
    sig("orc_train_get_grad", None, vp, i32, pf)
    sig("orc_train_batch", f32, vp, pf, pf, pf, f32)
    sig("orc_train_gradcheck", f64, i32, i32, i32, i32, i32, i32, i32, i32, u64, i32)
    sig("orc_example_new", vp, i32, i32, i32, i32, f64, f32, i32, i32, i32, i32, u64)
    sig("orc_example_free", None, vp)
    sig("orc_example_turn", i32, vp, pi, pi)
    sig("orc_example_root_children", i32, vp, pi, C.POINTER(C.c_uint32), pf, i32)
    sig("orc_example_nn_evals", i64, vp)
    sig("orc_arena_random_move", i32, vp, u64, i32)
    sig("orc_mcts_new", vp, vp, i32, f32, i32, i32, i32, i32, u32, f32, i32, f32, i32, i32, i32, i32, u64)
    sig("orc_mcts_free", None, vp)
    sig("orc_mcts_set_callback", None, vp, INFER_CB, vp, i32)
    sig("orc_mcts_set_game", None, vp, vp)
    sig("orc_mcts_search", i32, vp, i32)
    sig("orc_mcts_policies", i32, vp, vp, pf, i32)
    sig("orc_mcts_root_children", i32, vp, pi, C.POINTER(C.c_uint32), pf, pf, i32)
    sig("orc_mcts_stats", None, vp, C.POINTER(C.c_int64))
    _LIB = L
    return L


def _pf(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Game:
    """game.State handle (game/state.go:125-156)."""

    def __init__(self, kind, m, n, k=0, komi=0.0, _h=None):
        self.kind, self.m, self.n, self.k, self.komi = kind, m, n, k, komi
        self.h = _h if _h is not None else lib().orc_game_new(kind, m, n, k, komi)
        assert self.h

    def __del__(self):
        try:
            lib().orc_game_free(self.h)
        except Exception:
            pass

    @property
    def cells(self):
        return self.m * self.n

    def set_board(self, b):
        a = np.ascontiguousarray(b, dtype=np.int32)
        lib().orc_game_set_board(self.h, _pi(a), a.size)

    def board(self):
        a = np.zeros(self.cells, dtype=np.int32)
        lib().orc_game_get_board(self.h, _pi(a))
        return a

    def set_to_move(self, p):
        lib().orc_game_set_to_move(self.h, p)

    def to_move(self):
        return lib().orc_game_to_move(self.h)

    def move_number(self):
        return lib().orc_game_move_number(self.h)

    def passes(self):
        return lib().orc_game_passes(self.h)

    def hash(self):
        return lib().orc_game_hash(self.h)

    def check(self, player, move):
        return bool(lib().orc_game_check(self.h, player, move))

    def apply(self, player, move):
        lib().orc_game_apply(self.h, player, move)

    def komi_apply(self, player, move):
        return lib().orc_komi_apply(self.h, player, move)

    def wq_board_apply(self, player, move):
        return lib().orc_wq_board_apply(self.h, player, move)

    def wq_board_score(self, player):
        return lib().orc_wq_board_score(self.h, player)

    def score(self, player):
        return lib().orc_game_score(self.h, player)

    def ended(self):
        w = C.c_int32(0)
        e = lib().orc_game_ended(self.h, C.byref(w))
        return bool(e), w.value

    def is_winner(self, player):
        return bool(lib().orc_mnk_is_winner(self.h, player))

    def clone(self):
        return Game(self.kind, self.m, self.n, self.k, self.komi, _h=lib().orc_game_clone(self.h))

    def eq(self, other):
        return bool(lib().orc_game_eq(self.h, other.h))

    def reset(self):
        lib().orc_game_reset(self.h)

    def encode(self, enc):
        f = 18 if enc == ENC_WQ else 2
        out = np.zeros(f * self.cells, dtype=np.float32)
        r = lib().orc_game_encode(self.h, enc, _pf(out), out.size)
        assert r == out.size, r
        return out


class Net:
    """dual.Dual forward restatement (dualnet/dual.go:50-103)."""

    def __init__(self, K, L, FC, W, H, F, A, BatchSize=256, bn_mode=0, bn_eps=1e-5):
        self.conf = dict(K=K, SharedLayers=L, FC=FC, BatchSize=BatchSize, Width=W, Height=H, Features=F,
                         ActionSpace=A, bn_mode=bn_mode, bn_eps=bn_eps)
        self.h = lib().orc_net_new(K, L, FC, BatchSize, W, H, F, A, bn_mode, bn_eps)
        assert self.h, "invalid config"

    def __del__(self):
        try:
            lib().orc_net_free(self.h)
        except Exception:
            pass

    def num_params(self):
        return lib().orc_net_num_params(self.h)

    def param_name(self, i):
        return lib().orc_net_param_name(self.h, i).decode()

    def get_param(self, i):
        a = np.zeros(lib().orc_net_param_size(self.h, i), dtype=np.float32)
        lib().orc_net_get_param(self.h, i, _pf(a))
        return a

    def set_param(self, i, v):
        a = np.ascontiguousarray(v, dtype=np.float32)
        assert a.size == lib().orc_net_param_size(self.h, i)
        lib().orc_net_set_param(self.h, i, _pf(a))

    def set_bn_stats(self, bi, mean, var):
        m = np.ascontiguousarray(mean, dtype=np.float32)
        v = np.ascontiguousarray(var, dtype=np.float32)
        lib().orc_net_set_bn_stats(self.h, bi, _pf(m), _pf(v), m.size)

    def init_random(self, seed):
        lib().orc_net_init_random(self.h, seed)

    def flops_per_eval(self):
        return lib().orc_net_flops_per_eval(self.h)

    def infer(self, planes):
        c = self.conf
        x = np.ascontiguousarray(planes, dtype=np.float32).reshape(-1, c["Features"], c["Height"], c["Width"])
        B = x.shape[0]
        pol = np.zeros((B, c["ActionSpace"]), dtype=np.float32)
        val = np.zeros(B, dtype=np.float32)
        lib().orc_net_infer(self.h, _pf(x), B, _pf(pol), _pf(val))
        return pol, val


class Arena:
    """agogo.Arena for ONE game (arena.go:20-179) with two Agents, each owning an mcts.MCTS."""

    def __init__(self, kind, m, n, k=0, komi=0.0, enc=ENC_TWOPLANE, PUCT=1.0, M=None, N=None, RandomCount=0,
                 Budget=100, RandomMinVisits=0, RandomTemperature=0.0, DumbPass=True, ResignPercentage=0.0,
                 PassPreference=0, seed=1337, max_moves=0):
        self.kind, self.m, self.n = kind, m, n
        self.h = lib().orc_arena_new(kind, m, n, k, komi, enc, PUCT, M if M is not None else m,
                                     N if N is not None else n, RandomCount, Budget, RandomMinVisits,
                                     RandomTemperature, int(DumbPass), ResignPercentage, PassPreference, seed,
                                     max_moves)
        assert self.h
        self._keep = []

    def __del__(self):
        try:
            lib().orc_arena_free(self.h)
        except Exception:
            pass

    def set_inferencer(self, agent, kind, net=None, dummy_player=0, policy_len=0):
        r = lib().orc_arena_set_inferencer(self.h, agent, kind, net.h if net is not None else None, dummy_player,
                                           policy_len)
        assert r == 0
        if net is not None:
            self._keep.append(net)

    def set_callback(self, agent, fn, policy_len):
        """fn(planes: np.ndarray) -> (policy np.ndarray[policy_len], value float)"""

        def tramp(planes, n, policy, plen, value, user):
            x = np.ctypeslib.as_array(planes, shape=(n,)).copy()
            p, v = fn(x)
            out = np.ctypeslib.as_array(policy, shape=(plen,))
            out[:] = np.asarray(p, dtype=np.float32)[:plen]
            value[0] = float(v)

        cb = INFER_CB(tramp)
        self._keep.append(cb)
        lib().orc_arena_set_callback(self.h, agent, cb, None, policy_len)

    def begin(self, a_is_black=-1):
        lib().orc_arena_begin(self.h, a_is_black)

    def set_parallel(self, lanes):
        """MCTSConfig.Parallel: rounds of `lanes` simulations per tree (before begin())"""
        lib().orc_arena_set_parallel(self.h, int(lanes))

    def apply_move(self, move):
        """externally chosen move for the player to move: 1 continues, 0 game over, -1 illegal (nothing applied)"""
        return lib().orc_arena_apply_move(self.h, int(move))

    def random_move(self, seed, g):
        """synthetic opening move of game index g (orc_arena_random_move); returns the move or NO_MOVE (-32768)"""
        return lib().orc_arena_random_move(self.h, seed, g)

    def step(self, record=True):
        return bool(lib().orc_arena_step(self.h, int(record)))

    def play(self, n_moves=0, record=True):
        return lib().orc_arena_play(self.h, n_moves, int(record))

    def history(self):
        a = np.zeros(4096, dtype=np.int32)
        n = lib().orc_arena_history(self.h, _pi(a), a.size)
        return a[:n].copy()

    def state(self):
        board = np.zeros(self.m * self.n, dtype=np.int32)
        out = np.zeros(6, dtype=np.int32)
        lib().orc_arena_state(self.h, _pi(board), _pi(out))
        keys = ["to_move", "move_number", "passes", "ended", "winner", "a_is_black"]
        return board, dict(zip(keys, (int(x) for x in out)))

    def root_children(self, agent):
        cap = self.m * self.n + 2
        mv = np.zeros(cap, dtype=np.int32)
        vis = np.zeros(cap, dtype=np.uint32)
        bs = np.zeros(cap, dtype=np.float32)
        pr = np.zeros(cap, dtype=np.float32)
        n = lib().orc_arena_root_children(self.h, agent, _pi(mv), vis.ctypes.data_as(C.POINTER(C.c_uint32)), _pf(bs),
                                          _pf(pr), cap)
        return mv[:n].copy(), vis[:n].copy(), bs[:n].copy(), pr[:n].copy()

    def tree_stats(self, agent):
        out = np.zeros(5, dtype=np.int64)
        bs = C.c_float(0)
        lib().orc_arena_tree_stats(self.h, agent, out.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(bs))
        return dict(nn_evals=int(out[0]), playouts=int(out[1]), iters=int(out[2]), nodes=int(out[3]),
                    root_visits=int(out[4]), root_black_scores=bs.value)

    def examples(self):
        n = lib().orc_arena_num_examples(self.h)
        if n == 0:
            return np.zeros((0, 0), np.float32), np.zeros((0, 0), np.float32), np.zeros(0, np.float32)
        bl, pl = C.c_int32(0), C.c_int32(0)
        lib().orc_arena_example_sizes(self.h, C.byref(bl), C.byref(pl))
        B = np.zeros((n, bl.value), np.float32)
        P = np.zeros((n, pl.value), np.float32)
        V = np.zeros(n, np.float32)
        for i in range(n):
            v = C.c_float(0)
            lib().orc_arena_get_example(self.h, i, _pf(B[i]), _pf(P[i]), C.byref(v))
            V[i] = v.value
        return B, P, V


class Mcts:
    """mcts.MCTS on a caller-owned Game (mcts.New / SetGame / Search / Policies, tree.go:80-142, search.go:92)."""

    def __init__(self, game, enc=ENC_TWOPLANE, PUCT=1.0, M=None, N=None, RandomCount=0, Budget=100, RandomMinVisits=0,
                 RandomTemperature=0.0, DumbPass=True, ResignPercentage=0.0, PassPreference=0, lanes=1, inf=INF_HASH,
                 policy_len=0, seed=1337):
        self.game = game
        self.cells = game.cells
        self.h = lib().orc_mcts_new(game.h, enc, PUCT, M if M is not None else game.m, N if N is not None else game.n,
                                    RandomCount, Budget, RandomMinVisits, RandomTemperature, int(DumbPass), ResignPercentage,
                                    PassPreference, lanes, inf, policy_len, seed)
        assert self.h
        self._keep = []

    def __del__(self):
        try:
            lib().orc_mcts_free(self.h)
        except Exception:
            pass

    def set_callback(self, fn, policy_len):
        def tramp(planes, n, policy, plen, value, user):
            x = np.ctypeslib.as_array(planes, shape=(n,)).copy()
            p, v = fn(x)
            out = np.ctypeslib.as_array(policy, shape=(plen,))
            out[:] = np.asarray(p, dtype=np.float32)[:plen]
            value[0] = float(v)

        cb = INFER_CB(tramp)
        self._keep.append(cb)
        lib().orc_mcts_set_callback(self.h, cb, None, policy_len)

    def set_game(self, game):
        self.game = game
        lib().orc_mcts_set_game(self.h, game.h)

    def search(self, player):
        return lib().orc_mcts_search(self.h, int(player))

    def policies(self, game=None):
        g = game if game is not None else self.game
        out = np.zeros(self.cells + 2, np.float32)
        n = lib().orc_mcts_policies(self.h, g.h, _pf(out), out.size)
        return out[:n].copy()

    def root_children(self):
        cap = self.cells + 2
        mv = np.zeros(cap, np.int32)
        vis = np.zeros(cap, np.uint32)
        bs = np.zeros(cap, np.float32)
        pr = np.zeros(cap, np.float32)
        n = lib().orc_mcts_root_children(self.h, _pi(mv), vis.ctypes.data_as(C.POINTER(C.c_uint32)), _pf(bs), _pf(pr), cap)
        return mv[:n].copy(), vis[:n].copy(), bs[:n].copy(), pr[:n].copy()

    def stats(self):
        out = np.zeros(4, dtype=np.int64)
        lib().orc_mcts_stats(self.h, out.ctypes.data_as(C.POINTER(C.c_int64)))
        return dict(nn_evals=int(out[0]), playouts=int(out[1]), iters=int(out[2]), nodes=int(out[3]))


class ExampleSearch:
    """mcts/example_test.go:74-103 pattern: ONE mcts.MCTS searched alternately for both players."""

    def __init__(self, kind, m, n, k=0, komi=0.0, PUCT=1.0, Budget=200, inf=INF_SCRIPT, policy_len=0, first_player=BLACK,
                 seed=1):
        self.h = lib().orc_example_new(kind, m, n, k, komi, PUCT, Budget, inf, policy_len, first_player, seed)
        self.cells = m * n

    def __del__(self):
        try:
            lib().orc_example_free(self.h)
        except Exception:
            pass

    def turn(self):
        e, w = C.c_int32(0), C.c_int32(0)
        best = lib().orc_example_turn(self.h, C.byref(e), C.byref(w))
        return best, bool(e.value), w.value

    def root_children(self):
        cap = self.cells + 2
        mv = np.zeros(cap, np.int32)
        vis = np.zeros(cap, np.uint32)
        bs = np.zeros(cap, np.float32)
        n = lib().orc_example_root_children(self.h, _pi(mv), vis.ctypes.data_as(C.POINTER(C.c_uint32)), _pf(bs), cap)
        return [(int(mv[i]), int(vis[i]), float(bs[i])) for i in range(n)]

    def nn_evals(self):
        return lib().orc_example_nn_evals(self.h)


class TrainNet:
    """dual.Train restatement (dualnet/meta.go:16-54): full-shape learnables, training-mode BN, backward, SGD."""

    def __init__(self, K, L, FC, W, H, F, A, BatchSize, bn_eps=1e-5):
        self.conf = dict(K=K, SharedLayers=L, FC=FC, BatchSize=BatchSize, Width=W, Height=H, Features=F, ActionSpace=A)
        self.h = lib().orc_train_new(K, L, FC, BatchSize, W, H, F, A, bn_eps)
        assert self.h

    def __del__(self):
        try:
            lib().orc_train_free(self.h)
        except Exception:
            pass

    def num_params(self):
        return lib().orc_train_num_params(self.h)

    def param_name(self, i):
        return lib().orc_train_param_name(self.h, i).decode()

    def init_random(self, seed):
        lib().orc_train_init_random(self.h, seed)

    def get_param(self, i):
        a = np.zeros(lib().orc_train_param_size(self.h, i), np.float32)
        lib().orc_train_get_param(self.h, i, _pf(a))
        return a

    def set_param(self, i, v):
        a = np.ascontiguousarray(v, np.float32)
        assert a.size == lib().orc_train_param_size(self.h, i)
        lib().orc_train_set_param(self.h, i, _pf(a))

    def get_grad(self, i):
        a = np.zeros(lib().orc_train_param_size(self.h, i), np.float32)
        lib().orc_train_get_grad(self.h, i, _pf(a))
        return a

    def batch(self, planes, Pi, V, lr=0.0):
        x = np.ascontiguousarray(planes, np.float32)
        p = np.ascontiguousarray(Pi, np.float32)
        v = np.ascontiguousarray(V, np.float32)
        return float(lib().orc_train_batch(self.h, _pf(x), _pf(p), _pf(v), lr))


def rotate_board(board, m, n):
    """RotateBoard restatement (encoding_helper.go:80-107); returns None where the reference returns an error."""
    b = np.ascontiguousarray(board, np.float32)
    out = np.zeros_like(b)
    return out if lib().orc_rotate_board(_pf(b), m, n, _pf(out)) == 0 else None


class ExampleSet:
    """[]agogo.Example + rotation Augmenter + shuffleExamples/prepareExamples restatement (oracle/examples.hpp)."""

    def __init__(self, F, m, n, A1):
        self.F, self.m, self.n, self.A1 = F, m, n, A1
        self.h = lib().orc_exset_new(F, m, n, A1)

    def __del__(self):
        try:
            lib().orc_exset_free(self.h)
        except Exception:
            pass

    def __len__(self):
        return lib().orc_exset_size(self.h)

    def push(self, boards, policies, values):
        b = np.ascontiguousarray(boards, np.float32)
        p = np.ascontiguousarray(policies, np.float32)
        v = np.ascontiguousarray(values, np.float32)
        lib().orc_exset_push(self.h, _pf(b), _pf(p), _pf(v), v.size)

    def augment_rotate(self):
        return lib().orc_exset_augment_rotate(self.h) == 0

    def get(self):
        n = len(self)
        b = np.zeros((n, self.F * self.m * self.n), np.float32)
        p = np.zeros((n, self.A1), np.float32)
        v = np.zeros(n, np.float32)
        lib().orc_exset_get(self.h, _pf(b), _pf(p), _pf(v))
        return b, p, v

    def prepare(self, BatchSize, maxExamples=0, seed=1337):
        n = len(self)
        if maxExamples > 0 and n > maxExamples:
            n = maxExamples
        rows = (n // BatchSize) * BatchSize
        x = np.zeros((max(rows, 1), self.F, self.m, self.n), np.float32)
        p = np.zeros((max(rows, 1), self.A1), np.float32)
        v = np.zeros(max(rows, 1), np.float32)
        batches = lib().orc_exset_prepare(self.h, BatchSize, maxExamples, seed, _pf(x), _pf(p), _pf(v))
        return batches, x[:rows], p[:rows], v[:rows]


def learn_run(kind, m, n, k, komi, enc, K, L, FC, BatchSize, F, A, PUCT, Budget, threshold, seed, iters, episodes, nniters, arenaGames,
              sp_inf=(0, 0), eval_inf=(0, 0), RandomCount=0, maxExamples=0, augment=False):
    """AZ.Learn restated (oracle/learn.hpp): list of per-epoch dicts; fewer than iters epochs = the reference's "batches is nil" error"""
    out = np.zeros((iters, 12), np.float32)
    r = lib().orc_learn_run(kind, m, n, k, float(komi), enc, K, L, FC, BatchSize, F, A, float(PUCT), Budget, RandomCount, float(threshold),
                            maxExamples, int(augment), sp_inf[0], sp_inf[1], eval_inf[0], eval_inf[1], seed, iters, episodes, nniters,
                            arenaGames, _pf(out))
    assert r >= 0, "orc_learn_run: invalid configuration"
    keys = ("epoch", "examples", "batches", "cost", "a_wins", "a_loss", "a_draw", "b_wins", "b_loss", "b_draw", "killedA", "a_id")
    return [dict(zip(keys, (float(v) if k_ == "cost" else int(v) for k_, v in zip(keys, row)))) for row in out[:r]]
