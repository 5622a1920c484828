"""ctypes binding of libagz.so (include/agz.h).  No compute happens in Python and nothing here falls
back to the CPU: if the HIP library or a GPU is missing, calls raise AgzError."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# constants from include/agz.h
NONE, BLACK, WHITE = 0, 1, 2
PASS, RESIGN = -1, -2
NO_MOVE = -32768
GAME_MNK, GAME_C4, GAME_KOMI, GAME_WQ = 0, 1, 2, 3
ENC_TWOPLANE, ENC_WQ = 0, 1
INF_NET, INF_DUMMY, INF_SCRIPT, INF_HASH, INF_UNIFORM, INF_CALLBACK = 0, 1, 2, 3, 4, 5
BN_DEGENERATE_EPS, BN_RUNNING, BN_IDENTITY = 0, 1, 2
COMPUTE_F32_MFMA, COMPUTE_BF16X3, COMPUTE_FP16X2, COMPUTE_WINO, COMPUTE_AUTO, COMPUTE_WINO_H2 = 0, 1, 2, 3, 4, 5
COMPUTE_FORCE = 0x100
PROF_CONV, PROF_HEADS, PROF_SELECT, PROF_EXPAND, PROF_MOVE, PROF_CONV_INIT = 0, 1, 2, 3, 4, 5
PROF_WINO_IN, PROF_WINO_GEMM, PROF_WINO_OUT = 6, 7, 8
DONT_PREFER_PASS, PREFER_PASS, DONT_RESIGN = 0, 1, 2
POOL_STRICT, POOL_STOP_SEARCH, POOL_GROW = 0, 1, 2


class AgzError(RuntimeError):
    pass


class NetConf(C.Structure):
    """agz_net_conf == dual.Config (dualnet/config.go:4-16)."""
    _fields_ = [("K", C.c_int32), ("SharedLayers", C.c_int32), ("FC", C.c_int32), ("BatchSize", C.c_int32),
                ("Width", C.c_int32), ("Height", C.c_int32), ("Features", C.c_int32), ("ActionSpace", C.c_int32),
                ("bn_mode", C.c_int32), ("bn_eps", C.c_float)]


class GameConf(C.Structure):
    _fields_ = [("kind", C.c_int32), ("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("komi", C.c_float),
                ("max_moves", C.c_int32), ("encoder", C.c_int32)]


class MctsConf(C.Structure):
    """agz_mcts_conf == mcts.Config (mcts/tree.go:15-29), Timeout -> Budget simulations."""
    _fields_ = [("PUCT", C.c_float), ("M", C.c_int32), ("N", C.c_int32), ("RandomCount", C.c_int32),
                ("Budget", C.c_int32), ("RandomMinVisits", C.c_uint32), ("RandomTemperature", C.c_float),
                ("DumbPass", C.c_int32), ("ResignPercentage", C.c_float), ("PassPreference", C.c_int32)]


class ArenaStats(C.Structure):
    _fields_ = [("sims_total", C.c_int64), ("sims_nonnull", C.c_int64), ("nn_evals", C.c_int64),
                ("moves_played", C.c_int64), ("games_finished", C.c_int64), ("examples", C.c_int64),
                ("n_games", C.c_int32), ("n_active", C.c_int32), ("tree_full", C.c_int32), ("examples_dropped", C.c_int32),
                ("path_nodes", C.c_int64), ("children_read", C.c_int64)]


class State(C.Structure):
    """agz_state: a host-side game.State handed to mcts.SetGame (include/agz.h)."""
    _fields_ = [("board", C.POINTER(C.c_int32)), ("to_move", C.c_int32), ("n_moves", C.c_int32), ("passes", C.c_int32),
                ("hash", C.c_uint32), ("captures_black", C.c_float), ("captures_white", C.c_float),
                ("last_moves", C.POINTER(C.c_int32)), ("n_last_moves", C.c_int32),
                ("historical", C.POINTER(C.c_int32)), ("n_historical", C.c_int32)]


class GameState(C.Structure):
    _fields_ = [("to_move", C.c_int32), ("move_number", C.c_int32), ("passes", C.c_int32), ("ended", C.c_int32),
                ("winner", C.c_int32), ("a_is_black", C.c_int32), ("last_move", C.c_int32), ("reserved", C.c_int32),
                ("score_black", C.c_float), ("score_white", C.c_float)]


class LeafBatch(C.Structure):
    """agz_leaf_batch (include/agz.h): the leaves a host inferencer (AGZ_INF_CALLBACK) is handed in one call"""
    _fields_ = [("n", C.c_int32), ("features", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("policy_len", C.c_int32),
                ("planes", C.POINTER(C.c_float)), ("board", C.POINTER(C.c_int32)), ("to_move", C.POINTER(C.c_int32)),
                ("move_number", C.POINTER(C.c_int32)), ("game", C.POINTER(C.c_int32)), ("policy", C.POINTER(C.c_float)),
                ("value", C.POINTER(C.c_float))]


INFER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(LeafBatch))


def make_infer_fn(py):
    """agz_infer_fn around a Python function  leaves -> (policy [n, policy_len], value [n]);  leaves = dict(planes [n, F, H, W], board
    [n, H*W], to_move [n], move_number [n], game [n]).  An exception inside `py` is reported to libagz as a failed call (the search aborts
    with AGZ_E_CALLBACK) and re-raised by the binding afterwards."""
    box = {"exc": None}

    def tramp(_user, bp):
        try:
            b = bp.contents
            n, F, H, W, pl = b.n, b.features, b.height, b.width, b.policy_len
            as_np = np.ctypeslib.as_array
            leaves = {"planes": as_np(b.planes, shape=(n, F, H, W)), "board": as_np(b.board, shape=(n, H * W)),
                      "to_move": as_np(b.to_move, shape=(n,)), "move_number": as_np(b.move_number, shape=(n,)),
                      "game": as_np(b.game, shape=(n,)), "policy_len": pl}
            pol, val = py(leaves)
            as_np(b.policy, shape=(n, pl))[...] = np.asarray(pol, np.float32).reshape(n, pl)
            as_np(b.value, shape=(n,))[...] = np.asarray(val, np.float32).reshape(n)
            return 0
        except BaseException as e:   # never let an exception unwind through the C frames
            box["exc"] = e
            return 1

    fn = INFER_FN(tramp)
    fn._box = box
    return fn


def lib_path():
    return os.environ.get("AGZ_LIB_PATH") or os.path.join(_HERE, "lib", "libagz.so")


def lib():
    """Load libagz.so (built in-tree by `make` / __graft_entry__.build())."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise AgzError("libagz.so not built (%s): run `make` or __graft_entry__.build(); there is no CPU fallback"
                       % path)
    L = C.CDLL(path)
    vp, i32, u64, f64 = C.c_void_p, C.c_int, C.c_uint64, C.c_double
    pvp = C.POINTER(C.c_void_p)
    pf, pi, pu, pu8 = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("agz_last_error", C.c_char_p)
    sig("agz_version", C.c_char_p)
    sig("agz_ctx_create", i32, i32, pvp)
    sig("agz_ctx_destroy", None, vp)
    sig("agz_ctx_sync", i32, vp)
    sig("agz_ctx_stream", vp, vp)
    sig("agz_ctx_prof_enable", i32, vp, i32)
    sig("agz_ctx_prof_read", i32, vp, i32, C.POINTER(C.c_int64), C.POINTER(C.c_double))
    sig("agz_net_create", i32, vp, C.POINTER(NetConf), pvp)
    sig("agz_net_destroy", None, vp)
    sig("agz_net_num_params", i32, vp)
    sig("agz_net_param_info", i32, vp, i32, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t))
    sig("agz_net_set_param", i32, vp, i32, pf, C.c_size_t)
    sig("agz_net_get_param", i32, vp, i32, pf, C.c_size_t)
    sig("agz_net_set_bn_stats", i32, vp, i32, pf, pf, C.c_size_t)
    sig("agz_net_init_random", i32, vp, u64)
    sig("agz_net_commit", i32, vp)
    sig("agz_net_infer", i32, vp, pf, i32, pf, pf)
    sig("agz_net_infer_dev", i32, vp, vp, i32, vp, vp)
    sig("agz_host_alloc", i32, vp, C.c_size_t, pvp)
    sig("agz_host_free", i32, vp, vp)
    sig("agz_mcts_to_dot", i32, vp, i32, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t))
    sig("agz_mcts_set_timeout_ms", i32, vp, i32)
    sig("agz_mcts_last_simulations", i32, vp, C.POINTER(C.c_int64))
    sig("agz_net_set_latency_mode", i32, vp, i32)
    sig("agz_net_set_tower_queues", i32, vp, i32)
    sig("agz_net_set_compute_mode", i32, vp, i32)
    sig("agz_net_flops_per_eval", f64, vp)
    sig("agz_net_save", i32, vp, C.c_char_p)
    sig("agz_net_load", i32, vp, C.c_char_p)
    sig("agz_arena_get_results", i32, vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64))
    sig("agz_trainer_create", i32, vp, C.POINTER(NetConf), pvp)
    sig("agz_trainer_destroy", None, vp)
    sig("agz_trainer_num_params", i32, vp)
    sig("agz_trainer_param_info", i32, vp, i32, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t))
    sig("agz_trainer_set_param", i32, vp, i32, pf, C.c_size_t)
    sig("agz_trainer_get_param", i32, vp, i32, pf, C.c_size_t)
    sig("agz_trainer_get_grad", i32, vp, i32, pf, C.c_size_t)
    sig("agz_trainer_init_random", i32, vp, u64)
    sig("agz_trainer_batch", i32, vp, pf, pf, pf, C.c_float, pf)
    sig("agz_trainer_forward_backward", i32, vp, pf, pf, pf, pf)
    sig("agz_trainer_forward_backward_dev", i32, vp, vp, vp, vp, pf)
    sig("agz_trainer_apply", i32, vp, C.c_float, C.c_float)
    sig("agz_trainer_grads_dev", i32, vp, pvp, C.POINTER(C.c_size_t))
    sig("agz_trainer_set_compute_mode", i32, vp, i32)
    sig("agz_train", i32, vp, pf, pf, pf, i32, i32, u64, pf)
    sig("agz_trainer_export", i32, vp, vp)
    sig("agz_trainer_save", i32, vp, C.c_char_p)
    sig("agz_trainer_load", i32, vp, C.c_char_p)
    sig("agz_arena_create", i32, vp, C.POINTER(GameConf), C.POINTER(MctsConf), i32, u64, i32, pvp)
    sig("agz_arena_destroy", None, vp)
    sig("agz_arena_set_inferencer", i32, vp, i32, i32, vp)
    sig("agz_arena_reset", i32, vp, pu8)
    sig("agz_arena_play", i32, vp, i32, i32)
    sig("agz_arena_selfplay", i32, vp, C.c_int64, i32)
    sig("agz_arena_set_parallel", i32, vp, i32)
    sig("agz_arena_begin_move", i32, vp)
    sig("agz_arena_simulate", i32, vp, i32)
    sig("agz_arena_end_move", i32, vp, i32)
    sig("agz_arena_apply_moves", i32, vp, pi)
    sig("agz_arena_get_stats", i32, vp, C.POINTER(ArenaStats))
    sig("agz_arena_get_game", i32, vp, i32, pi, C.POINTER(GameState))
    sig("agz_arena_get_history", i32, vp, i32, pi, i32, pi)
    sig("agz_arena_root_children", i32, vp, i32, i32, pi, pu, pf, pf, i32, pi)
    sig("agz_arena_tree_nodes", i32, vp, i32, i32, pi)
    sig("agz_arena_get_examples", i32, vp, pf, pf, pf, pi, i32, pi)
    sig("agz_arena_clear_examples", i32, vp)
    sig("agz_arena_drop_labelled_examples", i32, vp)
    sig("agz_arena_examples_dev", i32, vp, pvp, pvp, pvp, pi)
    i64, pi64 = C.c_int64, C.POINTER(C.c_int64)
    sig("agz_train_dev", i32, vp, vp, vp, vp, i32, i32, u64, pf)
    sig("agz_examples_create", i32, vp, i32, i32, i32, i32, pvp)
    sig("agz_examples_destroy", None, vp)
    sig("agz_examples_count", i32, vp, pi64)
    sig("agz_examples_clear", i32, vp)
    sig("agz_examples_append_arena", i32, vp, vp)
    sig("agz_examples_append_dev", i32, vp, vp, vp, vp, i64)
    sig("agz_examples_append_host", i32, vp, pf, pf, pf, i64)
    sig("agz_examples_get", i32, vp, pf, pf, pf, i64, pi64)
    sig("agz_examples_augment_rotate", i32, vp)
    sig("agz_examples_prepare", i32, vp, i32, i32, u64, pi)
    sig("agz_examples_tensors_dev", i32, vp, pvp, pvp, pvp, pi64, pi)
    sig("agz_examples_get_tensors", i32, vp, pf, pf, pf)
    sig("agz_examples_raw_dev", i32, vp, pvp, pvp, pvp)
    sig("agz_rotate_boards", i32, vp, pf, i32, i32, i32, pf)
    sig("agz_ctx_prof_set_stride", i32, vp, i32, i32)
    sig("agz_wino_stages", i32, vp, pf, pf, i32, i32, i32, i32, i32, pf, pf)
    sig("agz_wino_h2_tile", i32, i32, i32)
    sig("agz_net_set_wino_h2_form", i32, vp, i32)
    sig("agz_net_set_wino_h2_gemm", i32, vp, i32)
    sig("agz_net_min_same_batch", i32, vp, i32, i32, C.POINTER(C.c_int))
    sig("agz_arena_set_prep_compact", i32, vp, i32)
    sig("agz_trainer_set_dma_forward", i32, vp, i32)
    sig("agz_arena_last_prep_batch", i32, vp, C.POINTER(C.c_int), C.POINTER(C.c_int))
    sig("agz_arena_debug_counter", i32, vp, i32, C.POINTER(C.c_int64))
    sig("agz_arena_pool_capacity", i32, vp, C.POINTER(C.c_int), C.POINTER(C.c_int))
    sig("agz_wino_h2_chained", i32, i32, i32, i32)
    sig("agz_arena_random_moves", i32, vp, pi, u64)
    sig("agz_arena_set_state", i32, vp, i32, C.POINTER(State))
    sig("agz_arena_examples_labelled_dev", i32, vp, pvp)
    sig("agz_mcts_create", i32, vp, C.POINTER(GameConf), C.POINTER(MctsConf), u64, i32, pvp)
    sig("agz_mcts_destroy", None, vp)
    sig("agz_mcts_set_inferencer", i32, vp, i32, vp)
    sig("agz_mcts_set_inferencer_callback", i32, vp, INFER_FN, vp, i32)
    sig("agz_arena_set_inferencer_callback", i32, vp, i32, INFER_FN, vp, i32)
    sig("agz_arena_set_pool_policy", i32, vp, i32)
    sig("agz_mcts_set_pool_policy", i32, vp, i32)
    sig("agz_mcts_set_parallel", i32, vp, i32)
    sig("agz_mcts_set_game", i32, vp, C.POINTER(State))
    sig("agz_mcts_search", i32, vp, i32, pi)
    sig("agz_mcts_policies", i32, vp, pf, i32)
    sig("agz_mcts_root_children", i32, vp, pi, pu, pf, pf, i32, pi)
    sig("agz_mcts_nodes", i32, vp, pi)
    sig("agz_mcts_children", i32, vp, i32, pi, pi, pu, pf, pf, i32, pi)
    sig("agz_mcts_get_stats", i32, vp, C.POINTER(ArenaStats))
    sig("agz_mcts_reset", i32, vp)
    sig("agz_comm_init_all", i32, pvp, i32, pvp)
    sig("agz_comm_unique_id", i32, vp)
    sig("agz_comm_init_rank", i32, vp, i32, i32, vp, pvp)
    sig("agz_comm_destroy", None, vp)
    sig("agz_comm_rank", i32, vp)
    sig("agz_comm_size", i32, vp)
    sig("agz_examples_allgather", i32, vp, vp)
    sig("agz_trainer_allreduce", i32, vp, vp)
    sig("agz_trainer_forward_backward_allreduce", i32, vp, vp, pf, pf, pf, pf)
    sig("agz_trainer_forward_backward_allreduce_dev", i32, vp, vp, vp, vp, vp, pf)
    sig("agz_comm_debug_fail_slice", i32, vp, i32)
    _LIB = L
    return L


def _check(r, what=""):
    if r != 0:
        raise AgzError("%s failed (%d): %s" % (what, r, lib().agz_last_error().decode(errors="replace")))


def _pf(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Ctx:
    def __init__(self, device=0):
        self.h = C.c_void_p()
        self._children = []
        _check(lib().agz_ctx_create(device, C.byref(self.h)), "agz_ctx_create")

    def _adopt(self, child):
        import weakref
        self._children.append(weakref.ref(child))

    def close(self):
        """destroys dependants (arenas, nets) first: their handles hold a pointer to this ctx"""
        if self.h:
            kids = [r() for r in self._children]
            for k in sorted([k for k in kids if k is not None], key=lambda k: 0 if isinstance(k, Arena) else 1):  # arenas first
                k.close()
            self._children = []
            for p in list(getattr(self, "_pinned", {}).values()):
                lib().agz_host_free(self.h, C.c_void_p(p))
            self._pinned = {}
            lib().agz_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def host_array(self, shape, dtype=np.float32):
        """a numpy array over page-locked host memory (agz_host_alloc); freed when the ctx closes or by host_free(array)"""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        _check(lib().agz_host_alloc(self.h, max(n, 1), C.byref(p)), "agz_host_alloc")
        arr = np.ctypeslib.as_array((C.c_char * max(n, 1)).from_address(p.value)).view(dtype)[: int(np.prod(shape))].reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None and self.h:
            _check(lib().agz_host_free(self.h, C.c_void_p(p)), "agz_host_free")

    def sync(self):
        _check(lib().agz_ctx_sync(self.h), "agz_ctx_sync")

    def stream(self):
        return lib().agz_ctx_stream(self.h)

    def prof_enable(self, on=True, classes=None):
        """classes: iterable of PROF_* to record (None: all)"""
        if on and classes is not None:
            mask = 0
            for k in classes:
                mask |= 1 << (k + 1)
            _check(lib().agz_ctx_prof_enable(self.h, mask), "agz_ctx_prof_enable")
            return
        return self._prof_enable_all(on)

    def prof_set_stride(self, klass, stride):
        """bracket only every stride-th launch of class klass (agz_debug.h)"""
        _check(lib().agz_ctx_prof_set_stride(self.h, int(klass), int(stride)), "agz_ctx_prof_set_stride")

    def _prof_enable_all(self, on=True):
        _check(lib().agz_ctx_prof_enable(self.h, int(on)), "agz_ctx_prof_enable")

    def prof_read(self, klass):
        n, ms = C.c_int64(0), C.c_double(0)
        _check(lib().agz_ctx_prof_read(self.h, klass, C.byref(n), C.byref(ms)), "agz_ctx_prof_read")
        return n.value, ms.value


class Net:
    """dual.Dual + dual.Inferencer over libagz (dualnet/dual.go, dualnet/meta.go:125-190)."""

    def __init__(self, ctx, K, SharedLayers, FC, Width, Height, Features, ActionSpace, BatchSize=256,
                 bn_mode=BN_DEGENERATE_EPS, bn_eps=1e-5):
        self.ctx = ctx
        self.conf = NetConf(K, SharedLayers, FC, BatchSize, Width, Height, Features, ActionSpace, bn_mode, bn_eps)
        self.h = C.c_void_p()
        _check(lib().agz_net_create(ctx.h, C.byref(self.conf), C.byref(self.h)), "agz_net_create")
        ctx._adopt(self)

    def close(self):
        if self.h and self.ctx.h:
            lib().agz_net_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_params(self):
        return lib().agz_net_num_params(self.h)

    def param_info(self, i):
        name = C.create_string_buffer(128)
        n = C.c_size_t(0)
        _check(lib().agz_net_param_info(self.h, i, name, 128, C.byref(n)), "agz_net_param_info")
        return name.value.decode(), n.value

    def set_param(self, i, v):
        a = np.ascontiguousarray(v, dtype=np.float32).ravel()
        _check(lib().agz_net_set_param(self.h, i, _pf(a), a.size), "agz_net_set_param")

    def get_param(self, i):
        _, n = self.param_info(i)
        a = np.zeros(n, dtype=np.float32)
        _check(lib().agz_net_get_param(self.h, i, _pf(a), a.size), "agz_net_get_param")
        return a

    def set_bn_stats(self, bi, mean, var):
        m = np.ascontiguousarray(mean, dtype=np.float32)
        v = np.ascontiguousarray(var, dtype=np.float32)
        _check(lib().agz_net_set_bn_stats(self.h, bi, _pf(m), _pf(v), m.size), "agz_net_set_bn_stats")

    def init_random(self, seed):
        _check(lib().agz_net_init_random(self.h, seed), "agz_net_init_random")

    def commit(self):
        _check(lib().agz_net_commit(self.h), "agz_net_commit")

    def infer(self, planes):
        c = self.conf
        x = np.ascontiguousarray(planes, dtype=np.float32).reshape(-1, c.Features, c.Height, c.Width)
        B = x.shape[0]
        pol = np.zeros((B, c.ActionSpace), dtype=np.float32)
        val = np.zeros(B, dtype=np.float32)
        _check(lib().agz_net_infer(self.h, _pf(x), B, _pf(pol), _pf(val)), "agz_net_infer")
        return pol, val

    def set_compute_mode(self, mode):
        _check(lib().agz_net_set_compute_mode(self.h, int(mode)), "agz_net_set_compute_mode")

    def set_latency_mode(self, on=True):
        _check(lib().agz_net_set_latency_mode(self.h, int(on)), "agz_net_set_latency_mode")

    def set_tower_queues(self, queues):
        _check(lib().agz_net_set_tower_queues(self.h, int(queues)), "agz_net_set_tower_queues")

    def set_wino_h2_form(self, form):
        """agz_debug.h A/B hook: -1 auto (chained block where the shape allows), 0 three-kernel block, 1 chained"""
        _check(lib().agz_net_set_wino_h2_form(self.h, int(form)), "agz_net_set_wino_h2_form")

    def set_wino_h2_gemm(self, variant):
        """agz_debug.h A/B hook: 0 default, 1 wino_gemm_h2g_kernel, 2 wino_gemm_h2p_kernel (persistent; K = 256)"""
        _check(lib().agz_net_set_wino_h2_gemm(self.h, int(variant)), "agz_net_set_wino_h2_gemm")

    def min_same_batch(self, n, G):
        """agz_debug.h: the smallest batch >= n whose per-board outputs equal those of a G-board batch bit for bit"""
        b = C.c_int(0)
        _check(lib().agz_net_min_same_batch(self.h, int(n), int(G), C.byref(b)), "agz_net_min_same_batch")
        return b.value

    def infer_dev(self, planes_ptr, B, policy_ptr, value_ptr):
        """device pointers (ints); asynchronous on the ctx stream"""
        _check(lib().agz_net_infer_dev(self.h, C.c_void_p(planes_ptr), B, C.c_void_p(policy_ptr),
                                       C.c_void_p(value_ptr)), "agz_net_infer_dev")

    def flops_per_eval(self):
        return lib().agz_net_flops_per_eval(self.h)

    def save(self, path):
        _check(lib().agz_net_save(self.h, os.fsencode(path)), "agz_net_save")

    def load(self, path):
        _check(lib().agz_net_load(self.h, os.fsencode(path)), "agz_net_load")


class Trainer:
    """dual.Train on device (dualnet/meta.go:16-54): full-shape learnables, training-mode BN, backward, vanilla SGD."""

    def __init__(self, ctx, K, SharedLayers, FC, Width, Height, Features, ActionSpace, BatchSize, bn_eps=1e-5):
        self.ctx = ctx
        self.conf = NetConf(K, SharedLayers, FC, BatchSize, Width, Height, Features, ActionSpace, 0, bn_eps)
        self.h = C.c_void_p()
        _check(lib().agz_trainer_create(ctx.h, C.byref(self.conf), C.byref(self.h)), "agz_trainer_create")
        ctx._adopt(self)

    def close(self):
        if self.h and self.ctx.h:
            lib().agz_trainer_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_params(self):
        return lib().agz_trainer_num_params(self.h)

    def param_info(self, i):
        name = C.create_string_buffer(128)
        n = C.c_size_t(0)
        _check(lib().agz_trainer_param_info(self.h, i, name, 128, C.byref(n)), "agz_trainer_param_info")
        return name.value.decode(), n.value

    def set_param(self, i, v):
        a = np.ascontiguousarray(v, dtype=np.float32).ravel()
        _check(lib().agz_trainer_set_param(self.h, i, _pf(a), a.size), "agz_trainer_set_param")

    def get_param(self, i):
        a = np.zeros(self.param_info(i)[1], np.float32)
        _check(lib().agz_trainer_get_param(self.h, i, _pf(a), a.size), "agz_trainer_get_param")
        return a

    def get_grad(self, i):
        a = np.zeros(self.param_info(i)[1], np.float32)
        _check(lib().agz_trainer_get_grad(self.h, i, _pf(a), a.size), "agz_trainer_get_grad")
        return a

    def init_random(self, seed):
        _check(lib().agz_trainer_init_random(self.h, seed), "agz_trainer_init_random")

    def batch(self, planes, pi, v, lr=0.1):
        x = np.ascontiguousarray(planes, np.float32)
        p = np.ascontiguousarray(pi, np.float32)
        vv = np.ascontiguousarray(v, np.float32)
        c = C.c_float(0)
        _check(lib().agz_trainer_batch(self.h, _pf(x), _pf(p), _pf(vv), lr, C.byref(c)), "agz_trainer_batch")
        return c.value

    def forward_backward(self, planes, pi, v):
        x = np.ascontiguousarray(planes, np.float32)
        p = np.ascontiguousarray(pi, np.float32)
        vv = np.ascontiguousarray(v, np.float32)
        c = C.c_float(0)
        _check(lib().agz_trainer_forward_backward(self.h, _pf(x), _pf(p), _pf(vv), C.byref(c)), "agz_trainer_forward_backward")
        return c.value

    def forward_backward_dev(self, planes_ptr, pi_ptr, v_ptr, want_cost=True):
        c = C.c_float(0)
        _check(lib().agz_trainer_forward_backward_dev(self.h, C.c_void_p(planes_ptr), C.c_void_p(pi_ptr), C.c_void_p(v_ptr),
                                                      C.byref(c) if want_cost else None), "agz_trainer_forward_backward_dev")
        return c.value if want_cost else None

    def apply(self, lr=0.1, grad_scale=1.0):
        _check(lib().agz_trainer_apply(self.h, lr, grad_scale), "agz_trainer_apply")

    def set_compute_mode(self, mode):
        _check(lib().agz_trainer_set_compute_mode(self.h, int(mode)), "agz_trainer_set_compute_mode")

    def set_dma_forward(self, on=True):
        """agz_debug.h A/B hook: bit 0 WINO_H2 forward convolutions through the DMA GEMM on pre-split planes (default) or the staging-split
        kernel; bit 2 the first form of the head kernels; bit 3 weight images per layer in line instead of at the start of the step;
        bit 4 no side stream; bit 5 k_conv_h2dma instead of k_conv_h2dma3; bits 6, 7 timing-only decomposition of the latter (wrong results)"""
        _check(lib().agz_trainer_set_dma_forward(self.h, int(on)), "agz_trainer_set_dma_forward")

    def grads_dev(self):
        ptr, n = C.c_void_p(), C.c_size_t(0)
        _check(lib().agz_trainer_grads_dev(self.h, C.byref(ptr), C.byref(n)), "agz_trainer_grads_dev")
        return ptr.value, n.value

    def train(self, Xs, policies, values, batches, iterations, seed=1337):
        """dual.Train; arrays are shuffled in place"""
        assert Xs.dtype == np.float32 and policies.dtype == np.float32 and values.dtype == np.float32
        c = C.c_float(0)
        _check(lib().agz_train(self.h, _pf(Xs), _pf(policies), _pf(values), batches, iterations, seed, C.byref(c)), "agz_train")
        return c.value

    def train_dev(self, Xs_ptr, policies_ptr, values_ptr, batches, iterations, seed=1337):
        """dual.Train over device tensors (see Examples.tensors_dev); nothing is copied to the host"""
        c = C.c_float(0)
        _check(lib().agz_train_dev(self.h, C.c_void_p(Xs_ptr), C.c_void_p(policies_ptr), C.c_void_p(values_ptr), batches,
                                   iterations, seed, C.byref(c)), "agz_train_dev")
        return c.value

    def export(self, net):
        _check(lib().agz_trainer_export(self.h, net.h), "agz_trainer_export")

    def save(self, path):
        _check(lib().agz_trainer_save(self.h, str(path).encode()), "agz_trainer_save")

    def load(self, path):
        _check(lib().agz_trainer_load(self.h, str(path).encode()), "agz_trainer_load")


class Arena:
    """n_games x agogo.Arena (arena.go:20-179): batched self-play with per-agent mcts.MCTS trees on device."""

    def __init__(self, ctx, kind, m, n, k=0, komi=0.0, encoder=ENC_TWOPLANE, n_games=1, seed=1337, max_nodes=0,
                 max_moves=0, PUCT=1.0, M=None, N=None, RandomCount=0, Budget=100, RandomMinVisits=0,
                 RandomTemperature=0.0, DumbPass=True, ResignPercentage=0.0, PassPreference=DONT_PREFER_PASS):
        self.ctx = ctx
        self.kind, self.m, self.n, self.n_games = kind, m, n, n_games
        self.features = 18 if encoder == ENC_WQ else 2
        self.action_space = n if kind == GAME_C4 else m * n
        self.gconf = GameConf(kind, m, n, k, komi, max_moves, encoder)
        self.mconf = MctsConf(PUCT, M if M is not None else m, N if N is not None else n, RandomCount, Budget,
                              RandomMinVisits, RandomTemperature, int(DumbPass), ResignPercentage, PassPreference)
        self.h = C.c_void_p()
        self._nets = []
        _check(lib().agz_arena_create(ctx.h, C.byref(self.gconf), C.byref(self.mconf), n_games, seed, max_nodes,
                                      C.byref(self.h)), "agz_arena_create")
        ctx._adopt(self)

    def close(self):
        if self.h and self.ctx.h:
            lib().agz_arena_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_inferencer(self, agent, kind, net=None):
        _check(lib().agz_arena_set_inferencer(self.h, agent, kind, net.h if net is not None else None),
               "agz_arena_set_inferencer")
        if net is not None:
            self._nets.append(net)

    def set_inferencer_callback(self, agent, fn, policy_len):
        """AGZ_INF_CALLBACK: agent `agent` evaluates its leaves through the Python function fn(leaves) -> (policy, value) (make_infer_fn)"""
        cfn = make_infer_fn(fn)
        _check(lib().agz_arena_set_inferencer_callback(self.h, agent, cfn, None, int(policy_len)), "agz_arena_set_inferencer_callback")
        self._nets.append(cfn)      # keeps the trampoline alive as long as the arena

    def set_pool_policy(self, policy):
        """POOL_STRICT (a full node pool fails play / selfplay; the default with an explicit max_nodes), POOL_STOP_SEARCH (the reference's
        MAXTREESIZE rule: the search ends early, the game goes on) or POOL_GROW (pools re-allocated before a search could outgrow them; the default
        when the library sized the pools, max_nodes = 0) — include/agz.h"""
        _check(lib().agz_arena_set_pool_policy(self.h, int(policy)), "agz_arena_set_pool_policy")

    def reset(self, a_is_black=None):
        if a_is_black is None:
            _check(lib().agz_arena_reset(self.h, None), "agz_arena_reset")
        else:
            a = np.ascontiguousarray(a_is_black, dtype=np.uint8)
            assert a.size == self.n_games
            _check(lib().agz_arena_reset(self.h, a.ctypes.data_as(C.POINTER(C.c_uint8))), "agz_arena_reset")

    def play(self, n_moves=0, record=True):
        _check(lib().agz_arena_play(self.h, n_moves, int(record)), "agz_arena_play")

    def selfplay(self, n_games_target, record=True):
        _check(lib().agz_arena_selfplay(self.h, n_games_target, int(record)), "agz_arena_selfplay")

    def set_parallel(self, lanes):
        _check(lib().agz_arena_set_parallel(self.h, int(lanes)), "agz_arena_set_parallel")

    def begin_move(self):
        _check(lib().agz_arena_begin_move(self.h), "agz_arena_begin_move")

    def simulate(self, k):
        _check(lib().agz_arena_simulate(self.h, k), "agz_arena_simulate")

    def end_move(self, record=True):
        _check(lib().agz_arena_end_move(self.h, int(record)), "agz_arena_end_move")

    def apply_moves(self, moves):
        """externally chosen moves (one per game) instead of a search: the opponent's reply in a tournament"""
        m = np.ascontiguousarray(moves, dtype=np.int32)
        assert m.size == self.n_games
        _check(lib().agz_arena_apply_moves(self.h, _pi(m)), "agz_arena_apply_moves")

    def stats(self):
        s = ArenaStats()
        _check(lib().agz_arena_get_stats(self.h, C.byref(s)), "agz_arena_get_stats")
        return {f: getattr(s, f) for f, _ in ArenaStats._fields_}

    def random_moves(self, n_moves, seed=1337):
        """synthetic openings: game g plays n_moves[g] uniformly random legal moves (agz_arena_random_moves)"""
        m = np.ascontiguousarray(n_moves, dtype=np.int32)
        assert m.size == self.n_games
        _check(lib().agz_arena_random_moves(self.h, _pi(m), seed), "agz_arena_random_moves")

    def set_state(self, g, **kw):
        st, keep = make_state(self.m * self.n, **kw)
        _check(lib().agz_arena_set_state(self.h, g, C.byref(st)), "agz_arena_set_state")
        del keep

    def results(self):
        a, b, d = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _check(lib().agz_arena_get_results(self.h, C.byref(a), C.byref(b), C.byref(d)), "agz_arena_get_results")
        return {"a_wins": a.value, "b_wins": b.value, "draws": d.value}

    def game(self, g):
        board = np.zeros(self.m * self.n, dtype=np.int32)
        st = GameState()
        _check(lib().agz_arena_get_game(self.h, g, _pi(board), C.byref(st)), "agz_arena_get_game")
        return board, {f: getattr(st, f) for f, _ in GameState._fields_ if f != "reserved"}

    def history(self, g):
        a = np.zeros(4096, dtype=np.int32)
        n = C.c_int32(0)
        _check(lib().agz_arena_get_history(self.h, g, _pi(a), a.size, C.byref(n)), "agz_arena_get_history")
        return a[:n.value].copy()

    def root_children(self, g, agent):
        cap = self.m * self.n + 2
        mv = np.zeros(cap, dtype=np.int32)
        vis = np.zeros(cap, dtype=np.uint32)
        bs = np.zeros(cap, dtype=np.float32)
        pr = np.zeros(cap, dtype=np.float32)
        n = C.c_int32(0)
        _check(lib().agz_arena_root_children(self.h, g, agent, _pi(mv), vis.ctypes.data_as(C.POINTER(C.c_uint32)),
                                             _pf(bs), _pf(pr), cap, C.byref(n)), "agz_arena_root_children")
        k = n.value
        return mv[:k].copy(), vis[:k].copy(), bs[:k].copy(), pr[:k].copy()

    def set_prep_compact(self, on=True):
        """agz_debug.h: prepareRoot's forward on the packed batch of the roots that need it (default) or on the whole arena batch"""
        _check(lib().agz_arena_set_prep_compact(self.h, int(on)), "agz_arena_set_prep_compact")

    def last_prep_batch(self):
        """agz_debug.h: (boards the last begin_move's forward ran on, roots it had to evaluate)"""
        b, r = C.c_int(0), C.c_int(0)
        _check(lib().agz_arena_last_prep_batch(self.h, C.byref(b), C.byref(r)), "agz_arena_last_prep_batch")
        return b.value, r.value

    def pool_capacity(self):
        """agz_debug.h: (nodes per pool, re-allocations by POOL_GROW)"""
        c, g = C.c_int(0), C.c_int(0)
        _check(lib().agz_arena_pool_capacity(self.h, C.byref(c), C.byref(g)), "agz_arena_pool_capacity")
        return c.value, g.value

    def max_path_nodes(self):
        """agz_debug.h AGZ_CNT_PATHMAX: nodes on the longest descent since the last reset"""
        v = C.c_int64(0)
        _check(lib().agz_arena_debug_counter(self.h, 15, C.byref(v)), "agz_arena_debug_counter")
        return v.value

    def tree_nodes(self, g, agent):
        n = C.c_int32(0)
        _check(lib().agz_arena_tree_nodes(self.h, g, agent, C.byref(n)), "agz_arena_tree_nodes")
        return n.value

    def examples(self):
        n = C.c_int32(0)
        _check(lib().agz_arena_get_examples(self.h, None, None, None, None, 0, C.byref(n)), "agz_arena_get_examples")
        k = n.value
        cells = self.m * self.n
        planes = np.zeros((k, self.features * cells), np.float32)
        policy = np.zeros((k, self.action_space + 1), np.float32)
        value = np.zeros(k, np.float32)
        gidx = np.zeros(k, np.int32)
        if k:
            _check(lib().agz_arena_get_examples(self.h, _pf(planes), _pf(policy), _pf(value), _pi(gidx), k,
                                                C.byref(n)), "agz_arena_get_examples")
        return planes, policy, value, gidx

    def clear_examples(self):
        _check(lib().agz_arena_clear_examples(self.h), "agz_arena_clear_examples")


def make_state(cells, board, to_move, n_moves=0, passes=0, hash=0, captures=(0.0, 0.0), last_moves=(), historical=()):
    """agz_state from numpy pieces; returns (struct, keep-alive list)"""
    b = np.ascontiguousarray(board, dtype=np.int32).reshape(-1)
    assert b.size == cells
    lm = np.ascontiguousarray(last_moves, dtype=np.int32).reshape(-1)
    hs = np.ascontiguousarray(historical, dtype=np.int32).reshape(-1)
    assert hs.size % cells == 0
    st = State()
    st.board = _pi(b)
    st.to_move, st.n_moves, st.passes, st.hash = int(to_move), int(n_moves), int(passes), int(hash) & 0xFFFFFFFF
    st.captures_black, st.captures_white = float(captures[0]), float(captures[1])
    st.last_moves = _pi(lm) if lm.size else None
    st.n_last_moves = lm.size
    st.historical = _pi(hs) if hs.size else None
    st.n_historical = hs.size // cells
    return st, [b, lm, hs]


class Mcts:
    """mcts.MCTS (mcts/tree.go:80-142, search.go:92): one search tree on a caller-owned game.State."""

    def __init__(self, ctx, kind, m, n, k=0, komi=0.0, encoder=ENC_TWOPLANE, seed=1337, max_nodes=0, max_moves=0, PUCT=1.0,
                 M=None, N=None, RandomCount=0, Budget=100, RandomMinVisits=0, RandomTemperature=0.0, DumbPass=True,
                 ResignPercentage=0.0, PassPreference=DONT_PREFER_PASS):
        self.ctx = ctx
        self.kind, self.m, self.n = kind, m, n
        self.action_space = n if kind == GAME_C4 else m * n
        self.gconf = GameConf(kind, m, n, k, komi, max_moves, encoder)
        self.mconf = MctsConf(PUCT, M if M is not None else m, N if N is not None else n, RandomCount, Budget,
                              RandomMinVisits, RandomTemperature, int(DumbPass), ResignPercentage, PassPreference)
        self.h = C.c_void_p()
        self._nets = []
        _check(lib().agz_mcts_create(ctx.h, C.byref(self.gconf), C.byref(self.mconf), seed, max_nodes, C.byref(self.h)),
               "agz_mcts_create")
        ctx._adopt(self)

    def close(self):
        if self.h and self.ctx.h:
            lib().agz_mcts_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_inferencer(self, kind, net=None):
        _check(lib().agz_mcts_set_inferencer(self.h, kind, net.h if net is not None else None), "agz_mcts_set_inferencer")
        if net is not None:
            self._nets.append(net)

    def set_inferencer_callback(self, fn, policy_len):
        """mcts.New(game, conf, nn) with a caller-supplied Inferencer: fn(leaves) -> (policy, value) (make_infer_fn)"""
        cfn = make_infer_fn(fn)
        _check(lib().agz_mcts_set_inferencer_callback(self.h, cfn, None, int(policy_len)), "agz_mcts_set_inferencer_callback")
        self._nets.append(cfn)

    def set_pool_policy(self, policy):
        _check(lib().agz_mcts_set_pool_policy(self.h, int(policy)), "agz_mcts_set_pool_policy")

    def set_parallel(self, lanes):
        _check(lib().agz_mcts_set_parallel(self.h, int(lanes)), "agz_mcts_set_parallel")

    def set_game(self, **kw):
        st, keep = make_state(self.m * self.n, **kw)
        _check(lib().agz_mcts_set_game(self.h, C.byref(st)), "agz_mcts_set_game")
        del keep

    def search(self, player):
        best = C.c_int32(0)
        _check(lib().agz_mcts_search(self.h, int(player), C.byref(best)), "agz_mcts_search")
        return best.value

    def policies(self):
        p = np.zeros(self.action_space + 1, np.float32)
        _check(lib().agz_mcts_policies(self.h, _pf(p), p.size), "agz_mcts_policies")
        return p

    def root_children(self):
        cap = self.m * self.n + 2
        mv = np.zeros(cap, dtype=np.int32)
        vis = np.zeros(cap, dtype=np.uint32)
        bs = np.zeros(cap, dtype=np.float32)
        pr = np.zeros(cap, dtype=np.float32)
        n = C.c_int32(0)
        _check(lib().agz_mcts_root_children(self.h, _pi(mv), vis.ctypes.data_as(C.POINTER(C.c_uint32)), _pf(bs), _pf(pr), cap,
                                            C.byref(n)), "agz_mcts_root_children")
        k = n.value
        return mv[:k].copy(), vis[:k].copy(), bs[:k].copy(), pr[:k].copy()

    def nodes(self):
        n = C.c_int32(0)
        _check(lib().agz_mcts_nodes(self.h, C.byref(n)), "agz_mcts_nodes")
        return n.value

    def set_timeout_ms(self, ms):
        """mcts.Config.Timeout (tree.go:18): > 0 = search by wall clock (not deterministic); 0 = exactly Budget simulations"""
        _check(lib().agz_mcts_set_timeout_ms(self.h, int(ms)), "agz_mcts_set_timeout_ms")

    def last_simulations(self):
        n = C.c_int64(0)
        _check(lib().agz_mcts_last_simulations(self.h, C.byref(n)), "agz_mcts_last_simulations")
        return n.value

    def to_dot(self, max_nodes=0):
        """(*MCTS).ToDot (mcts/graph.go:34): Graphviz text of the live tree.  max_nodes = 0 (default, as the reference and the Go shim):
        the whole tree — hundreds of MB for a deep 19x19 tree; max_nodes > 0: the first max_nodes nodes (a top of the tree)"""
        need = C.c_size_t(0)
        _check(lib().agz_mcts_to_dot(self.h, int(max_nodes), None, 0, C.byref(need)), "agz_mcts_to_dot")
        buf = C.create_string_buffer(need.value)
        _check(lib().agz_mcts_to_dot(self.h, int(max_nodes), buf, need.value, C.byref(need)), "agz_mcts_to_dot")
        return buf.value.decode("utf-8")

    def children(self, node=0):
        """(*MCTS).Children(of): ids, moves, visits, blackScores, priors of the children of any node (0 = root)"""
        cap = self.m * self.n + 2
        ids = np.zeros(cap, dtype=np.int32)
        mv = np.zeros(cap, dtype=np.int32)
        vis = np.zeros(cap, dtype=np.uint32)
        bs = np.zeros(cap, dtype=np.float32)
        pr = np.zeros(cap, dtype=np.float32)
        n = C.c_int32(0)
        _check(lib().agz_mcts_children(self.h, int(node), _pi(ids), _pi(mv), vis.ctypes.data_as(C.POINTER(C.c_uint32)), _pf(bs),
                                       _pf(pr), cap, C.byref(n)), "agz_mcts_children")
        k = n.value
        return ids[:k].copy(), mv[:k].copy(), vis[:k].copy(), bs[:k].copy(), pr[:k].copy()

    def stats(self):
        s = ArenaStats()
        _check(lib().agz_mcts_get_stats(self.h, C.byref(s)), "agz_mcts_get_stats")
        return {f: getattr(s, f) for f, _ in ArenaStats._fields_}

    def reset(self):
        _check(lib().agz_mcts_reset(self.h), "agz_mcts_reset")


class Comm:
    """agz_comm: an RCCL communicator over the devices of agz_ctx handles."""

    def __init__(self, h, ctx):
        self.h, self.ctx = h, ctx
        ctx._adopt(self)

    @staticmethod
    def init_all(ctxs):
        n = len(ctxs)
        arr = (C.c_void_p * n)(*[c.h for c in ctxs])
        out = (C.c_void_p * n)()
        _check(lib().agz_comm_init_all(arr, n, out), "agz_comm_init_all")
        return [Comm(C.c_void_p(out[i]), ctxs[i]) for i in range(n)]

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(lib().agz_comm_unique_id(buf), "agz_comm_unique_id")
        return buf.raw

    @staticmethod
    def init_rank(ctx, n_ranks, rank, id_bytes):
        out = C.c_void_p()
        buf = C.create_string_buffer(bytes(id_bytes), 128)
        _check(lib().agz_comm_init_rank(ctx.h, n_ranks, rank, buf, C.byref(out)), "agz_comm_init_rank")
        return Comm(out, ctx)

    def close(self):
        if self.h and self.ctx.h:
            lib().agz_comm_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rank(self):
        return lib().agz_comm_rank(self.h)

    def size(self):
        return lib().agz_comm_size(self.h)

    def allgather_examples(self, ex):
        _check(lib().agz_examples_allgather(self.h, ex.h), "agz_examples_allgather")

    def allreduce_trainer(self, trainer):
        _check(lib().agz_trainer_allreduce(self.h, trainer.h), "agz_trainer_allreduce")

    def forward_backward_allreduce(self, trainer, planes, pi, v):
        """the data-parallel step's gradients: forward / backward on this rank's batch with every slice of the flat gradient buffer
        summed over the ranks while the rest of the backward runs (agz_trainer_forward_backward_allreduce)"""
        x = np.ascontiguousarray(planes, np.float32)
        p = np.ascontiguousarray(pi, np.float32)
        vv = np.ascontiguousarray(v, np.float32)
        c = C.c_float(0)
        _check(lib().agz_trainer_forward_backward_allreduce(self.h, trainer.h, _pf(x), _pf(p), _pf(vv), C.byref(c)),
               "agz_trainer_forward_backward_allreduce")
        return c.value

    def debug_fail_slice(self, k):
        """agz_debug.h: this rank's next data-parallel step fails right before slice k (failure injection for the N > 1 tests)"""
        _check(lib().agz_comm_debug_fail_slice(self.h, int(k)), "agz_comm_debug_fail_slice")

    def forward_backward_allreduce_dev(self, trainer, planes_ptr, pi_ptr, v_ptr, want_cost=True):
        c = C.c_float(0)
        _check(lib().agz_trainer_forward_backward_allreduce_dev(self.h, trainer.h, C.c_void_p(planes_ptr), C.c_void_p(pi_ptr), C.c_void_p(v_ptr),
                                                                C.byref(c) if want_cost else None), "agz_trainer_forward_backward_allreduce_dev")
        return c.value if want_cost else None


class Examples:
    """[]agogo.Example on device + Augmenter / shuffleExamples / prepareExamples (agogo.go:110-133, 211-257)."""

    def __init__(self, ctx, Features, Height, Width, PolicyLen):
        self.ctx = ctx
        self.F, self.H, self.W, self.A1 = Features, Height, Width, PolicyLen
        self.h = C.c_void_p()
        _check(lib().agz_examples_create(ctx.h, Features, Height, Width, PolicyLen, C.byref(self.h)), "agz_examples_create")
        ctx._adopt(self)

    def close(self):
        if self.h and self.ctx.h:
            lib().agz_examples_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        n = C.c_int64(0)
        _check(lib().agz_examples_count(self.h, C.byref(n)), "agz_examples_count")
        return n.value

    def clear(self):
        _check(lib().agz_examples_clear(self.h), "agz_examples_clear")

    def append_arena(self, arena):
        _check(lib().agz_examples_append_arena(self.h, arena.h), "agz_examples_append_arena")

    def append_dev(self, planes_ptr, policy_ptr, value_ptr, n):
        _check(lib().agz_examples_append_dev(self.h, C.c_void_p(planes_ptr), C.c_void_p(policy_ptr), C.c_void_p(value_ptr), n),
               "agz_examples_append_dev")

    def append_host(self, planes, policy, value):
        p = np.ascontiguousarray(planes, np.float32)
        q = np.ascontiguousarray(policy, np.float32)
        v = np.ascontiguousarray(value, np.float32)
        n = v.size
        assert p.size == n * self.F * self.H * self.W and q.size == n * self.A1
        _check(lib().agz_examples_append_host(self.h, _pf(p), _pf(q), _pf(v), n), "agz_examples_append_host")

    def get(self):
        n = len(self)
        p = np.zeros((n, self.F * self.H * self.W), np.float32)
        q = np.zeros((n, self.A1), np.float32)
        v = np.zeros(n, np.float32)
        k = C.c_int64(0)
        _check(lib().agz_examples_get(self.h, _pf(p), _pf(q), _pf(v), n, C.byref(k)), "agz_examples_get")
        return p, q, v

    def raw_dev(self):
        """device pointers of the raw store (Board, Policy, Value rows in append order)"""
        x, p, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().agz_examples_raw_dev(self.h, C.byref(x), C.byref(p), C.byref(v)), "agz_examples_raw_dev")
        return x.value, p.value, v.value

    def augment_rotate(self):
        _check(lib().agz_examples_augment_rotate(self.h), "agz_examples_augment_rotate")

    def prepare(self, BatchSize, maxExamples=0, seed=1337):
        b = C.c_int32(0)
        _check(lib().agz_examples_prepare(self.h, BatchSize, maxExamples, seed, C.byref(b)), "agz_examples_prepare")
        return b.value

    def tensors_dev(self):
        x, p, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
        rows, b = C.c_int64(0), C.c_int32(0)
        _check(lib().agz_examples_tensors_dev(self.h, C.byref(x), C.byref(p), C.byref(v), C.byref(rows), C.byref(b)),
               "agz_examples_tensors_dev")
        return x.value, p.value, v.value, rows.value, b.value

    def tensors(self):
        _, _, _, rows, _ = self.tensors_dev()
        x = np.zeros((rows, self.F, self.H, self.W), np.float32)
        p = np.zeros((rows, self.A1), np.float32)
        v = np.zeros(rows, np.float32)
        _check(lib().agz_examples_get_tensors(self.h, _pf(x), _pf(p), _pf(v)), "agz_examples_get_tensors")
        return x, p, v


def wino_h2_chained(H, W, K):
    """agz_debug.h: 1 when AGZ_COMPUTE_WINO_H2 takes the chained block on this shape"""
    return int(lib().agz_wino_h2_chained(H, W, K))


def wino_h2_tile(H, W):
    """measurement: Winograd tile size (4 or 5) COMPUTE_WINO_H2 uses on an H x W board (agz_debug.h)"""
    return int(lib().agz_wino_h2_tile(H, W))


def wino_stages(ctx, x, w):
    """diagnostics: Winograd F(4x4,3x3) input transform and transform-domain GEMMs. x [B,H,W,C], w [N,C,3,3] -> V [36,T,C], M [36,T,N]"""
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    B, H, W, Cc = x.shape
    N = w.shape[0]
    T = B * ((H + 3) // 4) * ((W + 3) // 4)
    V = np.zeros((36, T, Cc), np.float32)
    M = np.zeros((36, T, N), np.float32)
    _check(lib().agz_wino_stages(ctx.h, _pf(x), _pf(w), B, H, W, Cc, N, _pf(V), _pf(M)), "agz_wino_stages")
    return V, M


def rotate_boards(ctx, boards, m, n):
    """RotateBoard (encoding_helper.go:80-107) on a stack of m x n boards"""
    b = np.ascontiguousarray(boards, np.float32)
    count = b.size // (m * n)
    out = np.zeros_like(b)
    _check(lib().agz_rotate_boards(ctx.h, _pf(b), count, m, n, _pf(out)), "agz_rotate_boards")
    return out
