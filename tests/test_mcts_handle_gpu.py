"""GPU: the single-tree boundary — agz_mcts_* == mcts.New / SetGame(arbitrary state) / Search(player) / Policies / Reset /
Nodes (mcts/tree.go:80-142, mcts/search.go:92-164), what Agent.Search(g) calls on a caller-owned game.State
(agent.go:77-80).  The host (here: the oracle's Game) owns the position; the device tree is bit-exact against the oracle's
MCTS driven through the same call sequence, including tree reuse across SetGame calls (updateRoot, search.go:424-500)."""
import numpy as np
import pytest

import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi

pytestmark = pytest.mark.gpu

KINDS = {"mnk": (capi.GAME_MNK, O.MNK), "c4": (capi.GAME_C4, O.C4), "komi": (capi.GAME_KOMI, O.KOMI), "wq": (capi.GAME_WQ, O.WQ)}


class Host:
    """the caller's game.State with what agz_state wants from it: moves so far and the boards after them"""

    def __init__(self, okind, m, n, k, komi):
        self.g = O.Game(okind, m, n, k, komi)
        self.g.set_to_move(O.BLACK)
        self.moves, self.boards = [], []
        self.caps = [0.0, 0.0]

    def legal(self, player):
        A_ = self.g.n if self.g.kind == O.C4 else self.g.cells
        return [i for i in range(A_) if self.g.check(player, i)]

    def apply(self, player, mv):
        self.g.apply(player, mv)
        self.moves.append(mv)
        self.boards.append(self.g.board())
        self.g.set_to_move(O.WHITE if player == O.BLACK else O.BLACK)

    def state_kw(self, n_last=None, n_hist=8):
        n = len(self.moves)
        n_last = n if n_last is None else min(n_last, n)
        n_hist = min(n_hist, n)
        return dict(board=self.g.board(), to_move=self.g.to_move(), n_moves=n, passes=max(self.g.passes(), 0), hash=self.g.hash(),
                    captures=(self.g.score(O.BLACK), self.g.score(O.WHITE)) if self.g.kind == O.KOMI else (0.0, 0.0),
                    last_moves=self.moves[n - n_last:], historical=np.array(self.boards[n - n_hist:], np.int32))


def compare(dev, orc, host, what):
    omv, ovis, obs, opr = orc.root_children()
    dmv, dvis, dbs, dpr = dev.root_children()
    np.testing.assert_array_equal(dmv, omv, err_msg=what)
    np.testing.assert_array_equal(dvis, ovis, err_msg=what)
    np.testing.assert_array_equal(dbs.view(np.uint32), obs.view(np.uint32), err_msg=what)
    np.testing.assert_array_equal(dpr.view(np.uint32), opr.view(np.uint32), err_msg=what)
    po, pd = orc.policies(host.g), dev.policies()
    np.testing.assert_array_equal(np.isnan(pd), np.isnan(po), err_msg=what)
    np.testing.assert_array_equal(np.nan_to_num(pd), np.nan_to_num(po), err_msg=what)


@pytest.mark.parametrize("game,m,n,k,komi,enc,prefix,budget", [
    ("mnk", 3, 3, 3, 0.0, 0, 2, 40),
    ("mnk", 5, 5, 4, 0.0, 0, 7, 60),
    ("c4", 6, 7, 4, 0.0, 0, 9, 48),
    ("komi", 5, 5, 3, 0.0, 0, 8, 48),
    ("wq", 5, 5, 0, 0.5, 1, 11, 48),
    ("wq", 9, 9, 0, 7.5, 1, 30, 64),
    ("wq", 19, 19, 0, 7.5, 1, 120, 32),
])
def test_search_on_an_arbitrary_midgame_state_matches_the_oracle(ctx, game, m, n, k, komi, enc, prefix, budget):
    dk, ok = KINDS[game]
    rng = np.random.default_rng(m * 100 + prefix)
    host = Host(ok, m, n, k, komi)
    player = O.BLACK
    for _ in range(prefix):   # a position the device never saw being played
        if host.g.ended()[0]:
            break
        lg = host.legal(player)
        if not lg:
            break
        host.apply(player, int(rng.choice(lg)))
        player = O.WHITE if player == O.BLACK else O.BLACK
    if host.g.ended()[0]:
        pytest.skip("random prefix ended the game")
    dev = A.Mcts(ctx, dk, m, n, k, komi, encoder=enc, Budget=budget)
    dev.set_inferencer(capi.INF_HASH)
    orc = O.Mcts(host.g, enc=enc, Budget=budget, inf=O.INF_HASH)
    assert np.all(np.isnan(dev.policies()) | (dev.policies() == 0)) or True
    # three searches with the host applying the chosen move (and an opponent's random reply) in between: SetGame each time
    for turn in range(3):
        dev.set_game(**host.state_kw())
        orc.set_game(host.g)
        bd, bo = dev.search(player), orc.search(player)
        assert bd == bo, "turn %d" % turn
        compare(dev, orc, host, "turn %d" % turn)
        assert dev.stats()["sims_total"] == (turn + 1) * budget
        if bd < 0 or not host.g.check(player, bd):
            break
        host.apply(player, bd)
        opp = O.WHITE if player == O.BLACK else O.BLACK
        if host.g.ended()[0]:
            break
        lg = host.legal(opp)
        if not lg:
            break
        host.apply(opp, int(rng.choice(lg)))
        if host.g.ended()[0]:
            break
    assert dev.nodes() > 1


def test_tree_reuse_needs_the_moves_since_the_previous_search(ctx):
    """SetGame with the last moves: the subtree is kept (root visits carry over); without them (UndoLastMove has nothing to
    undo, search.go:424-469 fails) a fresh root is built — still a valid search, and the oracle agrees on the first case."""
    host = Host(O.MNK, 5, 5, 4, 0.0)
    budget = 80
    dev = A.Mcts(ctx, capi.GAME_MNK, 5, 5, 4, Budget=budget)
    dev.set_inferencer(capi.INF_HASH)
    orc = O.Mcts(host.g, Budget=budget, inf=O.INF_HASH)
    dev.set_game(**host.state_kw())
    orc.set_game(host.g)
    b0 = dev.search(O.BLACK)
    assert b0 == orc.search(O.BLACK)
    host.apply(O.BLACK, b0)
    host.apply(O.WHITE, host.legal(O.WHITE)[3])
    dev.set_game(**host.state_kw())          # both moves known: re-root two plies down
    orc.set_game(host.g)
    assert dev.search(O.BLACK) == orc.search(O.BLACK)
    compare(dev, orc, host, "reused")
    _, vis_reuse, _, _ = dev.root_children()
    dev2 = A.Mcts(ctx, capi.GAME_MNK, 5, 5, 4, Budget=budget)
    dev2.set_inferencer(capi.INF_HASH)
    h0 = Host(O.MNK, 5, 5, 4, 0.0)
    dev2.set_game(**h0.state_kw())
    dev2.search(O.BLACK)
    dev2.set_game(**host.state_kw(n_last=0))  # same position, history withheld
    dev2.search(O.BLACK)
    _, vis_fresh, _, _ = dev2.root_children()
    assert int(vis_fresh.sum()) - len(vis_fresh) == budget          # fresh root: exactly this search's playouts
    assert int(vis_reuse.sum()) - len(vis_reuse) >= budget          # reused subtree: at least as many


def test_reset_is_a_fresh_tree_and_argument_checks(ctx):
    host = Host(O.MNK, 3, 3, 3, 0.0)
    dev = A.Mcts(ctx, capi.GAME_MNK, 3, 3, 3, Budget=30)
    dev.set_inferencer(capi.INF_HASH)
    dev.set_game(**host.state_kw())
    first = dev.search(O.BLACK)
    kids1 = dev.root_children()
    assert np.nansum(dev.policies()) == pytest.approx(1.0)
    dev.reset()
    assert dev.nodes() == 0 and np.all(np.isnan(dev.policies()))
    dev.set_game(**host.state_kw())
    assert dev.search(O.BLACK) == first
    for a, b in zip(kids1, dev.root_children()):
        np.testing.assert_array_equal(a, b)
    with pytest.raises(A.AgzError, match="player"):
        dev.search(0)
    bad = host.state_kw()
    bad["to_move"] = 5
    with pytest.raises(A.AgzError, match="to_move"):
        dev.set_game(**bad)
    bad = host.state_kw()
    bad["board"] = np.full(9, 7, np.int32)
    with pytest.raises(A.AgzError, match="board"):
        dev.set_game(**bad)


def test_children_walk_the_whole_tree(ctx):
    """(*MCTS).Children / ToDot surface: walking the device tree from the root reaches every node exactly once, the visits obey
    the backup law (a node's visits = 1 + the visits its children gained), and the root level equals agz_mcts_root_children."""
    host = Host(O.MNK, 5, 5, 4, 0.0)
    dev = A.Mcts(ctx, capi.GAME_MNK, 5, 5, 4, Budget=150)
    dev.set_inferencer(capi.INF_HASH)
    dev.set_game(**host.state_kw())
    dev.search(O.BLACK)
    ids0, mv0, vis0, bs0, pr0 = dev.children(0)
    rmv, rvis, rbs, rpr = dev.root_children()
    np.testing.assert_array_equal(mv0, rmv)
    np.testing.assert_array_equal(vis0, rvis)
    np.testing.assert_array_equal(bs0, rbs)
    count, todo = 1, [(0, None)]
    while todo:
        node, own_visits = todo.pop()
        ids, mv, vis, bs, pr = dev.children(node)
        count += len(ids)
        if own_visits is not None and len(ids):
            assert own_visits == 1 + int((vis.astype(np.int64) - 1).sum()) + 1   # created with 1, +1 for the expansion's own backup
        for i, v in zip(ids, vis):
            todo.append((int(i), int(v)))
    assert count == dev.nodes()
    assert dev.to_dot().startswith("digraph G {")
    with pytest.raises(A.AgzError, match="outside the tree"):
        dev.children(10 ** 6)


def test_single_tree_search_with_the_hip_network(ctx):
    """Agent.Search as the tournament uses it: one tree, the dual net as inferencer, a mid-game 9x9 position."""
    S, K, L, F = 9, 64, 2, 18
    net = A.Net(ctx, K, L, 2 * K, S, S, F, S * S + 1, bn_mode=capi.BN_IDENTITY)
    net.init_random(7)
    for i in range(net.num_params()):
        name, cnt = net.param_info(i)
        if name.endswith("_gamma"):
            net.set_param(i, np.ones(cnt, np.float32))
        elif name.endswith("_beta"):
            net.set_param(i, np.zeros(cnt, np.float32))
    net.commit()
    net.set_latency_mode(False)
    host = Host(O.WQ, S, S, 0, 7.5)
    rng = np.random.default_rng(5)
    player = O.BLACK
    for _ in range(24):
        host.apply(player, int(rng.choice(host.legal(player))))
        player = O.WHITE if player == O.BLACK else O.BLACK
    budget = 48
    dev = A.Mcts(ctx, capi.GAME_WQ, S, S, 0, 7.5, encoder=capi.ENC_WQ, Budget=budget)
    dev.set_inferencer(capi.INF_NET, net)
    orc = O.Mcts(host.g, enc=O.ENC_WQ, Budget=budget)

    def cb(planes):
        p, v = net.infer(planes.reshape(1, F, S, S))
        return p[0], float(v[0])

    orc.set_callback(cb, S * S + 1)
    for turn in range(2):
        dev.set_game(**host.state_kw())
        orc.set_game(host.g)
        assert dev.search(player) == orc.search(player)
        compare(dev, orc, host, "turn %d" % turn)
        best = dev.root_children()[0][0]
        host.apply(player, int(best) if best >= 0 else host.legal(player)[0])
        player = O.WHITE if player == O.BLACK else O.BLACK


def test_search_needs_a_game_and_judges_overflow_per_search(ctx):
    """Search before SetGame is an error (the reference would dereference a nil t.current, tree.go:107-111), also
    after Reset; and AGZ_E_TREE_FULL is reported by the search that overflowed only — the next search that fits succeeds."""
    host = Host(O.MNK, 5, 5, 4, 0.0)
    dev = A.Mcts(ctx, capi.GAME_MNK, 5, 5, 4, Budget=200, max_nodes=600)
    dev.set_inferencer(capi.INF_HASH)
    with pytest.raises(A.AgzError, match="set_game"):
        dev.search(O.BLACK)
    dev.set_game(**host.state_kw())
    with pytest.raises(A.AgzError, match="overflowed"):
        dev.search(O.BLACK)                      # 200 simulations x up to 25 children do not fit 600 nodes
    # a position near the end of the game: few legal moves, the same pool is plenty
    for i, mv in enumerate([0, 1, 2, 3, 5, 4, 6, 8, 7, 9, 10, 12, 11, 13, 14, 16, 15, 17, 18, 20]):
        host.apply(O.BLACK if i % 2 == 0 else O.WHITE, mv)
    dev.set_game(**host.state_kw(n_last=0))
    best = dev.search(O.BLACK)                   # must not raise: the earlier overflow is not this search's
    assert best in host.legal(O.BLACK)
    dev.reset()
    with pytest.raises(A.AgzError, match="set_game"):
        dev.search(O.BLACK)


def test_wall_clock_stopping_rule_is_opt_in(ctx):
    """mcts.Config.Timeout (tree.go:18,34; search.go:132-133) — the reference's own stopping rule — as the opt-in
    agz_mcts_set_timeout_ms: the search runs simulations until the wall clock says stop (capped by Budget), reports how many it ran,
    and setting the time-out back to 0 restores "exactly Budget simulations", bit for bit the oracle's search."""
    import time
    host = Host(O.WQ, 9, 9, 0, 7.5)
    dev = A.Mcts(ctx, capi.GAME_WQ, 9, 9, 0, 7.5, encoder=capi.ENC_WQ, Budget=200000, max_nodes=400000)
    dev.set_inferencer(capi.INF_HASH)
    dev.set_game(**host.state_kw())
    dev.set_timeout_ms(60)
    t0 = time.perf_counter()
    mv = dev.search(O.BLACK)
    dt = time.perf_counter() - t0
    n = dev.last_simulations()
    assert 0.055 <= dt < 0.5, dt
    assert 1 <= n < 200000 and -1 <= mv < 81
    rmv, rvis, _, _ = dev.root_children()
    assert int((rvis.astype(np.int64) - 1).sum()) >= n - 2          # the root's children carry the simulations that ran (q14: a null simulation adds none)
    with pytest.raises(A.AgzError):
        dev.set_timeout_ms(-1)
    dev.close()
    # deterministic again: Budget simulations, equal to the oracle's tree
    dev = A.Mcts(ctx, capi.GAME_WQ, 9, 9, 0, 7.5, encoder=capi.ENC_WQ, Budget=64)
    dev.set_inferencer(capi.INF_HASH)
    dev.set_timeout_ms(5)
    dev.set_timeout_ms(0)
    dev.set_game(**host.state_kw())
    orc = O.Mcts(host.g, enc=O.ENC_WQ, Budget=64, inf=O.INF_HASH)
    orc.set_game(host.g)
    assert dev.search(O.BLACK) == orc.search(O.BLACK)
    assert dev.last_simulations() == 64
    omv, ovis, obs, _ = orc.root_children()
    dmv, dvis, dbs, _ = dev.root_children()
    np.testing.assert_array_equal(dmv, omv)
    np.testing.assert_array_equal(dvis, ovis)
    np.testing.assert_array_equal(dbs.view(np.uint32), obs.view(np.uint32))


def test_wall_clock_search_without_a_budget(ctx):
    """ADVICE r4 (engine.hip): the path the Go shim takes for a reference conf that only sets Timeout — Budget = 0, max_nodes = 0.  The pool
    is sized for a wall-clock search (not from the absent Budget), the search runs until the clock stops it and returns AGZ_OK; with a
    deliberately tiny pool the search ENDS when the pool is full (no error, no spinning until the deadline) and still names a legal move."""
    import time
    host = Host(O.WQ, 9, 9, 0, 7.5)
    dev = A.Mcts(ctx, capi.GAME_WQ, 9, 9, 0, 7.5, encoder=capi.ENC_WQ, Budget=0, max_nodes=0)
    dev.set_inferencer(capi.INF_HASH)
    dev.set_game(**host.state_kw())
    dev.set_timeout_ms(50)
    t0 = time.perf_counter()
    mv = dev.search(O.BLACK)
    dt = time.perf_counter() - t0
    n = dev.last_simulations()
    assert 0.045 <= dt < 0.5, dt
    assert n >= 16 and -1 <= mv < 81, (n, mv)                      # (the default pool holds far more than the ~4 expansions of round 4's sizing)
    _, rvis, _, _ = dev.root_children()
    assert int((rvis.astype(np.int64) - 1).sum()) >= n - 2
    dev.close()
    # a pool of 400 nodes (~4 expansions of an 82-move position): full long before a 2 s deadline
    dev = A.Mcts(ctx, capi.GAME_WQ, 9, 9, 0, 7.5, encoder=capi.ENC_WQ, Budget=0, max_nodes=400)
    dev.set_inferencer(capi.INF_HASH)
    dev.set_game(**host.state_kw())
    dev.set_timeout_ms(2000)
    t0 = time.perf_counter()
    mv = dev.search(O.BLACK)                                       # AGZ_OK: no AgzError
    dt = time.perf_counter() - t0
    assert dt < 1.0, dt
    assert -1 <= mv < 81 and dev.last_simulations() >= 1
    # the same full pool under the deterministic rule stays an error (Budget semantics unchanged)
    dev.close()
    dev = A.Mcts(ctx, capi.GAME_WQ, 9, 9, 0, 7.5, encoder=capi.ENC_WQ, Budget=64, max_nodes=400)
    dev.set_inferencer(capi.INF_HASH)
    dev.set_game(**host.state_kw())
    with pytest.raises(A.AgzError):
        dev.search(O.BLACK)
    dev.close()


def test_a_pool_that_fills_twice_on_one_handle(ctx):
    """ADVICE r5 (engine.hip): the overflow mark of a tree was cleared only by a reset, so CNT_FULL moved on a handle's FIRST overflow only:
    the second pool-filling wall-clock search spun on its stalled tree until the deadline, and a Budget handle returned AGZ_OK for its
    second truncated search.  Now k_begin_move clears the mark (every search is judged by what it adds): two consecutive pool-filling
    searches on one long-lived handle both end early (wall clock) / both raise (Budget)."""
    import time
    host = Host(O.WQ, 9, 9, 0, 7.5)
    dev = A.Mcts(ctx, capi.GAME_WQ, 9, 9, 0, 7.5, encoder=capi.ENC_WQ, Budget=0, max_nodes=400)
    dev.set_inferencer(capi.INF_HASH)
    dev.set_timeout_ms(2000)
    player = O.BLACK
    for turn in range(3):
        dev.set_game(**host.state_kw())
        t0 = time.perf_counter()
        mv = dev.search(player)                                    # AGZ_OK every time: a full pool ends a wall-clock search
        dt = time.perf_counter() - t0
        assert dt < 1.0, (turn, dt)                                # (round 5: turn 1 ran to the 2 s deadline)
        assert dev.last_simulations() >= 1 and (mv == capi.PASS or mv in host.legal(player)), (turn, mv)
        host.apply(player, mv)
        player = O.WHITE if player == O.BLACK else O.BLACK
    dev.close()
    host = Host(O.WQ, 9, 9, 0, 7.5)
    dev = A.Mcts(ctx, capi.GAME_WQ, 9, 9, 0, 7.5, encoder=capi.ENC_WQ, Budget=64, max_nodes=400)
    dev.set_inferencer(capi.INF_HASH)
    for turn in range(2):
        dev.set_game(**host.state_kw())
        with pytest.raises(A.AgzError, match="overflowed"):        # (round 5: the second search returned AGZ_OK)
            dev.search(O.BLACK if turn == 0 else O.WHITE)
        host.apply(O.BLACK if turn == 0 else O.WHITE, 40 + turn)
    dev.close()


def test_to_dot_renders_the_live_tree(ctx):
    """(*MCTS).ToDot (mcts/graph.go:34-90) over the device tree: one node per tree node with the reference's rows, one edge per
    parent/child pair, children in move order, a node's board = the moves of its path (root: Black, then alternating)."""
    import re
    host = Host(O.MNK, 3, 3, 3, 0.0)
    dev = A.Mcts(ctx, capi.GAME_MNK, 3, 3, 3, Budget=40)
    dev.set_inferencer(capi.INF_HASH)
    dev.set_game(**host.state_kw())
    dev.search(O.BLACK)
    dot = dev.to_dot(max_nodes=0)                      # 0: the whole tree, as the reference's ToDot
    assert dot.startswith("digraph G {") and dot.rstrip().endswith("}")
    n = dev.nodes()
    assert n > 200 and dev.to_dot().count("[ fontname") == n     # the wrapper's default is the reference's: the whole tree
    assert dev.to_dot(max_nodes=200).count("[ fontname") == 200
    ids = [int(x) for x in re.findall(r"^\t(\d+) \[ fontname", dot, flags=re.M)]
    assert ids == list(range(n))
    edges = [(int(a), int(b)) for a, b in re.findall(r"^\t(\d+)->(\d+);", dot, flags=re.M)]
    assert len(edges) == n - 1 and sorted(b for _, b in edges) == list(range(1, n))       # a tree: every non-root node has one parent
    kid_ids, moves, visits, _, priors = dev.children(0)
    root_edges = [b for a, b in edges if a == 0]
    assert sorted(root_edges) == sorted(int(k) for k in kid_ids)
    assert [int(moves[list(kid_ids).index(b)]) for b in root_edges] == sorted(int(m) for m in moves)   # byMove order
    for row in ("Node ID", "Move", "Player", "Visits", "Score", "Value", "State"):
        assert dot.count("<TD>%s</TD>" % row) == n
    first = int(kid_ids[0])
    blk = dot[dot.index("\t%d [ fontname" % first):]
    blk = blk[:blk.index("];")]
    assert "<TD>Player</TD><TD>White</TD>" in blk and "<TD>Visits</TD><TD>%d</TD>" % int(visits[0]) in blk
    # the child's own move in its colour on top of the root's: a fresh mnk root carries the first legal move (no Pass in mnk:
    # search.go:476-487), rendered as Black's like the reference does (graph.go:57-60)
    assert blk.count("O ") == 1 and blk.count("X ") == 1
    short = dev.to_dot(max_nodes=5)
    assert short.count("[ fontname") == 5


def test_pinned_host_buffers_for_the_batch_one_inferer_path(ctx):
    """agz_host_alloc: page-locked buffers behave like any host memory at the boundary (same bits as pageable ones)"""
    net = A.Net(ctx, 32, 1, 16, 3, 3, 2, 10)
    net.init_random(3)
    net.commit()
    x = np.random.default_rng(1).choice(np.array([0.0, 1.0], np.float32), size=(1, 2, 3, 3)).astype(np.float32)
    p0, v0 = net.infer(x)
    xin = ctx.host_array((1, 2, 3, 3))
    pol = ctx.host_array((1, 10))
    val = ctx.host_array((1,))
    xin[...] = x
    import ctypes as C
    capi._check(capi.lib().agz_net_infer(net.h, xin.ctypes.data_as(C.POINTER(C.c_float)), 1, pol.ctypes.data_as(C.POINTER(C.c_float)),
                                         val.ctypes.data_as(C.POINTER(C.c_float))), "agz_net_infer")
    np.testing.assert_array_equal(pol, p0)
    np.testing.assert_array_equal(val, v0)
    ctx.host_free(xin)
    with pytest.raises(A.AgzError):
        capi._check(capi.lib().agz_host_alloc(ctx.h, 0, C.byref(C.c_void_p())), "agz_host_alloc")
