// Package agzhip binds libagz.so (MI355X self-play hot path) into gorgonia/agogo through cgo.
//
// NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no Go toolchain.  This is the binding a
// maintainer adds on the reference side (see INTEGRATION.md).  Every exported Go type implements the
// reference interface it replaces:
//
//	Inferencer    -> agogo.Inferer            (datatypes.go:56-59)   batch-1 path for Agent.Infer
//	BatchedArena  -> the role of Arena.Play / AZ.SelfPlay (arena.go:80-179, agogo.go:93-97) for N games
//	MCTS          -> *mcts.MCTS's method set  (mcts/tree.go:80-142, search.go:92): SetGame / Search / Policies / Reset / Nodes
//	                 on the caller's own game.State — what Agent.Search (agent.go:77-80) drives
//	Trainer/Train -> dual.Train               (dualnet/meta.go:16-54)
//	Comm          -> the example gather + gradient sum around dual.Train on n GPUs (agogo.go:118-133) over RCCL
//
// Threading: an agz_ctx is not thread-safe; every method locks the OS thread for the duration of the call
// and serialises on the Ctx mutex (under `-tags cuda` agogo itself runs a single VM, const_cuda.go:5).
package agzhip

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../agogo_amd/lib -lagz
#include <stdlib.h>
#include "agz.h"

// the host-inferencer trampoline (AGZ_INF_CALLBACK): exported from Go below, handed to libagz as an agz_infer_fn
// (declaration only: a cgo preamble next to //export directives may not hold definitions)
extern int agzGoInfer(void* user, agz_leaf_batch* batch);
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"runtime/cgo"
	"sync"
	"time"
	"unsafe"

	"github.com/gorgonia/agogo"
	dual "github.com/gorgonia/agogo/dualnet"
	"github.com/gorgonia/agogo/game"
	"github.com/gorgonia/agogo/mcts"
)

func lastErr(code C.int) error {
	if code == 0 {
		return nil
	}
	return errors.New(C.GoString(C.agz_last_error()))
}

// Ctx owns one HIP device + stream.
type Ctx struct {
	mu sync.Mutex
	h  *C.agz_ctx
}

func NewCtx(device int) (*Ctx, error) {
	c := &Ctx{}
	if err := lastErr(C.agz_ctx_create(C.int(device), &c.h)); err != nil {
		return nil, err
	}
	return c, nil
}

func (c *Ctx) Close() error { c.mu.Lock(); defer c.mu.Unlock(); C.agz_ctx_destroy(c.h); c.h = nil; return nil }

func (c *Ctx) enter() func() {
	runtime.LockOSThread()
	c.mu.Lock()
	return func() { c.mu.Unlock(); runtime.UnlockOSThread() }
}

// Net is a device-resident dual network.
type Net struct {
	ctx  *Ctx
	h    *C.agz_net
	conf dual.Config
}

// NewNet mirrors dual.New + Init (dualnet/dual.go:33-47) and copies the learnables of d in Model() order
// exactly like dual.Infer's copy loop (dualnet/meta.go:141-146).  bnMode selects the BatchNorm inference
// reading (C.AGZ_BN_*).
func NewNet(ctx *Ctx, d *dual.Dual, bnMode int) (*Net, error) {
	defer ctx.enter()()
	cc := C.agz_net_conf{
		K: C.int32_t(d.K), SharedLayers: C.int32_t(d.SharedLayers), FC: C.int32_t(d.FC), BatchSize: C.int32_t(d.BatchSize),
		Width: C.int32_t(d.Width), Height: C.int32_t(d.Height), Features: C.int32_t(d.Features),
		ActionSpace: C.int32_t(d.ActionSpace), bn_mode: C.int32_t(bnMode), bn_eps: 1e-5,
	}
	n := &Net{ctx: ctx, conf: d.Config}
	if err := lastErr(C.agz_net_create(ctx.h, &cc, &n.h)); err != nil {
		return nil, err
	}
	for i, node := range d.Model() {
		data := node.Value().Data().([]float32)
		if err := lastErr(C.agz_net_set_param(n.h, C.int(i), (*C.float)(unsafe.Pointer(&data[0])), C.size_t(len(data)))); err != nil {
			return nil, err
		}
	}
	if err := lastErr(C.agz_net_commit(n.h)); err != nil {
		return nil, err
	}
	return n, nil
}

// SetComputeMode selects the dual-block conv arithmetic: ComputeF32MFMA (default), ComputeBF16X3 (exact 3-way bf16
// split, six bf16 MFMAs per product; same parity tolerance, ~1.6x faster at self-play batch sizes) or ComputeFP16X2.
func (n *Net) SetComputeMode(mode int) error {
	defer n.ctx.enter()()
	return lastErr(C.agz_net_set_compute_mode(n.h, C.int(mode)))
}

const (
	ComputeF32MFMA = C.AGZ_COMPUTE_F32_MFMA
	ComputeBF16X3  = C.AGZ_COMPUTE_BF16X3
	ComputeFP16X2  = C.AGZ_COMPUTE_FP16X2 // opt-in: range-managed 2-way fp16 split, 3 MFMAs per product
	ComputeWino    = C.AGZ_COMPUTE_WINO   // opt-in: Winograd F(4x4,3x3) with bf16x3 products, ~1.9x BF16X3 at self-play batch sizes
	ComputeWinoH2  = C.AGZ_COMPUTE_WINO_H2 // Winograd F(5x5,3x3) / F(4x4,3x3) with range-managed fp16x2 products: the fastest at self-play batch sizes
	ComputeForce   = C.AGZ_COMPUTE_FORCE  // flag: keep the mode's tower at every batch size
)

func (n *Net) Close() error { defer n.ctx.enter()(); C.agz_net_destroy(n.h); n.h = nil; return nil }

// Inferencer implements agogo.Inferer over a Net (the batch-1 path; the batched path is BatchedArena).  The board, policy and
// value of a call travel through page-locked staging buffers (agz_host_alloc): Go memory cannot be pinned, a pageable buffer
// costs a driver bounce copy per direction on the one-board-per-call path (agent.go:60-74).
type Inferencer struct {
	*Net
	in, pol, val unsafe.Pointer // pinned: [Features*H*W], [ActionSpace], [1] floats
}

var _ agogo.Inferer = (*Inferencer)(nil)

// NewInferencer is dual.Infer's role (dualnet/meta.go:125): an inference handle over committed weights.
func NewInferencer(n *Net) (*Inferencer, error) {
	defer n.ctx.enter()()
	m := &Inferencer{Net: n}
	nin := C.size_t(n.conf.Features * n.conf.Height * n.conf.Width * 4)
	free := func() { // a later allocation failed: give back what was taken (agz_host_free accepts nil)
		for _, p := range []unsafe.Pointer{m.in, m.pol, m.val} {
			C.agz_host_free(n.ctx.h, p)
		}
	}
	if err := lastErr(C.agz_host_alloc(n.ctx.h, nin, &m.in)); err != nil {
		return nil, err
	}
	if err := lastErr(C.agz_host_alloc(n.ctx.h, C.size_t(n.conf.ActionSpace*4), &m.pol)); err != nil {
		free()
		return nil, err
	}
	if err := lastErr(C.agz_host_alloc(n.ctx.h, 4, &m.val)); err != nil {
		free()
		return nil, err
	}
	return m, nil
}

// Infer evaluates one encoded board (dualnet/meta.go:168-190).  The returned slice is freshly allocated —
// the reference returns a slice aliasing the VM output (meta.go:186-189), a latent race not reproduced here.
func (m *Inferencer) Infer(board []float32) (policy []float32, value float32, err error) {
	if m.Net == nil || m.in == nil || m.pol == nil || m.val == nil {
		return nil, 0, errors.New("agzhip: Inferencer without staging buffers (use NewInferencer; a literal Inferencer{Net: n} has none)")
	}
	nin := m.conf.Features * m.conf.Height * m.conf.Width
	if len(board) != nin {
		return nil, 0, errors.New("agzhip: Infer wants Features*Height*Width floats")
	}
	defer m.ctx.enter()()
	copy(unsafe.Slice((*float32)(m.in), nin), board)
	err = lastErr(C.agz_net_infer(m.h, (*C.float)(m.in), 1, (*C.float)(m.pol), (*C.float)(m.val)))
	policy = make([]float32, m.conf.ActionSpace)
	copy(policy, unsafe.Slice((*float32)(m.pol), m.conf.ActionSpace))
	return policy, *(*float32)(m.val), err // Agent.Infer panics on err (agent.go:66-71): behaviour preserved by the caller
}

// Close releases the staging buffers (agogo.Inferer.Close).  OWNERSHIP: the Net stays with its owner — this Close shadows the
// embedded Net.Close on purpose (an Inferer handed to agogo must not tear down weights other agents still use); whoever made the
// Net closes it with n.Close() (or m.Net.Close()).
func (m *Inferencer) Close() error {
	defer m.ctx.enter()()
	for _, p := range []unsafe.Pointer{m.in, m.pol, m.val} {
		C.agz_host_free(m.ctx.h, p)
	}
	m.in, m.pol, m.val = nil, nil, nil
	return nil
}

// GameKind maps an in-tree game.State to its device implementation; arbitrary user games are not supported.
type GameKind int

const (
	MNK  GameKind = C.AGZ_GAME_MNK
	C4   GameKind = C.AGZ_GAME_C4
	Komi GameKind = C.AGZ_GAME_KOMI
	WQ   GameKind = C.AGZ_GAME_WQ
)

// BatchedArena plays nGames self-play games concurrently on one GPU.
type BatchedArena struct {
	ctx    *Ctx
	h      *C.agz_arena
	m, n   int
	feats  int
	action int
	nGames, budget, maxMoves int
}

// NewBatchedArena mirrors MakeArena (arena.go:42-70) for nGames games: conf is the reference mcts.Config with
// Timeout replaced by exactly conf.Budget simulations per move.
func NewBatchedArena(ctx *Ctx, kind GameKind, m, n, k int, komi float32, encoder int, conf mcts.Config, nGames int, seed uint64) (*BatchedArena, error) {
	defer ctx.enter()()
	gc := C.agz_game_conf{kind: C.int32_t(kind), m: C.int32_t(m), n: C.int32_t(n), k: C.int32_t(k), komi: C.float(komi), encoder: C.int32_t(encoder)}
	dumb := 0
	if conf.DumbPass {
		dumb = 1
	}
	mc := C.agz_mcts_conf{PUCT: C.float(conf.PUCT), M: C.int32_t(conf.M), N: C.int32_t(conf.N), RandomCount: C.int32_t(conf.RandomCount),
		Budget: C.int32_t(conf.Budget), RandomMinVisits: C.uint32_t(conf.RandomMinVisits), RandomTemperature: C.float(conf.RandomTemperature),
		DumbPass: C.int32_t(dumb), ResignPercentage: C.float(conf.ResignPercentage), PassPreference: C.int32_t(conf.PassPreference)}
	a := &BatchedArena{ctx: ctx, m: m, n: n, nGames: nGames, budget: int(conf.Budget), maxMoves: 2 * m * n}
	a.feats = 2
	if encoder == C.AGZ_ENC_WQ {
		a.feats = 18
	}
	a.action = m * n
	if kind == C4 {
		a.action = n
	}
	if err := lastErr(C.agz_arena_create(ctx.h, &gc, &mc, C.int(nGames), C.uint64_t(seed), 0, &a.h)); err != nil {
		return nil, err
	}
	return a, nil
}

// SetAgents installs the two agents' networks (Agent.NN + SwitchToInference, agent.go:42-57); nil selects the
// reference's dummyInferer (agogo.go:83-87).
func (a *BatchedArena) SetAgents(na, nb *Net) error {
	defer a.ctx.enter()()
	for i, n := range []*Net{na, nb} {
		kind, h := C.int(C.AGZ_INF_DUMMY), (*C.agz_net)(nil)
		if n != nil {
			kind, h = C.int(C.AGZ_INF_NET), n.h
		}
		if err := lastErr(C.agz_arena_set_inferencer(a.h, C.int(i), kind, h)); err != nil {
			return err
		}
	}
	return nil
}

// SelfPlay plays every game to its end and returns the recorded examples (AZ.SelfPlay × nGames, agogo.go:93-97).
func (a *BatchedArena) SelfPlay() ([]agogo.Example, error) {
	defer a.ctx.enter()()
	if err := lastErr(C.agz_arena_reset(a.h, nil)); err != nil {
		return nil, err
	}
	if err := lastErr(C.agz_arena_play(a.h, 0, 1)); err != nil {
		return nil, err
	}
	var n C.int
	if err := lastErr(C.agz_arena_get_examples(a.h, nil, nil, nil, nil, 0, &n)); err != nil || n == 0 {
		return nil, err
	}
	bl, pl := a.feats*a.m*a.n, a.action+1
	planes, policy, value := make([]float32, int(n)*bl), make([]float32, int(n)*pl), make([]float32, int(n))
	if err := lastErr(C.agz_arena_get_examples(a.h, (*C.float)(unsafe.Pointer(&planes[0])), (*C.float)(unsafe.Pointer(&policy[0])),
		(*C.float)(unsafe.Pointer(&value[0])), nil, n, &n)); err != nil {
		return nil, err
	}
	ex := make([]agogo.Example, int(n))
	for i := range ex {
		ex[i] = agogo.Example{Board: planes[i*bl : (i+1)*bl], Policy: policy[i*pl : (i+1)*pl], Value: value[i]}
	}
	return ex, nil
}

func (a *BatchedArena) Close() error { defer a.ctx.enter()(); C.agz_arena_destroy(a.h); a.h = nil; return nil }

// Examples is []agogo.Example kept in HBM: the rotation Augmenter (RotateBoard, encoding_helper.go:80-107), the
// maxExamples cut (agogo.go:118-121) and prepareExamples (agogo.go:211-249) run on the device; only 4-byte row indices
// ever touch the host.  Feed the prepared tensors to a trainer with agz_train_dev (dual.Train, dualnet/meta.go:16-54).
type Examples struct {
	ctx *Ctx
	h   *C.agz_examples
}

func NewExamples(ctx *Ctx, features, height, width, policyLen int) (*Examples, error) {
	defer ctx.enter()()
	e := &Examples{ctx: ctx}
	if err := lastErr(C.agz_examples_create(ctx.h, C.int(features), C.int(height), C.int(width), C.int(policyLen), &e.h)); err != nil {
		return nil, err
	}
	return e, nil
}

// SelfPlayInto = `ex = append(ex, a.SelfPlay()...)` for all games of the batch, device to device (agogo.go:110-114).
func (a *BatchedArena) SelfPlayInto(ex *Examples) error {
	defer a.ctx.enter()()
	if err := lastErr(C.agz_arena_reset(a.h, nil)); err != nil {
		return err
	}
	if err := lastErr(C.agz_arena_play(a.h, 0, 1)); err != nil {
		return err
	}
	return lastErr(C.agz_examples_append_arena(ex.h, a.h))
}

func (e *Examples) AugmentRotate() error { defer e.ctx.enter()(); return lastErr(C.agz_examples_augment_rotate(e.h)) }

// Prepare returns the number of batches (0: "batches is nil", agogo.go:123-125).
func (e *Examples) Prepare(batchSize, maxExamples int, seed uint64) (int, error) {
	defer e.ctx.enter()()
	var b C.int
	err := lastErr(C.agz_examples_prepare(e.h, C.int(batchSize), C.int(maxExamples), C.uint64_t(seed), &b))
	return int(b), err
}

func (e *Examples) Close() error { defer e.ctx.enter()(); C.agz_examples_destroy(e.h); e.h = nil; return nil }

// ---- tournament use: Agent.Search against an outside opponent (agent.go:76-81, BASELINE configs[4]) ----
// The game state and both trees live on the device.  Search() = mcts.Search for the side to move (begin_move, Budget
// simulations, end_move): it returns the chosen move and plays it.  Opponent(move) applies the outside player's reply
// (State.Check'ed on the device); the next Search re-roots the tree over both plies (updateRoot, search.go:424-500).

// Search runs one move decision for every unfinished game of the arena and returns the moves played.
func (a *BatchedArena) Search() ([]game.Single, error) {
	defer a.ctx.enter()()
	if err := lastErr(C.agz_arena_begin_move(a.h)); err != nil {
		return nil, err
	}
	if err := lastErr(C.agz_arena_simulate(a.h, C.int(a.budget))); err != nil {
		return nil, err
	}
	if err := lastErr(C.agz_arena_end_move(a.h, 0)); err != nil {
		return nil, err
	}
	out := make([]game.Single, a.nGames)
	buf := make([]int32, a.maxMoves+4)
	for g := 0; g < a.nGames; g++ {
		var n C.int
		if err := lastErr(C.agz_arena_get_history(a.h, C.int(g), (*C.int32_t)(unsafe.Pointer(&buf[0])), C.int(len(buf)), &n)); err != nil {
			return nil, err
		}
		if n > 0 {
			out[g] = game.Single(buf[n-1])
		}
	}
	return out, nil
}

// SetParallel runs the simulations of every tree in rounds of `lanes` (1..16) whose leaves are one network batch: the
// deterministic, lane-ordered form of the reference's NumCPU goroutines sharing a tree (search.go:112-131).  1 = sequential.
func (a *BatchedArena) SetPoolPolicy(policy int) error {
	defer a.ctx.enter()()
	return lastErr(C.agz_arena_set_pool_policy(a.h, C.int(policy)))
}

func (a *BatchedArena) SetParallel(lanes int) error {
	defer a.ctx.enter()()
	return lastErr(C.agz_arena_set_parallel(a.h, C.int(lanes)))
}

// Opponent applies the outside player's moves (one per game; ignored for finished games).
func (a *BatchedArena) Opponent(moves []game.Single) error {
	defer a.ctx.enter()()
	m := make([]int32, a.nGames)
	for i := range m {
		m[i] = int32(moves[i])
	}
	return lastErr(C.agz_arena_apply_moves(a.h, (*C.int32_t)(unsafe.Pointer(&m[0]))))
}

// ---- mcts.MCTS: one device tree on the caller's own game.State (Agent.Search, agent.go:77-80) -------------------------
// MCTS has *mcts.MCTS's method set: `agent.MCTS = agzhip.NewMCTS(...)` and Agent.Search keeps reading
// `a.MCTS.SetGame(g); return a.MCTS.Search(a.Player)`.  The game stays a Go game.State; SetGame ships the position (board,
// to-move, hash, the moves since the start via LastMove/UndoLastMove on a clone, the last 8 historical boards for WQEncoder).
type MCTS struct {
	ctx        *Ctx
	h          *C.agz_mcts
	cells      int
	action     int
	current    game.State
	host       *hostInferencer // NewMCTSInferer / NewMCTSInferencer: the caller's network behind AGZ_INF_CALLBACK
	hostHandle cgo.Handle
}

// NewMCTS mirrors mcts.New(game, conf, nn) (tree.go:80-103): kind names the device implementation of g's rules, nn the
// network (nil: the reference's dummyInferer).
func NewMCTS(ctx *Ctx, kind GameKind, g game.State, k int, komi float32, encoder int, conf mcts.Config, nn *Net, seed uint64) (*MCTS, error) {
	defer ctx.enter()()
	m, n := g.BoardSize()
	gc := C.agz_game_conf{kind: C.int32_t(kind), m: C.int32_t(m), n: C.int32_t(n), k: C.int32_t(k), komi: C.float(komi), encoder: C.int32_t(encoder)}
	dumb := 0
	if conf.DumbPass {
		dumb = 1
	}
	mc := C.agz_mcts_conf{PUCT: C.float(conf.PUCT), M: C.int32_t(conf.M), N: C.int32_t(conf.N), RandomCount: C.int32_t(conf.RandomCount),
		Budget: C.int32_t(conf.Budget), RandomMinVisits: C.uint32_t(conf.RandomMinVisits), RandomTemperature: C.float(conf.RandomTemperature),
		DumbPass: C.int32_t(dumb), ResignPercentage: C.float(conf.ResignPercentage), PassPreference: C.int32_t(conf.PassPreference)}
	t := &MCTS{ctx: ctx, cells: m * n, action: g.ActionSpace(), current: g}
	if err := lastErr(C.agz_mcts_create(ctx.h, &gc, &mc, C.uint64_t(seed), C.int(poolNodes(conf, g.ActionSpace(), nn != nil)), &t.h)); err != nil {
		return nil, err
	}
	kindInf, h := C.int(C.AGZ_INF_DUMMY), (*C.agz_net)(nil)
	if nn != nil {
		kindInf, h = C.int(C.AGZ_INF_NET), nn.h
	}
	if err := lastErr(C.agz_mcts_set_inferencer(t.h, kindInf, h)); err != nil {
		C.agz_mcts_destroy(t.h) // (the handle exists from here on: no error path may leak it)
		return nil, err
	}
	// mcts.Config.Timeout is what a reference caller actually sets (Budget is inert there, search.go:183-185): a conf without a
	// Budget keeps the reference's own stopping rule — search by wall clock, not deterministic; with a Budget the search runs
	// exactly Budget simulations (the declared, bit-reproducible semantics)
	if conf.Budget <= 0 && conf.Timeout > 0 {
		if err := lastErr(C.agz_mcts_set_timeout_ms(t.h, timeoutMs(conf.Timeout))); err != nil {
			C.agz_mcts_destroy(t.h)
			return nil, err
		}
	}
	return t, nil
}

// poolNodes: the node pool of a tree whose conf has no Budget (the reference's callers set Timeout only).  Sized from the time-out —
// a search cannot expand more leaves than it has time for: a batch-1 evaluation of a residual tower takes >= 0.25 ms on the device
// (4000 simulations/s), a synthetic inferencer ~16 us (60000/s) — times ActionSpace+1 children per expansion, at least 64 expansions,
// at most the library's 8 M-node ceiling.  A conf with a Budget returns 0: the library sizes the pool from the Budget.  (Round 5 let the
// library default to 65536 expansions for every such handle: about a GB of device memory per Agent.)  A pool that fills before the clock
// runs out ENDS the search with the best move so far (agz_mcts_search).
func poolNodes(conf mcts.Config, actionSpace int, withNet bool) int {
	if conf.Budget > 0 {
		return 0
	}
	rate := 60000.0
	if withNet {
		rate = 4000.0
	}
	d := conf.Timeout
	if d <= 0 {
		d = 100 * time.Millisecond // the reference's DefaultConfig
	}
	exp := int(d.Seconds()*rate) + 64
	nodes := int64(exp) * int64(actionSpace+1)
	if nodes > 8000000 {
		nodes = 8000000
	}
	return int(nodes)
}

// ---- mcts.New(game, conf, nn Inferencer) with ANY Inferencer (mcts/mcts.go:15-18) ---------------------------------------------------
// The device search hands the leaves of a simulation to the host between its two kernels (AGZ_INF_CALLBACK, include/agz.h): the shim
// turns each leaf into what the caller's network takes.
//
//	NewMCTSInferer    — an agogo.Inferer (datatypes.go:56-59): Infer(encoded planes) — the leaf arrives ENCODED (the tree's encoder ran on
//	                    the device: what Agent.Infer computes with a.Enc(g), agent.go:60-74), so a gorgonia dualnet or any other
//	                    network plugs in unchanged;
//	NewMCTSInferencer — an mcts.Inferencer: Infer(state game.State) — the leaf arrives as a read-only game.State (LeafState below:
//	                    Board, ToMove, MoveNumber, BoardSize, ActionSpace, Hash; the mutating methods panic), which is what the
//	                    reference's own test inferencers read (mcts/example_test.go:40-72 switches on state.MoveNumber()).
type hostInferencer struct {
	byPlanes agogo.Inferer
	byState  mcts.Inferencer
	action   int
	err      error
}

// LeafState is the game.State view of one leaf of a device search (read-only).
type LeafState struct {
	m, n, action int
	board        []game.Colour
	toMove       game.Player
	moveNumber   int
}

func (l *LeafState) BoardSize() (int, int)   { return l.m, l.n }
func (l *LeafState) Board() []game.Colour    { return l.board }
func (l *LeafState) ActionSpace() int        { return l.action }
func (l *LeafState) ToMove() game.Player     { return l.toMove }
func (l *LeafState) MoveNumber() int         { return l.moveNumber }
func (l *LeafState) Passes() int             { return 0 }
func (l *LeafState) Handicap() int           { return 0 }
func (l *LeafState) AdditionalScore() float32 { return 0 }
func (l *LeafState) Hash() game.Zobrist { // FNV-1a over the cells (the device keeps its own zobrist key; a leaf's is not shipped)
	h := uint32(2166136261)
	for _, c := range l.board {
		h = (h ^ uint32(c)) * 16777619
	}
	return game.Zobrist(h)
}
func (l *LeafState) LastMove() game.PlayerMove         { panic("agzhip.LeafState: LastMove is not available on a device leaf") }
func (l *LeafState) Score(game.Player) float32         { panic("agzhip.LeafState: Score is not available on a device leaf") }
func (l *LeafState) Ended() (bool, game.Player)        { return false, game.None }
func (l *LeafState) SetToMove(game.Player)             { panic("agzhip.LeafState is read-only") }
func (l *LeafState) Check(game.PlayerMove) bool        { panic("agzhip.LeafState is read-only") }
func (l *LeafState) Apply(game.PlayerMove) game.State  { panic("agzhip.LeafState is read-only") }
func (l *LeafState) Reset()                            { panic("agzhip.LeafState is read-only") }
func (l *LeafState) Historical(int) []game.Colour      { panic("agzhip.LeafState: history is not shipped with a leaf (the encoded planes carry it)") }
func (l *LeafState) UndoLastMove()                     { panic("agzhip.LeafState is read-only") }
func (l *LeafState) Fwd()                              { panic("agzhip.LeafState is read-only") }
func (l *LeafState) Eq(game.State) bool                { return false }
func (l *LeafState) Clone() game.State                 { c := *l; c.board = append([]game.Colour(nil), l.board...); return &c }

//export agzGoInfer
func agzGoInfer(user unsafe.Pointer, b *C.agz_leaf_batch) C.int {
	inf := cgo.Handle(uintptr(user)).Value().(*hostInferencer)
	n, f, h, w, pl := int(b.n), int(b.features), int(b.height), int(b.width), int(b.policy_len)
	planes := unsafe.Slice((*float32)(unsafe.Pointer(b.planes)), n*f*h*w)
	boards := unsafe.Slice((*int32)(unsafe.Pointer(b.board)), n*h*w)
	toMove := unsafe.Slice((*int32)(unsafe.Pointer(b.to_move)), n)
	moveNo := unsafe.Slice((*int32)(unsafe.Pointer(b.move_number)), n)
	polOut := unsafe.Slice((*float32)(unsafe.Pointer(b.policy)), n*pl)
	valOut := unsafe.Slice((*float32)(unsafe.Pointer(b.value)), n)
	for i := 0; i < n; i++ {
		var policy []float32
		var value float32
		if inf.byPlanes != nil {
			var err error
			if policy, value, err = inf.byPlanes.Infer(planes[i*f*h*w : (i+1)*f*h*w]); err != nil {
				inf.err = err
				return 1 // the search aborts with AGZ_E_CALLBACK; Search reports inf.err
			}
		} else {
			ls := &LeafState{m: h, n: w, action: inf.action, toMove: game.Player(toMove[i]), moveNumber: int(moveNo[i]), board: make([]game.Colour, h*w)}
			for q := range ls.board {
				ls.board[q] = game.Colour(boards[i*h*w+q])
			}
			policy, value = inf.byState.Infer(ls)
		}
		copy(polOut[i*pl:(i+1)*pl], policy) // (a shorter policy leaves zeros; the LAST entry of the row is the pass probability, search.go:276)
		valOut[i] = value
	}
	return 0
}

func newMCTSHost(ctx *Ctx, kind GameKind, g game.State, k int, komi float32, encoder int, conf mcts.Config, inf *hostInferencer, policyLen int, seed uint64) (*MCTS, error) {
	t, err := NewMCTS(ctx, kind, g, k, komi, encoder, conf, nil, seed)
	if err != nil {
		return nil, err
	}
	defer ctx.enter()()
	inf.action = g.ActionSpace()
	t.host = inf
	t.hostHandle = cgo.NewHandle(inf)
	if err := lastErr(C.agz_mcts_set_inferencer_callback(t.h, C.agz_infer_fn(C.agzGoInfer), unsafe.Pointer(uintptr(t.hostHandle)), C.int(policyLen))); err != nil {
		t.hostHandle.Delete()
		C.agz_mcts_destroy(t.h)
		return nil, err
	}
	return t, nil
}

// NewMCTSInferer: mcts.New with an agogo.Inferer as the network (policyLen = the length of the policy it returns: ActionSpace + 1).
func NewMCTSInferer(ctx *Ctx, kind GameKind, g game.State, k int, komi float32, encoder int, conf mcts.Config, nn agogo.Inferer, policyLen int, seed uint64) (*MCTS, error) {
	return newMCTSHost(ctx, kind, g, k, komi, encoder, conf, &hostInferencer{byPlanes: nn}, policyLen, seed)
}

// NewMCTSInferencer: mcts.New(game, conf, nn) for any mcts.Inferencer (the reference's signature, tree.go:80).
func NewMCTSInferencer(ctx *Ctx, kind GameKind, g game.State, k int, komi float32, encoder int, conf mcts.Config, nn mcts.Inferencer, policyLen int, seed uint64) (*MCTS, error) {
	return newMCTSHost(ctx, kind, g, k, komi, encoder, conf, &hostInferencer{byState: nn}, policyLen, seed)
}

// timeoutMs: a positive duration below one millisecond is one millisecond (0 would silently mean "exactly Budget simulations").
func timeoutMs(d time.Duration) C.int {
	if d <= 0 {
		return 0
	}
	if d < time.Millisecond {
		return 1
	}
	return C.int(d / time.Millisecond)
}

// SetTimeout: mcts.Config.Timeout on a live tree (0 restores "exactly Budget simulations").
func (t *MCTS) SetTimeout(d time.Duration) error {
	defer t.ctx.enter()()
	return lastErr(C.agz_mcts_set_timeout_ms(t.h, timeoutMs(d)))
}

// Log (mcts/debug.go:40, release.go): the reference's release build logs nothing; the device tree keeps no log either.  Present so that
// *MCTS has the whole method set Agent / Arena use on their tree (SetGame, Search, Policies, Reset, Log — INTEGRATION.md section 2).
func (t *MCTS) Log() string { return "" }

// SetGame (tree.go:120-124).
func (t *MCTS) SetGame(g game.State) { t.current = g }

// SetPoolPolicy: what a full node pool means — PoolStrict (default: Search reports it) or PoolStopSearch (the reference's MAXTREESIZE rule:
// the search of that move stops, the game goes on; search.go:23,78,229).
const (
	PoolStrict     = 0
	PoolStopSearch = 1
	PoolGrow       = 2 // pools re-allocated before a search could outgrow them: results unchanged, no max_nodes to choose
)

func (t *MCTS) SetPoolPolicy(policy int) error {
	defer t.ctx.enter()()
	return lastErr(C.agz_mcts_set_pool_policy(t.h, C.int(policy)))
}

// SetParallel: lanes per round (see BatchedArena.SetParallel).
func (t *MCTS) SetParallel(lanes int) error {
	defer t.ctx.enter()()
	return lastErr(C.agz_mcts_set_parallel(t.h, C.int(lanes)))
}

// ship serialises t.current into an agz_state.
func (t *MCTS) ship() error {
	g := t.current
	board := make([]int32, t.cells)
	for i, c := range g.Board() {
		board[i] = int32(c)
	}
	// the moves played so far, oldest first: LastMove / UndoLastMove on a clone (what newRootState itself does, search.go:429-440)
	nMoves := g.MoveNumber()
	tmp := g.Clone().(game.State)
	var rev []int32
	for i := 0; i < nMoves; i++ {
		lm := tmp.LastMove()
		if lm.Single.IsPass() && tmp.MoveNumber() == 0 {
			break
		}
		rev = append(rev, int32(lm.Single))
		tmp.UndoLastMove()
	}
	last := make([]int32, len(rev))
	for i := range rev {
		last[len(rev)-1-i] = rev[i]
	}
	// State.Historical(i): the boards after the last (up to 8) moves, oldest first
	nh := nMoves
	if nh > 8 {
		nh = 8
	}
	hist := make([]int32, nh*t.cells)
	for q := 0; q < nh; q++ {
		for i, c := range g.Historical(nMoves - nh + q) {
			hist[q*t.cells+i] = int32(c)
		}
	}
	passes := g.Passes()
	if passes < 0 {
		passes = 0
	}
	st := C.agz_state{board: (*C.int32_t)(unsafe.Pointer(&board[0])), to_move: C.int32_t(g.ToMove()), n_moves: C.int32_t(nMoves),
		passes: C.int32_t(passes), hash: C.uint32_t(g.Hash()),
		captures_black: C.float(g.Score(game.Player(game.Black))), captures_white: C.float(g.Score(game.Player(game.White)))}
	if len(last) > 0 {
		st.last_moves, st.n_last_moves = (*C.int32_t)(unsafe.Pointer(&last[0])), C.int32_t(len(last))
	}
	if nh > 0 {
		st.historical, st.n_historical = (*C.int32_t)(unsafe.Pointer(&hist[0])), C.int32_t(nh)
	}
	return lastErr(C.agz_mcts_set_game(t.h, &st))
}

// Search (search.go:92-164).  Like the reference it panics when the search cannot run (Agent.Infer panics on an inferer error,
// agent.go:66-71).
func (t *MCTS) Search(player game.Player) game.Single {
	defer t.ctx.enter()()
	if err := t.ship(); err != nil {
		panic(err)
	}
	var best C.int32_t
	if err := lastErr(C.agz_mcts_search(t.h, C.int(player), &best)); err != nil {
		panic(err)
	}
	return game.Single(best)
}

// Policies (tree.go:128-142) of the game set last.  g is accepted for signature compatibility (Arena.Play passes the game it
// just searched, arena.go:108).
func (t *MCTS) Policies(g game.State) []float32 {
	defer t.ctx.enter()()
	out := make([]float32, t.action+1)
	if err := lastErr(C.agz_mcts_policies(t.h, (*C.float)(unsafe.Pointer(&out[0])), C.int(len(out)))); err != nil {
		panic(err)
	}
	return out
}

// Nodes (tree.go:126).
func (t *MCTS) Nodes() int {
	defer t.ctx.enter()()
	var n C.int
	C.agz_mcts_nodes(t.h, &n)
	return int(n)
}

// ToDot renders the live tree as Graphviz text (mcts/graph.go:34-90).
func (t *MCTS) ToDot() string {
	defer t.ctx.enter()()
	var need C.size_t
	if C.agz_mcts_to_dot(t.h, 0, nil, 0, &need) != 0 || need == 0 {
		return ""
	}
	buf := make([]byte, int(need))
	if C.agz_mcts_to_dot(t.h, 0, (*C.char)(unsafe.Pointer(&buf[0])), need, &need) != 0 {
		return ""
	}
	return string(buf[:len(buf)-1])
}

// Child is one entry of Children: what (*MCTS).Log / ToDot print per node (node.go:56-68).
type Child struct {
	ID          int // feeds further Children calls
	Move        game.Single
	Visits      uint32
	BlackScores float32
	Prior       float32
}

// Children (mcts/unsafe_safe.go:15): the children of any node of the live tree (0 = the root).
func (t *MCTS) Children(of int) []Child {
	defer t.ctx.enter()()
	cap := t.cells + 2
	ids, moves := make([]int32, cap), make([]int32, cap)
	visits := make([]uint32, cap)
	bs, pr := make([]float32, cap), make([]float32, cap)
	var n C.int
	if err := lastErr(C.agz_mcts_children(t.h, C.int(of), (*C.int32_t)(unsafe.Pointer(&ids[0])), (*C.int32_t)(unsafe.Pointer(&moves[0])),
		(*C.uint32_t)(unsafe.Pointer(&visits[0])), (*C.float)(unsafe.Pointer(&bs[0])), (*C.float)(unsafe.Pointer(&pr[0])), C.int(cap), &n)); err != nil {
		panic(err)
	}
	out := make([]Child, int(n))
	for i := range out {
		out[i] = Child{ID: int(ids[i]), Move: game.Single(moves[i]), Visits: visits[i], BlackScores: bs[i], Prior: pr[i]}
	}
	return out
}

// Reset (tree.go:249-276), completed to "a fresh tree" (what Arena.Play does next, arena.go:140-141,175-176).
func (t *MCTS) Reset() { defer t.ctx.enter()(); C.agz_mcts_reset(t.h) }

func (t *MCTS) Close() error {
	defer t.ctx.enter()()
	C.agz_mcts_destroy(t.h)
	t.h = nil
	if t.host != nil {
		t.hostHandle.Delete()
		t.host = nil
	}
	return nil
}

// ---- dual.Train (dualnet/meta.go:16-54) ----------------------------------------------------------------------------------
// Trainer holds the training graph of a Dual (full batch-shaped learnables in Model() order, train.hip).
type Trainer struct {
	ctx *Ctx
	h   *C.agz_trainer
	d   *dual.Dual
}

// NewTrainer copies d's learnables (Model() order, full shapes) to the device.
func NewTrainer(ctx *Ctx, d *dual.Dual) (*Trainer, error) {
	defer ctx.enter()()
	cc := C.agz_net_conf{
		K: C.int32_t(d.K), SharedLayers: C.int32_t(d.SharedLayers), FC: C.int32_t(d.FC), BatchSize: C.int32_t(d.BatchSize),
		Width: C.int32_t(d.Width), Height: C.int32_t(d.Height), Features: C.int32_t(d.Features),
		ActionSpace: C.int32_t(d.ActionSpace), bn_mode: C.AGZ_BN_DEGENERATE_EPS, bn_eps: 1e-5,
	}
	t := &Trainer{ctx: ctx, d: d}
	if err := lastErr(C.agz_trainer_create(ctx.h, &cc, &t.h)); err != nil {
		return nil, err
	}
	for i, node := range d.Model() {
		data := node.Value().Data().([]float32)
		if err := lastErr(C.agz_trainer_set_param(t.h, C.int(i), (*C.float)(unsafe.Pointer(&data[0])), C.size_t(len(data)))); err != nil {
			return nil, err
		}
	}
	return t, nil
}

// Train is dual.Train(d, Xs, policies, values, batches, iterations) on the device; afterwards the learnables are copied back
// into d.Model() so the rest of AZ.Learn (SwitchToInference, Save) sees the trained network.  Xs [rows, F, H, W], policies
// [rows, ActionSpace], values [rows] as flat float32 slices (tensor.Dense.Data()).
func (t *Trainer) Train(Xs, policies, values []float32, batches, iterations int, seed uint64) error {
	rows := batches * t.d.BatchSize
	if batches < 1 || len(Xs) < rows*t.d.Features*t.d.Height*t.d.Width || len(policies) < rows*t.d.ActionSpace || len(values) < rows {
		return fmt.Errorf("agzhip: Train: %d batches of %d rows need %d / %d / %d values, got %d / %d / %d", batches, t.d.BatchSize,
			rows*t.d.Features*t.d.Height*t.d.Width, rows*t.d.ActionSpace, rows, len(Xs), len(policies), len(values))
	}
	defer t.ctx.enter()()
	var cost C.float
	if err := lastErr(C.agz_train(t.h, (*C.float)(unsafe.Pointer(&Xs[0])), (*C.float)(unsafe.Pointer(&policies[0])),
		(*C.float)(unsafe.Pointer(&values[0])), C.int(batches), C.int(iterations), C.uint64_t(seed), &cost)); err != nil {
		return err
	}
	for i, node := range t.d.Model() {
		data := node.Value().Data().([]float32)
		if err := lastErr(C.agz_trainer_get_param(t.h, C.int(i), (*C.float)(unsafe.Pointer(&data[0])), C.size_t(len(data)))); err != nil {
			return err
		}
	}
	return nil
}

// TrainDev is dual.Train over the tensors an Examples set prepared on the device (no host copy of the examples).
func (t *Trainer) TrainDev(ex *Examples, iterations int, seed uint64) error {
	defer t.ctx.enter()()
	var xs, pi, v *C.float
	var rows C.int64_t
	var batches C.int
	if err := lastErr(C.agz_examples_tensors_dev(ex.h, &xs, &pi, &v, &rows, &batches)); err != nil {
		return err
	}
	var cost C.float
	return lastErr(C.agz_train_dev(t.h, xs, pi, v, batches, C.int(iterations), C.uint64_t(seed), &cost))
}

// Export is dual.Infer's copy loop (meta.go:141-146): row 0 of every learnable into an inference Net.
// SetComputeMode selects the arithmetic of training's GEMMs: ComputeF32MFMA (default), ComputeBF16X3 (forward, data- and
// weight-gradient GEMMs on the bf16 pipe) or ComputeWinoH2 (Winograd forward / data-gradient convolutions; 1.9x the default's step rate).
func (t *Trainer) SetComputeMode(mode int) error {
	defer t.ctx.enter()()
	return lastErr(C.agz_trainer_set_compute_mode(t.h, C.int(mode)))
}

func (t *Trainer) Export(n *Net) error { defer t.ctx.enter()(); return lastErr(C.agz_trainer_export(t.h, n.h)) }

func (t *Trainer) Close() error { defer t.ctx.enter()(); C.agz_trainer_destroy(t.h); t.h = nil; return nil }

// ---- n GPUs: games shard with no data-path collective; RCCL only around dual.Train (agogo.go:118-133) --------------------
// Comm is one rank of an RCCL communicator over the devices of n Ctx handles of THIS process.  Collective methods must be
// called concurrently for all n ranks, one goroutine (locked to its OS thread by enter()) per Ctx.
type Comm struct {
	ctx *Ctx
	h   *C.agz_comm
}

// NewComms = ncclCommInitAll over the ctxs' devices.
func NewComms(ctxs []*Ctx) ([]*Comm, error) {
	n := len(ctxs)
	hs := make([]*C.agz_ctx, n)
	for i, c := range ctxs {
		hs[i] = c.h
	}
	out := make([]*C.agz_comm, n)
	if err := lastErr(C.agz_comm_init_all((**C.agz_ctx)(unsafe.Pointer(&hs[0])), C.int(n), (**C.agz_comm)(unsafe.Pointer(&out[0])))); err != nil {
		return nil, err
	}
	comms := make([]*Comm, n)
	for i := range comms {
		comms[i] = &Comm{ctx: ctxs[i], h: out[i]}
	}
	return comms, nil
}

// AllGatherExamples: every rank's set becomes the union of all ranks' self-play examples, in rank order — the `ex` that
// AZ.Learn shuffles, truncates and tensorises (agogo.go:118-122).
func (c *Comm) AllGatherExamples(ex *Examples) error {
	defer c.ctx.enter()()
	return lastErr(C.agz_examples_allgather(c.h, ex.h))
}

// AllReduceGradients + the averaged SGD step of one data-parallel training batch.
func (c *Comm) AllReduceGradients(t *Trainer, lr float32) error {
	defer c.ctx.enter()()
	if err := lastErr(C.agz_trainer_allreduce(c.h, t.h)); err != nil {
		return err
	}
	return lastErr(C.agz_trainer_apply(t.h, C.float(lr), C.float(1.0/float32(C.agz_comm_size(c.h)))))
}

// BatchStep: one data-parallel dual.Train inner step (meta.go:33-40) on this rank's batch — forward / backward with every slice of the
// flat gradient buffer summed over the ranks while the rest of the backward runs (agz_trainer_forward_backward_allreduce), then the
// averaged SGD step.  Every rank calls it for the same step.
func (c *Comm) BatchStep(t *Trainer, planes, pi, v []float32, lr float32) (cost float32, err error) {
	// (lengths first: the C side reads exactly one batch from each slice, and an error returned here comes BEFORE this rank enters the
	// step's collectives — its peers must be stopped by the caller, as after any programming error on one rank)
	b := t.d.BatchSize
	if want := b * t.d.Features * t.d.Height * t.d.Width; len(planes) < want {
		return 0, fmt.Errorf("agzhip: BatchStep: planes has %d values, one batch is %d", len(planes), want)
	}
	if want := b * t.d.ActionSpace; len(pi) < want {
		return 0, fmt.Errorf("agzhip: BatchStep: pi has %d values, one batch is %d", len(pi), want)
	}
	if len(v) < b {
		return 0, fmt.Errorf("agzhip: BatchStep: v has %d values, one batch is %d", len(v), b)
	}
	defer c.ctx.enter()()
	var cc C.float
	if err := lastErr(C.agz_trainer_forward_backward_allreduce(c.h, t.h, (*C.float)(unsafe.Pointer(&planes[0])), (*C.float)(unsafe.Pointer(&pi[0])),
		(*C.float)(unsafe.Pointer(&v[0])), &cc)); err != nil {
		return 0, err
	}
	return float32(cc), lastErr(C.agz_trainer_apply(t.h, C.float(lr), C.float(1.0/float32(C.agz_comm_size(c.h)))))
}

func (c *Comm) Close() error { defer c.ctx.enter()(); C.agz_comm_destroy(c.h); c.h = nil; return nil }

var _ = game.Pass // keep the import: game.Single values cross the ABI as int32 (-1 pass, -2 resign)

// Constants mirrored for callers (include/agz.h).
const (
	BNDegenerateEps = int(C.AGZ_BN_DEGENERATE_EPS)
	BNRunning       = int(C.AGZ_BN_RUNNING)
	BNIdentity      = int(C.AGZ_BN_IDENTITY)
	EncTwoPlane     = int(C.AGZ_ENC_TWOPLANE)
	EncWQ           = int(C.AGZ_ENC_WQ)
)
