// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of gorgonia/agogo's game package.
// Nothing under agogo_amd/ may include, link or call this; only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg use it, and only as the checker.
//
// Each function cites the reference file:line it follows (paths relative to the agogo tree).
// The Go reference cannot be built here (no Go toolchain); this restatement is pinned against the
// reference's own known-answer tables (tests/golden/*.json, transcribed from game/*/…_test.go).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace oracle {

// game/state.go:7-13
typedef int32_t Colour;
enum : Colour { None = 0, Black = 1, White = 2 };
typedef Colour Player;
// game/state.go:110-122
typedef int32_t Single;
static const Single Pass = -1;
static const Single Resign = -2;
struct PlayerMove {
  Player player;
  Single move;
};

inline Player Opponent(Player p) {  // game/komi/game.go:412-420, mnk.go:297-305 (panic on None)
  if (p == White) return Black;
  if (p == Black) return White;
  throw std::runtime_error("Unreachable: opponent of None");
}

// FNV-1a 32 over the %v rendering of each colour: game/mnk/mnk.go:76-82, game/c4/game.go:203-210,
// game/state.go:15-27 ("None"/"Black"/"White").
inline uint32_t fnv1a_board(const std::vector<Colour>& b) {
  static const char* names[3] = {"None", "Black", "White"};
  uint32_t h = 2166136261u;
  for (Colour c : b) {
    const char* s = (c >= 0 && c <= 2) ? names[c] : "";
    for (; *s; ++s) {
      h ^= (uint8_t)*s;
      h *= 16777619u;
    }
  }
  return h;
}

// The build's RNG (Go's math/rand stream is not reproducible without Go: SURVEY App. A q3).
struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed = 0) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  uint32_t next_u32() { return (uint32_t)(next() >> 32); }
  int32_t int31() { return (int32_t)(next() >> 33); }               // rand.Int31 analogue
  float float32() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }  // [0,1)
  double float64() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

// game.State: game/state.go:125-156
struct State : std::enable_shared_from_this<State> {
  virtual ~State() {}
  virtual void BoardSize(int& m, int& n) const = 0;
  virtual const std::vector<Colour>& Board() const = 0;
  virtual int ActionSpace() const = 0;
  virtual uint32_t Hash() const = 0;
  virtual Player ToMove() const = 0;
  virtual int Passes() const = 0;
  virtual int MoveNumber() const = 0;
  virtual PlayerMove LastMove() const = 0;
  virtual int Handicap() const { return 0; }
  virtual float Score(Player p) const = 0;
  virtual float AdditionalScore() const = 0;
  virtual bool Ended(Player* winner) const = 0;
  virtual void SetToMove(Player p) = 0;
  virtual bool Check(PlayerMove m) const = 0;
  virtual std::shared_ptr<State> Apply(PlayerMove m) = 0;
  virtual void Reset() = 0;
  virtual const std::vector<Colour>& Historical(int i) const = 0;
  virtual void UndoLastMove() = 0;
  virtual void Fwd() = 0;
  virtual bool Eq(const State* other) const = 0;
  virtual std::shared_ptr<State> Clone() const = 0;
  virtual int Kind() const = 0;  // AGZ_GAME_* (not in the reference; lets the C API describe a state)
};
typedef std::shared_ptr<State> StatePtr;

// ------------------------------------------------------------------------------------------------
// game/mnk/mnk.go
struct MNK : State {
  std::vector<Colour> board;
  int m, n, k;
  Player nextToMove = None;
  std::vector<PlayerMove> history;
  std::vector<std::vector<Colour>> historical;
  int histPtr = 0;

  MNK(int m_, int n_, int k_) : board((size_t)m_ * n_, None), m(m_), n(n_), k(k_) {}  // mnk.go:36-45
  int Kind() const override { return 0; }
  void BoardSize(int& a, int& b) const override { a = m; b = n; }
  const std::vector<Colour>& Board() const override { return board; }
  const std::vector<Colour>& Historical(int i) const override { return historical.at(i); }  // :74
  uint32_t Hash() const override { return fnv1a_board(board); }                              // :76-82
  int ActionSpace() const override { return m * n; }                                         // :84
  void SetToMove(Player p) override { nextToMove = p; }
  Player ToMove() const override { return nextToMove; }
  PlayerMove LastMove() const override {  // :90-95
    if (!history.empty()) return history.at(histPtr - 1);
    return PlayerMove{None, Pass};
  }
  int Passes() const override { return -1; }                     // :98
  int MoveNumber() const override { return (int)history.size(); }  // :100
  bool Check(PlayerMove mv) const override {                     // :102-120
    if (mv.move == Resign) return true;
    if (mv.move == Pass) return false;
    if (mv.move >= (int)board.size()) return false;
    if (board[mv.move] != None) return false;
    return true;
  }
  StatePtr Apply(PlayerMove mv) override {  // :122-142 (in place, returns itself)
    if (!Check(mv)) return shared_from_this();
    std::vector<Colour> hb = board;  // copy BEFORE the move
    board.at(mv.move) = mv.player;
    histPtr++;
    if ((int)history.size() < histPtr)
      history.push_back(mv);
    else
      history[histPtr - 1] = mv;
    historical.push_back(hb);
    nextToMove = Opponent(mv.player);
    return shared_from_this();
  }
  float Score(Player p) const override {  // :147-155
    if (isWinner(p)) return 1;
    if (isWinner(Opponent(p))) return -2;
    return 0;
  }
  float AdditionalScore() const override { return 0; }
  bool Ended(Player* winner) const override {  // :161-174
    if (isWinner(Black)) { *winner = Black; return true; }
    if (isWinner(White)) { *winner = White; return true; }
    for (Colour c : board)
      if (c == None) { *winner = None; return false; }
    *winner = None;
    return true;
  }
  void Reset() override {  // :176-182 (historical and nextToMove are NOT cleared)
    for (auto& c : board) c = None;
    history.clear();
    histPtr = 0;
  }
  void UndoLastMove() override {  // :184-189
    if (!history.empty()) {
      board.at(history.at(histPtr - 1).move) = None;
      histPtr--;
    }
  }
  void Fwd() override {  // :191-195
    if (!history.empty()) histPtr++;
  }
  bool Eq(const State* other) const override {  // :197-211 (board only)
    const MNK* ot = dynamic_cast<const MNK*>(other);
    if (!ot) return false;
    if (board.size() != ot->board.size()) return false;
    for (size_t i = 0; i < board.size(); i++)
      if (board[i] != ot->board[i]) return false;
    return true;
  }
  StatePtr Clone() const override {  // :213-224 (historical is copied into a zero-length slice: no-op)
    auto r = std::make_shared<MNK>(m, n, k);
    r->board = board;
    r->history = history;
    r->nextToMove = nextToMove;
    r->histPtr = histPtr;
    return r;
  }
  bool isWinner(Player p) const {  // :226-295, quirks kept (row count-- ; column trailing run ; no wrap check)
    Colour colour = p;
    for (int i = 0; i < m; i++) {
      int rowCount = 0;
      for (int j = 0; j < n; j++) {
        if (board[i * n + j] == colour) rowCount++; else rowCount--;
      }
      if (rowCount >= k) return true;
    }
    for (int j = 0; j < n; j++) {
      int count = 0;
      for (int i = 0; i * n + j < (int)board.size(); i++) {
        if (board[i * n + j] == colour) count++; else count = 0;
      }
      if (count >= k) return true;
    }
    for (int i = 0; i < m; i++) {
      for (int j = 0; n - j > n - k && j < n; j++) {
        int idx = i * n + j;
        int diagCount = 0;
        while (board[idx] == colour) {
          diagCount++;
          if (diagCount >= k) return true;
          idx = idx + n + 1;
          if (idx >= m * n) break;
        }
      }
    }
    for (int i = 0; i < m; i++) {
      for (int j = n - 1; j >= k - 1; j--) {
        int idx = i * n + j;
        int diagCount = 0;
        while (board[idx] == colour) {
          diagCount++;
          if (diagCount >= k) return true;
          idx = idx + n - 1;
          if (idx >= m * n) break;
        }
      }
    }
    return false;
  }
};

// ------------------------------------------------------------------------------------------------
// game/c4/c4.go + game/c4/game.go.  Reference quirks kept: Apply never flips
// nextToMove and never increments moveCount; pass is always legal; Clone pads history by 2.
struct C4 : State {
  int rows, cols, nwin;
  std::vector<Colour> data;  // row-major rows x cols, row 0 on top
  std::vector<PlayerMove> history;
  std::vector<std::vector<Colour>> historical;
  Player nextToMove = None;
  int histPtr = 0, moveCount = 0, passCount = 0;

  C4(int r, int c, int n) : rows(r), cols(c), nwin(n), data((size_t)r * c, None) {}  // game.go:25-35
  int Kind() const override { return 1; }
  Colour at(int r, int c) const { return data[r * cols + c]; }
  void BoardSize(int& a, int& b) const override { a = rows; b = cols; }
  void SetToMove(Player p) override { nextToMove = p; }
  Player ToMove() const override { return nextToMove; }
  PlayerMove LastMove() const override {  // game.go:43-48
    if (!history.empty()) return history.at(histPtr - 1);
    return PlayerMove{None, -1};
  }
  int Passes() const override { return 0; }               // game.go:50
  int MoveNumber() const override { return moveCount + 1; }  // game.go:52
  // c4.go:59-70
  bool boardCheck(PlayerMove mv, int* row, int* col) const {
    if (mv.move == Pass) { *row = -1; *col = -1; return true; }
    *col = mv.move;
    if (*col < 0 || *col >= cols) return false;  // Go would panic on the index; treated as illegal
    for (*row = rows - 1; *row >= 0; (*row)--)
      if (at(*row, *col) == None) return true;
    return false;
  }
  bool Check(PlayerMove mv) const override { int r, c; return boardCheck(mv, &r, &c); }  // game.go:54
  StatePtr Apply(PlayerMove mv) override {  // game.go:56-73
    std::vector<Colour> hb = data;
    int r, c;
    bool ok = true;
    if (mv.move != Pass) {  // c4.go:47-57
      ok = boardCheck(mv, &r, &c);
      if (ok) data[r * cols + c] = mv.player;
    }
    if (ok) {
      history.push_back(mv);
      historical.push_back(hb);
      histPtr++;
    }
    if (mv.move == Pass) passCount++; else passCount = 0;
    return shared_from_this();
  }
  Colour checkWin() const {  // c4.go:82-192
    // vertical
    for (int x = 0; x < cols; x++) for (int y = 0; y < rows; y++) {
      Colour c = at(y, x); if (c == None) continue; bool w = true;
      for (int i = 0; i < nwin; i++) { if (y + i < rows) { if (at(y + i, x) != c) w = false; } else w = false; }
      if (w) return c;
    }
    // horizontal
    for (int x = 0; x < cols; x++) for (int y = 0; y < rows; y++) {
      Colour c = at(y, x); if (c == None) continue; bool w = true;
      for (int i = 0; i < nwin; i++) { if (x + i < cols) { if (at(y, x + i) != c) w = false; } else w = false; }
      if (w) return c;
    }
    // TLBR (down-left)
    for (int x = 0; x < cols; x++) for (int y = 0; y < rows; y++) {
      Colour c = at(y, x); if (c == None) continue; bool w = true;
      for (int i = 0; i < nwin; i++) { if (x - i >= 0 && y + i < rows) { if (at(y + i, x - i) != c) w = false; } else w = false; }
      if (w) return c;
    }
    // TRBL (down-right)
    for (int x = 0; x < cols; x++) for (int y = 0; y < rows; y++) {
      Colour c = at(y, x); if (c == None) continue; bool w = true;
      for (int i = 0; i < nwin; i++) { if (x + i < cols && y + i < rows) { if (at(y + i, x + i) != c) w = false; } else w = false; }
      if (w) return c;
    }
    return None;
  }
  float Score(Player p) const override {  // game.go:75-84
    Colour w = checkWin();
    if (w == p) return 1;
    if (w == None) return 0;
    return -1;
  }
  void UndoLastMove() override {  // game.go:86-98 (buggy in the reference; kept for completeness)
    histPtr--;
    PlayerMove last = history.at(histPtr - 1);
    int col = last.move, row;
    for (row = rows - 1; row >= 0; row--)
      if (at(row, col) == None) { row--; break; }
    data.at(row * cols + col) = None;
  }
  void Fwd() override { if (!history.empty()) histPtr++; }  // game.go:100-104
  bool Eq(const State* other) const override {              // game.go:106-140 (board compared with itself)
    const C4* ot = dynamic_cast<const C4*>(other);
    if (!ot) return false;
    if (histPtr != ot->histPtr) return false;
    if (moveCount != ot->moveCount) return false;
    if (history.size() != ot->history.size()) return false;
    if (historical.size() != ot->historical.size()) return false;
    for (size_t i = 0; i < history.size(); i++)
      if (ot->history[i].player != history[i].player || ot->history[i].move != history[i].move) return false;
    for (size_t i = 0; i < historical.size(); i++)
      for (size_t j = 0; j < historical[i].size(); j++)
        if (ot->historical[i].at(j) != historical[i][j]) return false;
    return true;
  }
  StatePtr Clone() const override {  // game.go:142-160 (history/historical padded by 2)
    auto r = std::make_shared<C4>(rows, cols, nwin);
    r->data = data;
    r->history = history;
    r->history.resize(history.size() + 2, PlayerMove{None, 0});
    r->historical = historical;
    r->historical.resize(historical.size() + 2);
    r->nextToMove = nextToMove;
    r->histPtr = histPtr;
    r->moveCount = moveCount;
    r->passCount = passCount;
    return r;
  }
  float AdditionalScore() const override { return 0; }
  bool Ended(Player* winner) const override {  // game.go:164-183
    Colour w = checkWin();
    if (w != None) { *winner = w; return true; }
    *winner = None;
    if (passCount > 2) return true;
    for (Colour c : data) if (c == None) return false;
    return true;
  }
  void Reset() override {  // game.go:187-197
    for (auto& c : data) c = None;
    historical.clear(); history.clear();
    histPtr = 0; moveCount = 0; passCount = 0; nextToMove = None;
  }
  int ActionSpace() const override { return cols; }                 // game.go:199
  const std::vector<Colour>& Board() const override { return data; }  // game.go:201
  uint32_t Hash() const override { return fnv1a_board(data); }      // game.go:203-210
  const std::vector<Colour>& Historical(int i) const override { return historical.at(i); }
};

// ------------------------------------------------------------------------------------------------
// The capture engine shared by game/komi/game.go:316-402 and game/wq/wq.go:205-290.
// Coordinates follow the reference: c=(X,Y); cell(X,Y) = board[X*stride + Y]; valid iff X<xmax, Y<ymax.
// komi uses stride = m, xmax = m, ymax = n and Itol(c) = (c/m, c%m) (game/naughty.go:9-19 strides rows by m;
// komi/game.go:267-275), which coincides with the usual row-major geometry only for square boards.
struct Coord { int X, Y; };
struct CaptureEngine {
  const std::vector<Colour>* board;
  int stride, xmax, ymax;
  bool valid(Coord c) const { return c.X < xmax && c.X >= 0 && c.Y < ymax && c.Y >= 0; }  // isCoordValid
  Colour at(Coord c) const { return (*board)[c.X * stride + c.Y]; }
  static Coord adj(Coord c, int i) {  // adjacents {0,1},{1,0},{0,-1},{-1,0}: komi/game.go:404-409
    static const int dx[4] = {0, 1, 0, -1}, dy[4] = {1, 0, -1, 0};
    return Coord{c.X + dx[i], c.Y + dy[i]};
  }
  // nolib: komi/game.go:348-402, wq.go:237-290. Returns the liberty-less group containing c (treating
  // `potential` as filled), or empty if it has a liberty.
  std::vector<Coord> nolib(Coord c, Coord potential) const {
    std::vector<Coord> retVal;
    bool found = true;
    std::vector<Coord> founds{c};
    while (found) {
      found = false;
      std::vector<Coord> group;
      for (Coord f : founds) {
        for (int i = 0; i < 4; i++) {
          Coord a = adj(f, i);
          if (!valid(a)) continue;
          if (at(a) == None && !(a.X == potential.X && a.Y == potential.Y)) return {};
          if (at(f) != at(a)) continue;
          bool potentialGroup = true;
          for (Coord g : group) if (g.X == a.X && g.Y == a.Y) { potentialGroup = false; break; }
          if (potentialGroup)
            for (Coord l : retVal) if (l.X == a.X && l.Y == a.Y) { potentialGroup = false; break; }
          if (potentialGroup) { group.push_back(a); found = true; }
        }
      }
      retVal.insert(retVal.end(), founds.begin(), founds.end());
      founds = group;
    }
    return retVal;
  }
  // check: komi/game.go:316-345, wq.go:205-234.  Returns false on suicide. captures may hold duplicates
  // when one dead group is touched on two sides — kept.
  bool check(Coord c, Player player, std::vector<Coord>* captures) const {
    captures->clear();
    for (int i = 0; i < 4; i++) {
      Coord a = adj(c, i);
      if (!valid(a)) continue;
      if (at(a) == Opponent(player)) {
        std::vector<Coord> nl = nolib(a, c);
        captures->insert(captures->end(), nl.begin(), nl.end());
      }
    }
    if (!captures->empty()) return true;
    std::vector<Coord> suicides = nolib(c, Coord{-5, -5});
    return suicides.empty();
  }
};

// Zobrist tables (komi/zobrist.go:24-68, wq/zobrist.go:24-55).  The reference seeds from wall-clock time;
// the build seeds SplitMix64 with a fixed seed.  table[2*i+0] = Black key of cell i, [2*i+1] = White key.
struct Zobrist {
  std::vector<int32_t> table;
  int32_t hash = 0;
  void update(PlayerMove m) {
    if (m.player == Black) hash ^= table.at(2 * m.move);
    else if (m.player == White) hash ^= table.at(2 * m.move + 1);
  }
};
static const uint64_t kZobristSeed = 1337;

// ------------------------------------------------------------------------------------------------
// game/komi/game.go — capture-k Go variant
struct Komi : State {
  std::vector<Colour> board;
  int m, n;
  float k;
  Player nextToMove = Black;
  std::vector<PlayerMove> history;
  std::vector<std::vector<Colour>> historical;
  int histPtr = 0;
  float ws = 0, bs = 0;
  Zobrist z;
  int taken = 0;
  bool err = false;

  static Zobrist makeZobrist(int m, int n) {  // komi/zobrist.go:32-43: only the first size+1 entries are drawn
    Zobrist z;
    int size = m * n;
    z.table.assign((size_t)2 * size, 0);
    SplitMix64 r(kZobristSeed);
    for (int i = 0; i < size + 1 && i < 2 * size; i++) z.table[i] = r.int31();
    return z;
  }
  Komi(int m_, int n_, int k_) : board((size_t)m_ * n_, None), m(m_), n(n_), k((float)k_), z(makeZobrist(m_, n_)) {}
  int Kind() const override { return 2; }
  CaptureEngine eng() const { return CaptureEngine{&board, m, m, n}; }
  Coord Itol(Single c) const { return Coord{c / m, c % m}; }  // game.go:267-271 (divides by m)
  Single Ltoi(Coord c) const { return c.X * m + c.Y; }        // game.go:274
  void BoardSize(int& a, int& b) const override { a = m; b = n; }
  const std::vector<Colour>& Board() const override { return board; }
  const std::vector<Colour>& Historical(int i) const override { return historical.at(i); }
  uint32_t Hash() const override { return (uint32_t)z.hash; }
  int ActionSpace() const override { return m * n; }
  void SetToMove(Player p) override { nextToMove = p; }
  Player ToMove() const override { return nextToMove; }
  PlayerMove LastMove() const override {
    if (!history.empty()) return history.at(histPtr - 1);
    return PlayerMove{None, Pass};
  }
  int Passes() const override { return -1; }
  int MoveNumber() const override { return (int)history.size(); }
  bool checkCaptures(PlayerMove mv, std::vector<Single>* caps) const {  // game.go:316-345
    if (mv.move == Pass) return false;
    std::vector<Coord> cc;
    bool ok = eng().check(Itol(mv.move), mv.player, &cc);
    caps->clear();
    if (ok) for (Coord c : cc) caps->push_back(Ltoi(c));
    return ok;
  }
  bool Check(PlayerMove mv) const override {  // game.go:78-102
    if (mv.move == Resign) return true;
    if (mv.move == Pass) return false;
    if (mv.move >= (int)board.size() || mv.move < 0) return false;
    if (board[mv.move] != None) return false;
    std::vector<Single> caps;
    return checkCaptures(mv, &caps);
  }
  // game.go:277-313. Returns false on error.
  bool applyMove(PlayerMove mv, int* ntaken) {
    *ntaken = 0;
    if (!(mv.player == Black || mv.player == White)) return false;
    if (mv.move == Pass) return false;
    if (mv.move >= m * m || mv.move < 0 || mv.move >= (int)board.size()) return false;  // `>= g.m*g.m` in the reference
    if (board[mv.move] != None) return false;
    std::vector<Single> caps;
    if (!checkCaptures(mv, &caps)) return false;
    board[mv.move] = mv.player;
    z.update(mv);
    for (Single p : caps) {
      board[p] = None;
      z.update(PlayerMove{Opponent(mv.player), p});
    }
    *ntaken = (int)caps.size();
    return true;
  }
  StatePtr Apply(PlayerMove mv) override {  // game.go:104-130
    bool ok = applyMove(mv, &taken);
    err = !ok;
    if (!ok) return shared_from_this();
    histPtr++;
    if ((int)history.size() < histPtr) history.push_back(mv); else history[histPtr - 1] = mv;
    historical.push_back(board);  // copied AFTER the move
    nextToMove = Opponent(mv.player);
    if (mv.player == Black) bs += (float)taken; else if (mv.player == White) ws += (float)taken;
    return shared_from_this();
  }
  float Score(Player p) const override {
    if (p == White) return ws;
    if (p == Black) return bs;
    throw std::runtime_error("unreachable");
  }
  float AdditionalScore() const override { return 0; }
  bool Ended(Player* winner) const override {  // game.go:145-187
    if (ws >= k) { *winner = White; return true; }
    if (bs >= k) { *winner = Black; return true; }
    bool cur = false, opp = false;
    for (int i = 0; i < (int)board.size(); i++)
      if (board[i] == None && Check(PlayerMove{nextToMove, i})) { cur = true; break; }
    for (int i = 0; i < (int)board.size(); i++)
      if (board[i] == None && Check(PlayerMove{Opponent(nextToMove), i})) { opp = true; break; }
    if (cur && opp) { *winner = None; return false; }
    if (ws > bs) { *winner = White; return true; }
    if (bs > ws) { *winner = Black; return true; }
    *winner = None;
    return true;
  }
  void Reset() override {  // game.go:189-199
    for (auto& c : board) c = None;
    history.clear(); historical.clear();
    histPtr = 0; nextToMove = Black; ws = 0; bs = 0;
    z = makeZobrist(m, n);
  }
  void UndoLastMove() override {  // game.go:201-206 (does not restore captures)
    if (!history.empty()) { board.at(history.at(histPtr - 1).move) = None; histPtr--; }
  }
  void Fwd() override { if (!history.empty()) histPtr++; }
  bool Eq(const State* other) const override {  // game.go:214-231
    const Komi* ot = dynamic_cast<const Komi*>(other);
    if (!ot) return false;
    if (nextToMove != ot->nextToMove || board.size() != ot->board.size() ||
        (history.size() != ot->history.size() &&
         (history.size() > 0 && ot->history.size() > 0 && (histPtr - 1) != (ot->histPtr - 1))))
      return false;
    for (size_t i = 0; i < board.size(); i++)
      if (board[i] != ot->board[i]) return false;
    return true;
  }
  StatePtr Clone() const override {  // game.go:233-250 (ws/bs are NOT copied)
    auto r = std::make_shared<Komi>(m, n, (int)k);
    r->board = board;
    r->history = history;
    r->historical = historical;
    r->nextToMove = nextToMove;
    r->histPtr = histPtr;
    r->z = z;
    return r;
  }
};

// ------------------------------------------------------------------------------------------------
// game/wq/wq.go — Board (faithful), and game/wq/game.go — Game (completed: the reference panics in
// Score/Reset/UndoLastMove/Fwd and cannot apply a pass; rules chosen are stated in DESIGN.md).
struct WQBoard {
  int size;
  std::vector<Colour> data;
  Zobrist z;
  explicit WQBoard(int s) : size(s), data((size_t)s * s, None) {  // wq.go:60-70, zobrist.go:31-42
    z.table.assign((size_t)2 * s * s, 0);
    SplitMix64 r(kZobristSeed);
    for (auto& t : z.table) t = r.int31();
  }
  CaptureEngine eng() const { return CaptureEngine{&data, size, size, size}; }
  // wq.go:205-234
  bool check(PlayerMove mv, std::vector<Single>* caps) const {
    Coord c{mv.move / size, mv.move % size};
    std::vector<Coord> cc;
    bool ok = eng().check(c, mv.player, &cc);
    caps->clear();
    if (ok) for (Coord q : cc) caps->push_back(q.X * size + q.Y);
    return ok;
  }
  // wq.go:141-171: returns captures count or -1 on error
  int Apply(PlayerMove mv) {
    if (!(mv.player == Black || mv.player == White)) return -1;
    if (mv.move >= size * size || mv.move < 0) return -1;
    if (data[mv.move] != None) return -1;
    std::vector<Single> caps;
    if (!check(mv, &caps)) return -1;
    data[mv.move] = mv.player;
    z.update(mv);
    for (Single p : caps) {
      data[p] = None;
      z.update(PlayerMove{Opponent(mv.player), p});
    }
    return (int)(uint8_t)caps.size();  // byte(len(captures))
  }
  // wq.go:173-202 — the reference flood fill with adjacents {-size, 1, size, 1} and bound `a >= size`
  // (buggy, but pinned by the reference KATs wq_test.go:33-196).
  float Score(Player player) const {
    Colour colour = player;
    std::vector<char> bd(data.size(), 0);
    std::vector<int32_t> q;
    size_t head = 0;
    int32_t adjacents[4] = {-size, 1, size, 1};
    float reachable = 0;
    for (int32_t i = 0; i < (int32_t)data.size(); i++)
      if (data[i] == colour) { reachable++; bd[i] = 1; q.push_back(i); }
    while (head < q.size()) {
      int32_t i = q[head++];
      for (int32_t ad : adjacents) {
        int32_t a = i + ad;
        if (a >= size || a < 0) continue;
        if (!bd[a] && data[a] == None) { reachable++; bd[a] = 1; q.push_back(a); }
      }
    }
    return reachable;
  }
  // Tromp-Taylor area score of `player` (stones + empty regions bordered only by player): used by the
  // completed Game.Score (the reference's Game.Score panics, game.go:180).
  float AreaScore(Player player) const {
    int N = size * size;
    std::vector<int> seen(N, 0);
    float total = 0;
    for (int i = 0; i < N; i++) if (data[i] == player) total += 1;
    for (int s = 0; s < N; s++) {
      if (data[s] != None || seen[s]) continue;
      std::vector<int> st{s};
      seen[s] = 1;
      int cnt = 0;
      bool tb = false, tw = false;
      while (!st.empty()) {
        int c = st.back(); st.pop_back(); cnt++;
        int x = c / size, y = c % size;
        const int dx[4] = {0, 1, 0, -1}, dy[4] = {1, 0, -1, 0};
        for (int d = 0; d < 4; d++) {
          int nx = x + dx[d], ny = y + dy[d];
          if (nx < 0 || ny < 0 || nx >= size || ny >= size) continue;
          int a = nx * size + ny;
          if (data[a] == Black) tb = true;
          else if (data[a] == White) tw = true;
          else if (!seen[a]) { seen[a] = 1; st.push_back(a); }
        }
      }
      if (player == Black && tb && !tw) total += cnt;
      if (player == White && tw && !tb) total += cnt;
    }
    return total;
  }
};

struct WQ : State {
  WQBoard board;
  std::vector<PlayerMove> history;
  std::vector<std::vector<Colour>> historical;  // board AFTER each move (komi convention, komi/game.go:111-121)
  Player nextToMove = Black;
  float komi;
  int moveCount = 0, passes = 0, histPtr = 0, handicap = 0;
  uint8_t captures[2] = {0, 0};
  // pre-move snapshots for UndoLastMove (completion)
  struct Snap { std::vector<Colour> data; int32_t hash; int passes; uint8_t cap[2]; Player next; };
  std::vector<Snap> undo;

  WQ(int size, int handicap_, double komi_) : board(size), komi((float)komi_), handicap(handicap_) {}  // game.go:28-38
  int Kind() const override { return 3; }
  void BoardSize(int& a, int& b) const override { a = board.size; b = board.size; }
  const std::vector<Colour>& Board() const override { return board.data; }
  const std::vector<Colour>& Historical(int i) const override { return historical.at(i); }
  uint32_t Hash() const override { return (uint32_t)board.z.hash; }
  int ActionSpace() const override { return (int)board.data.size(); }
  void SetToMove(Player p) override { nextToMove = p; }
  Player ToMove() const override { return nextToMove; }
  PlayerMove LastMove() const override {  // game.go:54-59
    if (!history.empty() && histPtr > 0) return history.at(histPtr - 1);
    return PlayerMove{None, -1};
  }
  int Passes() const override { return passes; }
  int MoveNumber() const override { return histPtr; }  // == len(history) except between Undo and Fwd
  int Handicap() const override { return handicap; }
  bool Check(PlayerMove mv) const override {  // game.go:65-79 (no ko / superko)
    if (mv.move == Resign) return true;
    if (mv.move == Pass) return true;
    if (mv.move >= (int)board.data.size() || mv.move < 0) return false;
    if (board.data[mv.move] != None) return false;  // Board.Apply rejects occupied cells (wq.go:152)
    std::vector<Single> caps;
    return board.check(mv, &caps);
  }
  StatePtr Apply(PlayerMove mv) override {  // game.go:81-92 clones; completion: passes, historical
    auto ns = std::static_pointer_cast<WQ>(Clone());
    ns->undo.push_back(Snap{ns->board.data, ns->board.z.hash, ns->passes, {ns->captures[0], ns->captures[1]}, ns->nextToMove});
    if (mv.move == Pass) {
      ns->passes++;
    } else {
      int c = ns->board.Apply(mv);
      if (c >= 0) ns->captures[mv.player - 1] += (uint8_t)c;
      ns->passes = 0;
    }
    ns->nextToMove = Opponent(mv.player);
    ns->history.resize(ns->histPtr);
    ns->history.push_back(mv);
    ns->historical.resize(ns->histPtr);
    ns->historical.push_back(ns->board.data);
    ns->histPtr++;
    ns->moveCount++;
    return ns;
  }
  float Score(Player p) const override { return board.AreaScore(p); }
  float AdditionalScore() const override { return komi; }
  bool Ended(Player* winner) const override {  // game.go:94-115
    *winner = None;
    if (passes < 2) return false;
    float w = Score(White), b = Score(Black);
    // the reference compares raw scores; komi enters only through mcts' combinedScore (utils.go:62-67).
    // Completion: komi is added to White so that Ended() and combinedScore agree.
    w += komi;
    if (w == b) *winner = None; else if (w > b) *winner = White; else *winner = Black;
    return true;
  }
  void Reset() override {
    for (auto& c : board.data) c = None;
    board.z.hash = 0;
    history.clear(); historical.clear(); undo.clear();
    nextToMove = Black; moveCount = 0; passes = 0; histPtr = 0; captures[0] = captures[1] = 0;
  }
  void UndoLastMove() override {  // completion: full restore
    if (histPtr <= 0) return;
    const Snap& s = undo.at(histPtr - 1);
    board.data = s.data; board.z.hash = s.hash; passes = s.passes;
    captures[0] = s.cap[0]; captures[1] = s.cap[1]; nextToMove = s.next;
    histPtr--; moveCount--;
  }
  void Fwd() override {  // completion: re-apply history[histPtr]
    if (histPtr >= (int)history.size()) return;
    PlayerMove mv = history[histPtr];
    if (mv.move == Pass) passes++; else { int c = board.Apply(mv); if (c >= 0) captures[mv.player - 1] += (uint8_t)c; passes = 0; }
    nextToMove = Opponent(mv.player);
    histPtr++; moveCount++;
  }
  bool Eq(const State* other) const override {  // game.go:123-161 (board + counters + history prefix)
    const WQ* ot = dynamic_cast<const WQ*>(other);
    if (!ot) return false;
    if (nextToMove != ot->nextToMove || komi != ot->komi || moveCount != ot->moveCount || passes != ot->passes ||
        handicap != ot->handicap)
      return false;
    if (captures[0] != ot->captures[0] || captures[1] != ot->captures[1]) return false;
    if (board.size != ot->board.size || board.z.hash != ot->board.z.hash) return false;
    for (size_t i = 0; i < board.data.size(); i++) if (board.data[i] != ot->board.data[i]) return false;
    for (int i = 0, j = 0; i < histPtr && j < ot->histPtr; i++, j++)
      if (history[i].player != ot->history[j].player || history[i].move != ot->history[j].move) return false;
    return true;
  }
  StatePtr Clone() const override {  // game.go:163-175
    auto r = std::make_shared<WQ>(board.size, handicap, komi);
    r->board = board;
    r->history = history; r->historical = historical; r->undo = undo;
    r->nextToMove = nextToMove; r->moveCount = moveCount; r->passes = passes; r->histPtr = histPtr;
    r->captures[0] = captures[0]; r->captures[1] = captures[1];
    return r;
  }
};

}  // namespace oracle
