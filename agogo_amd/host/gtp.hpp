// GTP (Go Text Protocol 2) front end over the device engine — SURVEY 8(f) row 4: tournament use of Agent.Search.
// Mirrors the reference's engine skeleton internal/gtp/gtp.go (command set :12-31, `[id] command args` parsing with the
// id optional and the line lower-cased :83-112, replies "= [id] result\n\n" / "? [id] error\n\n" :139-154); the
// reference's Generate hook (gtp.go:42) is Agent.Search, here BatchedArena-style begin_move/simulate/end_move on ONE game
// held on the device, and `play` is agz_arena_apply_moves (State.Check'ed on the device).
// Replies without an id are byte-identical to the reference's (its Test_General strings are pinned in tests/test_gtp_protocol.py);
// with an id the reference prints "= 7 result" (gtp.go:150), which GTP controllers reject — the spec's "=7 result" is used.
// Host-side glue only: no GPU code here.  Moves must alternate colours (the device game alternates; GTP's free-form
// "same colour twice" is answered with an error rather than by inventing passes).
#pragma once
#include <cctype>
#include <functional>
#include <istream>
#include <memory>
#include <ostream>
#include <sstream>
#include <string>
#include <vector>

#include "agogo.hpp"

namespace gtp {

// The protocol without a game (the reference's gtp.New(nil, name, version, nil)): administrative commands, parsing, framing.
struct Protocol {
  std::string name, version;
  Protocol(const std::string& n = "agz-hip", const std::string& v = "r01") : name(n), version(v) {}
  virtual ~Protocol() {}
  static const std::vector<std::string>& known() {
    static const std::vector<std::string> k = {"protocol_version", "name", "version", "known_command", "list_commands", "quit", "boardsize",
                                               "clear_board", "komi", "play", "genmove", "showboard"};
    return k;
  }
  // board commands; the game-less protocol has none to offer
  virtual bool board_command(const std::string&, const std::vector<std::string>&, std::string* out) { *out = "no game attached"; return false; }

  // one command -> one reply (without the id prefix); returns false on failure with the message in *out
  bool command(const std::string& cmd, const std::vector<std::string>& args, std::string* out, bool* quit) {
    *out = "";
    bool is_known = false;
    for (const std::string& k : known()) is_known = is_known || k == cmd;
    if (!is_known) { *out = "Unknown command \"" + cmd + "\""; return false; }   // gtp.go:103
    if (cmd == "protocol_version") { *out = "2"; return true; }
    if (cmd == "name") { *out = name; return true; }
    if (cmd == "version") { *out = version; return true; }
    if (cmd == "list_commands") { for (const std::string& k : known()) { if (!out->empty()) *out += "\n"; *out += k; } return true; }
    if (cmd == "known_command") { bool f = false; for (const std::string& k : known()) f = f || (!args.empty() && args[0] == k); *out = f ? "true" : "false"; return true; }
    if (cmd == "quit") { *quit = true; return true; }
    return board_command(cmd, args, out);
  }

  // the protocol loop (internal/gtp/gtp.go:66-81,139-154)
  void run(std::istream& in, std::ostream& os) {
    std::string line;
    bool quit = false;
    while (!quit && std::getline(in, line)) {
      size_t hash = line.find('#');
      if (hash != std::string::npos) line.erase(hash);
      for (char& ch : line) ch = (char)std::tolower((unsigned char)ch);
      std::istringstream ss(line);
      std::vector<std::string> tok;
      for (std::string t; ss >> t;) tok.push_back(t);
      if (tok.empty()) continue;
      std::string id;
      if (std::isdigit((unsigned char)tok[0][0])) { id = tok[0]; tok.erase(tok.begin()); }
      if (tok.empty()) continue;   // an id alone is ignored (gtp.go:96-98)
      std::string cmd = tok[0], reply;
      tok.erase(tok.begin());
      bool ok;
      try { ok = command(cmd, tok, &reply, &quit); } catch (const std::exception& e) { ok = false; reply = e.what(); }
      os << (ok ? "=" : "?") << id << (reply.empty() ? "" : " ") << reply << "\n\n";
      os.flush();
    }
  }
};

struct Engine : Protocol {
  agz::Ctx& ctx;
  dual::Dual& net;
  mcts::Config mc;
  int size;
  float komi = 7.5f;
  int lanes = 1;
  std::unique_ptr<agogo::Arena> arena;

  Engine(agz::Ctx& c, dual::Dual& n, const mcts::Config& m, int boardsize, int lanes_ = 1) : ctx(c), net(n), mc(m), size(boardsize), lanes(lanes_) { clear(); }

  void clear() {
    arena.reset(new agogo::Arena(ctx, AGZ_GAME_WQ, size, size, 0, komi, AGZ_ENC_WQ, mc, 1, 1337));
    arena->SetAgents(&net, &net);
    arena->SetParallel(lanes);
    uint8_t a_black = 1;
    agz::check(agz_arena_reset(arena->h, &a_black), "clear_board");
  }
  // --- vertices: letters skip 'i', row 1 is the bottom row (GTP 2 spec 2.11)
  bool parse_vertex(const std::string& v, int32_t* mv) const {
    if (v == "pass") { *mv = AGZ_PASS; return true; }
    if (v == "resign") { *mv = AGZ_RESIGN; return true; }
    if (v.size() < 2 || !std::isalpha((unsigned char)v[0])) return false;
    char c = (char)std::tolower((unsigned char)v[0]);
    if (c == 'i') return false;
    int col = c - 'a' - (c > 'i' ? 1 : 0);
    int row = 0;
    for (size_t k = 1; k < v.size(); k++) { if (!std::isdigit((unsigned char)v[k])) return false; row = row * 10 + (v[k] - '0'); }
    if (col < 0 || col >= size || row < 1 || row > size) return false;
    *mv = (size - row) * size + col;
    return true;
  }
  std::string vertex(int32_t mv) const {
    if (mv == AGZ_PASS) return "pass";
    if (mv == AGZ_RESIGN) return "resign";
    int r = mv / size, c = mv % size;
    char letter = (char)('A' + c + (c >= 8 ? 1 : 0));
    return std::string(1, letter) + std::to_string(size - r);
  }
  static int colour(const std::string& s) { return (s == "b" || s == "black") ? AGZ_BLACK : (s == "w" || s == "white") ? AGZ_WHITE : 0; }
  agz_game_state state() const { agz_game_state st{}; agz::check(agz_arena_get_game(arena->h, 0, nullptr, &st), "state"); return st; }

  bool board_command(const std::string& cmd, const std::vector<std::string>& args, std::string* out) override {
    if (cmd == "boardsize") {
      if (args.empty() || std::atoi(args[0].c_str()) != size) { *out = "unacceptable size"; return false; }   // the network fixes the size
      return true;
    }
    if (cmd == "clear_board") { clear(); return true; }
    if (cmd == "komi") { if (args.empty()) { *out = "syntax error"; return false; } komi = (float)std::atof(args[0].c_str()); clear(); return true; }
    if (cmd == "showboard") {
      std::vector<int32_t> b((size_t)size * size);
      agz_game_state st{};
      agz::check(agz_arena_get_game(arena->h, 0, b.data(), &st), "showboard");
      std::ostringstream os;
      for (int r = 0; r < size; r++) { os << "\n"; for (int c = 0; c < size; c++) os << (b[(size_t)r * size + c] == AGZ_BLACK ? 'X' : b[(size_t)r * size + c] == AGZ_WHITE ? 'O' : '.'); }
      *out = os.str();
      return true;
    }
    if (cmd == "play" || cmd == "genmove") {
      int col = args.empty() ? 0 : colour(args[0]);
      if (!col) { *out = "syntax error"; return false; }
      agz_game_state st = state();
      if (st.ended) { *out = "game is over"; return false; }
      if (st.to_move != col) { *out = "it is the other colour's turn"; return false; }
      if (cmd == "play") {
        int32_t mv;
        if (args.size() < 2 || !parse_vertex(args[1], &mv)) { *out = "syntax error"; return false; }
        if (agz_arena_apply_moves(arena->h, &mv) != AGZ_OK) { *out = "illegal move"; return false; }
        return true;
      }
      arena->Search(mc.Budget);
      int32_t hist[1024];
      int n = 0;
      agz::check(agz_arena_get_history(arena->h, 0, hist, 1024, &n), "genmove");
      *out = vertex(n > 0 ? hist[n - 1] : AGZ_PASS);
      return true;
    }
    *out = "not implemented";
    return false;
  }
};

}  // namespace gtp
