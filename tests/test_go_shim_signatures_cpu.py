"""CPU: the Go shim's method sets against the reference's own source.

go/agzhip cannot meet a Go compiler in this image, so nothing type-checks `*agzhip.MCTS` against the interface INTEGRATION.md section 2 puts on
`Agent.MCTS`, or `*agzhip.Inferencer` against `agogo.Inferer`.  This test parses the method declarations on both sides with a small
signature parser (receiver, name, parameter TYPES, result TYPES — names dropped) and asserts that every method the reference calls on its
tree / requires of an inferer exists in the shim with the same types.  The reference half is skipped where /root/reference is absent (the
GPU box); the shim-only checks (gofmt's import order, no leaked handle on NewMCTS's error paths) always run."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SHIM = os.path.join(ROOT, "go", "agzhip", "agzhip.go")


def _split(args):
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _types(args):
    """'policy []float32, value float32, err error' -> ['[]float32', 'float32', 'error'];  'game.State' -> ['game.State'];
    'a, b int' -> ['int', 'int']"""
    items = _split(args)
    types, pending = [], 0
    for it in items:
        toks = it.split()
        if len(toks) == 1:
            # either a bare type (unnamed list) or a name sharing the next item's type
            if any(len(x.split()) > 1 for x in items):
                pending += 1
            else:
                types.append(toks[0])
        else:
            types.extend([" ".join(toks[1:])] * (pending + 1))
            pending = 0
    return types


METHOD = re.compile(r"^func \((\w+) (\*?\w+)\) (\w+)\(([^)]*)\)\s*(\([^)]*\)|[^\s{]+)?\s*\{", re.M)


def methods(src, recv_types):
    out = {}
    for m in METHOD.finditer(src):
        if m.group(2) in recv_types:
            res = (m.group(5) or "").strip()
            res = res[1:-1] if res.startswith("(") else res
            out[m.group(3)] = (_types(m.group(4)), _types(res) if res else [])
    return out


def iface(src, name):
    body = re.search(r"type %s interface \{(.*?)\n\}" % name, src, re.S).group(1)
    out = {}
    for line in body.splitlines():
        line = line.split("//")[0].strip()
        m = re.match(r"(\w+)\(([^)]*)\)\s*(\([^)]*\)|\S+)?$", line)
        if m:
            res = (m.group(3) or "").strip()
            res = res[1:-1] if res.startswith("(") else res
            out[m.group(1)] = (_types(m.group(2)), _types(res) if res else [])
    return out


def test_signature_parser():
    assert _types("policy []float32, value float32, err error") == ["[]float32", "float32", "error"]
    assert _types("a, b int") == ["int", "int"]
    assert _types("game.State") == ["game.State"]
    assert _types("") == []
    src = "func (t *MCTS) Search(player game.Player) (retVal game.Single) {\nfunc (l lumberjack) Log() string { return \"\" }\n"
    assert methods(src, ("*MCTS", "lumberjack")) == {"Search": (["game.Player"], ["game.Single"]), "Log": ([], ["string"])}


def test_shim_source_hygiene():
    src = open(SHIM).read()
    std = re.search(r'import \(\n((?:\t"[^"]+"\n)+)\n', src).group(1).split()
    assert std == sorted(std), "gofmt: standard-library imports are sorted: %r" % std
    # NewMCTS: after agz_mcts_create succeeded every error return destroys the handle
    body = src[src.index("func NewMCTS("):src.index("// timeoutMs")]
    after = body[body.index("if err := lastErr(C.agz_mcts_create"):]
    for m in re.finditer(r"return nil, err", after):
        seg = after[:m.start()]
        last_if = seg.rfind("if err :=")
        if "agz_mcts_create" in seg[last_if:]:
            continue                                            # the create call's own failure: nothing to destroy
        assert "C.agz_mcts_destroy(t.h)" in seg[last_if:], "an error path of NewMCTS leaks the handle"
    assert "func timeoutMs(" in src and "return 1" in src[src.index("func timeoutMs("):src.index("func (t *MCTS) SetTimeout")]


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference sources are not on this box")
def test_mcts_and_inferer_method_sets_match_the_reference():
    shim = open(SHIM).read()
    tree = "".join(open(os.path.join(REF, "mcts", f)).read() for f in ("tree.go", "search.go", "graph.go", "release.go"))
    ref = methods(tree, ("*MCTS", "lumberjack"))
    got = methods(shim, ("*MCTS",))
    # what agent.go:78-79 and arena.go:107,140-141,200-202 call on Agent.MCTS (INTEGRATION.md's `searcher`), plus the tree's read-outs
    for name in ("SetGame", "Search", "Policies", "Reset", "Log", "Nodes", "ToDot"):
        assert name in ref, "the reference has no %s" % name
        assert name in got, "go/agzhip's MCTS lacks %s" % name
        assert got[name] == ref[name], "%s: shim %r, reference %r" % (name, got[name], ref[name])
    # the uses themselves are still there (a reference update that changes them must be seen here)
    agent, arena = open(os.path.join(REF, "agent.go")).read(), open(os.path.join(REF, "arena.go")).read()
    assert re.search(r"MCTS\s+\*mcts\.MCTS", agent) and "a.MCTS.SetGame(g)" in agent and "a.MCTS.Search(a.Player)" in agent
    assert arena.count("mcts.New(") == 4 and ".MCTS.Policies(a.game)" in arena and "a.A.MCTS.Reset()" in arena and "a.A.MCTS.Log()" in arena
    # INTEGRATION.md's interface = exactly those five, with the reference's types
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    want = iface(doc.replace("  \t", "\t").replace("\n  ", "\n"), "searcher")
    assert set(want) == {"SetGame", "Search", "Policies", "Reset", "Log"}, sorted(want)
    for name, sig in want.items():
        assert sig == ref[name], "INTEGRATION.md searcher.%s: %r, reference %r" % (name, sig, ref[name])
    # agogo.Inferer (datatypes.go:56-59) = Infer + io.Closer;  *agzhip.Inferencer must satisfy it
    inf = iface(open(os.path.join(REF, "datatypes.go")).read(), "Inferer")
    assert inf == {"Infer": (["[]float32"], ["[]float32", "float32", "error"])}, inf     # (io.Closer is embedded: Close() error)
    gi = methods(shim, ("*Inferencer",))
    assert gi["Infer"] == inf["Infer"] and gi["Close"] == ([], ["error"])
    # mcts.New's parameter list (tree.go:80) is what NewMCTS mirrors after the device-side arguments
    assert re.search(r"func New\(game game\.State, conf Config, nn Inferencer\) \*MCTS", tree)
    assert re.search(r"func NewMCTS\(ctx \*Ctx, kind GameKind, g game\.State, .*conf mcts\.Config, nn \*Net, seed uint64\) \(\*MCTS, error\)", shim)


def _c_prototypes():
    """name -> number of parameters, for every function include/agz.h declares"""
    hdr = open(os.path.join(ROOT, "include", "agz.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for m in re.finditer(r"^[\w \*]+?\b(agz_\w+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.M | re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(_split(args))
    return protos


def _go_calls(src):
    """(name, number of arguments) of every C.agz_xxx(...) call in the shim, by balanced-parenthesis scanning"""
    out = []
    for m in re.finditer(r"C\.(agz_\w+)\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        args = src[m.end():i - 1].strip()
        out.append((m.group(1), 0 if not args else len(_split(args)), src.count("\n", 0, m.start()) + 1))
    return out


def test_every_cgo_call_passes_as_many_arguments_as_the_header_declares():
    """A poor man's type check of the cgo calls (no Go toolchain here): every C.agz_* call names a function of include/agz.h and passes
    exactly as many arguments as its prototype has parameters."""
    protos = _c_prototypes()
    assert len(protos) > 90 and protos["agz_net_infer"] == 5 and protos["agz_last_error"] == 0, len(protos)
    calls = _go_calls(open(SHIM).read())
    assert len(calls) > 60
    hdr = open(os.path.join(ROOT, "include", "agz.h")).read()
    typedefs = set(re.findall(r"typedef\s+[^;]*?\(\s*\*\s*(agz_\w+)\s*\)", hdr)) | set(re.findall(r"}\s*(agz_\w+)\s*;", hdr))
    assert "agz_infer_fn" in typedefs
    for name, n, line in calls:
        if name in typedefs:        # C.agz_infer_fn(x): a conversion to a type of the header, not a call
            assert n == 1
            continue
        assert name in protos, "agzhip.go:%d calls C.%s, which include/agz.h does not declare" % (line, name)
        assert n == protos[name], "agzhip.go:%d: C.%s called with %d argument(s), the prototype has %d" % (line, name, n, protos[name])
