"""CPU: the C-ABI library loads, exports every symbol include/agz.h declares, validates arguments, and fails
LOUDLY without a GPU (no CPU fallback anywhere in the product path)."""
import ctypes as C
import os
import re

import pytest

import agogo_amd as A
from agogo_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(headers=("agz.h", "agz_debug.h")):
    """every function include/*.h declares (the drop-in surface agz.h and the measurement/test hooks agz_debug.h)"""
    names = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(agz_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_debug_hooks_stay_out_of_the_product_header():
    prod = declared_symbols(("agz.h",))
    assert not [n for n in prod if "prof" in n or "wino_stages" in n], "test/measurement hooks belong in agz_debug.h"
    assert "agz_mcts_search" in prod and "agz_comm_init_all" in prod


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 35
    L = C.CDLL(capi.lib_path())
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_binding_covers_header():
    L = capi.lib()
    for n in declared_symbols():
        assert getattr(L, n).argtypes is not None or n in ("agz_last_error", "agz_version"), n


def test_version_string():
    assert b"gfx950" in capi.lib().agz_version()


def test_struct_layouts_match_header():
    # agz_net_conf: 9 int32 + 1 float ; agz_game_conf: 4 int32 + float + 2 int32 ; agz_mcts_conf: 40 bytes
    assert C.sizeof(capi.NetConf) == 40
    assert C.sizeof(capi.GameConf) == 28
    assert C.sizeof(capi.MctsConf) == 40
    assert C.sizeof(capi.ArenaStats) == 80
    assert C.sizeof(capi.GameState) == 40


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(A.AgzError) as e:
        A.Ctx(0)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_null_arguments_are_rejected():
    L = capi.lib()
    assert L.agz_ctx_create(0, None) == -1  # AGZ_E_INVALID
    assert b"NULL" in L.agz_last_error() or b"out" in L.agz_last_error()
    assert L.agz_net_create(None, None, None) == -1
    assert L.agz_arena_create(None, None, None, 1, 0, 0, None) == -1
    assert L.agz_net_num_params(None) == 0


def test_product_path_never_touches_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may reference oracle/"""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "agogo_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"oracle_lib|liboracle|#include\s+\"[^\"]*oracle/", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
    mk = open(os.path.join(ROOT, "Makefile")).read()
    assert "liboracle" not in mk.split("$(OUT)/libagz.so: $(OBJS)")[1].split("\n")[1]


def test_header_is_plain_c99_and_links(tmp_path):
    """cgo compiles include/agz.h with a C compiler: the header must be valid C99 (no C++-isms), and a C program using it
    must link against libagz.so.  Without a GPU the one call it makes fails loudly (no CPU fallback)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi_c.c"
    src.write_text('''#include "agz.h"
#include <stdio.h>
#include <string.h>
int main(void) {
  agz_ctx* c = 0;
  agz_net_conf nc = {3, 3, 8, 1, 3, 3, 2, 10, AGZ_BN_DEGENERATE_EPS, 1e-5f};
  agz_game_conf gc = {AGZ_GAME_MNK, 3, 3, 3, 0.0f, 0, AGZ_ENC_TWOPLANE};
  agz_mcts_conf mc = {1.0f, 3, 3, 0, 10, 0, 0.0f, 1, 0.0f, AGZ_DONT_PREFER_PASS};
  int r = agz_ctx_create(0, &c);
  (void)nc; (void)gc; (void)mc;
  printf("%d|%s\\n", r, r ? agz_last_error() : "ok");
  if (r == 0) agz_ctx_destroy(c);
  return 0;
}
''')
    exe = tmp_path / "abi_c"
    lib_dir = os.path.join(root, "agogo_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), str(src),
                           "-o", str(exe), "-L", lib_dir, "-lagz", "-Wl,-rpath," + lib_dir])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60).stdout.strip()
    code, msg = out.split("|", 1)
    import torch
    if torch.cuda.is_available():
        assert code == "0"
    else:
        assert int(code) < 0 and "no CPU fallback" in msg


def test_go_shim_only_references_declared_symbols():
    """go/agzhip cannot be compiled in this image (no Go toolchain): at least every C.agz_* / C.AGZ_* it uses must exist in
    include/agz.h, and its braces must balance."""
    src = open(os.path.join(ROOT, "go", "agzhip", "agzhip.go")).read()
    hdr = open(os.path.join(ROOT, "include", "agz.h")).read()
    code = re.sub(r"//[^\n]*", "", src)
    code = re.sub(r"/\*.*?\*/", "", code, flags=re.S)
    code = re.sub(r'"(\\.|[^"\\])*"', '""', code)
    for a, b in ("{}", "()", "[]"):
        assert code.count(a) == code.count(b), (a, b)
    used = set(re.findall(r"C\.(agz_\w+|AGZ_\w+)", src))
    assert len(used) > 30
    missing = sorted(u for u in used if not re.search(r"\b%s\b" % re.escape(u), hdr))
    assert not missing, missing


def test_environment_switches_are_few_documented_and_tested():
    """at most eight environment switches in the library, every one of them documented in include/agz.h (with
    the test that runs it) and actually set by some test."""
    import glob
    srcs = glob.glob(os.path.join(ROOT, "agogo_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "agogo_amd", "csrc", "*.hpp"))
    knobs = set()
    for f in srcs:
        knobs |= set(re.findall(r'getenv\("(AGZ_\w+)"\)', open(f).read()))
    assert 0 < len(knobs) <= 8, sorted(knobs)
    hdr = open(os.path.join(ROOT, "include", "agz.h")).read()
    tests = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "tests", "*.py")) if not f.endswith("test_capi_cpu.py"))
    for k in sorted(knobs):
        assert re.search(r"\b%s\b" % k, hdr), "%s is not documented in include/agz.h" % k
        assert re.search(r"\b%s\b" % k, tests), "%s is not exercised by any test" % k
    documented = set(re.findall(r"^ \*   (AGZ_\w+)=", hdr, flags=re.M))
    assert documented == knobs, (sorted(documented), sorted(knobs))
