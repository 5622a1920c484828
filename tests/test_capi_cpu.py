"""CPU: the C-ABI library loads, exports every symbol include/agz.h declares, validates arguments, and fails
LOUDLY without a GPU (no CPU fallback anywhere in the product path)."""
import ctypes as C
import os
import re

import pytest

import agogo_amd as A
from agogo_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "agz.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(agz_[a-z0-9_]+)\s*\(", src)))


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
    assert C.sizeof(capi.ArenaStats) == 64
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
