"""CPU: the address arithmetic of the chained Winograd GEMMs, checked on the code the kernels run.

agogo_amd/csrc/gemm_maps.hpp holds every offset of wino_gemm_h2g_kernel (the measured GEMM) and wino_gemm_h2p_kernel (the persistent
one) as plain constexpr functions; conv_wino_h2c.hpp computes its DMA source offsets, LDS fragment addresses, M store addresses, ring
slots, counted waits and work list ONLY through them.  tests/cpp/gemm_maps_check.cpp includes the same header under g++ and checks
properties (every unit lands once and where the fragment reads expect it; 16 distinct bank slots per ds_read_b128 lane group; M stores
cover the tile once in 256-byte runs; the ring never overwrites an unread stage; the waits cover the stage they are for; the work list
deals every unit once, a team's slabs on one XCD).  Editing an offset in the header — or replacing a helper call in a kernel by a
literal, which the second test forbids — turns this red without a GPU.  (Replaces round 4's numpy restatements)"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gemm_maps_properties(tmp_path):
    exe = str(tmp_path / "gemm_maps_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "gemm_maps_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "GEMM_MAPS OK" in out.stdout, out.stdout[-3000:]


def _kernel_body(src, name):
    i = src.index("void %s(" % name)
    j = src.index("{", i)
    depth, k = 0, j
    while True:
        depth += {"{": 1, "}": -1}.get(src[k], 0)
        if depth == 0:
            return src[j:k + 1]
        k += 1


def test_the_kernels_take_their_offsets_from_the_header():
    src = open(os.path.join(ROOT, "agogo_amd", "csrc", "conv_wino_h2c.hpp")).read()
    for name, need in (("wino_gemm_h2g_kernel", ("maps::h2c_dma_src", "maps::h2c_dma_dst", "maps::h2c_wave_part", "maps::h2c_frag", "maps::mc_index", "maps::mfma_row")),
                       ("wino_gemm_h2p_kernel", ("maps::h2c_dma_src", "maps::h2c_dma_dst", "maps::h2c_wave_part", "maps::h2c_frag", "maps::mc_index", "maps::mfma_row",
                                                 "maps::h2p_team", "maps::h2p_u0", "maps::h2p_v_base", "maps::h2p_slot_ahead", "maps::h2p_slot_next", "maps::h2p_kk_ahead",
                                                 "maps::h2p_tiles_ahead", "maps::h2p_wait_early", "maps::h2p_wait_steady", "maps::h2p_early"))):
        body = _kernel_body(src, name)
        body = re.sub(r"//[^\n]*", "", body)
        for fn in need:
            assert fn in body, "%s no longer calls %s" % (name, fn)
        # no hand-written swizzle / image arithmetic beside the helpers: no XOR, no 16-byte-unit shifts, no literal waits
        assert "^" not in body, "%s: an XOR outside gemm_maps.hpp" % name
        assert not re.search(r"<<\s*4\b", body), "%s: a unit shift outside gemm_maps.hpp" % name
        assert not re.search(r"vmcnt\(\d", body), "%s: a literal wait count" % name
    # the device code includes the very header the check compiles
    assert '#include "gemm_maps.hpp"' in open(os.path.join(ROOT, "agogo_amd", "csrc", "net.hip")).read()
