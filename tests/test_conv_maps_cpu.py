"""CPU: the address arithmetic of the trainer's nine-tap DMA forward convolution, checked on the code the kernel runs.

agogo_amd/csrc/conv_maps.hpp holds the image geometry of k_conv_h2dma3 (train.hip) — padded-pixel index of a GEMM row, image row of an
output pixel under a tap, rows a tile's image needs, the DMA instruction's row / source unit / landing, the LDS swizzle — as plain
constexpr functions; the kernel and the launcher's fit rule compute these ONLY through them.  tests/cpp/conv_maps_check.cpp includes the
same header under g++, replays the kernel's data movement on tagged planes and checks that every MFMA fragment read of every output pixel,
tap and k half fetches the unit the convolution's definition names (whole tiles, board crossings, the partial last tile), that the fit
rule counts exactly the rows read, and the reads' bank spread.  Six one-token mutations of the header each turn it red."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_conv_maps_properties(tmp_path):
    exe = str(tmp_path / "conv_maps_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "conv_maps_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "CONV_MAPS OK" in out.stdout, out.stdout[-3000:]


def test_a_mutated_header_fails_the_check(tmp_path):
    """the check is not vacuous: a wrong tap offset, swizzle, source unit, image base, row count or DMA row each fail it"""
    hdr = open(os.path.join(ROOT, "agogo_amd", "csrc", "conv_maps.hpp")).read()
    muts = [("return ky * Wp + kx;", "return ky * Wp + kx + 1;"), ("((row >> 2) & 3)) << 4)); }", "((row >> 1) & 3)) << 4)); }"),
            ("((lane >> 4) & 3)) << 4; }", "((lane >> 3) & 3)) << 4; }"), ("return (int)pix_first - Wp - 1;", "return (int)pix_first - Wp;"),
            ("2 * Wp + 3; }", "2 * Wp + 1; }"), ("return 16 * j + (lane >> 2);", "return 16 * j + (lane >> 3);")]
    os.makedirs(tmp_path / "agogo_amd" / "csrc")
    os.makedirs(tmp_path / "tests" / "cpp")
    src = open(os.path.join(ROOT, "tests", "cpp", "conv_maps_check.cpp")).read()
    (tmp_path / "tests" / "cpp" / "conv_maps_check.cpp").write_text(src)
    for old, new in muts:
        assert hdr.count(old) == 1, old
        (tmp_path / "agogo_amd" / "csrc" / "conv_maps.hpp").write_text(hdr.replace(old, new))
        exe = str(tmp_path / "chk")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, str(tmp_path / "tests" / "cpp" / "conv_maps_check.cpp")])
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert out.returncode != 0, "mutation %r passes the check" % new


def _kernel_body(src, name):
    i = src.index("void %s(" % name)
    j = src.index("{", i)
    depth, k = 0, j
    while True:
        depth += {"{": 1, "}": -1}.get(src[k], 0)
        if depth == 0:
            return src[j:k + 1]
        k += 1


def test_the_kernel_takes_its_offsets_from_the_header():
    src = open(os.path.join(ROOT, "agogo_amd", "csrc", "train.hip")).read()
    assert '#include "conv_maps.hpp"' in src
    body = re.sub(r"//[^\n]*", "", _kernel_body(src, "k_conv_h2dma3"))
    for fn in ("cmaps::dma_src_unit", "cmaps::dma_row", "cmaps::dma_dst", "cmaps::cd3_base", "cmaps::cd3_row0", "cmaps::cd3_tap", "cmaps::lds_off"):
        assert fn in body, "k_conv_h2dma3 no longer calls %s" % fn
    assert "^" not in body, "k_conv_h2dma3: an XOR outside conv_maps.hpp"
    assert not re.search(r"<<\s*4\b", body), "k_conv_h2dma3: a unit shift outside conv_maps.hpp"
    # the launcher's fit rule is the header's
    fit = src[src.index("bool conv3_layer(int l)"):]
    fit = fit[:fit.index("return conv3_span <= CD3_IMG;")]
    assert "cmaps::cd3_rows" in fit and "cmaps::pix" in fit
