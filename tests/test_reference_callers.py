"""The reference's OWN callers compile unchanged against this library.

include/compat/{genasm_gpu,genasm_cpu,util}.hpp carry the reference's header names and declare its namespaces on
top of the C ABI, including `enabled_algorithm_log` as an assignable object (src/genasm_gpu.hpp:6,
src/library_example.cu:91-92, src/tests.cu:791-792).  The source that is compiled is read from /root/reference at
test time — it exists in the build container only, nothing of it is stored here or travels to the GPU box, so these
tests skip there (the GPU run of the same four call shapes is tests/test_cpp_shim.py with examples/library_example.cpp).
"""
import os
import subprocess

import pytest

import scrooge_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"
EXAMPLE = os.path.join(REF_SRC, "library_example.cu")

needs_reference = pytest.mark.skipif(not os.path.exists(EXAMPLE), reason="reference sources are only in the build container")


def _gxx(args, stdin_path=None, **kw):
    libdir = os.path.join(ROOT, "scrooge_amd")
    cmd = ["g++", "-std=c++17", "-Wall"] + args + ["-L" + libdir, "-lscrooge_amd", "-Wl,-rpath," + libdir]
    if stdin_path is None:
        return subprocess.run(cmd, capture_output=True, text=True, **kw)
    with open(stdin_path) as fh:       # fed through stdin: `#include "x.hpp"` must not find the reference's x.hpp next to the file
        return subprocess.run(cmd, stdin=fh, capture_output=True, text=True, **kw)


@needs_reference
def test_reference_library_example_compiles_unchanged(tmp_path):
    scrooge_amd.build_library()
    exe = str(tmp_path / "ref_library_example")
    p = _gxx(["-x", "c++", "-I" + os.path.join(ROOT, "include", "compat"), "-I" + os.path.join(ROOT, "include"),
              "-o", exe, "-"], stdin_path=EXAMPLE)
    assert p.returncode == 0, p.stderr
    # no GPU here: the program must fail loudly (the shim throws), not produce CPU results
    lib = scrooge_amd.load_library()
    if lib.scrg_device_count() == 0:
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode != 0 and "edit_distance" not in r.stdout
        assert "no usable HIP device" in r.stderr


@needs_reference
def test_reference_library_example_with_the_references_own_util_hpp(tmp_path):
    """Inside the reference tree the maintainer keeps src/util.hpp: the compat headers must coexist with it (the shim
    does not declare Genome_t & co. a second time)."""
    scrooge_amd.build_library()
    only = tmp_path / "hdr"
    only.mkdir()
    for h in ("genasm_gpu.hpp", "genasm_cpu.hpp"):
        (only / h).write_text('#include "%s"\n' % os.path.join(ROOT, "include", "compat", h))
    p = _gxx(["-x", "c++", "-I" + str(only), "-I" + REF_SRC, "-I" + os.path.join(ROOT, "include"),
              "-o", str(tmp_path / "exe"), "-"], stdin_path=EXAMPLE)
    assert p.returncode == 0, p.stderr


CALLER = r"""
#include "genasm_gpu.hpp"
#include "genasm_cpu.hpp"
#include <cstdio>
int main()
{
    // assignable and readable, like the reference's extern bool (src/tests.cu:791-792)
    bool verbose = true;
    genasm_cpu::enabled_algorithm_log = verbose;
    if (!genasm_gpu::enabled_algorithm_log) return 10;             // one switch behind both names
    genasm_gpu::enabled_algorithm_log = false;
    if (genasm_cpu::enabled_algorithm_log) return 11;
    bool copy = genasm_gpu::enabled_algorithm_log;
    if (copy) return 12;
    // overload resolution as against the reference's header: NULL for the out-parameter, defaulted arguments
    std::vector<std::string> t = {"ACGTACGT"}, q = {"ACGTACG"};
    Genome_t g;
    g.content = "ACGTACGT";
    std::vector<Read_t> reads;
    try {
        (void)genasm_gpu::align_all(t, q, NULL);
        (void)genasm_gpu::align_all(g, reads, NULL);
        (void)genasm_cpu::align_all(t, q, 4);
        (void)genasm_cpu::align_all(g, reads);
        long long ns = 0;
        (void)genasm_cpu::align_all(g, reads, 2, &ns);
        (void)measure_ns([&]() { (void)genasm_gpu::align_all(t, q); });
    } catch (const std::exception& e) {
        std::printf("threw: %s\n", e.what());
        return 2;
    }
    std::printf("ran\n");
    return 0;
}
"""


def test_compat_headers_offer_the_reference_surface(tmp_path):
    """Runs everywhere (no reference needed): the switch object and the overload set of the compat headers."""
    scrooge_amd.build_library()
    src = tmp_path / "caller.cpp"
    src.write_text(CALLER)
    exe = str(tmp_path / "caller")
    p = _gxx(["-Werror", "-Wno-conversion-null", "-I" + os.path.join(ROOT, "include", "compat"),
              "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe])
    assert p.returncode == 0, p.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    has_gpu = scrooge_amd.load_library().scrg_device_count() > 0
    assert r.returncode == (0 if has_gpu else 2), r.stdout + r.stderr


# ---- the reference's third export: __global__ genasm_gpu::ascii_to_twobit_strings (src/genasm_gpu.hpp:9) ----
TWOBIT_SRC = os.path.join(ROOT, "tests", "proto", "twobit_caller.hip")
TWOBIT_EXE = "/tmp/scrg_twobit_caller"
# the six strings of the reference's own test (src/tests.cu:583-590), then mixed case and lengths around a workgroup's stride
TWOBIT_INPUTS = ["", "A", "ACGT", "ACGTA", "AAAAAAAACCCCCCCCGGGGGGGGTTTTTTTT", "AAAAAAAACCCCCCCCGGGGGGGGTTTTTTTTA",
                 "acgtTGCA", "T" * 127, "GATTACA" * 37, "C" * 128 + "G", "TTTTGGGGCCCCAAAA" * 64 + "ACG"]


def _twobit_expected(seq):
    """4 bases per byte, the first base of a quad in bits 7..6, the tail zero-padded (src/genasm_gpu.cu:640-669)."""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    out = bytearray((len(seq) + 3) // 4)
    for k, ch in enumerate(seq.upper()):
        out[k // 4] |= code[ch] << (6 - 2 * (k % 4))
    return out.hex()


def build_twobit_caller():
    """hipcc, against include/compat only (the kernel is compiled in the CALLER's translation unit from the compat header)."""
    scrooge_amd.build_library()
    libdir = os.path.join(ROOT, "scrooge_amd")
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-Wall", "-Werror",
                        "-I" + os.path.join(ROOT, "include", "compat"), "-I" + os.path.join(ROOT, "include"), TWOBIT_SRC,
                        "-L" + libdir, "-lscrooge_amd", "-Wl,-rpath," + libdir, "-o", TWOBIT_EXE], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr


def test_third_export_compiles_with_hipcc_against_the_compat_headers():
    """Runs everywhere: a tests.cu:582-647-shaped caller of the exported kernel compiles and links for gfx950."""
    build_twobit_caller()
    assert os.path.exists(TWOBIT_EXE)


@pytest.mark.gpu
def test_third_export_packs_the_references_strings():
    """The kernel launched as src/tests.cu:626 launches it (<<<32, 32>>>), on the six strings of src/tests.cu:583-590 and more."""
    build_twobit_caller()
    r = subprocess.run([TWOBIT_EXE] + [s or "-" for s in TWOBIT_INPUTS], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split("\n")[:len(TWOBIT_INPUTS)] == [_twobit_expected(s) for s in TWOBIT_INPUTS]


def test_third_export_header_in_two_translation_units(tmp_path):
    """The compat header defines the kernel (`static`, a kernel cannot be `inline`): a caller made of several translation units
    that all include it must still link (no duplicate symbols), for gfx950, with hipcc."""
    scrooge_amd.build_library()
    libdir = os.path.join(ROOT, "scrooge_amd")
    a, b = tmp_path / "a.hip", tmp_path / "b.hip"
    a.write_text('#include "genasm_gpu.hpp"\nvoid launch_b(int, long long*, char**, char**);\n'
                 'int main() { genasm_gpu::ascii_to_twobit_strings<<<1, 32>>>(0, nullptr, nullptr, nullptr); launch_b(0, nullptr, nullptr, nullptr);'
                 ' return hipDeviceSynchronize() == hipSuccess ? 0 : 1; }\n')
    b.write_text('#include "genasm_gpu.hpp"\nvoid launch_b(int n, long long* l, char** x, char** y) { genasm_gpu::ascii_to_twobit_strings<<<4, 64>>>(n, l, x, y); }\n')
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-Wall", "-Werror", "-Wno-unused-result",
                        "-I" + os.path.join(ROOT, "include", "compat"), "-I" + os.path.join(ROOT, "include"), str(a), str(b),
                        "-L" + libdir, "-lscrooge_amd", "-Wl,-rpath," + libdir, "-o", str(tmp_path / "two_tu")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
