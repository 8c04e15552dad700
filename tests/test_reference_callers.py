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
