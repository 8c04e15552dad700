"""CPU tests of the C-ABI boundary: the library builds for gfx950, loads, and
exports every symbol include/scrooge_amd.h declares.  No compute calls here."""
import os
import re

import pytest

import scrooge_amd
from scrooge_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    scrooge_amd.build_library()
    return scrooge_amd.load_library()


def declared_symbols(header="scrooge_amd.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(scrg_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(api.EXPORTED_SYMBOLS) == names


def test_interface_version_is_one_number(lib):
    """scrg_abi_version(): the header's SCRG_ABI_VERSION, the binding's constant and what the library reports are the same
    number — an entry point that changes its arguments under the same name (scrg_decode_edit_stream gained the capacity of
    its output array) bumps it, and a binding refuses a library of another version instead of passing shifted arguments."""
    hdr = open(os.path.join(ROOT, "include", "scrooge_amd.h")).read()
    want = int(re.search(r"#define\s+SCRG_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert lib.scrg_abi_version() == want == api.SCRG_ABI_VERSION
    assert "check_abi()" in open(os.path.join(ROOT, "include", "scrooge_amd.hpp")).read()


def test_io_header_symbols_exported(lib):
    from scrooge_amd import io as sio
    names = declared_symbols("scrooge_amd_io.h")
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(sio.IO_SYMBOLS) == names


def test_defaults_match_reference_knobs(lib):
    p = api.Params()
    lib.scrg_params_default(p)
    assert (p.W, p.O) == (64, 33)        # src/genasm_cpu.cpp:7-9
    r = api.Params()
    assert lib.scrg_params_resolve(p, r) == 0
    assert (r.W, r.O, r.lanes_per_pair) == (64, 33, 1) and r.lds_rows > 0 and r.waves_per_cu > 0
    p.W, p.O = 128, 65
    assert lib.scrg_params_resolve(p, r) == 0 and r.lanes_per_pair == 1       # one pair per lane for every W
    p.W, p.O = 256, 129
    assert lib.scrg_params_resolve(p, r) == 0 and r.lanes_per_pair == 1
    p.lanes_per_pair = 32
    assert lib.scrg_params_resolve(p, r) == 0 and r.lanes_per_pair == 32 and r.lds_rows > 0
    p.lanes_per_pair = 0
    p.W, p.O = 257, 129
    assert lib.scrg_params_resolve(p, r) != 0
    p.W, p.O = 64, 0                     # (O = 0: the reference's no-overlap special case, src/genasm_cpu.cpp:104-110 — one pair per lane only)
    assert lib.scrg_params_resolve(p, r) == 0 and (r.W, r.O, r.lanes_per_pair) == (64, 0, 1)
    p.lanes_per_pair = 8
    assert lib.scrg_params_resolve(p, r) != 0
    p.lanes_per_pair = 0
    p.W, p.O = 64, 64
    assert lib.scrg_params_resolve(p, r) != 0
    p.W, p.O = 64, -1
    assert lib.scrg_params_resolve(p, r) != 0
    p.W, p.O = 64, 33
    p.stranded = 1                       # minus-strand pairs from one packed copy of the read: the one-pair-per-lane kernels
    assert lib.scrg_params_resolve(p, r) == 0 and r.stranded == 1
    p.O = 2
    assert lib.scrg_params_resolve(p, r) == 0
    p.lanes_per_pair = 8
    assert lib.scrg_params_resolve(p, r) != 0
    p.lanes_per_pair = 0
    p.O, p.stranded = 33, 2
    assert lib.scrg_params_resolve(p, r) != 0


def test_struct_layouts():
    import ctypes as C
    assert C.sizeof(api.Run) == 2        # CigarEntry_t, src/util.hpp:43-46
    assert C.sizeof(api.PairDesc) == 48
    assert C.sizeof(api.Params) == 48    # 12 x int32 (scrooge_amd.h: scrg_params; `stranded` since ABI 7)


def test_status_strings(lib):
    for s in range(0, 7):
        assert lib.scrg_status_string(s)


def test_no_device_is_an_error_not_a_fallback(lib):
    if lib.scrg_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(scrooge_amd.ScroogeError) as e:
        scrooge_amd.Aligner(0)
    assert e.value.status == api.SCRG_ERR_NO_DEVICE


def test_product_does_not_reference_oracle():
    """The shipped path must never import, link or load anything under oracle/."""
    pkg = os.path.join(ROOT, "scrooge_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), os.path.join(d, f)


def test_unknown_experiment_switches_are_rejected(lib):
    """scrg_params.reserved[] in the SHIPPED library: nothing passes.  The selections between formulations that give
    identical results (32: no diagonal-major path for lanes_per_pair = 8; 256: the table-in-HBM kernel where a
    table-in-registers one would serve; 512 / 1024: the default kernel as two wavefronts per 64 pairs / as one) belong to the
    test build (-DSCRG_SELECT, ab_libs/lib_select.so), the scheduling switches (1, 64, 128), the counters (reserved[1]) and the
    ablation switches (2, 4, 8, 16) to the profiling builds (-DSCRG_STATS / -DSCRG_ABLATE, scripts/ab.sh); here they, and
    anything an uninitialised struct might hold, are SCRG_ERR_INVALID_ARG — and the kernels and the kernel selection contain
    none of that code (scrg_build_flags() == 0)."""
    import ctypes as C
    assert lib.scrg_build_flags() == 0, "the in-tree library must be the shipped build (no -DSCRG_SELECT / -DSCRG_STATS / -DSCRG_ABLATE)"
    p, out = api.Params(), api.Params()
    for flags, ok in ((0, True), (32, False), (256, False), (32 | 256, False), (512, False), (1024, False), (1, False), (64 | 1, False), (128, False),
                      (64, False), (2, False), (4, False), (8, False), (16, False), (0x7fffffff, False), (-1, False), (2048, False)):
        lib.scrg_params_default(C.byref(p))
        p.reserved[0] = flags
        assert (lib.scrg_params_resolve(C.byref(p), C.byref(out)) == api.SCRG_OK) == ok, flags
    for r1 in (1, -1, 7):
        lib.scrg_params_default(C.byref(p))
        p.reserved[1] = r1
        assert lib.scrg_params_resolve(C.byref(p), C.byref(out)) == api.SCRG_ERR_INVALID_ARG, r1


def test_the_test_build_accepts_the_four_selections_and_nothing_else():
    """ab_libs/lib_select.so — the same sources with -DSCRG_SELECT, what the parity tests that compare formulations load
    (tests/conftest.py: aligner_select): the four result-neutral selections pass, every other switch and the counters do not."""
    import ctypes as C
    scrooge_amd.build_library(variant="select")
    sel = scrooge_amd.load_library("select")
    assert sel.scrg_build_flags() == 4 and sel.scrg_abi_version() == api.SCRG_ABI_VERSION
    p, out = api.Params(), api.Params()
    for flags, ok in ((0, True), (32, True), (256, True), (32 | 256, True), (512, True), (1024, True), (1, False), (64, False), (128, False),
                      (2, False), (4, False), (8, False), (16, False), (2048, False), (-1, False)):
        sel.scrg_params_default(C.byref(p))
        p.reserved[0] = flags
        assert (sel.scrg_params_resolve(C.byref(p), C.byref(out)) == api.SCRG_OK) == ok, flags
    sel.scrg_params_default(C.byref(p))
    p.reserved[1] = 1
    assert sel.scrg_params_resolve(C.byref(p), C.byref(out)) == api.SCRG_ERR_INVALID_ARG


def test_shipped_kernels_have_no_experiment_plumbing():
    """The experiment plumbing is compiled out, not just switched off: in the shipped build SCRG_TIMING / SCRG_SW / SCRG_ABL /
    SCRG_SEL are the constant false (genasm_kernels.h), no kernel source tests args.debug or args.stats except through them,
    and the kernel selection (scrg_api.cpp) tests scrg_params.reserved[0] only through SCRG_SEL."""
    import re
    csrc = os.path.join(ROOT, "scrooge_amd", "csrc")
    hdr = open(os.path.join(csrc, "genasm_kernels.h")).read()
    shipped = hdr[hdr.index("#else", hdr.index("#ifdef SCRG_STATS\n#define SCRG_TIMING")):]
    assert "#define SCRG_TIMING(args) false" in shipped and "#define SCRG_SW(args, bit) false" in shipped
    assert "#define SCRG_ABL(args, bit) false" in hdr and "#define SCRG_SEL(flags, bit) false" in hdr
    assert "constexpr int32_t SCRG_ALLOWED_SWITCHES = 0;" in hdr
    for f in os.listdir(csrc):
        if not f.endswith((".hip", ".cpp")):
            continue
        txt = re.sub(r"//[^\n]*", "", open(os.path.join(csrc, f)).read())
        for m in re.finditer(r"[^\n;]*\ba\.debug\b[^;\n]*", txt):
            assert "SCRG_SEL(a.debug" in m.group(0) or "a.debug = " in m.group(0), (f, m.group(0))
        for m in re.finditer(r"if\s*\(\s*a\.stats\b", txt):
            raise AssertionError("%s tests a.stats directly: %s" % (f, m.group(0)))
        for m in re.finditer(r"[^\n;]*reserved\[0\]\s*&[^;\n]*", txt):
            assert "SCRG_ALLOWED_SWITCHES" in m.group(0), (f, m.group(0))        # (the one place: what resolve rejects)
