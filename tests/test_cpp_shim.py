"""The header-only C++ shim (include/scrooge_amd.hpp) keeps the reference's
align_all signatures (src/genasm_gpu.hpp:7-8): it must compile with plain g++
against the C ABI, and on a GPU produce the reference's answers."""
import os
import subprocess

import pytest

import scrooge_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = "/tmp/scrg_library_example"

EXPECTED = [
    "pairwise cigar=7= edit_distance=0",
    "pairwise cigar=4=4D4=4I4= edit_distance=8",
    "pairwise_timed cigar=7= edit_distance=0",
    "pairwise_timed cigar=4=4D4=4I4= edit_distance=8",
    "pairwise_timed kernel_ns>0=1",
    "mapping cigar=7= edit_distance=0",
    "mapping cigar=3X4= edit_distance=3",
    "mapping cigar=12= edit_distance=0",
    "mapping_timed cigar=7= edit_distance=0",
    "mapping_timed cigar=3X4= edit_distance=3",
    "mapping_timed cigar=12= edit_distance=0",
    "mapping_timed kernel_ns>0=1",
    "resident cigar=7= edit_distance=0",
    "resident cigar=3X4= edit_distance=3",
    "resident cigar=12= edit_distance=0",
    "resident cigar=12= edit_distance=0",
]


def build_example():
    scrooge_amd.build_library()
    libdir = os.path.join(ROOT, "scrooge_amd")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "library_example.cpp"),
                           "-L" + libdir, "-lscrooge_amd", "-Wl,-rpath," + libdir, "-o", EXE])


def test_shim_compiles_with_gxx_and_fails_loudly_without_gpu():
    build_example()
    lib = scrooge_amd.load_library()
    if lib.scrg_device_count() > 0:
        pytest.skip("GPU present; covered by the gpu test")
    p = subprocess.run([EXE], capture_output=True, text=True)
    assert p.returncode == 2
    assert "no usable HIP device" in p.stderr


@pytest.mark.gpu
def test_shim_matches_reference_answers():
    build_example()
    p = subprocess.run([EXE], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert p.stdout.strip().splitlines() == EXPECTED


@pytest.mark.gpu
def test_shim_converts_large_results_in_parallel():
    """More than 8 MB of CIGAR text: the strings of the reference's result type are filled by several threads
    (include/scrooge_amd.hpp, detail::to_alignments); every pair against the C ABI's own arrays (tests/proto/shim_large.cpp)."""
    scrooge_amd.build_library()
    libdir = os.path.join(ROOT, "scrooge_amd")
    exe = "/tmp/scrg_shim_large"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "proto", "shim_large.cpp"), "-L" + libdir, "-lscrooge_amd", "-Wl,-rpath," + libdir,
                           "-pthread", "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    out = p.stdout.strip()
    assert out.endswith("mismatches=0") and int(out.split("text_mb=")[1].split()[0]) >= 8, out


PIPE_EXE = "/tmp/scrg_pipeline_example"


def build_pipeline_example():
    scrooge_amd.build_library()
    libdir = os.path.join(ROOT, "scrooge_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-Wall",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "pipeline_example.cpp"),
                           "-L" + libdir, "-lscrooge_amd", "-Wl,-rpath," + libdir, "-o", PIPE_EXE])


def test_pipeline_example_builds():
    """examples/pipeline_example.cpp: the device-pointer layer driven from C++ over two handles and two
    streams."""
    build_pipeline_example()


@pytest.mark.gpu
def test_pipeline_example_runs():
    build_pipeline_example()
    p = subprocess.run([PIPE_EXE], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.strip().endswith("mismatches=0")
