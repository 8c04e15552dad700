import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_pairs():
    return load_golden("pairs_w64_o33.json")


@pytest.fixture(scope="session")
def golden_mapping():
    return load_golden("mapping_w64_o33.json")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def aligner():
    """The HIP path.  Fails (does not skip, does not fall back) if the library or GPU is missing."""
    import scrooge_amd
    scrooge_amd.build_library()
    a = scrooge_amd.Aligner(0)
    yield a
    a.close()


@pytest.fixture(scope="session")
def aligner_select():
    """The TEST build of the same sources (-DSCRG_SELECT, ab_libs/lib_select.so: scrg_params.reserved[0] selects between
    formulations that give identical results; the shipped library accepts no switch) next to the shipped one — for the parity
    tests that compare formulations.  Built here if the snapshot did not bring it."""
    import scrooge_amd
    scrooge_amd.build_library(variant="select")
    a = scrooge_amd.Aligner(0, variant="select")
    assert a.lib.scrg_build_flags() == 4            # SCRG_BUILD_SELECT and nothing else: no counters, no ablations
    yield a
    a.close()
