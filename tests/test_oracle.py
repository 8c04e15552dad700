"""CPU tests: the oracle restatement is pinned to the reference's answers.

(1) known edit distances from the reference's own test (src/tests.cu:246),
(2) every committed golden fixture (generated from the unmodified reference
    CPU path by tests/golden/make_golden.py),
(3) when oracle/_ref/libgenasm_ref.so is present, fresh random inputs straight
    against the reference build.
"""
import numpy as np
import pytest

from oracle.pyoracle import Reference
from scrooge_amd import synth


def test_known_edit_distances(oracle):
    ref = "AAAACCCCGGGGTTTT"
    reads = ["CCCCGGGGTTTTAAAA", "AAAACCCCGGGGTTTT", "ACCCCGG", "AAAAGGGGAAAATTTT",
             "AAAAAAAAAAAAAAAA", "ATTAACGCCTTT", "TTTTAAAACCCCGGGGTTTTAAAA", "",
             "T" * 44 + "AAAACCCCGGGGTTTTAAAA"]
    eds, cigars, _, _ = oracle.align([ref] * len(reads), reads)
    assert eds == [8, 0, 3, 8, 12, 6, 8, 0, 48]          # src/tests.cu:246
    assert cigars[8] == "31I13I16=2I2I"                   # runs are not merged across windows
    assert cigars[7] == ""


def test_oracle_matches_golden_pairs(oracle, golden_pairs):
    cases = golden_pairs["cases"]
    eds, cigars, _, _ = oracle.align([c["text"] for c in cases], [c["read"] for c in cases],
                                     W=golden_pairs["W"], O=golden_pairs["O"])
    for c, e, g in zip(cases, eds, cigars):
        assert e == c["ed"], c["group"]
        assert g == c["cigar"], c["group"]


@pytest.mark.parametrize("name", ["pairs_w32_o17.json", "pairs_w64_o2.json", "pairs_w48_o24.json", "pairs_w64_o40.json",
                                  "pairs_w128_o65.json", "pairs_w96_o49.json",
                                  "pairs_w256_o129.json", "pairs_w192_o97.json", "pairs_w128_o20.json",
                                  "pairs_w200_o50.json"])
def test_oracle_matches_golden_other_knobs(oracle, name):
    """Fixtures from the reference built with its own -DCLI_W/-DCLI_K/-DCLI_O switches."""
    from tests.conftest import load_golden
    g = load_golden(name)
    cases = g["cases"]
    eds, cigars, _, _ = oracle.align([c["text"] for c in cases], [c["read"] for c in cases], W=g["W"], O=g["O"])
    assert eds == [c["ed"] for c in cases]
    assert cigars == [c["cigar"] for c in cases]


def test_oracle_matches_golden_mapping(oracle, golden_mapping):
    g = golden_mapping
    texts, reads = [], []
    for r, cands in zip(g["reads"], g["candidates"]):
        for s in cands:
            texts.append(g["genome"][s:])      # genome suffix, genasm_cpu.cpp:512-514
            reads.append(r)
    eds, cigars, _, _ = oracle.align(texts, reads)
    assert eds == g["ed"]
    assert cigars == g["cigar"]


def test_oracle_rejects_non_acgt(oracle):
    with pytest.raises(ValueError):
        oracle.align(["ACGN"], ["ACG"])


def test_oracle_threads_agree(oracle):
    t, q = synth.make_pairs(40, 300, "ont", seed=3)
    a = oracle.align(t, q, threads=1)
    b = oracle.align(t, q, threads=4)
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]


def test_oracle_stats_shape(oracle):
    t, q = synth.make_pairs(4, 1000, "ont", seed=9)
    eds, cigars, st, _ = oracle.align(t, q)
    assert st["windows"] >= 4 * (1000 // 31)
    assert st["dc_cells"] >= st["windows"] * 2
    assert st["runs"] == sum(len([ch for ch in c if not ch.isdigit()]) for c in cigars)


@pytest.mark.skipif(not Reference.available(), reason="oracle/_ref not built (reference sources absent)")
def test_oracle_matches_reference_build_random(oracle):
    ref = Reference()
    rng = np.random.Generator(np.random.PCG64(99))
    T, Q = [], []
    for _ in range(400):
        T.append(synth.random_seq(int(rng.integers(0, 150)), rng))
        Q.append(synth.random_seq(int(rng.integers(0, 150)), rng))
    for prof, L, n in [("ont", 700, 30), ("pacbio15", 1500, 10), ("illumina", 150, 100)]:
        t, q = synth.make_pairs(n, L, prof, seed=L)
        T += t
        Q += q
    e1, c1, _, _ = oracle.align(T, Q)
    e2, c2, _ = ref.align(T, Q)
    assert e1 == e2
    assert c1 == c2
