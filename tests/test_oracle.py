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
                                  "pairs_w200_o50.json",
                                  "pairs_w16_o0.json", "pairs_w24_o0.json", "pairs_w40_o0.json", "pairs_w64_o0.json", "pairs_w128_o0.json"])
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


def test_rows_interface_with_per_row_lengths(oracle):
    """The array interface bench.py uses (a text slot and a read slot per row, results as run arrays) with every row's own
    lengths — the mixed-length batches of bench.py's other_configs — gives what the per-pair interface gives, through the
    restatement and, where it is built, through the reference itself (oracle/ref_driver.cpp: ref_align_rows_var)."""
    import numpy as np
    from scrooge_amd import synth
    t, q = synth.make_pairs(40, 600, "ont", seed=77)
    rows = np.zeros((40, 1536), np.uint8)
    tl, rl = [], []
    for k in range(40):
        a, b = t[k][:300 + 9 * k], q[k][:200 + 10 * k]
        if k == 7:
            a, b = b"", b[:50]
        if k == 9:
            b = b""
        rows[k, :len(a)] = np.frombuffer(a, np.uint8)
        rows[k, 768:768 + len(b)] = np.frombuffer(b, np.uint8)
        tl.append(len(a))
        rl.append(len(b))
    texts = [bytes(rows[k, :tl[k]]) for k in range(40)]
    reads = [bytes(rows[k, 768:768 + rl[k]]) for k in range(40)]
    for W, O in ((64, 33), (128, 65)):
        eds, cigars, _, _ = oracle.align(texts, reads, W=W, O=O)
        e, off, runs, st, _ = oracle.align_rows(rows, 0, 768, 768, 768, W=W, O=O, threads=2, text_lens=tl, read_lens=rl)
        got = ["".join("%d%s" % (runs[j, 0], chr(runs[j, 1])) for j in range(int(off[k]), int(off[k + 1]))) for k in range(40)]
        assert list(e) == eds and got == cigars
        if Reference.available(W, O):
            e2, off2, runs2, _ = Reference(W, O).align_rows(rows, 0, 768, 768, 768, threads=2, text_lens=tl, read_lens=rl)
            assert list(e2) == eds and (off2 == off).all() and (runs2 == runs).all()
    # a length beyond its slot is an error, not an overrun
    with pytest.raises(RuntimeError):
        oracle.align_rows(rows, 0, 768, 768, 768, text_lens=[769] * 40, read_lens=rl)
