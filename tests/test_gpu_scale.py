"""GPU parity at larger scale and on the other BASELINE.json configurations
(smaller pair counts than the bench, sized so the oracle finishes in seconds),
plus size-independent properties on every result."""
import numpy as np
import pytest

from scrooge_amd import synth
from tests.cigar_check import validate

pytestmark = pytest.mark.gpu


def _same(alns, eds, cigars):
    bad = [k for k, (a, e, c) in enumerate(zip(alns, eds, cigars)) if a.edit_distance != e or a.cigar != c]
    assert not bad, "%d/%d differ, first %d" % (len(bad), len(eds), bad[0])


def test_many_distinct_pairs_fill_every_wave(aligner, oracle):
    """> 2816*8 distinct pairs: every persistent wavefront runs all its slots and refills them
    (this is the regime where an LDS/flat ordering bug once hid from the small tests)."""
    t, q = synth.make_pairs(30000, 600, "ont", seed=77)
    eds, cigars, _, _ = oracle.align(t, q, threads=16)
    _same(aligner.align_pairs(t, q), eds, cigars)
    _same(aligner.align_pairs(t, q, lds_rows=4), eds, cigars)        # heavy HBM spill of R rows


def test_config1_illumina_150bp(aligner, oracle):
    """BASELINE configs[0]: 1k x 150 bp Illumina-like pairs."""
    t, q = synth.make_pairs(1000, 150, "illumina", seed=42)
    eds, cigars, _, _ = oracle.align(t, q)
    alns = aligner.align_pairs(t, q)
    _same(alns, eds, cigars)
    assert max(eds) <= 15


def test_config3_read_mapping_candidates(aligner, oracle):
    """BASELINE configs[2] in miniature: one synthetic chromosome, 150 bp reads, 4 candidates
    each (true locus, two shifted loci, one random locus), text = genome suffix."""
    rng = np.random.Generator(np.random.PCG64(9))
    G = 200000
    genome = synth.random_seq(G, rng)
    gcodes = np.searchsorted(synth.BASES, np.frombuffer(genome, dtype=np.uint8)).astype(np.uint8)
    reads, cands, texts_flat, reads_flat = [], [], [], []
    for _ in range(3000):
        start = int(rng.integers(0, G - 400))
        r = synth.BASES[synth.mutate(gcodes[start:start + 200], 0.01, (90, 5, 5), rng)[:150]].tobytes()
        c = [start, max(0, start - int(rng.integers(1, 4))), start + int(rng.integers(1, 4)),
             int(rng.integers(0, G - 10))]
        reads.append(r)
        cands.append(c)
        for s in c:
            texts_flat.append(genome[s:s + 400])      # enough of the suffix for a 150 bp read
            reads_flat.append(r)
    eds, cigars, _, _ = oracle.align(texts_flat, reads_flat, threads=16)
    alns = aligner.align_mapping(genome, reads, cands)
    _same(alns, eds, cigars)


def test_config5_long_noisy_reads(aligner, oracle):
    """BASELINE configs[4] in miniature: 50 kb PacBio-error reads at 15 % (multi-window traceback
    over ~1650 windows per pair, frequent window distances above the LDS rows)."""
    t, q = synth.make_pairs(24, 50000, "pacbio15", seed=50)
    eds, cigars, st, _ = oracle.align(t, q, threads=16)
    _same(aligner.align_pairs(t, q), eds, cigars)
    _same(aligner.align_pairs(t, q, lanes_per_pair=64), eds, cigars)
    assert st["windows"] / 24 > 1500


def test_properties_at_bench_shape(aligner):
    """Size-independent invariants (validateCigarString, src/tests.cu:27-169) on 10 kb ONT pairs
    without consulting the oracle."""
    t, q = synth.make_pairs(300, 10000, "ont", seed=4242)
    alns = aligner.align_pairs(t, q)
    for text, read, a in zip(t, q, alns):
        assert validate(text, read, a.cigar, a.edit_distance) is None
        assert 600 < a.edit_distance < 1500


def test_ascii_to_twobit_reference_layout(aligner):
    """Mirrors ascii_to_two_bit_correctness_test (src/tests.cu:582-647): 4 bases per byte,
    first base in bits 7..6, tail zero padded; strings incl. empty, 1, 4, 5, 32, 33 bases."""
    import torch
    strings = [b"", b"A", b"ACGT", b"ACGTA", b"ACGTACGTACGTACGTACGTACGTACGTACGT",
               b"ACGTACGTACGTACGTACGTACGTACGTACGTA", b"ttgacca" * 37]
    code = {65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}
    want = []
    for s in strings:
        out = bytearray((len(s) + 3) // 4)
        for k, ch in enumerate(s):
            out[k // 4] |= code[ch] << (6 - 2 * (k % 4))
        want.append(bytes(out))
    dev = torch.device("cuda", 0)
    a_off = np.cumsum([0] + [len(s) for s in strings])[:-1]
    t_off = np.cumsum([0] + [len(w) for w in want])[:-1]
    ascii_t = torch.tensor(list(b"".join(strings)), dtype=torch.uint8, device=dev)
    lens = torch.tensor([len(s) for s in strings], dtype=torch.int64, device=dev)
    aoff = torch.tensor(a_off, dtype=torch.int64, device=dev)
    toff = torch.tensor(t_off, dtype=torch.int64, device=dev)
    out = torch.full((sum(len(w) for w in want) + 8,), 0xEE, dtype=torch.uint8, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    aligner.set_stream(0)
    aligner.ascii_to_twobit(len(strings), lens, aoff, ascii_t, toff, out, bad)
    torch.cuda.synchronize()
    aligner.use_own_stream()
    got = bytes(out.cpu().numpy().tobytes())
    assert got[: len(b"".join(want))] == b"".join(want)
    assert got[len(b"".join(want)):] == b"\xee" * 8      # nothing written past the end
    assert int(bad.item()) == 0
