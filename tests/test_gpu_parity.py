"""GPU parity: the HIP path, called through the C ABI, against the committed
reference fixtures and against the oracle on seeded random inputs.  Bit-exact:
edit distance and CIGAR text must be identical."""
import numpy as np
import pytest

import scrooge_amd

from scrooge_amd import synth

pytestmark = pytest.mark.gpu

MAPPINGS = [1, 8, 64, 16, 4, 32]     # lanes per pair; 1 = one pair per lane (the default), 64 = one pair per wavefront


def _check(alns, eds, cigars, tag=""):
    assert len(alns) == len(eds)
    bad = [k for k, (a, e, c) in enumerate(zip(alns, eds, cigars))
           if a.edit_distance != e or a.cigar != c]
    assert not bad, "%s: %d/%d pairs differ, first %d: got (%d,%s) want (%d,%s)" % (
        tag, len(bad), len(eds), bad[0], alns[bad[0]].edit_distance, alns[bad[0]].cigar[:80],
        eds[bad[0]], cigars[bad[0]][:80])


@pytest.mark.parametrize("g", MAPPINGS)
def test_golden_pairs(aligner, golden_pairs, g):
    cases = golden_pairs["cases"]
    alns = aligner.align_pairs([c["text"] for c in cases], [c["read"] for c in cases],
                               lanes_per_pair=g)
    _check(alns, [c["ed"] for c in cases], [c["cigar"] for c in cases], "golden g=%d" % g)


@pytest.mark.parametrize("g", [1, 8, 64])
def test_golden_mapping(aligner, golden_mapping, g):
    gm = golden_mapping
    alns = aligner.align_mapping(gm["genome"], gm["reads"], gm["candidates"], lanes_per_pair=g)
    _check(alns, gm["ed"], gm["cigar"], "mapping g=%d" % g)


@pytest.mark.parametrize("name", ["pairs_w32_o17.json", "pairs_w48_o24.json", "pairs_w64_o40.json", "pairs_w64_o2.json"])
def test_golden_other_knobs(aligner, name):
    """Reference fixtures at other W/O (the reference recompiles for these; here they are runtime values)."""
    from tests.conftest import load_golden
    g = load_golden(name)
    cases = g["cases"]
    for lanes in (1, 8, 64):          # (W-O > 31: the 64-bit-row variants of both formulations)
        alns = aligner.align_pairs([c["text"] for c in cases], [c["read"] for c in cases], W=g["W"], O=g["O"],
                                   lanes_per_pair=lanes)
        _check(alns, [c["ed"] for c in cases], [c["cigar"] for c in cases], "%s g=%d" % (name, lanes))


def test_known_edit_distances(aligner):
    """src/tests.cu:224-271 (cpu/gpu_algorithm_correctness_test)"""
    ref = "AAAACCCCGGGGTTTT"
    reads = ["CCCCGGGGTTTTAAAA", "AAAACCCCGGGGTTTT", "ACCCCGG", "AAAAGGGGAAAATTTT",
             "AAAAAAAAAAAAAAAA", "ATTAACGCCTTT", "TTTTAAAACCCCGGGGTTTTAAAA", "",
             "T" * 44 + "AAAACCCCGGGGTTTTAAAA"]
    alns = aligner.align_mapping(ref, reads, [[0]] * len(reads))
    assert [a.edit_distance for a in alns] == [8, 0, 3, 8, 12, 6, 8, 0, 48]


def test_library_example(aligner):
    """src/library_example.cu:11-31"""
    alns = aligner.align_pairs(["ACGTACGT"], ["ACGTACG"])
    assert alns == [("7=", 0)]


def test_empty_batch_and_empty_reads(aligner):
    assert aligner.align_pairs([], []) == []
    alns = aligner.align_pairs(["ACGT", "", ""], ["", "", "ACG"])
    assert alns[0] == ("", 0) and alns[1] == ("", 0)
    assert alns[2] == ("3I", 3)


def test_non_acgt_is_an_error(aligner):
    import scrooge_amd
    with pytest.raises(scrooge_amd.ScroogeError) as e:
        aligner.align_pairs(["ACGTN"], ["ACGT"])
    assert e.value.status == 2


@pytest.mark.parametrize("g", MAPPINGS)
@pytest.mark.parametrize("lds_rows", [16, 3])
def test_random_vs_oracle(aligner, oracle, g, lds_rows):
    """Seeded random pairs incl. unrelated sequences (window distances up to 64, which
    exercises the HBM spill rows when lds_rows is small)."""
    rng = np.random.Generator(np.random.PCG64(1000 + g))
    T, Q = [], []
    for _ in range(300):
        T.append(synth.random_seq(int(rng.integers(0, 260)), rng))
        Q.append(synth.random_seq(int(rng.integers(0, 260)), rng))
    for prof, L, n in [("ont", 1500, 40), ("pacbio15", 3000, 12), ("illumina", 150, 200)]:
        t, q = synth.make_pairs(n, L, prof, seed=g + L)
        T += t
        Q += q
    eds, cigars, _, _ = oracle.align(T, Q, threads=8)
    alns = aligner.align_pairs(T, Q, lanes_per_pair=g, lds_rows=lds_rows)
    _check(alns, eds, cigars, "random g=%d rows=%d" % (g, lds_rows))


@pytest.mark.parametrize("sort", [0, 1])
def test_result_order_is_input_order(aligner, oracle, sort):
    t, q = [], []
    for L in [50, 900, 10, 400, 0, 77, 1300]:
        a, b = synth.make_pairs(3, max(L, 1), "ont", seed=L + 1)
        t += a
        q += [x[:L] for x in b]
    eds, cigars, _, _ = oracle.align(t, q)
    _check(aligner.align_pairs(t, q, sort_by_length=sort), eds, cigars, "order sort=%d" % sort)


@pytest.mark.parametrize("w,o", [(32, 17), (64, 40), (48, 24), (17, 3), (64, 63), (2, 1), (33, 2), (31, 1)])
def test_other_window_settings(aligner, oracle, w, o):
    """W/O are runtime parameters here (compile-time macros in the reference,
    src/genasm_cpu.cpp:22-35); the short-read setting of the paper is W=32, O=17."""
    t, q = synth.make_pairs(60, 200, "ont", seed=w * 100 + o)
    rng = np.random.Generator(np.random.PCG64(w))
    for _ in range(60):
        t.append(synth.random_seq(int(rng.integers(0, 120)), rng))
        q.append(synth.random_seq(int(rng.integers(0, 120)), rng))
    eds, cigars, _, _ = oracle.align(t, q, W=w, O=o)
    _check(aligner.align_pairs(t, q, W=w, O=o), eds, cigars, "W=%d O=%d" % (w, o))
    _check(aligner.align_pairs(t, q, W=w, O=o, lanes_per_pair=8), eds, cigars, "W=%d O=%d g8" % (w, o))
    _check(aligner.align_pairs(t, q, W=w, O=o, lanes_per_pair=64), eds, cigars, "W=%d O=%d g64" % (w, o))


@pytest.mark.parametrize("w,o", [(64, 2), (64, 20), (64, 32), (40, 5), (33, 1), (64, 1)])
@pytest.mark.parametrize("g", [1, 8, 64, 16])
def test_small_overlap_uses_wide_storage(aligner, oracle, w, o, g):
    """W-O > 31 (e.g. the reference's O sweep down to small overlaps, scripts/profile.py:88-100; README's
    W=64,O=2 row): the traceback may consume up to W-O characters per window, so whole 64-bit entries of
    all columns are stored (kernel variant WIDE; one pair per lane: genasm_lane_mw_kernel, 64-bit rows, table in HBM)."""
    t, q = synth.make_pairs(50, 400, "ont", seed=w * 31 + o)
    rng = np.random.Generator(np.random.PCG64(w + o))
    for _ in range(80):
        t.append(synth.random_seq(int(rng.integers(0, 200)), rng))
        q.append(synth.random_seq(int(rng.integers(0, 200)), rng))
    eds, cigars, _, _ = oracle.align(t, q, W=w, O=o)
    _check(aligner.align_pairs(t, q, W=w, O=o, lanes_per_pair=g), eds, cigars, "W=%d O=%d g=%d" % (w, o, g))
    _check(aligner.align_pairs(t, q, W=w, O=o, lanes_per_pair=g, lds_rows=3), eds, cigars, "W=%d O=%d g=%d spill" % (w, o, g))


def test_long_reads_10kb(aligner, oracle):
    t, q = synth.make_pairs(64, 10000, "ont", seed=42)
    eds, cigars, st, _ = oracle.align(t, q, threads=8)
    _check(aligner.align_pairs(t, q), eds, cigars, "10kb")
    _check(aligner.align_pairs(t, q, lanes_per_pair=8), eds, cigars, "10kb g8")
    _check(aligner.align_pairs(t, q, lanes_per_pair=64), eds, cigars, "10kb g64")


@pytest.mark.parametrize("name", ["pairs_w16_o0.json", "pairs_w24_o0.json", "pairs_w40_o0.json", "pairs_w64_o0.json", "pairs_w128_o0.json"])
def test_golden_no_overlap(aligner, oracle, name):
    """O = 0 (the reference's special case src/genasm_cpu.cpp:104-110; its O sweep reaches it for W < 32, scripts/profile.py:92-93):
    fixtures from the reference built with -DCLI_O=0.  W <= 31: the default kernel; 32..63: the two-halves kernel; 64, 128: the
    kernel with the table in HBM (the stop bit is row W).  Runs and edit streams; 256/0 and a mixed batch against the oracle; the
    GenASM-row mappings refuse O = 0."""
    import scrooge_amd
    from tests.conftest import load_golden
    g = load_golden(name)
    cases = g["cases"]
    T, Q = [c["text"] for c in cases], [c["read"] for c in cases]
    alns = aligner.align_pairs(T, Q, W=g["W"], O=0)
    _check(alns, [c["ed"] for c in cases], [c["cigar"] for c in cases], name)
    with pytest.raises(scrooge_amd.ScroogeError):
        aligner.align_pairs(T[:4], Q[:4], W=g["W"], O=0, lanes_per_pair=64 if g["W"] > 64 else 8)
    if name == "pairs_w128_o0.json":
        t, q = synth.make_pairs(60, 1500, "ont", seed=5)
        for W in (256, 31, 63, 100):
            eds, cigars, _, _ = oracle.align(t, q, W=W, O=0, threads=8)
            _check(aligner.align_pairs(t, q, W=W, O=0), eds, cigars, "W=%d O=0" % W)


@pytest.mark.parametrize("name", ["pairs_w128_o65.json", "pairs_w96_o49.json", "pairs_w256_o129.json",
                                  "pairs_w192_o97.json", "pairs_w128_o20.json", "pairs_w200_o50.json"])
@pytest.mark.parametrize("g", [1, 32, 64])
def test_golden_windows_over_64(aligner, name, g):
    """Reference fixtures built with -DCLI_W=96 ... 256 (the reference's bitvector<N> path,
    src/bitvector.hpp:45-48): one pair per lane with multi-word difference vectors (genasm_lane_mw_kernel.hip, the
    default) and the GenASM-row kernel with multi-word entries (genasm_kernel_multiword.hip; lds_rows: its spill path)."""
    from tests.conftest import load_golden
    gd = load_golden(name)
    cases = gd["cases"]
    alns = aligner.align_pairs([c["text"] for c in cases], [c["read"] for c in cases], W=gd["W"], O=gd["O"],
                               lanes_per_pair=g)
    _check(alns, [c["ed"] for c in cases], [c["cigar"] for c in cases], "%s g=%d" % (name, g))
    alns = aligner.align_pairs([c["text"] for c in cases], [c["read"] for c in cases], W=gd["W"], O=gd["O"],
                               lanes_per_pair=g, lds_rows=5)
    _check(alns, [c["ed"] for c in cases], [c["cigar"] for c in cases], "%s g=%d spill" % (name, g))


@pytest.mark.parametrize("w,o", [(128, 65), (96, 49), (100, 40), (65, 2), (128, 127), (127, 64), (80, 41),
                                 (256, 129), (256, 1), (129, 65), (192, 64), (255, 200), (160, 81), (130, 2), (200, 9)])
@pytest.mark.parametrize("g", [1, 32, 64])
def test_windows_over_64_vs_oracle(aligner, oracle, w, o, g):
    t, q = synth.make_pairs(40, 900, "ont", seed=w * 7 + o)
    a, b = synth.make_pairs(10, 2500, "pacbio15", seed=w + o)
    t, q = t + a, q + b
    rng = np.random.Generator(np.random.PCG64(w * 3 + o))
    for _ in range(100):
        t.append(synth.random_seq(int(rng.integers(0, 600)), rng))
        q.append(synth.random_seq(int(rng.integers(0, 600)), rng))
    eds, cigars, _, _ = oracle.align(t, q, W=w, O=o, threads=8)
    _check(aligner.align_pairs(t, q, W=w, O=o, lanes_per_pair=g), eds, cigars, "W=%d O=%d g=%d" % (w, o, g))
    _check(aligner.align_pairs(t, q, W=w, O=o, lanes_per_pair=g, lds_rows=4), eds, cigars,
           "W=%d O=%d g=%d spill" % (w, o, g))


@pytest.mark.parametrize("w,o", [(64, 2), (64, 1), (64, 16), (64, 32), (63, 20), (40, 5), (33, 1), (128, 65), (96, 49), (80, 41),
                                 (112, 57), (128, 96), (100, 40), (65, 2), (127, 64)])
def test_table_in_two_halves(aligner, aligner_select, oracle, w, o):
    """32 <= W-O <= 63, W <= 128 (the reference's small-overlap and W > 64 sweep points, scripts/profile.py:88-100,
    180-185): genasm_lane_wide_kernel builds the window's table in two halves of 32 columns in registers.  Runs that
    cross from the first half into the second are one run (long matches on low-error reads, long gaps), walks that end
    in the first half never enter the second, texts that end inside a window take the short-window variant.  The
    kernel it replaces for these W/O (table in HBM: reserved[0] = 256 in the test build of the library) gives the same results."""
    t, q = synth.make_pairs(150, 2000, "ont", seed=w * 13 + o)
    a, b = synth.make_pairs(40, 2500, "pacbio15", seed=w + o + 1)
    c, d = synth.make_pairs(300, 300, "illumina", seed=w + o + 2)       # long match runs across the halves
    t, q = t + a + c, q + b + d
    rng = np.random.Generator(np.random.PCG64(w * 5 + o))
    for _ in range(100):                       # unrelated sequences, ragged and empty inputs
        t.append(synth.random_seq(int(rng.integers(0, 500)), rng))
        q.append(synth.random_seq(int(rng.integers(0, 500)), rng))
    for _ in range(40):                        # low-complexity sequences: long insertion / deletion runs, many ties
        t.append(bytes(rng.choice(np.frombuffer(b"AC", np.uint8), int(rng.integers(1, 400)))))
        q.append(bytes(rng.choice(np.frombuffer(b"AC", np.uint8), int(rng.integers(1, 400)))))
    for _ in range(40):                        # one long gap
        s = synth.random_seq(int(rng.integers(200, 700)), rng)
        cut, gap = int(rng.integers(10, 150)), int(rng.integers(10, 60))
        t += [s, s[:cut] + s[cut + gap:]]
        q += [s[:cut] + s[cut + gap:], s]
    t += [b"", b"ACGT", b"A" * 300, b"A" * 10, b"ACGT" * 100]
    q += [b"ACGT", b"", b"A" * 10, b"A" * 300, b"TGCA" * 100]
    eds, cigars, _, _ = oracle.align(t, q, W=w, O=o, threads=8)
    _check(aligner.align_pairs(t, q, W=w, O=o), eds, cigars, "W=%d O=%d" % (w, o))
    # (the kernel selection switch exists in the test build only: ab_libs/lib_select.so, conftest.aligner_select)
    p = aligner_select.make_params(W=w, O=o)
    p.reserved[0] = 256
    keep = aligner_select.params
    aligner_select.params = p
    try:
        _check(aligner_select.align_pairs(t, q), eds, cigars, "W=%d O=%d, table in HBM" % (w, o))
    finally:
        aligner_select.params = keep


@pytest.mark.parametrize("w,o", [(256, 129), (160, 81), (192, 97), (224, 113), (128, 20), (128, 1), (129, 65), (130, 2), (200, 100), (255, 128),
                                 (191, 64), (192, 128), (65, 1), (256, 192), (161, 81),
                                 # W > 128 with W-O <= 63 (the reference's O sweeps at --override_W, scripts/profile.py:88-100): 1 to 4 parts
                                 (256, 200), (192, 150), (130, 100), (256, 255), (160, 144), (129, 66), (256, 224)])
def test_table_in_parts(aligner, aligner_select, oracle, w, o):
    """64 <= W-O <= 127 (the reference's large-window sweep points, scripts/profile.py:180-185: W = 160 ... 256 with
    O = W/2 + 1; two-word table rows, src/bitvector.hpp:45-48): genasm_lane_parts_kernel builds the window's table in parts of
    16 columns in registers, each part re-swept from a checkpoint of the difference vectors.  Runs that cross from one part into
    the next are one run (long matches, long gaps), walks that end early never enter the later parts, texts that end inside a
    window skip the chunks past their end, insertion runs of more than 64 rows (unrelated sequences) cross the words of a row.
    Vectors of 2, 3 and 4 words.  The kernel it replaces for these W/O (table in HBM: reserved[0] = 256 in the test build of the library) gives the same results."""
    t, q = synth.make_pairs(120, 3000, "ont", seed=w * 13 + o)
    a, b = synth.make_pairs(40, 3500, "pacbio15", seed=w + o + 1)
    c, d = synth.make_pairs(200, 600, "illumina", seed=w + o + 2)       # long match runs across the parts
    t, q = t + a + c, q + b + d
    rng = np.random.Generator(np.random.PCG64(w * 5 + o))
    for _ in range(100):                       # unrelated sequences, ragged and empty inputs
        t.append(synth.random_seq(int(rng.integers(0, 900)), rng))
        q.append(synth.random_seq(int(rng.integers(0, 900)), rng))
    for _ in range(40):                        # low-complexity sequences: long insertion / deletion runs, many ties
        t.append(bytes(rng.choice(np.frombuffer(b"AC", np.uint8), int(rng.integers(1, 700)))))
        q.append(bytes(rng.choice(np.frombuffer(b"AC", np.uint8), int(rng.integers(1, 700)))))
    for _ in range(40):                        # one long gap
        s = synth.random_seq(int(rng.integers(300, 1200)), rng)
        cut, gap = int(rng.integers(10, 250)), int(rng.integers(10, 120))
        t += [s, s[:cut] + s[cut + gap:]]
        q += [s[:cut] + s[cut + gap:], s]
    for _ in range(20):                        # reads much longer than their texts: whole windows of insertions
        t.append(synth.random_seq(int(rng.integers(0, 60)), rng))
        q.append(synth.random_seq(int(rng.integers(300, 800)), rng))
    t += [b"", b"ACGT", b"A" * 600, b"A" * 10, b"ACGT" * 200]
    q += [b"ACGT", b"", b"A" * 10, b"A" * 600, b"TGCA" * 200]
    eds, cigars, _, _ = oracle.align(t, q, W=w, O=o, threads=8)
    _check(aligner.align_pairs(t, q, W=w, O=o), eds, cigars, "W=%d O=%d" % (w, o))
    # (the kernel selection switch exists in the test build only: ab_libs/lib_select.so, conftest.aligner_select)
    p = aligner_select.make_params(W=w, O=o)
    p.reserved[0] = 256
    keep = aligner_select.params
    aligner_select.params = p
    try:
        _check(aligner_select.align_pairs(t, q), eds, cigars, "W=%d O=%d, table in HBM" % (w, o))
    finally:
        aligner_select.params = keep


@pytest.mark.parametrize("w,o", [(64, 33), (64, 40), (64, 60), (33, 2), (32, 17), (2, 1), (48, 24), (17, 9), (63, 32)])
def test_one_and_two_wavefronts_per_window_agree(aligner, aligner_select, oracle, w, o):
    """The default table (W <= 64, W-O <= 31) exists as one wavefront per 64 pairs (genasm_lane_kernel) and with a window's
    work split over a producer and a consumer wavefront (genasm_lane_split_kernel: what a launch that cannot fill the SIMDs
    takes by default, i.e. every small batch in this test suite).  In the test build reserved[0] = 512 / 1024 force one or the other: both must
    give the CPU checker's results — long and short reads, unrelated and low-complexity sequences, ragged and empty inputs
    (a pair of no windows is handed over as first and last at once), more pairs than one wavefront holds (lanes refill from
    the queue while their neighbours are in the middle of a pair)."""
    t, q = synth.make_pairs(200, 1500, "ont", seed=w * 11 + o)
    a, b = synth.make_pairs(60, 2500, "pacbio15", seed=w + o + 5)
    c, d = synth.make_pairs(300, 200, "illumina", seed=w + o + 6)
    t, q = t + a + c, q + b + d
    rng = np.random.Generator(np.random.PCG64(w * 7 + o))
    for _ in range(150):
        t.append(synth.random_seq(int(rng.integers(0, 500)), rng))
        q.append(synth.random_seq(int(rng.integers(0, 500)), rng))
    for _ in range(40):
        t.append(bytes(rng.choice(np.frombuffer(b"AC", np.uint8), int(rng.integers(1, 400)))))
        q.append(bytes(rng.choice(np.frombuffer(b"AC", np.uint8), int(rng.integers(1, 400)))))
    t += [b"", b"ACGT", b"A" * 300, b"A" * 10, b"ACGT" * 100, b"", b"ACGT"] + [b"ACGT"] * 70
    q += [b"ACGT", b"", b"A" * 10, b"A" * 300, b"TGCA" * 100, b"", b"A"] + [b""] * 70
    eds, cigars, _, _ = oracle.align(t, q, W=w, O=o, threads=8)
    for flags, name in ((512, "two wavefronts per window"), (1024, "one wavefront"), (0, "default")):
        al = aligner_select if flags else aligner          # (the switches exist in the test build only; the default is the shipped library's)
        p = al.make_params(W=w, O=o)
        p.reserved[0] = flags
        keep = al.params
        al.params = p
        try:
            _check(al.align_pairs(t, q), eds, cigars, "W=%d O=%d, %s" % (w, o, name))
            _check(al.align_pairs(t, q, sort_by_length=0), eds, cigars, "W=%d O=%d, %s, caller order" % (w, o, name))
        finally:
            al.params = keep


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_window_settings_drawn_at_random(aligner, oracle, seed):
    """The whole (W, O) plane, not the hand-picked points of the tests above: 40 settings per seed with 2 <= W <= 256 and 0 <= O < W
    (runtime parameters here; compile-time macros in the reference, src/genasm_cpu.cpp:22-35, 104-110), a third of them next to the
    borders between the kernels (W-O = 31/32, 63/64, 127/128; W = 64/65, 128/129), on related, unrelated, low-complexity, ragged and
    empty pairs — edit distance and CIGAR of the one-pair-per-lane kernels and of two GenASM-row mappings against the oracle."""
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    settings = []
    while len(settings) < 40:
        if len(settings) % 3 == 0:
            tb = int(rng.choice([31, 32, 33, 63, 64, 65, 127, 128, 129, 1, 2]))       # W - O
            W = int(rng.choice([64, 65, 128, 129, 256, 63, 127, 255, int(rng.integers(2, 257))]))
            O = W - tb
        else:
            W = int(rng.integers(2, 257))
            O = int(rng.integers(0, W))
        if 2 <= W <= 256 and 0 <= O < W and (W, O) not in settings:
            settings.append((W, O))
    t, q = synth.make_pairs(24, 700, "ont", seed=seed)
    t2, q2 = synth.make_pairs(12, 500, "pacbio15", seed=seed + 50)
    t, q = t + t2, q + q2
    for _ in range(16):
        t.append(synth.random_seq(int(rng.integers(0, 300)), rng))
        q.append(synth.random_seq(int(rng.integers(0, 300)), rng))
    for _ in range(8):
        t.append(bytes(rng.choice(np.frombuffer(b"AC", np.uint8), int(rng.integers(1, 400)))))
        q.append(bytes(rng.choice(np.frombuffer(b"AC", np.uint8), int(rng.integers(1, 400)))))
    t += [b"", b"ACGT", b"A" * 300]
    q += [b"ACGT", b"", b"A" * 290]
    for W, O in settings:
        eds, cigars, _, _ = oracle.align(t, q, W=W, O=O, threads=8)
        _check(aligner.align_pairs(t, q, W=W, O=O), eds, cigars, "W=%d O=%d" % (W, O))
        if O >= 1:                                   # the GenASM-row mappings (they refuse O = 0: their traceback reads R[i + 1])
            for g in ((8, 64) if W <= 64 else (32, 64)):
                _check(aligner.align_pairs(t, q, W=W, O=O, lanes_per_pair=g), eds, cigars, "W=%d O=%d g%d" % (W, O, g))


def test_windows_over_64_limits(aligner):
    import scrooge_amd
    with pytest.raises(scrooge_amd.ScroogeError):
        aligner.align_pairs(["ACGT"], ["ACGT"], W=128, O=65, lanes_per_pair=8)   # one pair per lane, or slots of 32 or 64 lanes
    with pytest.raises(scrooge_amd.ScroogeError):
        aligner.align_pairs(["ACGT"], ["ACGT"], W=257, O=129)
    assert aligner.align_pairs(["ACGT"], ["ACGT"], W=256, O=129) == [("4=", 0)]


@pytest.mark.parametrize("lds_rows", [13, 16, 12, 6])
def test_diagonal_path_on_and_off(aligner, aligner_select, oracle, lds_rows):
    """Full W=64 windows with a small distance take the diagonal-major path of the G=8 kernel (carry-chain
    rows, clz traceback), everything else the column-major path; rounds of both kinds interleave inside a
    pair and share the CIGAR staging ring.  Same results with the path on, off (reserved[0] = 32, test build) and with
    row budgets that change how many rows it may use (lds_rows >= 13: 16 compacted rows, else lds_rows)."""
    T, Q = synth.make_pairs(300, 3000, "ont", seed=5)
    a, b = synth.make_pairs(100, 3000, "pacbio15", seed=6)       # many windows beyond 15 edits: fall back mid-pair
    c, d = synth.make_pairs(400, 500, "illumina", seed=7)
    T, Q = T + a + c, Q + b + d
    eds, cigars, _, _ = oracle.align(T, Q, threads=8)
    _check(aligner.align_pairs(T, Q, lanes_per_pair=8, lds_rows=lds_rows), eds, cigars, "diag on rows=%d" % lds_rows)
    p = aligner_select.make_params(lanes_per_pair=8, lds_rows=lds_rows)
    p.reserved[0] = 32
    keep = aligner_select.params
    aligner_select.params = p
    try:
        _check(aligner_select.align_pairs(T, Q), eds, cigars, "diag off rows=%d" % lds_rows)
    finally:
        aligner_select.params = keep


def test_array_results_match_object_results(aligner):
    """align_pairs(arrays=True) returns the same edit distances and CIGAR text as the per-pair objects."""
    t, q = synth.make_pairs(200, 700, "ont", seed=12)
    t += [b"ACGT", b""]
    q += [b"", b"ACG"]
    objs = aligner.align_pairs(t, q)
    arr = aligner.align_pairs(t, q, arrays=True)
    assert arr["edit_distance"].tolist() == [a.edit_distance for a in objs]
    off = arr["cigar_offset"]
    for i, a in enumerate(objs):
        assert arr["cigar_text"][int(off[i]):int(off[i + 1]) - 1].decode() == a.cigar
    ro = arr["run_offset"]
    for i in (0, 7, 199):
        seg = arr["runs"][int(ro[i]):int(ro[i + 1])]
        assert "".join("%d%s" % (c, chr(o)) for c, o in seg) == objs[i].cigar
    assert aligner.align_pairs([], [], arrays=True)["edit_distance"].shape == (0,)


def test_resident_genome(aligner, golden_mapping, oracle):
    """scrg_genome_set + scrg_align_mapping_resident: the genome is staged and packed once; batches of different
    sizes (the sequence array grows in between: the genome moves device-to-device), the same results as
    scrg_align_mapping; a call that brings its own sequences evicts it and the next resident call fails."""
    import scrooge_amd
    gm = golden_mapping
    aligner.set_genome(gm["genome"])
    try:
        small = aligner.align_mapping(None, gm["reads"][:3], gm["candidates"][:3])
        n_small = sum(len(c) for c in gm["candidates"][:3])
        _check(small, gm["ed"][:n_small], gm["cigar"][:n_small], "resident, small batch")
        _check(aligner.align_mapping(None, gm["reads"], gm["candidates"]), gm["ed"], gm["cigar"], "resident, full batch")
        # a much larger batch against the same resident genome
        rng = np.random.Generator(np.random.PCG64(3))
        G = len(gm["genome"])
        reads, cands, texts, qs = [], [], [], []
        for _ in range(3000):
            s = int(rng.integers(0, max(1, G - 200)))
            r = gm["genome"][s:s + 150]
            r = r if isinstance(r, bytes) else r.encode()
            reads.append(r)
            cands.append([s, max(0, s - 2)])
            for c in cands[-1]:
                g = gm["genome"][c:c + 400]
                texts.append(g if isinstance(g, bytes) else g.encode())
                qs.append(r)
        eds, cigars, _, _ = oracle.align(texts, qs, threads=8)
        _check(aligner.align_mapping(None, reads, cands), eds, cigars, "resident, large batch")
        _check(aligner.align_mapping(None, gm["reads"], gm["candidates"]), gm["ed"], gm["cigar"], "resident, again")
        # a pairwise call in between leaves the genome where it is (it sits in front of the chunks' sequence regions)
        assert aligner.align_pairs(["ACGT"], ["ACGT"]) == [("4=", 0)]
        big_t, big_q = synth.make_pairs(300, 4000, "ont", seed=31)          # large enough to make the sequence array grow
        eds2, cig2, _, _ = oracle.align(big_t, big_q, threads=8)
        _check(aligner.align_pairs(big_t, big_q), eds2, cig2, "pairs between resident batches")
        _check(aligner.align_mapping(None, gm["reads"], gm["candidates"]), gm["ed"], gm["cigar"], "resident, after pairwise calls")
        aligner.clear_genome()
        with pytest.raises(scrooge_amd.ScroogeError):
            aligner.align_mapping(None, gm["reads"], gm["candidates"])
    finally:
        aligner.clear_genome()
    _check(aligner.align_mapping(gm["genome"], gm["reads"], gm["candidates"]), gm["ed"], gm["cigar"], "after clear")


def test_dense_run_patterns(aligner, oracle):
    """Windows with a run boundary at (almost) every column — every other base substituted, inserted or deleted —
    fill the CIGAR staging ring as fast as it can be filled (up to ~60 runs per window): ring wrap, piece flushes
    inside the emission loop and slices that overflow are all on this path."""
    rng = np.random.Generator(np.random.PCG64(5))
    sub = {65: 67, 67: 71, 71: 84, 84: 65}
    T, Q = [], []
    for L in (64, 200, 1000, 5000):
        for _ in range(6):
            t = synth.random_seq(L + 40, rng)
            q = bytearray(t[:L])
            for k in range(int(rng.integers(0, 2)), L, 2):          # X = X = ...
                q[k] = sub[q[k]]
            T.append(t), Q.append(bytes(q))
            q2 = bytearray()
            for k in range(L):                                       # an insertion after every other base
                q2.append(t[k])
                if k % 2:
                    q2.append(sub[t[k]])
            T.append(t), Q.append(bytes(q2))
            T.append(t), Q.append(bytes(t[k] for k in range(L) if k % 3))     # a deletion every third base
            mix = bytearray()
            for k in range(L):                                       # substitutions, insertions and deletions interleaved
                r = k % 6
                if r == 1:
                    mix.append(sub[t[k]])
                elif r == 3:
                    mix += bytes([t[k], sub[t[k]]])
                elif r != 5:
                    mix.append(t[k])
            T.append(t), Q.append(bytes(mix))
    eds, cigars, _, _ = oracle.align(T, Q, threads=8)
    assert max(c.count("X") + c.count("I") + c.count("D") for c in cigars) > 1500
    for g in (1, 8):
        _check(aligner.align_pairs(T, Q, lanes_per_pair=g), eds, cigars, "dense runs g=%d" % g)
    e32, c32, _, _ = oracle.align(T, Q, W=32, O=17, threads=8)
    _check(aligner.align_pairs(T, Q, W=32, O=17), e32, c32, "dense runs W=32")


def test_multi_device_entry_points(aligner, golden_pairs, golden_mapping, oracle):
    """scrg_align_pairs_multi / scrg_align_mapping_multi (one call, several GPUs, one host thread pair per device): the one
    GPU of the test box listed four times — four device states, sixteen streams, chunks dealt round-robin — must give
    the results of the single-device call, in caller order: goldens, ragged batches against the oracle, reverse strand."""
    g = golden_pairs
    T = [c["text"] for c in g["cases"]]
    Q = [c["read"] for c in g["cases"]]
    for devs in ([0], [0, 0, 0, 0]):
        _check(aligner.align_pairs_multi(devs, T, Q), [c["ed"] for c in g["cases"]], [c["cigar"] for c in g["cases"]], "golden pairs %s" % devs)
        gm = golden_mapping
        _check(aligner.align_mapping_multi(devs, gm["genome"], gm["reads"], gm["candidates"]), gm["ed"], gm["cigar"], "golden mapping %s" % devs)
    # a ragged batch in no particular order: 5000 pairs of 0..3 kb -> many chunks on each state, results permuted back
    rng = np.random.Generator(np.random.PCG64(77))
    T, Q = [], []
    for L in rng.integers(0, 3000, 5000):
        t, q = synth.make_pair(int(L), 0.1, (23, 31, 46), rng, 0.15) if L else (np.zeros(3, np.uint8), np.zeros(0, np.uint8))
        T.append(synth.BASES[t].tobytes())
        Q.append(synth.BASES[q].tobytes())
    eds, cigars, _, _ = oracle.align(T, Q, threads=8)
    _check(aligner.align_pairs_multi([0, 0, 0, 0], T, Q), eds, cigars, "ragged, four states")
    _check(aligner.align_pairs_multi([0, 0], T, Q, sort_by_length=0), eds, cigars, "ragged, unsorted issue order")
    _check(aligner.align_pairs(T, Q), eds, cigars, "ragged, one handle")
    # reverse-strand candidates through the multi entry point
    genome = synth.random_seq(30000, rng)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads, cands, revs, texts, qs = [], [], [], [], []
    for k in range(300):
        s0 = int(rng.integers(0, 29000))
        r = genome[s0:s0 + int(rng.integers(40, 400))]
        rev = k % 3 == 0
        reads.append(r.translate(comp)[::-1] if rev else r)
        cands.append([s0, max(0, s0 - 2)])
        revs.append([1 if rev else 0] * 2)
        for st in cands[-1]:
            texts.append(genome[st:st + 600])
            qs.append(r)
    eds, cigars, _, _ = oracle.align(texts, qs, threads=8)
    _check(aligner.align_mapping_multi([0, 0, 0], genome, reads, cands, reverse=revs), eds, cigars, "stranded mapping, three states")
    with pytest.raises(scrooge_amd.ScroogeError):
        aligner.align_pairs_multi([0, 99], T[:10], Q[:10])          # no such device: an error, not a smaller job


def test_output_selection(aligner, oracle):
    """scrg_params.outputs: text only / runs only leave the other array empty (it never crosses PCIe); the part that is
    asked for is identical to the full result."""
    T, Q = synth.make_pairs(700, 1500, "ont", seed=12)
    Q[3] = b""
    full = aligner.align_pairs(T, Q, arrays=True)
    text = aligner.align_pairs(T, Q, arrays=True, outputs=1)
    runs = aligner.align_pairs(T, Q, arrays=True, outputs=2)
    assert (text["edit_distance"] == full["edit_distance"]).all() and text["cigar_text"] == full["cigar_text"]
    assert (text["cigar_offset"] == full["cigar_offset"]).all() and text["runs"].shape[0] == 0 and int(text["run_offset"].max()) == 0
    assert (runs["edit_distance"] == full["edit_distance"]).all() and (runs["runs"] == full["runs"]).all()
    assert (runs["run_offset"] == full["run_offset"]).all() and int(runs["cigar_offset"].max()) == 0
    with pytest.raises(scrooge_amd.ScroogeError):
        aligner.align_pairs(T[:4], Q[:4], outputs=3)
