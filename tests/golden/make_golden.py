#!/usr/bin/env python3
"""Regenerate tests/golden/*.json from the UNMODIFIED reference CPU path.

Runs only in the build container (needs oracle/_ref/libgenasm_ref.so, built by
`make -C oracle` from /root/reference/src/genasm_cpu.cpp).  The fixtures are
data: input sequences and the (edit distance, CIGAR) the reference returned at
its default knobs W=64, K=64, O=33 (src/genasm_cpu.cpp:7-9).

Known-answer inputs come from the reference's own tests:
  src/tests.cu:236-246 (reference AAAACCCCGGGGTTTT, 9 reads, expected EDs),
  src/tests.cu:276-283 (8 library-interface pairs),
  src/library_example.cu:12-13.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Reference  # noqa: E402
from scrooge_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
        f.write("\n")


def main():
    ref = Reference()
    cases = []

    def add(group, texts, reads):
        eds, cigs, _ = ref.align(texts, reads)
        for t, q, e, c in zip(texts, reads, eds, cigs):
            t = t.decode() if isinstance(t, bytes) else t
            q = q.decode() if isinstance(q, bytes) else q
            cases.append({"group": group, "text": t, "read": q, "ed": e, "cigar": c})

    # --- the reference's own known-answer inputs -------------------------
    g = "AAAACCCCGGGGTTTT"
    reads = ["CCCCGGGGTTTTAAAA", "AAAACCCCGGGGTTTT", "ACCCCGG", "AAAAGGGGAAAATTTT",
             "AAAAAAAAAAAAAAAA", "ATTAACGCCTTT", "TTTTAAAACCCCGGGGTTTTAAAA", "",
             "T" * 44 + "AAAACCCCGGGGTTTTAAAA"]
    add("tests_cu_known_ed", [g] * len(reads), reads)
    assert [c["ed"] for c in cases] == [8, 0, 3, 8, 12, 6, 8, 0, 48]  # src/tests.cu:246
    add("library_example", ["ACGTACGT"], ["ACGTACG"])
    lib_pairs = [  # (query, text), src/tests.cu:276-283
        ("ACGT", "ACGT"),
        ("CAAATCTATTAAGTCAAACGGTCCGTAAGCTAGAACCTCCTGCCGTGTAAGTTACGACGTGGTCGAGTTACTTTCGTTCTTATTAACACAATGTCCATCA", "CAAACCTATCAAGTCAAACGGTCCGTAGCTACACCTCCTGCCGTGTAAAGTTACGACGTGGTTGAGTTACTTTCGTTCTTATTAACAACAATGTTCCATCA"),
        ("CGGCGAAGGAATTAATTACAAGCCTTGTACACTTGCATATTCTTCTGCAACAGGGCCCCGGCTCCGTCCTACCTCGGTTTACTGTGACTCACTTGAGCGA", "CGGCGAAGGAATAATTACAAGCCTGTATCACTTGCATATTCGTTCTGCAACAGGCCCGGCTCCGTCTACGCTGGTTTACTGTGACTCACTTGAGCGA"),
        ("ACAGTGGAAATGTCGCGGAAGGGTAGCAGTAGAACTTAATCAGAGAGATTACCTCGCGTAGTTGAAGTCTTGACGGGCGCATTGGACATAACAAACATAC", "ACGTGGACATGTCGCGGAAGGATAGCAGTAGAACTTAATCAGAGAATTACCTCGCGTAGTTGAACTCTTGACGGCGCGATGTGGACCTAACAAACATAC"),
        ("AACCCACGGTCTTCTCTGGTTTCGAACTTACAATCGTGAGCCCATCCGTACTTTCATGTTTCTTAAGATGGCAAGACAGAAATATAATTAGGCCGGGAGC", "AACCCACGGTCTTCTCTGGTTTCGAATTAGCAATCGTCGAGCCGCATCCGTACTTTCATGTTTCCTTAAGATGGCCAGAACAGAAATAATTAGGCCGGGAGC"),
        ("TTTGCTTAGCCGAGCTATGCGGAACTAGAGCACCGGAGGTTTGTGTGGTCACTAGAATGACAAGGTCTCTGATCAGATATAACTCTTCGGGTTTGCGTAA", "TTTGCTTAGCCGAGCTATCCCGGAACAGACACCGGAGGTTTGAGTGGTCACTAGAATGACAAGGTATCTGATCAGATACAACTTCTTCGGGCTTTGCGTAA"),
        ("GATGTACAGTCTCGAAAACCAAGTCTAGGACCAATTCCAACCTTATAATCCAGATTTACCATTATGACAACCGCAGAAGAGAAACTAATCGTCCAAAAGA", "GATGTGCAGTCTCGAAAACCAAGTCTAGGACCAGATTCCAACCTTTTAACCCAGAGTTACCAGAGACAACCGCAGAAGAGAAACTAATCGTCCAAAAGA"),
        ("TCCTGCGCGCGAAGGGGACATTGCAGGGCAAAGCAATGGCTAGATAGCCTCATACTGAGACGATAAATGGCGTTGGACACCGGAGAAAAGACCCCGCCGA", "TCTGCGCGCGAAGGGGACATAGCAGGCAAAGCAATGGCTAGATAGCCTCATACTGAGAGATAAATGGCGTTGGCCACCGGAGCAAAAGACCCCGCCG"),
    ]
    add("tests_cu_library_interface", [p[1] for p in lib_pairs], [p[0] for p in lib_pairs])

    # --- seeded random coverage --------------------------------------------
    rng = np.random.Generator(np.random.PCG64(20261002))
    T, Q = [], []
    for _ in range(150):  # unrelated sequences: large window distances, ragged ends
        T.append(synth.random_seq(int(rng.integers(0, 200)), rng))
        Q.append(synth.random_seq(int(rng.integers(0, 200)), rng))
    add("random_unrelated", T, Q)

    T, Q = [], []
    for _ in range(150):  # related pairs, error 0..40 %, text sometimes shorter than needed
        L = int(rng.integers(1, 400))
        err = float(rng.random() * 0.4)
        ratio = tuple(float(x) + 0.05 for x in rng.random(3))
        slack = float(rng.choice([-0.3, 0.0, 0.15, 0.5]))
        t, q = synth.make_pair(L, err, ratio, rng, slack=max(slack, 0.0))
        if slack < 0:
            t = t[: int(len(t) * 0.7)]
        T.append(synth.BASES[t].tobytes())
        Q.append(synth.BASES[q].tobytes())
    add("random_related", T, Q)

    T, Q = [], []
    for _ in range(20):  # low-complexity sequences: many equally good paths, tie-breaking
        lt, lq = int(rng.integers(1, 160)), int(rng.integers(1, 160))
        alpha = synth.BASES[rng.integers(0, 4, 2)]
        T.append(alpha[rng.integers(0, 2, lt)].tobytes())
        Q.append(alpha[rng.integers(0, 2, lq)].tobytes())
    add("low_complexity", T, Q)

    t, q = synth.make_pairs(6, 97, "ont", seed=5)
    add("lower_case", [x.lower() for x in t], [x.decode().swapcase().encode() for x in q])

    for prof, L, n, seed in [("illumina", 150, 40, 11), ("ont", 1000, 12, 12),
                             ("pacbio", 1000, 6, 13), ("pacbio15", 2000, 4, 14),
                             ("ont", 10000, 3, 15)]:
        t, q = synth.make_pairs(n, L, prof, seed=seed)
        add("%s_%d" % (prof, L), t, q)

    dump("pairs_w64_o33.json", {"W": 64, "O": 33, "cases": cases})

    # --- read-mapping surface: genome suffix semantics ------------------------
    genome = synth.random_seq(3000, rng)
    reads, cands = [], []
    for k in range(24):
        L = int(rng.integers(20, 200))
        start = int(rng.integers(0, 3000 - 10))
        src = np.frombuffer(genome[start:start + L + 40], dtype=np.uint8)
        codes = np.searchsorted(synth.BASES, src).astype(np.uint8)
        q = synth.mutate(codes, 0.08, (1, 1, 1), rng)[:L]
        reads.append(synth.BASES[q].tobytes())
        c = [start]
        for _ in range(int(rng.integers(0, 4))):
            c.append(int(np.clip(start + rng.integers(-6, 7), 0, 2999)) if rng.random() < 0.5
                     else int(rng.integers(0, 3000)))
        cands.append(c)
    reads.append(b"")            # empty read
    cands.append([5])
    reads.append(reads[0])       # read without candidates
    cands.append([])
    reads.append(genome[2990:3000] + b"ACGTACGT")  # runs off the end of the genome
    cands.append([2990, 2999])
    eds, cigs, _ = ref.align_mapping(genome, reads, cands)
    dump("mapping_w64_o33.json", {
        "W": 64, "O": 33, "genome": genome.decode(),
        "reads": [r.decode() for r in reads], "candidates": cands,
        "ed": eds, "cigar": cigs})
    print("pairs cases:", len(cases), " mapping pairs:", len(eds))

    # --- other knob settings: the same reference sources built with -DCLI_KNOBS -DCLI_W/-DCLI_K/-DCLI_O
    #     (src/genasm_cpu.cpp:22-35); W=32,O=17 is the paper's short-read setting, O=2 the README's sweep row
    for W, O in [(32, 17), (64, 2), (48, 24), (64, 40), (128, 65), (96, 49)]:
        r2 = Reference(W, O)
        rng2 = np.random.Generator(np.random.PCG64(1000 * W + O))
        T, Q = synth.make_pairs(25, 300, "ont", seed=W * 7 + O)
        t2, q2 = synth.make_pairs(10, 150, "illumina", seed=W + O)
        T, Q = T + t2, Q + q2
        for _ in range(45):
            T.append(synth.random_seq(int(rng2.integers(0, 180)), rng2))
            Q.append(synth.random_seq(int(rng2.integers(0, 180)), rng2))
        eds2, cigs2, _ = r2.align(T, Q)
        dump("pairs_w%d_o%d.json" % (W, O), {"W": W, "O": O, "cases": [
            {"group": "knobs", "text": t.decode(), "read": q.decode(), "ed": e, "cigar": c}
            for t, q, e, c in zip(T, Q, eds2, cigs2)]})
        print("W=%d O=%d cases:" % (W, O), len(T))

    # --- windows of more than 128 characters (four-word bitvectors) and W=128 with a small overlap
    #     (the traceback then reads past character 63): longer inputs so that several windows chain
    for W, O in [(256, 129), (192, 97), (128, 20), (200, 50)]:
        r2 = Reference(W, O)
        rng2 = np.random.Generator(np.random.PCG64(1000 * W + O))
        T, Q = synth.make_pairs(20, 1200, "ont", seed=W * 7 + O)
        t2, q2 = synth.make_pairs(8, 600, "pacbio15", seed=W + O)
        T, Q = T + t2, Q + q2
        for _ in range(40):
            T.append(synth.random_seq(int(rng2.integers(0, 500)), rng2))
            Q.append(synth.random_seq(int(rng2.integers(0, 500)), rng2))
        eds2, cigs2, _ = r2.align(T, Q)
        dump("pairs_w%d_o%d.json" % (W, O), {"W": W, "O": O, "cases": [
            {"group": "knobs", "text": t.decode(), "read": q.decode(), "ed": e, "cigar": c}
            for t, q, e, c in zip(T, Q, eds2, cigs2)]})
        print("W=%d O=%d cases:" % (W, O), len(T))


if __name__ == "__main__":
    main()
