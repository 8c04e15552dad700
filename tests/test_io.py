"""Rows of SURVEY.md §8(f): the read-mapping front door (FASTA/FASTQ/MAF/PAF readers,
left-extension, multi-chromosome offsets), affine re-scoring and the CIGAR validator.
CPU tests compare the loader with the reference's own readers (oracle/_ref, when present);
the GPU test runs a loaded job end to end, including reverse-strand candidates."""
import ctypes as C
import os

import numpy as np
import pytest

import scrooge_amd
from scrooge_amd import io as sio
from scrooge_amd import synth
from tests.cigar_check import validate

COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def revcomp(b):
    return b.translate(COMP)[::-1]


def make_dataset(tmp, seed=3, n_reads=40, multi=True, paf=True):
    """Synthetic FASTA (2 chromosomes, wrapped lines, lower-case stretch), FASTQ and seeds."""
    rng = np.random.Generator(np.random.PCG64(seed))
    chroms = [("chr1 first test contig", synth.random_seq(6000, rng)), ("chr2", synth.random_seq(4000, rng))]
    if not multi:
        chroms = chroms[:1]
    fa = os.path.join(tmp, "genome.fa")
    with open(fa, "w") as f:
        for name, seq in chroms:
            f.write(">%s\n" % name)
            s = seq.decode()
            s = s[:100].lower() + s[100:]
            for k in range(0, len(s), 70):
                f.write(s[k:k + 70] + "\r\n")
    fq = os.path.join(tmp, "reads.fq")
    seeds = os.path.join(tmp, "seeds.paf" if paf else "seeds.maf")
    truth = []
    with open(fq, "w") as f, open(seeds, "w") as s:
        for r in range(n_reads):
            ci = int(rng.integers(0, len(chroms)))
            name, seq = chroms[ci]
            L = int(rng.integers(60, 400))
            start = int(rng.integers(0, len(seq) - L - 60))
            codes = np.searchsorted(synth.BASES, np.frombuffer(seq[start:start + L + 50], dtype=np.uint8)).astype(np.uint8)
            read = synth.BASES[synth.mutate(codes, 0.06, (1, 1, 1), rng)[:L]].tobytes()
            fwd = bool(rng.random() < 0.7)
            clip = int(rng.integers(0, 12))            # unaligned read prefix the seed does not cover
            rname = "read%d" % r
            stored = read if fwd else revcomp(read)
            f.write("@%s\n%s\n+\n%s\n" % (rname, stored.decode(), "I" * L))
            qs, qe = (clip, L) if fwd else (0, L - clip)
            if paf:
                s.write("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t60\n" % (
                    rname, L, qs, qe, "+" if fwd else "-", name.split()[0], len(seq), start + clip, start + L,
                    L - clip, L - clip))
            else:
                s.write("a\ns ref %d %d + %d %s\ns %s %d %d %s %d %s\n\n" % (
                    start + clip, L - clip, len(seq), "ACGT", rname, qs, qe - qs, "+" if fwd else "-", L, "ACGT"))
            truth.append((rname, ci, start, fwd, read))
    return fa, fq, seeds, chroms, truth


def test_affine_score_matches_reference_rule():
    # get_alignment_score (src/cpu_baseline.cpp:694-725) with costs 2,4,4,2
    assert sio.affine_score("10=") == 20
    assert sio.affine_score("") == 0
    assert sio.affine_score("4=1X4=") == 8 - 4 + 8
    assert sio.affine_score("3=2I3=") == 6 - (4 + 2 * 2) + 6
    assert sio.affine_score("2I3D") == -(4 + 2 * 2 + 3 * 2)          # adjacent I and D share one opening
    assert sio.affine_score("2I1=3D") == -(4 + 4) + 2 - (4 + 6)
    assert sio.affine_score("31I13I16=2I2I") == -(4 + 2 * 44) + 32 - (4 + 2 * 4)
    with pytest.raises(scrooge_amd.ScroogeError):
        sio.affine_score("3Z")


def test_validator_agrees_with_python_checker():
    text, read = "AAAACCCCGGGGTTTT", "AAAAGGGGAAAATTTT"
    assert sio.validate_alignment(text, read, "4=4D4=4I4=", 8) == 0
    assert sio.validate_alignment(text, read, "4=4D4=4I4=", 7) == 6
    assert sio.validate_alignment(text, read, "4=4D4=4I3=", 8) == 3
    assert sio.validate_alignment(text, read, "4=4D4X4I4=", 12) == 5
    assert sio.validate_alignment(text, read, "0=4=4D4=4I4=", 8) == 2
    assert sio.validate_alignment(text, read, "4=4D4=4I4=9D", 17) == 4
    assert sio.validate_alignment(text, read, "4=oops", 0) == 1
    assert sio.validate_alignment(text.lower(), read, "4=4D4=4I4=", 8) == 0
    rng = np.random.Generator(np.random.PCG64(1))
    from oracle.pyoracle import Oracle
    t, q = synth.make_pairs(50, 300, "ont", seed=8)
    eds, cigars, _, _ = Oracle().align(t, q)
    for a, b, e, c in zip(t, q, eds, cigars):
        assert sio.validate_alignment(a, b, c, e) == 0
        assert validate(a, b, c, e) is None


@pytest.mark.parametrize("paf", [True, False])
@pytest.mark.parametrize("multi", [True, False])
def test_job_loader_semantics(tmp_path, paf, multi):
    if not paf and multi:
        pytest.skip("the MAF reader knows a single reference named 'ref' (src/util.cpp:207-211)")
    fa, fq, seeds, chroms, truth = make_dataset(str(tmp_path), multi=multi, paf=paf)
    job = sio.Job(fa, fq, seeds, reverse_strand=1)
    genome, reads, cands, names = job.views()
    assert genome.upper() == b"".join(c[1] for c in chroms)
    assert job.n_chromosomes == len(chroms) and job.n_reads == len(truth)
    lens = [len(r) for r in reads]
    assert lens == sorted(lens, reverse=True)                      # src/tests.cu:375-377
    by_name = {t[0]: t for t in truth}
    offs = np.cumsum([0] + [len(c[1]) for c in chroms])
    k = 0
    for r, cs, nm in zip(reads, cands, names):
        _, ci, start, fwd, orig = by_name[nm]
        assert len(cs) == 1
        s, rev = cs[0]
        assert rev == (not fwd)
        assert s == offs[ci] + start                               # left-extended back over the clipped prefix
        assert (r if fwd else revcomp(r)) == orig
        cname, cstart, clen = job.pair_chromosome(k)
        assert cname == chroms[ci][0] and cstart == start and clen == len(chroms[ci][1])
        k += 1
    fwd_only = sio.Job(fa, fq, seeds)                              # reference behaviour: '-' dropped
    assert fwd_only.n_pairs == sum(1 for t in truth if t[3])
    capped = sio.Job(fa, fq, seeds, read_length_cap=100, inflation=2)
    assert capped.n_reads == 2 * len(truth)
    assert max(len(r) for r in capped.views()[1]) == 100


def test_loader_errors(tmp_path):
    fa, fq, seeds, _, _ = make_dataset(str(tmp_path))
    with pytest.raises(scrooge_amd.ScroogeError) as e:
        sio.Job(fa, fq, os.path.join(str(tmp_path), "missing.paf"))
    assert e.value.status == sio.SCRG_ERR_IO
    bad = os.path.join(str(tmp_path), "seeds.txt")
    open(bad, "w").write("x")
    with pytest.raises(scrooge_amd.ScroogeError) as e:
        sio.Job(fa, fq, bad)
    assert e.value.status == sio.SCRG_ERR_FORMAT
    unk = os.path.join(str(tmp_path), "unk.paf")
    open(unk, "w").write("nosuchread\t10\t0\t10\t+\tchr1\t6000\t5\t15\t10\t10\t60\n")
    with pytest.raises(scrooge_amd.ScroogeError) as e:
        sio.Job(fa, fq, unk)
    assert "unknown read" in str(e.value)


def _ref_dump(fa, fq, seeds, out):
    from oracle.pyoracle import Reference
    lib = C.CDLL(Reference.PATH)
    lib.ref_dump_job.restype = C.c_int
    lib.ref_dump_job.argtypes = [C.c_char_p] * 4
    assert lib.ref_dump_job(fa.encode(), fq.encode(), seeds.encode(), out.encode()) == 0
    recs = {}
    genome = None
    for line in open(out):
        f = line.rstrip("\n").split("\t")
        if f[0] == "genome":
            genome = f[1]
        else:
            recs[f[0]] = (f[1], sorted(int(x) for x in f[2].split(",") if x))
    return genome, recs


@pytest.mark.parametrize("paf,multi", [(True, False), (False, False)])
def test_loader_matches_reference_readers(tmp_path, paf, multi):
    """Against read_genome / read_fastq_and_seed_locations of the unmodified src/util.cpp.
    (Single chromosome: the reference looks chromosomes up by the full FASTA header,
    src/util.cpp:294-296, which PAF target names never equal when the header has a comment.)"""
    from oracle.pyoracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref not built")
    fa, fq, seeds, chroms, truth = make_dataset(str(tmp_path), multi=multi, paf=paf, seed=11)
    g_ref, recs = _ref_dump(fa, fq, seeds, os.path.join(str(tmp_path), "ref_dump.txt"))
    job = sio.Job(fa, fq, seeds)
    genome, reads, cands, names = job.views()
    assert genome.decode() == g_ref
    assert sorted(names) == sorted(recs)
    for r, cs, nm in zip(reads, cands, names):
        assert r.decode() == recs[nm][0]
        assert sorted(s for s, _ in cs) == recs[nm][1]


@pytest.mark.gpu
def test_job_end_to_end_with_reverse_strand(tmp_path, aligner, oracle):
    fa, fq, seeds, chroms, truth = make_dataset(str(tmp_path), n_reads=300, seed=21)
    job = sio.Job(fa, fq, seeds, reverse_strand=1)
    genome, reads, cands, names = job.views()
    paf_out = os.path.join(str(tmp_path), "out.paf")
    alns = job.align(aligner, out_path=paf_out)
    texts, qs = [], []
    for r, cs in zip(reads, cands):
        for s, rev in cs:
            texts.append(genome[s:s + len(r) + 200].upper())
            qs.append((revcomp(r) if rev else r).upper())
    eds, cigars, _, _ = oracle.align(texts, qs)
    assert [a.edit_distance for a in alns] == eds
    assert [a.cigar for a in alns] == cigars
    assert np.mean(eds) < 0.12 * np.mean([len(q) for q in qs])       # true loci: error-rate sized distances
    lines = open(paf_out).read().splitlines()
    assert len(lines) == len(alns)
    f = lines[0].split("\t")
    assert f[0] == names[0] and f[4] in "+-" and f[-1] == "cg:Z:" + alns[0].cigar
    sam_out = os.path.join(str(tmp_path), "out.sam")
    job.align(aligner, out_path=sam_out, fmt="sam")
    body = [l for l in open(sam_out) if not l.startswith("@")]
    assert len(body) == len(alns) and body[0].split("\t")[5] == alns[0].cigar


@pytest.mark.gpu
def test_cli_end_to_end(tmp_path, capsys):
    """`python -m scrooge_amd.cli` = the reference's `tests --reference= --reads= --seeds=` harness."""
    from scrooge_amd import cli
    fa, fq, seeds, chroms, truth = make_dataset(str(tmp_path), n_reads=120, seed=33)
    out = os.path.join(str(tmp_path), "cli.paf")
    rc = cli.main(["--reference=" + fa, "--reads=" + fq, "--seeds=" + seeds, "--out=" + out, "--reverse_strand",
                   "--validate", "--dataset_inflation=2"])
    text = capsys.readouterr().out
    assert rc == 0
    assert "GPU kernel ran at" in text and "validated 240 alignments, 0 failed" in text
    assert len(open(out).read().splitlines()) == 240
