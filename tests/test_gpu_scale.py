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
    _same(aligner.align_pairs(t, q), eds, cigars)                                # one pair per lane (the default)
    _same(aligner.align_pairs(t, q, lanes_per_pair=8), eds, cigars)
    _same(aligner.align_pairs(t, q, lanes_per_pair=8, lds_rows=4), eds, cigars)  # heavy HBM spill of R rows


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


def _mapping_workload(n_reads, G, seed):
    """BASELINE configs[2]: one synthetic chromosome of G bases, n_reads x 150 bp reads with ~1 % errors (0.9 % substitutions,
    one deletion or one insertion in 7.5 % of the reads each), 4 candidates per read: the true locus, two loci shifted by
    1-3 bases, one random locus."""
    rng = np.random.Generator(np.random.PCG64(seed))
    L = 150
    gcodes = rng.integers(0, 4, G, dtype=np.uint8)
    starts = rng.integers(0, G - 400, n_reads)
    codes = np.empty((n_reads, L), np.uint8)
    j = np.arange(L)[None, :]
    for lo in range(0, n_reads, 125000):                  # in pieces: bounded temporaries
        hi = min(n_reads, lo + 125000)
        m = hi - lo
        pos = rng.integers(1, L - 1, (m, 1))
        u = rng.random((m, 2))
        is_del = u[:, 0:1] < 0.075
        is_ins = (u[:, 1:2] < 0.075) & ~is_del
        src = j + (is_del & (j >= pos)) - (is_ins & (j > pos))
        c = gcodes[starts[lo:hi, None] + src]
        sub = rng.random((m, L)) < 0.009
        c = np.where(sub, (c + rng.integers(1, 4, (m, L), dtype=np.uint8)) & 3, c)
        c = np.where(is_ins & (j == pos), rng.integers(0, 4, (m, L), dtype=np.uint8), c)
        codes[lo:hi] = c
    cand = np.stack([starts, np.maximum(0, starts - rng.integers(1, 4, n_reads)), starts + rng.integers(1, 4, n_reads),
                     rng.integers(0, G - 10, n_reads)], axis=1)
    return gcodes, codes, cand


def test_config3_full_size_host_api(aligner, oracle):
    """BASELINE configs[2] AT FULL SIZE through the host entry point the reference's mapping overload binds to
    (scrg_align_mapping = genasm_gpu::align_all(genome, reads, candidates), src/genasm_gpu.cu:1067-1200): a 100 Mbp
    chromosome, 1 M x 150 bp reads x 4 candidates = 4 M pairs in one call.  The first 12.5 k reads (50 k pairs) are
    compared with the oracle, CIGAR for CIGAR; every one of the 4 M results is held to the size-independent properties
    (the read is consumed exactly, the text is not overrun, the edit distance is the number of non-match columns, the
    rendered text has one letter per run) and, where the alignment has no gaps, to the Hamming distance of the two
    sequences; the same call against the resident genome returns the same arrays."""
    G, n_reads, L, n_c = 100_000_000, 1_000_000, 150, 4
    gcodes, codes, cand = _mapping_workload(n_reads, G, seed=2024)
    genome = synth.BASES[gcodes].tobytes()
    ascii_reads = synth.BASES[codes]
    reads = [ascii_reads[r].tobytes() for r in range(n_reads)]
    cands = cand.tolist()
    res = aligner.align_mapping(genome, reads, cands, arrays=True)
    n = n_reads * n_c
    assert res["edit_distance"].shape == (n,) and not res["status"].any()

    # -- oracle, CIGAR for CIGAR, on the first 50 k pairs
    k = 12500
    texts, qs = [], []
    for r in range(k):
        for s in cands[r]:
            texts.append(genome[s:s + 400])               # enough of the suffix for a 150 bp read
            qs.append(reads[r])
    eds, cigars, _, _ = oracle.align(texts, qs, threads=16)
    off = res["cigar_offset"]
    bad = [i for i in range(n_c * k) if int(res["edit_distance"][i]) != eds[i]
           or res["cigar_text"][int(off[i]):int(off[i + 1]) - 1].decode() != cigars[i]]
    assert not bad, "%d of %d differ from the oracle, first %d" % (len(bad), n_c * k, bad[0])

    # -- every pair: properties on the run arrays
    ro = res["run_offset"].astype(np.int64)
    assert (np.diff(ro) > 0).all()                        # no empty alignment: every read has 150 bases
    cnt = res["runs"][:, 0].astype(np.int64)
    op = res["runs"][:, 1]
    assert np.isin(op, np.frombuffer(b"=XID", np.uint8)).all() and (cnt > 0).all()
    is_m, is_x, is_i, is_d = (op == ord(c) for c in "=XID")
    seg = ro[:-1]
    read_used = np.add.reduceat(np.where(is_d, 0, cnt), seg)
    text_used = np.add.reduceat(np.where(is_i, 0, cnt), seg)
    edits = np.add.reduceat(np.where(is_m, 0, cnt), seg)
    assert (read_used == L).all()
    assert (text_used <= G - cand.reshape(-1)).all()
    assert (edits == res["edit_distance"]).all()
    text = np.frombuffer(res["cigar_text"], np.uint8)
    assert int(np.isin(text, np.frombuffer(b"=XID", np.uint8)).sum()) == int(ro[-1]) and int((text == 0).sum()) == n
    assert (text[res["cigar_offset"][1:].astype(np.int64) - 1] == 0).all()
    # gap-free alignments walk the diagonal: their edit distance is the Hamming distance of read and text
    gaps = np.add.reduceat((is_i | is_d).astype(np.int64), seg)
    flat = np.nonzero(gaps == 0)[0]
    assert len(flat) > 0.8 * n_reads                      # the true loci of the 85 % of reads without an insertion or deletion
    for lo in range(0, len(flat), 500000):
        f = flat[lo:lo + 500000]
        t = gcodes[cand.reshape(-1)[f, None] + np.arange(L)[None, :]]
        assert ((t != codes[f // n_c]).sum(axis=1) == res["edit_distance"][f]).all()
    ed = res["edit_distance"].reshape(n_reads, n_c)
    assert ed[:, 0].mean() < 3 and ed[:, 3].mean() > 50   # true locus vs random locus

    # -- the same batch against the genome kept on the device (scrg_genome_set + scrg_align_mapping_resident)
    aligner.set_genome(genome)
    try:
        res2 = aligner.align_mapping(None, reads, cands, arrays=True)
    finally:
        aligner.clear_genome()
    for key in ("edit_distance", "status", "run_offset", "runs", "cigar_offset"):
        assert np.array_equal(res[key], res2[key]), key
    assert res["cigar_text"] == res2["cigar_text"]


def test_config5_long_noisy_reads(aligner, oracle):
    """BASELINE configs[4] in miniature: 50 kb PacBio-error reads at 15 % (multi-window traceback
    over ~1650 windows per pair, frequent window distances above the LDS rows)."""
    t, q = synth.make_pairs(24, 50000, "pacbio15", seed=50)
    eds, cigars, st, _ = oracle.align(t, q, threads=16)
    _same(aligner.align_pairs(t, q), eds, cigars)
    _same(aligner.align_pairs(t, q, lanes_per_pair=8), eds, cigars)
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


def test_cigar_slice_overflow_is_reported_not_overrun(aligner, oracle):
    """Device-pointer API with deliberately tiny CIGAR slices: the pair's status flags the overflow,
    n_runs still counts every run, nothing is written outside the slice, edit distance stays exact."""
    import torch
    import scrooge_amd
    dev = torch.device("cuda", 0)
    t, q = synth.make_pairs(6, 800, "ont", seed=31)
    eds, cigars, _, _ = oracle.align(t, q)
    tw, rw = (len(t[0]) + 31) // 32, (800 + 31) // 32
    rows = np.zeros((6, (tw + rw) * 32), dtype=np.uint8)
    for k in range(6):
        rows[k, :len(t[k])] = np.frombuffer(t[k], dtype=np.uint8)
        rows[k, tw * 32: tw * 32 + 800] = np.frombuffer(q[k], dtype=np.uint8)
    ascii_t = torch.from_numpy(rows).to(dev)
    seq = torch.zeros(6 * (tw + rw) + scrooge_amd.api.SEQ_PAD_WORDS, dtype=torch.int64, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    aligner.set_stream(0)
    try:
        aligner.pack_planar(ascii_t.view(-1), seq, bad)
        caps = [16, 32, 4096, 16, 48, 4096]                      # multiples of 16; 3 of them far too small
        offs = np.cumsum([0] + [c + 16 for c in caps])[:-1]       # 16-run guard band after every slice
        desc = torch.tensor([[k * (tw + rw) * 32, len(t[k]), (k * (tw + rw) + tw) * 32, 800, int(offs[k]), caps[k]]
                             for k in range(6)], dtype=torch.int64, device=dev)
        total = int(offs[-1] + caps[-1] + 16)
        runs = torch.full((total * 2,), 0xAB, dtype=torch.uint8, device=dev)
        ed = torch.zeros(6, dtype=torch.int64, device=dev)
        nr = torch.zeros(6, dtype=torch.int32, device=dev)
        st = torch.zeros(6, dtype=torch.int32, device=dev)
        aligner.align_device(6, seq, desc, runs, ed, nr, st)
        torch.cuda.synchronize()
    finally:
        aligner.use_own_stream()
    n_true = [len([c for c in cg if not c.isdigit()]) for cg in cigars]
    assert ed.cpu().tolist() == eds
    assert nr.cpu().tolist() == n_true
    assert st.cpu().tolist() == [1 if n_true[k] > caps[k] else 0 for k in range(6)]
    assert sum(st.cpu().tolist()) == 4
    h = runs.cpu().numpy()
    for k in range(6):
        guard = h[2 * (offs[k] + caps[k]): 2 * (offs[k] + caps[k] + 16)]
        assert (guard == 0xAB).all(), "pair %d wrote past its slice" % k
        if n_true[k] <= caps[k]:
            seg = h[2 * offs[k]: 2 * (offs[k] + n_true[k])]
            assert "".join("%d%s" % (seg[2 * j], chr(seg[2 * j + 1])) for j in range(n_true[k])) == cigars[k]


def test_extreme_inputs(aligner, oracle):
    """Window distances pinned at their maximum (no text left: m insertions per window, every row
    of R up to K=64 in use), all-mismatch pairs, homopolymers, and a 300 kb read."""
    rng = np.random.Generator(np.random.PCG64(77))
    big_t, big_q = synth.make_pairs(1, 300000, "ont", seed=123)
    T = [b"", b"ACGT", b"T" * 6000, b"A" * 3000, b"A" * 3000, b"AC" * 2000, big_t[0], b"G" * 10]
    Q = [synth.random_seq(5000, rng), synth.random_seq(3000, rng), b"A" * 5000, b"A" * 2500, b"A" * 3500,
         b"CA" * 1500, big_q[0], b"g" * 700]
    eds, cigars, st, _ = oracle.align(T, Q, threads=8)
    assert eds[0] == 5000 and eds[2] == 5000 and eds[3] == 0
    for g, rows in [(1, 0), (8, 13), (8, 2), (64, 13), (16, 5)]:
        _same(aligner.align_pairs(T, Q, lanes_per_pair=g, lds_rows=rows), eds, cigars)


def test_two_handles_on_two_streams_overlap(oracle):
    """Batches pipelined the way bench.py and INTEGRATION.md §4b do it: two handles, two streams, launches in
    flight at the same time (each handle owns its work queue and spill area).  Every launch's results are
    checked, including a handle reused while the other one is still running."""
    import torch
    import scrooge_amd
    dev = torch.device("cuda", 0)
    batches = []
    for b in range(4):
        t, q = synth.make_pairs(3000, 2000, "ont" if b % 2 == 0 else "pacbio15", seed=100 + b)
        eds, cigars, _, _ = oracle.align(t, q, threads=16)
        tw, rw = (max(len(x) for x in t) + 31) // 32, (2000 + 31) // 32
        rows = np.zeros((len(t), (tw + rw) * 32), dtype=np.uint8)
        for k in range(len(t)):
            rows[k, :len(t[k])] = np.frombuffer(t[k], dtype=np.uint8)
            rows[k, tw * 32: tw * 32 + len(q[k])] = np.frombuffer(q[k], dtype=np.uint8)
        batches.append((t, q, eds, cigars, tw, rw, torch.from_numpy(rows).to(dev)))
    als = [scrooge_amd.Aligner(0), scrooge_amd.Aligner(0)]
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    outs = []
    torch.cuda.synchronize()
    for b, (t, q, eds, cigars, tw, rw, ascii_t) in enumerate(batches):
        a, st = als[b % 2], streams[b % 2]
        a.set_stream(st.cuda_stream)
        n = len(t)
        with torch.cuda.stream(st):
            seq = torch.zeros(n * (tw + rw) + scrooge_amd.api.SEQ_PAD_WORDS, dtype=torch.int64, device=dev)
            bad = torch.zeros(1, dtype=torch.int32, device=dev)
            a.pack_planar(ascii_t.view(-1), seq, bad)
            cap = (2 * 2000 + 8 + 15) // 16 * 16
            idx = torch.arange(n, dtype=torch.int64, device=dev)
            desc = torch.stack([idx * (tw + rw) * 32, torch.tensor([len(x) for x in t], device=dev),
                                (idx * (tw + rw) + tw) * 32, torch.tensor([len(x) for x in q], device=dev),
                                idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
            runs = torch.empty(n * cap * 2, dtype=torch.uint8, device=dev)
            ed = torch.empty(n, dtype=torch.int64, device=dev)
            nr = torch.empty(n, dtype=torch.int32, device=dev)
            status = torch.empty(n, dtype=torch.int32, device=dev)
            a.align_device(n, seq, desc, runs, ed, nr, status)       # no synchronisation between the launches
        outs.append((runs, ed, nr, status, cap, seq, desc, bad))
    torch.cuda.synchronize()
    for (t, q, eds, cigars, tw, rw, _), (runs, ed, nr, status, cap, _, _, bad) in zip(batches, outs):
        assert int(bad.item()) == 0 and int(status.max().item()) == 0
        assert ed.cpu().tolist() == eds
        h, cnt = runs.cpu().numpy(), nr.cpu().tolist()
        for k in range(0, len(t), 7):
            seg = h[2 * k * cap: 2 * (k * cap + cnt[k])]
            assert "".join("%d%s" % (seg[2 * j], chr(seg[2 * j + 1])) for j in range(cnt[k])) == cigars[k]
    for a in als:
        a.close()


@pytest.mark.parametrize("gather_root,root_share", [("rotate", "auto"), ("0", "auto"), ("0", "equal"), ("0", "0"), ("0", "0.3")])
def test_bench_two_ranks_dry_run(gather_root, root_share):
    """bench.py's multi-rank control flow (two pipeline lanes, double-buffered gather, barriers, rank 0
    printing) with two processes on this one GPU: gloo through the host instead of RCCL (SCRG_BENCH_DRYRUN),
    so only the logic is checked, not the speed.  With the root fixed, rank 0 aligns a smaller share of the step's pairs
    (--root-share: 0.8 of an equal share at N = 2 by default; 0 = a root that only collects and decodes, what 'auto'
    gives at N = 8): its buffers are padded with empty reads, every slot still decodes to its own kernel's runs."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCRG_BENCH_DRYRUN="1")
    # plain `python bench.py --gpus 2`: bench.py starts its own two ranks (a fresh torch.distributed.run child, started
    # before this parent touches the GPU) — the form the driver uses; one case goes through an explicit launcher instead
    args = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--pairs", "6400", "--read-len", "2000",
            "--gather-root", gather_root, "--root-share", root_share]
    if (gather_root, root_share) == ("0", "0.3"):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", "29533"] + args
    else:
        cmd = [sys.executable] + args
        env.pop("WORLD_SIZE", None)
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]              # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["value"] > 0 and j["scaling"] == "weak"
    # both ranks' slots of the last step decoded on its root (rank 1 when the root rotates: 6 steps), the verdict sent to rank 0
    assert j["gather_check"] is True and j["config"]["gather"]["format"] == "edits"
    assert j["config"]["gather"]["root"] == ("rank 0" if gather_root == "0" else "step k to rank k mod N")
    sh = j["config"]["shards"]
    if gather_root == "rotate" or root_share == "equal":
        assert sh == "equal" and j["config"]["pairs_per_step_all_gpus"] == 12800
    else:
        want0 = {"auto": 5120, "0": 0, "0.3": 1920}[root_share]
        assert sh["rank_0"] == want0 and sh["other_ranks"] == (12800 - want0 + 63) // 64 * 64
        assert j["config"]["pairs_per_step_all_gpus"] == sh["rank_0"] + sh["other_ranks"] >= 12800
    assert j["roofline"]["kernel"].startswith("genasm_lane_kernel") and j["roofline"]["hbm"]["achieved"] > 0
    # the diagnostics of the multi-GPU step, in the same line (--diagnose auto = on for N > 1): the same steps under the other
    # policies, the gather alone, every peer's rate into rank 0, every rank's own N = 1-equivalent rate
    d = j["diagnose"]
    for key in ("root0_equal_shards", "root0_auto_shards", "rotating_root_equal_shards"):
        assert d[key]["value"] > 0 and d[key]["every_slot_decoded"] is True and d[key]["steps"] >= 2
    # the root-share plan at N = 2: rank 0 aligns 0.8 of an equal share; in a run whose buffers were sized for equal shards the
    # same split runs at the size that fits (the other rank aligns what the buffers hold)
    a_sh = d["root0_auto_shards"]["shards"]
    assert a_sh["rank_0"] < a_sh["other_ranks"] <= 7680 and 0.6 < a_sh["rank_0"] / a_sh["other_ranks"] < 0.7
    assert d["root0_equal_shards"]["shards"] == {"rank_0": 6400, "other_ranks": 6400}
    assert d["rotating_root_equal_shards"]["root"] == "step k to rank k mod N"
    assert d["gather_without_decode"] == j["gather_without_decode"] and j["gather_without_decode"]["value"] > 0
    ln = d["links"]
    assert ln["GBs_per_peer_all_at_once"] > 0 and ln["bytes_per_rank_and_gather"] == j["config"]["gather"]["bytes_per_rank_and_step"]
    assert [x["peer"] for x in ln["one_peer_at_a_time"]] == [1] and ln["one_peer_at_a_time"][0]["GBs"] > 0
    pg = j["per_gpu_value"]
    assert len(pg["per_rank"]) == 2 and pg["min"] > 0 and pg["pairs_per_rank_and_step"] == 6400
    assert d["efficiency_vs_per_gpu_value"]["configured"] > 0
    assert j["config"]["rccl_ranks"] == 2


@pytest.mark.parametrize("gather_root", ["rotate", "0"])
def test_bench_eight_ranks_dry_run(gather_root):
    """`python bench.py --gpus 8` as the driver starts it on an 8-GPU node, rehearsed on this one GPU (SCRG_BENCH_DRYRUN: all
    eight ranks on GPU 0, gloo through the host): the rotating root with eight buffer sets — every rank is the root of steps,
    produces its own slot in place and decodes all eight —, the fixed root with the root-share plan of N = 8 (rank 0 only collects
    and decodes), every slot decoded to its own kernel's runs, one line from rank 0 with the diagnostics of an N > 1 run in it."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCRG_BENCH_DRYRUN="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "17", "--warmup", "3", "--pairs", "1280",
                          "--read-len", "1500", "--gather-root", gather_root, "--diagnose-budget", "200", "--deadline", "500"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=700)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["steps"] == 17 and j["value"] > 0 and j["scaling"] == "weak" and j["gather_check"] is True
    assert j["config"]["rccl_ranks"] == 8 and j["config"]["gather"]["format"] == "edits"
    assert j["config"]["gather"]["root"] == ("rank 0" if gather_root == "0" else "step k to rank k mod N")
    if gather_root == "rotate":
        assert j["config"]["shards"] == "equal" and j["config"]["pairs_per_step_all_gpus"] == 8 * 1280
    else:
        assert j["config"]["shards"]["rank_0"] == 0                  # 'auto' at N = 8: the root only collects and decodes
    d = j["diagnose"]
    assert "error" not in d, d
    for key in ("root0_equal_shards", "root0_auto_shards", "rotating_root_equal_shards"):
        assert d[key]["value"] > 0 and d[key]["every_slot_decoded"] is True
    assert [x["peer"] for x in d["links"]["one_peer_at_a_time"]] == list(range(1, 8))
    assert len(j["per_gpu_value"]["per_rank"]) == 8 and j["per_gpu_value"]["min"] > 0


@pytest.mark.parametrize("fault", ["raise", "hang"])
def test_bench_line_survives_its_diagnostics(fault):
    """The diagnostics of an N > 1 run come last and under a budget: if they raise on a rank (the others then wait in a
    collective that rank never enters) or never return, rank 0 still prints the complete line with the error in `diagnose`
    and the job ends with exit code 0 (bench.py: guarded_diagnostics).  Two gloo ranks on this GPU, as in the dry run above."""
    import json, os, subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCRG_BENCH_DRYRUN="1", SCRG_BENCH_TEST_DIAG=fault)
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "6400",
                          "--read-len", "2000", "--diagnose-budget", "20"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["gather_check"] is True and j["roofline"]["hbm"]["achieved"] > 0
    assert "error" in j["diagnose"] and j["per_gpu_value"] is None
    if fault == "hang":
        assert "did not finish within --diagnose-budget" in j["diagnose"]["error"]
    assert "rest of the line is complete" in j["diagnose"]["error"]
    assert time.time() - t0 < 300


def test_full_bench_size_two_algorithms_agree(aligner, aligner_select):
    """BASELINE configs[1] at full size (100k x 10 kb ONT-error pairs, generated on the GPU like bench.py):
    the lane-per-pair kernel (difference vectors, the default) and the diagonal-major and column-major window
    paths of the G = 8 kernel (GenASM rows) are three independent formulations of the same table (the lane-per-pair kernel
    in both of its forms: at this size the library's choice — one wavefront per 64 pairs — and two forced, in the test build of the library); every edit
    distance, run count and run must be identical between them, the edit distances must respect the read
    length and error rate, 500 sampled pairs must validate against their sequences — and the first 20 000 pairs are
    compared, run for run, with the reference CPU path itself (oracle/_ref where it was built, else the restatement):
    the full-batch comparison that every default `bench.py` run does, here inside the test suite."""
    import torch
    import bench
    import scrooge_amd
    from tests.cigar_check import validate
    dev = torch.device("cuda", 0)
    n, L = 100000, 10000
    err, ratio = synth.PROFILES["ont"]
    rows, tw, rw, text_len = bench.device_pairs(torch, n, L, err, ratio, 4242, dev)
    n_ref = 20000
    sample = rows[:n_ref].cpu().numpy()
    seq = torch.zeros(n * (tw + rw) + scrooge_amd.api.SEQ_PAD_WORDS, dtype=torch.int64, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    aligner.set_stream(0)
    try:
        aligner.pack_planar(rows.view(-1), seq, bad)
        del rows
        cap = (2 * L + 8 + 15) // 16 * 16
        idx = torch.arange(n, dtype=torch.int64, device=dev)
        desc = torch.stack([idx * (tw + rw) * 32, torch.full_like(idx, text_len), (idx * (tw + rw) + tw) * 32,
                            torch.full_like(idx, L), idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
        res = []
        shipped = aligner
        for lanes, flags in ((1, 0), (1, 512), (8, 0), (8, 32)):           # 512: two wavefronts per 64 pairs (the library's choice at this size: one); 32: diagonal path off
            # (the selection switches exist in the test build only: the same sources with -DSCRG_SELECT, conftest.aligner_select)
            aligner = aligner_select if flags else shipped
            if flags:
                aligner.set_stream(0)
            p = aligner.make_params(lanes_per_pair=lanes)
            p.reserved[0] = flags
            keep, aligner.params = aligner.params, p
            try:
                runs = torch.zeros(n * cap * 2, dtype=torch.uint8, device=dev)
                ed = torch.empty(n, dtype=torch.int64, device=dev)
                nr = torch.empty(n, dtype=torch.int32, device=dev)
                st = torch.empty(n, dtype=torch.int32, device=dev)
                aligner.align_device(n, seq, desc, runs, ed, nr, st)
                cnt = nr.to(torch.int64)
                off = torch.cumsum(cnt, 0) - cnt
                dense = torch.empty(int(cnt.sum().item()) * 2, dtype=torch.uint8, device=dev)
                aligner.compact_runs(n, desc, runs, nr, off, dense)
                torch.cuda.synchronize()
                del runs
                res.append((ed, nr, st, dense, off))
            finally:
                aligner.params = keep
        aligner = shipped
        assert int(bad.item()) == 0
        # a fourth formulation of the output: the kernel's edit streams (one byte per edit and per window end), and from them —
        # by the decoder — the runs again, for all 100 000 pairs
        (ed0, nr0, st0, d0, off0) = res[0]
        slices = torch.zeros(n * cap * 2, dtype=torch.uint8, device=dev)
        ed_s = torch.empty(n, dtype=torch.int64, device=dev)
        ln = torch.empty(n, dtype=torch.int32, device=dev)
        st_s = torch.empty(n, dtype=torch.int32, device=dev)
        aligner.align_device_edits(n, seq, desc, slices, ed_s, ln, st_s)
        r4 = (ln.to(torch.int64) + 3) & -4
        boff = torch.cumsum(r4, 0) - r4
        stream = torch.zeros(int(r4.sum().item()) + 8, dtype=torch.uint8, device=dev)
        aligner.compact_runs(n, desc, slices, (r4 >> 1).to(torch.int32), boff >> 1, stream)
        del slices
        back = torch.zeros_like(d0)
        nbad = torch.zeros(1, dtype=torch.int32, device=dev)
        aligner.decode_edit_stream(n, stream, boff, ln, desc.view(-1)[3:], 6, off0, back, nr0, nbad)
        torch.cuda.synchronize()
        assert torch.equal(ed_s, ed0) and int(st_s.max()) == 0 and int(nbad.item()) == 0 and torch.equal(back, d0)
        is_edit = (stream >= 64).to(torch.int64)
        csum = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(is_edit, 0)])
        assert torch.equal(csum[boff + ln.to(torch.int64)] - csum[boff], ed0)          # one byte per edit
        assert 1.2 < float(ln.double().mean()) / float(ed0.double().mean()) < 1.5      # (+ a byte per window: 10 kb / 31 against ~970 edits)
        del stream, back
    finally:
        aligner.use_own_stream()
        aligner_select.use_own_stream()
    for (ed1, nr1, st1, d1, _) in res[1:]:
        assert int(st0.max()) == 0 and int(st1.max()) == 0
        assert torch.equal(ed0, ed1) and torch.equal(nr0, nr1) and torch.equal(d0, d1)
    # ~10 % of 10 kb on average (the greedy windows lose the diagonal in a few pairs, here as in the reference,
    # so there is no useful bound on the maximum)
    assert 700 < float(ed0.double().mean()) < 1300 and float((ed0 > 2000).double().mean()) < 0.01
    h, o, c = d0.cpu().numpy(), off0.cpu().tolist(), nr0.cpu().tolist()
    for k in range(500):
        seg = h[2 * o[k]: 2 * (o[k] + c[k])]
        cigar = "".join("%d%s" % (seg[2 * j], chr(seg[2 * j + 1])) for j in range(c[k]))
        text = bytes(sample[k, :text_len])
        read = bytes(sample[k, tw * 32: tw * 32 + L])
        assert validate(text, read, cigar, int(ed0[k])) is None
    # the reference CPU path on the first n_ref pairs: edit distances, run offsets and the runs as arrays
    import numpy as np
    from oracle.pyoracle import Oracle, Reference
    if Reference.available():
        e_cpu, off_cpu, runs_cpu, _ = Reference().align_rows(sample, 0, text_len, tw * 32, L, threads=16)
    else:
        e_cpu, off_cpu, runs_cpu, _, _ = Oracle().align_rows(sample, 0, text_len, tw * 32, L, threads=16)
    cnt = np.asarray(c[:n_ref], dtype=np.uint64)
    off_gpu = np.concatenate([np.zeros(1, np.uint64), np.cumsum(cnt, dtype=np.uint64)])
    assert (ed0[:n_ref].cpu().numpy() == e_cpu).all() and (off_gpu == off_cpu).all()
    assert np.array_equal(h[: 2 * int(off_gpu[n_ref])].reshape(-1, 2), runs_cpu)


def _device_align(aligner, torch, seq, desc, n, cap, **kw):
    dev = seq.device
    runs = torch.zeros(n * cap * 2, dtype=torch.uint8, device=dev)
    ed = torch.empty(n, dtype=torch.int64, device=dev)
    nr = torch.empty(n, dtype=torch.int32, device=dev)
    st = torch.empty(n, dtype=torch.int32, device=dev)
    aligner.align_device(n, seq, desc, runs, ed, nr, st, **kw)
    torch.cuda.synchronize()
    h, cnt = runs.cpu().numpy(), nr.cpu().tolist()
    cig = []
    for k in range(n):
        seg = h[2 * k * cap: 2 * (k * cap + cnt[k])]
        cig.append("".join("%d%s" % (seg[2 * j], chr(seg[2 * j + 1])) for j in range(cnt[k])))
    return ed.cpu().tolist(), cig, st.cpu().tolist()


def _revcomp(b):
    return b.translate(bytes.maketrans(b"ACGTacgt", b"TGCAtgca"))[::-1]


def _drawn_window_settings(count, seed):
    """(W, O) drawn from the whole plane 2 <= W <= 256, 0 <= O < W, every third one next to a border between the kernels."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    while len(out) < count:
        if len(out) % 3 == 0:
            W = int(rng.choice([64, 65, 128, 129, 256, 63, 127, 255, int(rng.integers(2, 257)), int(rng.integers(2, 257))]))
            O = W - int(rng.choice([31, 32, 33, 63, 64, 65, 127, 128, 129, 1, 2]))
        else:
            W = int(rng.integers(2, 257))
            O = int(rng.integers(0, W))
        if 2 <= W <= 256 and 0 <= O < W and (W, O) not in out:
            out.append((W, O))
    return out


@pytest.mark.parametrize("form", ["runs, small launch (two wavefronts per window)", "runs, one wavefront per window", "edit streams"])
@pytest.mark.parametrize("W,O", [(64, 33), (32, 17), (50, 25), (64, 2), (128, 65), (256, 129), (200, 50), (256, 1), (64, 0), (160, 120)]
                         + _drawn_window_settings(20, 606))
def test_reverse_strand_on_the_device(aligner, oracle, form, W, O):
    """SURVEY.md §8 f4 on the device-pointer layer (scrg_params.stranded, SCRG_READ_REVCOMP in scrg_pair_desc.read_off): pairs
    whose read is aligned as its reverse complement FROM THE ONE PACKED COPY of the read — forward and reverse candidates of the
    same read share its words — against the reference algorithm on the reverse-complemented string (the reference itself drops
    such candidates, src/tests.cu:346-355).  Ragged and empty reads, reads shorter than a window, both sequence layouts, runs
    (both forms of the default kernel) and edit streams; every one-pair-per-lane kernel (W-O <= 31; the two-halves kernel; the
    table in parts; the table in HBM, O = 0 included)."""
    if form.startswith("runs, small") and not (W <= 64 and W - O <= 31):
        pytest.skip("the two-wavefront form exists for the default table only")
    drawn = (W, O) in _drawn_window_settings(20, 606)[:20] and (W, O) not in [(64, 33), (32, 17), (50, 25), (64, 2), (128, 65), (256, 129), (200, 50), (256, 1), (64, 0), (160, 120)]
    import torch
    import scrooge_amd
    dev = torch.device("cuda", 0)
    rng = np.random.Generator(np.random.PCG64(W * 7 + O))
    n_reads = 96 if drawn else 333           # (the settings drawn at random: smaller batches)
    t, q = synth.make_pairs(n_reads, 900, "ont", seed=400 + W)
    for k in range(0, n_reads, 4):
        q[k] = q[k][: int(rng.integers(0, 900))]
    q[5], q[9], q[13] = b"", q[9][:1], q[13][:63]
    # every read twice: as it is against its text, and flagged "reverse complement" against a text made from its reverse complement
    texts, reads, want_reads, rev = [], [], [], []
    for k in range(n_reads):
        texts.append(t[k]); reads.append(k); want_reads.append(q[k]); rev.append(0)
        rc = _revcomp(q[k])
        t_rc = synth.BASES[synth.mutate(np.searchsorted(synth.BASES, np.frombuffer(rc + b"ACGT" * 30, dtype=np.uint8)).astype(np.uint8), 0.08, (23, 31, 46), rng)].tobytes()
        texts.append(t_rc); reads.append(k); want_reads.append(rc); rev.append(1)
    n = len(texts)
    eds, cigars, _, _ = oracle.align(texts, want_reads, W=W, O=O, threads=8)
    tw, rw = (max(len(x) for x in texts) + 31) // 32 + 1, (900 + 31) // 32
    cap = (2 * 900 + 8 + 15) // 16 * 16
    G = scrooge_amd.api.GROUP
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    kw = dict(W=W, O=O, stranded=1)
    if form.startswith("runs, one"):
        kw["waves_per_cu"] = 16
    aligner.set_stream(0)
    try:
        for layout in ("linear", "groups"):
            t_rows = np.zeros((n, tw * 32), dtype=np.uint8)
            r_rows = np.zeros((n_reads, rw * 32), dtype=np.uint8)
            for k in range(n):
                t_rows[k, :len(texts[k])] = np.frombuffer(texts[k], dtype=np.uint8)
            for k in range(n_reads):
                r_rows[k, :len(q[k])] = np.frombuffer(q[k], dtype=np.uint8)
            idx = torch.arange(n, dtype=torch.int64, device=dev)
            ridx = torch.tensor(reads, dtype=torch.int64, device=dev)
            flag = torch.tensor(rev, dtype=torch.int64, device=dev) << 63
            tl = torch.tensor([len(x) for x in texts], dtype=torch.int64, device=dev)
            ql = torch.tensor([len(q[k]) for k in reads], dtype=torch.int64, device=dev)
            if layout == "linear":
                seq = torch.zeros(n * tw + n_reads * rw + scrooge_amd.api.SEQ_PAD_WORDS, dtype=torch.int64, device=dev)
                aligner.pack_planar(torch.from_numpy(t_rows).to(dev).view(-1), seq[: n * tw], bad)
                aligner.pack_planar(torch.from_numpy(r_rows).to(dev).view(-1), seq[n * tw:], bad)
                t_off, r_off = idx * tw * 32, (n * tw + ridx * rw) * 32
                lay = {}
            else:
                tg, rg = (n + G - 1) // G, (n_reads + G - 1) // G
                seq = torch.zeros(tg * G * tw + rg * G * rw + scrooge_amd.api.SEQ_PAD_WORDS_GROUPS, dtype=torch.int64, device=dev)
                aligner.pack_planar_groups(torch.from_numpy(t_rows).to(dev).view(-1), n, tw, seq[: tg * G * tw], bad)
                aligner.pack_planar_groups(torch.from_numpy(r_rows).to(dev).view(-1), n_reads, rw, seq[tg * G * tw:], bad)
                t_off = ((idx // G) * tw * G + idx % G) * 32
                r_off = (tg * G * tw + (ridx // G) * rw * G + ridx % G) * 32
                lay = dict(text_stride_words=G, read_stride_words=G)
            desc = torch.stack([t_off, tl, r_off | flag, ql, idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
            assert int(bad.item()) == 0
            if form == "edit streams":
                slices = torch.zeros(n * cap * 2, dtype=torch.uint8, device=dev)
                ed = torch.empty(n, dtype=torch.int64, device=dev)
                ln = torch.empty(n, dtype=torch.int32, device=dev)
                st = torch.empty(n, dtype=torch.int32, device=dev)
                aligner.align_device_edits(n, seq, desc, slices, ed, ln, st, **kw, **lay)
                torch.cuda.synchronize()
                assert ed.cpu().tolist() == eds and int(st.max().item()) == 0, layout
                sl, lh = slices.cpu().numpy(), ln.cpu().tolist()
                for k in range(n):
                    stream = sl[2 * k * cap: 2 * k * cap + lh[k]].tobytes()
                    assert scrooge_amd.api.edit_stream_to_cigar(stream, len(want_reads[k]), W=W, O=O) == cigars[k], (layout, k, rev[k])
            else:
                got = _device_align(aligner, torch, seq, desc, n, cap, **kw, **lay)
                bad_k = [k for k in range(n) if (got[0][k], got[1][k]) != (eds[k], cigars[k])]
                assert not bad_k, (layout, bad_k[:10], [rev[k] for k in bad_k[:10]], [len(want_reads[k]) for k in bad_k[:10]])
                assert got[2] == [0] * n
        # the GenASM-row mappings refuse the parameter (the flag without it is a huge offset, not a strand)
        with pytest.raises(scrooge_amd.ScroogeError):
            _device_align(aligner, torch, seq, desc, n, cap, lanes_per_pair=8, stranded=1)
    finally:
        aligner.use_own_stream()


@pytest.mark.parametrize("wpr", [1, 2, 3, 7, 22, 673])
@pytest.mark.parametrize("n", [1, 63, 65, 300])
def test_group_packer_against_the_linear_one(aligner, n, wpr):
    """scrg_pack_planar_groups (a thread packs two words of a row: odd and even row widths, a last group that is not full) gives the
    words of scrg_pack_planar, re-arranged; padding rows and words are zero; bytes that are no base are counted by both."""
    import torch
    import scrooge_amd
    dev = torch.device("cuda", 0)
    G = scrooge_amd.api.GROUP
    rng = np.random.Generator(np.random.PCG64(n * 1000 + wpr))
    rows = rng.choice(np.frombuffer(b"ACGTacgt", np.uint8), (n, wpr * 32))
    for k in range(0, n, 2):                                   # ragged rows: zero padding
        rows[k, int(rng.integers(0, wpr * 32 + 1)):] = 0
    n_bad = min(5, n)
    for k in range(n_bad):                                     # one byte that is no base in each of the first rows
        rows[k, int(rng.integers(0, wpr * 32))] = ord("N")
    ascii_t = torch.from_numpy(rows).to(dev).view(-1)
    ng = (n + G - 1) // G
    lin = torch.zeros(n * wpr, dtype=torch.int64, device=dev)
    grp = torch.full((ng * G * wpr,), -1, dtype=torch.int64, device=dev)
    bad_l = torch.zeros(1, dtype=torch.int32, device=dev)
    bad_g = torch.zeros(1, dtype=torch.int32, device=dev)
    aligner.set_stream(0)
    try:
        aligner.pack_planar(ascii_t, lin, bad_l)
        aligner.pack_planar_groups(ascii_t, n, wpr, grp, bad_g)
        torch.cuda.synchronize()
    finally:
        aligner.use_own_stream()
    want = np.zeros((ng, wpr, G), dtype=np.int64)
    L = lin.cpu().numpy().reshape(n, wpr)
    for r in range(n):
        want[r // G, :, r % G] = L[r]
    assert np.array_equal(grp.cpu().numpy().reshape(ng, wpr, G), want)
    assert int(bad_l.item()) == n_bad and int(bad_g.item()) == n_bad


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000])
def test_lane_interleaved_layout(aligner, oracle, n):
    """scrg_pack_planar_groups + word strides of 64 (the layout bench.py times): ragged texts and reads, a pair
    count that is not a multiple of the group size, same results as the oracle and as the contiguous layout."""
    import torch
    import scrooge_amd
    dev = torch.device("cuda", 0)
    rng = np.random.Generator(np.random.PCG64(n))
    t, q = synth.make_pairs(n, 700, "ont", seed=1000 + n)
    for k in range(0, n, 3):                                   # ragged: cut some reads and texts short, a few empty
        q[k] = q[k][: int(rng.integers(0, 700))]
        t[k] = t[k][: int(rng.integers(0, len(t[k]) + 1))]
    eds, cigars, _, _ = oracle.align(t, q, threads=8)
    tw, rw = (max(len(x) for x in t) + 31) // 32 + 1, (700 + 31) // 32
    rows = np.zeros((n, (tw + rw) * 32), dtype=np.uint8)
    for k in range(n):
        rows[k, :len(t[k])] = np.frombuffer(t[k], dtype=np.uint8)
        rows[k, tw * 32: tw * 32 + len(q[k])] = np.frombuffer(q[k], dtype=np.uint8)
    ascii_t = torch.from_numpy(rows).to(dev)
    G, RW = scrooge_amd.api.GROUP, tw + rw
    cap = (2 * 700 + 8 + 15) // 16 * 16
    idx = torch.arange(n, dtype=torch.int64, device=dev)
    tl = torch.tensor([len(x) for x in t], dtype=torch.int64, device=dev)
    ql = torch.tensor([len(x) for x in q], dtype=torch.int64, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    aligner.set_stream(0)
    try:
        seq = torch.zeros((n + G - 1) // G * G * RW + scrooge_amd.api.SEQ_PAD_WORDS_GROUPS, dtype=torch.int64, device=dev)
        aligner.pack_planar_groups(ascii_t.view(-1), n, RW, seq, bad)
        first = (idx // G) * RW * G + idx % G
        desc = torch.stack([first * 32, tl, (first + tw * G) * 32, ql, idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
        got = _device_align(aligner, torch, seq, desc, n, cap, text_stride_words=G, read_stride_words=G)
        assert got == (eds, cigars, [0] * n)
        # the same batch in the contiguous layout
        seq2 = torch.zeros(n * RW + scrooge_amd.api.SEQ_PAD_WORDS, dtype=torch.int64, device=dev)
        aligner.pack_planar(ascii_t.view(-1), seq2, bad)
        desc2 = torch.stack([idx * RW * 32, tl, (idx * RW + tw) * 32, ql, idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
        assert _device_align(aligner, torch, seq2, desc2, n, cap) == got
        assert int(bad.item()) == 0
        # strides are a property of the one-pair-per-lane kernel only
        with pytest.raises(scrooge_amd.ScroogeError):
            _device_align(aligner, torch, seq, desc, n, cap, text_stride_words=G, read_stride_words=G, lanes_per_pair=8)
    finally:
        aligner.use_own_stream()


def test_packed_runs_round_trip(aligner, oracle):
    """scrg_compact_runs_packed (one byte per run, the RCCL transfer format) + scrg_unpack_runs reproduce
    scrg_compact_runs bit for bit, at odd destination offsets and with empty pairs; W-O > 63 is refused."""
    import torch
    import scrooge_amd
    dev = torch.device("cuda", 0)
    t, q = synth.make_pairs(300, 1500, "pacbio15", seed=77)
    q[5], t[9] = b"", b""                                   # no runs at all / insertions only
    n = len(t)
    tw, rw = (max(len(x) for x in t) + 31) // 32, (1500 + 31) // 32
    rows = np.zeros((n, (tw + rw) * 32), dtype=np.uint8)
    for k in range(n):
        rows[k, :len(t[k])] = np.frombuffer(t[k], dtype=np.uint8)
        rows[k, tw * 32: tw * 32 + len(q[k])] = np.frombuffer(q[k], dtype=np.uint8)
    ascii_t = torch.from_numpy(rows).to(dev)
    cap = (2 * 1500 + 8 + 15) // 16 * 16
    idx = torch.arange(n, dtype=torch.int64, device=dev)
    aligner.set_stream(0)
    try:
        seq = torch.zeros(n * (tw + rw) + scrooge_amd.api.SEQ_PAD_WORDS, dtype=torch.int64, device=dev)
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
        aligner.pack_planar(ascii_t.view(-1), seq, bad)
        desc = torch.stack([idx * (tw + rw) * 32, torch.tensor([len(x) for x in t], device=dev),
                            (idx * (tw + rw) + tw) * 32, torch.tensor([len(x) for x in q], device=dev),
                            idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
        runs = torch.zeros(n * cap * 2, dtype=torch.uint8, device=dev)
        ed = torch.empty(n, dtype=torch.int64, device=dev)
        nr = torch.empty(n, dtype=torch.int32, device=dev)
        st = torch.empty(n, dtype=torch.int32, device=dev)
        aligner.align_device(n, seq, desc, runs, ed, nr, st)
        cnt = nr.to(torch.int64)
        total = int(cnt.sum().item())
        for shift in (0, 1, 2, 3):                            # destination offsets of every alignment
            off = torch.cumsum(cnt, 0) - cnt + shift
            dense = torch.zeros((total + shift) * 2 + 8, dtype=torch.uint8, device=dev)
            packed = torch.zeros((total + shift + 7) // 8 * 8 + 8, dtype=torch.uint8, device=dev)
            back = torch.zeros((total + shift) * 2 + 8, dtype=torch.uint8, device=dev)
            aligner.compact_runs(n, desc, runs, nr, off, dense)
            aligner.compact_runs_packed(n, desc, runs, nr, off, packed)
            aligner.unpack_runs(total + shift, packed, back)
            torch.cuda.synchronize()
            assert torch.equal(back[2 * shift: 2 * (total + shift)], dense[2 * shift: 2 * (total + shift)])
            assert int(packed[total + shift:].max().item()) == 0            # nothing written past the end
        with pytest.raises(scrooge_amd.ScroogeError):
            aligner.compact_runs_packed(n, desc, runs, nr, off, packed, W=128, O=1)      # counts up to 127 do not fit 6 bits
    finally:
        aligner.use_own_stream()


@pytest.mark.parametrize("dec_kernel", ["lane", "wave", "quad"])
@pytest.mark.parametrize("W,O", [(64, 33), (64, 2), (40, 9), (128, 65), (200, 50), (256, 1), (256, 129), (192, 97), (128, 20)])
def test_edit_stream_round_trip(aligner, oracle, W, O, dec_kernel, monkeypatch):
    """(Both decoders — one pair per lane, one pair per wavefront: edit_stream_decode_kernel.hip — on the same streams.)
    scrg_encode_edit_stream (one byte per edit and per window end, the RCCL transfer format) against the definition of the
    format on the oracle's CIGARs, and scrg_decode_edit_stream back to the very runs scrg_compact_runs delivers; long
    error-free stretches, empty reads, empty texts, a tiny stream buffer."""
    monkeypatch.setenv("SCRG_DEC_KERNEL", dec_kernel)
    import torch
    import scrooge_amd
    from tests.test_edit_stream import py_encode
    dev = torch.device("cuda", 0)
    t, q = synth.make_pairs(300, 1500, "pacbio15", seed=78)
    t2, q2 = synth.make_pairs(40, 1400, "illumina", seed=79)          # stretches of > 63 matches between edits
    t, q = t + t2, q + q2
    q[5], t[9] = b"", b""                                   # no runs at all / insertions only
    q[11] = t[11][:1500]                                    # error-free: an empty stream
    n = len(t)
    tw, rw = (max(len(x) for x in t) + 31) // 32, (1500 + 31) // 32
    rows = np.zeros((n, (tw + rw) * 32), dtype=np.uint8)
    for k in range(n):
        rows[k, :len(t[k])] = np.frombuffer(t[k], dtype=np.uint8)
        rows[k, tw * 32: tw * 32 + len(q[k])] = np.frombuffer(q[k], dtype=np.uint8)
    ascii_t = torch.from_numpy(rows).to(dev)
    cap = (2 * 1500 + 8 + 15) // 16 * 16
    idx = torch.arange(n, dtype=torch.int64, device=dev)
    eds, cigars, _, _ = oracle.align(t, q, W=W, O=O)
    aligner.set_stream(0)
    try:
        seq = torch.zeros(n * (tw + rw) + scrooge_amd.api.SEQ_PAD_WORDS, dtype=torch.int64, device=dev)
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
        aligner.pack_planar(ascii_t.view(-1), seq, bad)
        desc = torch.stack([idx * (tw + rw) * 32, torch.tensor([len(x) for x in t], device=dev),
                            (idx * (tw + rw) + tw) * 32, torch.tensor([len(x) for x in q], device=dev),
                            idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
        runs = torch.zeros(n * cap * 2, dtype=torch.uint8, device=dev)
        ed = torch.empty(n, dtype=torch.int64, device=dev)
        nr = torch.empty(n, dtype=torch.int32, device=dev)
        st = torch.empty(n, dtype=torch.int32, device=dev)
        aligner.align_device(n, seq, desc, runs, ed, nr, st, W=W, O=O)
        assert ed.cpu().tolist() == eds
        want = [py_encode(c, W, O) for c in cigars]
        need = sum((len(w) + 3) // 4 * 4 for w in want)
        stream = torch.zeros(need + 64, dtype=torch.uint8, device=dev)
        s_off = torch.empty(n, dtype=torch.int64, device=dev)
        s_len = torch.empty(n, dtype=torch.int32, device=dev)
        tot = torch.empty(2, dtype=torch.int64, device=dev)
        aligner.encode_edit_stream(n, desc, runs, nr, stream, s_off, s_len, tot, W=W, O=O)
        torch.cuda.synchronize()
        assert tot.cpu().tolist() == [need, 0]
        assert int(stream[need:].max().item()) == 0                       # nothing written past the reserved bytes
        sh, oh, lh = stream.cpu().numpy().tobytes(), s_off.cpu().tolist(), s_len.cpu().tolist()
        assert sorted(o for o, l in zip(oh, lh) if l) == sorted(set(o for o, l in zip(oh, lh) if l))   # no two pairs share a start
        for k in range(n):
            assert oh[k] % 4 == 0 and sh[oh[k]: oh[k] + lh[k]] == want[k], k
            assert scrooge_amd.api.edit_stream_to_cigar(want[k], len(q[k]), W=W, O=O) == cigars[k]
        # back to runs on the GPU: counts first, then the runs themselves, identical to the compaction of the originals
        cnt32 = torch.empty(n, dtype=torch.int32, device=dev)
        nbad = torch.zeros(1, dtype=torch.int32, device=dev)
        aligner.decode_edit_stream(n, stream, s_off, s_len, desc.view(-1)[3:], 6, None, None, cnt32, nbad, W=W, O=O)
        assert torch.equal(cnt32, nr) and int(nbad.item()) == 0
        cnt = nr.to(torch.int64)
        total = int(cnt.sum().item())
        off = torch.cumsum(cnt, 0) - cnt
        dense = torch.zeros(total * 2 + 8, dtype=torch.uint8, device=dev)
        back = torch.zeros(total * 2 + 8, dtype=torch.uint8, device=dev)
        aligner.compact_runs(n, desc, runs, nr, off, dense)
        aligner.decode_edit_stream(n, stream, s_off, s_len, desc.view(-1)[3:], 6, off, back, cnt32, nbad, W=W, O=O)
        torch.cuda.synchronize()
        assert int(nbad.item()) == 0 and torch.equal(back, dense)
        # wrong read lengths are noticed (longer: more runs than the segment holds; shorter: the edits overrun the
        # read), and nothing is written past a pair's segment
        for rl, least in ((desc[:, 3].contiguous() + 1000, n), (desc[:, 3].contiguous() // 2, n // 2)):
            back.zero_()
            nbad.zero_()
            aligner.decode_edit_stream(n, stream, s_off, s_len, rl, 1, off, back, cnt32, nbad, W=W, O=O)
            assert int(nbad.item()) >= least and int(back[2 * total:].max().item()) == 0
        # the align kernel's own edit-stream output (scrg_align_device_edits): the same bytes in every pair's slice
        if True:              # (every W/O: genasm_lane_kernel<true> or genasm_lane_mw_kernel<.., true>)
            slices = torch.full((n * cap * 2,), 0xEE, dtype=torch.uint8, device=dev)
            ed2 = torch.empty(n, dtype=torch.int64, device=dev)
            ln2 = torch.empty(n, dtype=torch.int32, device=dev)
            st2 = torch.empty(n, dtype=torch.int32, device=dev)
            rc2 = torch.full((n,), -1, dtype=torch.int32, device=dev)
            aligner.align_device_edits(n, seq, desc, slices, ed2, ln2, st2, rc2, W=W, O=O)
            torch.cuda.synchronize()
            assert ed2.cpu().tolist() == eds and int(st2.max().item()) == 0 and ln2.cpu().tolist() == lh
            assert torch.equal(rc2, nr)                # the run count of the same alignment travels with the stream
            sl = slices.cpu().numpy().tobytes()
            for k in range(n):
                r4 = (lh[k] + 3) // 4 * 4
                assert sl[2 * k * cap: 2 * k * cap + r4] == want[k] + bytes(r4 - lh[k]), k      # zero up to the next dword
                assert sl[2 * k * cap + r4: 2 * k * cap + r4 + 4] in (b"\xee" * 4, b""), k        # and nothing after it
            # gathered with scrg_compact_runs (two bytes per "run") at 4-byte aligned offsets, then decoded
            r4 = (ln2.to(torch.int64) + 3) // 4 * 4
            boff = torch.cumsum(r4, 0) - r4
            dense_s = torch.zeros(int(r4.sum().item()) + 8, dtype=torch.uint8, device=dev)
            aligner.compact_runs(n, desc, slices, (r4 // 2).to(torch.int32), boff // 2, dense_s)
            back.zero_()
            nbad.zero_()
            aligner.decode_edit_stream(n, dense_s, boff, ln2, desc.view(-1)[3:], 6, off, back, nr, nbad, W=W, O=O)
            torch.cuda.synchronize()
            assert int(nbad.item()) == 0 and torch.equal(back, dense)
            # slices too small for the stream: reported, not overrun (the neighbouring slice stays intact)
            tiny = desc.clone()
            tiny[:, 5] = 16
            slices.fill_(0xEE)
            aligner.align_device_edits(n, seq, tiny, slices, ed2, ln2, st2, W=W, O=O)
            torch.cuda.synchronize()
            assert ln2.cpu().tolist() == lh and st2.cpu().tolist() == [1 if l > 32 else 0 for l in lh]
            sl = slices.cpu().numpy().tobytes()
            for k in range(n):
                assert sl[2 * k * cap: 2 * k * cap + min(32, (lh[k] + 3) // 4 * 4)] == (want[k] + bytes(3))[: min(32, (lh[k] + 3) // 4 * 4)], k
                assert sl[2 * k * cap + 32: 2 * (k + 1) * cap] == b"\xee" * (2 * cap - 32), k
        # streams that are not inside the buffer (offsets and lengths may come off a wire) are counted, never read; the
        # others still decode
        o_bad = s_off.clone()
        l_bad = s_len.clone()
        o_bad[0] = stream.numel() + 4096
        o_bad[1] = stream.numel() - 2
        l_bad[1] = 64
        l_bad[2] = 0x7fffffff
        back.zero_()
        nbad.zero_()
        aligner.decode_edit_stream(n, stream, o_bad, l_bad, desc.view(-1)[3:], 6, off, back, nr, nbad, W=W, O=O)
        torch.cuda.synchronize()
        lo = int(off[3].item())
        assert 1 <= int(nbad.item()) <= 3 and torch.equal(back[2 * lo: 2 * total], dense[2 * lo: 2 * total])
        assert int(back[2 * total:].max().item()) == 0
        with pytest.raises(scrooge_amd.ScroogeError):          # 16-byte alignment of the stream buffer
            aligner.decode_edit_stream(n, stream[4:], s_off, s_len, desc.view(-1)[3:], 6, off, back, nr, nbad, W=W, O=O)
        with pytest.raises(scrooge_amd.ScroogeError):          # the one-pair-per-lane kernels only
            aligner.align_device_edits(n, seq, desc, runs, ed, nr, st, W=W, O=O, lanes_per_pair=64)
        # a stream buffer that is too small: the pairs that do not fit are counted and marked, the others are intact
        small = torch.zeros(need // 2 // 4 * 4, dtype=torch.uint8, device=dev)
        aligner.encode_edit_stream(n, desc, runs, nr, small, s_off, s_len, tot, W=W, O=O)
        torch.cuda.synchronize()
        oh2, missing = s_off.cpu().tolist(), int(tot[1].item())
        assert missing > 0 and sum(1 for o in oh2 if o == -1) == missing and s_len.cpu().tolist() == lh
        sm = small.cpu().numpy().tobytes()
        assert all(sm[o: o + l] == w for o, l, w in zip(oh2, lh, want) if o != -1)
        # capacities that are not multiples of 4: a stream is stored as whole dwords, so a pair whose bytes
        # fit but whose dwords do not is "did not fit" — and nothing is ever written past the capacity handed over
        k0 = max(range(n), key=lambda k: lh[k])
        for capb in [c for c in range(max(lh[k0] - 5, 1), lh[k0] + 6) if c % 4]:
            guard = torch.full((capb + 4096 + 64,), 0xAB, dtype=torch.uint8, device=dev)
            aligner.encode_edit_stream(1, desc[k0:k0 + 1].contiguous(), runs, nr[k0:k0 + 1].contiguous(), guard[:capb], s_off, s_len, tot, W=W, O=O)
            torch.cuda.synchronize()
            gh = guard.cpu().numpy().tobytes()
            assert gh[capb:] == b"\xab" * (4096 + 64), capb
            fits = (lh[k0] + 3) // 4 * 4 <= capb
            assert (int(s_off[0].item()) == 0) == fits and int(tot[1].item()) == (0 if fits else 1), capb
            if fits:
                assert gh[:lh[k0]] == want[k0], capb
    finally:
        aligner.use_own_stream()


def test_mapping_shape_with_mixed_strides(aligner, oracle):
    """The read-mapping shape through the device API: one contiguous genome (text stride 1) shared by all candidates,
    the reads in lane-interleaved groups (read stride 64), both in one sequence array."""
    import torch
    import scrooge_amd
    dev = torch.device("cuda", 0)
    rng = np.random.Generator(np.random.PCG64(21))
    Gn = 50000
    genome = synth.random_seq(Gn, rng)
    gcodes = np.searchsorted(synth.BASES, np.frombuffer(genome, dtype=np.uint8)).astype(np.uint8)
    n = 777
    reads, starts, texts = [], [], []
    for _ in range(n):
        s = int(rng.integers(0, Gn - 400))
        r = synth.BASES[synth.mutate(gcodes[s:s + 220], 0.05, (23, 31, 46), rng)[:int(rng.integers(20, 180))]].tobytes()
        reads.append(r)
        starts.append(s + int(rng.integers(-2, 3)) if s > 2 else s)
        texts.append(genome[starts[-1]:])
    eds, cigars, _, _ = oracle.align([t[:600] for t in texts], reads, threads=8)
    G = scrooge_amd.api.GROUP
    gw = (Gn + 31) // 32
    rw = (180 + 31) // 32
    g_ascii = np.zeros(gw * 32, dtype=np.uint8)
    g_ascii[:Gn] = np.frombuffer(genome, dtype=np.uint8)
    r_ascii = np.zeros((n, rw * 32), dtype=np.uint8)
    for k in range(n):
        r_ascii[k, :len(reads[k])] = np.frombuffer(reads[k], dtype=np.uint8)
    n_groups = (n + G - 1) // G
    seq = torch.zeros(gw + n_groups * G * rw + scrooge_amd.api.SEQ_PAD_WORDS_GROUPS, dtype=torch.int64, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    cap = (2 * 180 + 8 + 15) // 16 * 16
    idx = torch.arange(n, dtype=torch.int64, device=dev)
    aligner.set_stream(0)
    try:
        aligner.pack_planar(torch.from_numpy(g_ascii).to(dev), seq[:gw], bad)
        aligner.pack_planar_groups(torch.from_numpy(r_ascii).to(dev).view(-1), n, rw, seq[gw:], bad)
        first = gw + (idx // G) * rw * G + idx % G
        st_t = torch.tensor(starts, dtype=torch.int64, device=dev)
        desc = torch.stack([st_t, Gn - st_t, first * 32, torch.tensor([len(r) for r in reads], device=dev),
                            idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
        got = _device_align(aligner, torch, seq, desc, n, cap, text_stride_words=1, read_stride_words=G)
        assert int(bad.item()) == 0
    finally:
        aligner.use_own_stream()
    assert got == (eds, cigars, [0] * n)


@pytest.mark.parametrize("fmt", ["edits", "edits-from-runs", "packed", "runs"])
def test_bench_gather_on_rccl_single_rank(fmt):
    """The gather path of bench.py on the real RCCL backend (a one-rank group, SCRG_BENCH_FORCE_GATHER): asynchronous
    collectives, four steps in flight, CIGARs as edit streams (the default: one collective per step, decoded and
    compared after the timed region), as packed runs (unpacked on a stream of its own ordered after the collective) or
    as scrg_run pairs; bench.py itself asserts that what rank 0 holds after the last step's gather is what its kernel
    produced."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCRG_BENCH_FORCE_GATHER="1", MASTER_PORT="29579")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "9", "--warmup", "2",
                          "--pairs", "20000", "--read-len", "3000", "--cpu-seconds", "0", "--gather-format", fmt],
                         env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith('{"metric"')][0])
    assert j["gather_check"] is True and j["value"] > 0 and j["config"]["gather"]["format"] == fmt
    assert j["config"]["backend"] == "nccl" and j["config"]["rccl_ranks"] == 1 and j["config"]["gather"]["root"] == "rank 0"
    # the contract step: CIGAR runs of every rank's pairs exist on the root when the clock stops (edit streams decoded inside
    # the timed region); the same steps without the decoding are reported next to it, never as the headline
    assert j["config"]["gather"]["decoded_to_runs_inside_timed_region"] == (fmt == "edits")
    if fmt == "edits":
        assert j["gather_without_decode"]["value"] > 0
    if fmt.startswith("edits"):
        assert 0.115 * 3000 < j["config"]["gather"]["stream_bytes_per_pair"] < 0.15 * 3000      # one byte per edit at 10 % error + one per window (3000 / 31)


def test_host_pipeline_under_random_calls():
    """The host entry points' pipeline (chunks, 4 or 8 slots, two collect threads, growing result arrays, device lists)
    under a series of calls of random sizes, read lengths, W/O, output selections and device lists, every result compared
    with the oracle (tests/tools/host_stress.py runs the long version: 550 calls clean)."""
    from tests.tools import host_stress
    host_stress.run(calls=24, seed=5, verbose=False)


@pytest.mark.parametrize("order", ["ascending", "descending"])
def test_host_results_grow_while_chunks_arrive(aligner, oracle, order):
    """A call with large result arrays (> 8 MB of runs, > 8 MB of text) in caller order (sort_by_length = 0).  Ascending read
    lengths: the first chunk's bytes per pair underestimate the call, so the result arrays are replaced (moved) while other
    chunks are being collected; descending: the estimate holds.  Every pair against the oracle, all three output selections."""
    t, q = [], []
    for L, n_distinct, rep in ((1000, 200, 20), (4000, 100, 30), (12000, 60, 50)):
        a, b = synth.make_pairs(n_distinct, L, "ont", seed=L + 5)
        t += a * rep
        q += b * rep
    if order == "descending":
        t, q = t[::-1], q[::-1]
    n = len(t)
    eds, cigars, _, _ = oracle.align(t, q, threads=16)
    want_text = b"".join(c.encode() + b"\0" for c in cigars)
    runs_of = {}
    for outputs in (0, 1, 2):
        r = aligner.align_pairs(t, q, arrays=True, outputs=outputs, sort_by_length=0)
        assert (r["edit_distance"] == np.array(eds)).all() and not r["status"].any()
        if outputs != 2:
            assert int(r["cigar_offset"][n]) == len(want_text) > (8 << 20)
            assert r["cigar_text"] == want_text, (order, outputs)
        if outputs != 1:
            assert int(r["run_offset"][n]) * 2 > (8 << 20)
            runs_of[outputs] = r["runs"].tobytes()
            ro = np.asarray(r["run_offset"]).astype(np.int64)
    assert runs_of[0] == runs_of[2]
    runs = np.frombuffer(runs_of[0], dtype=np.uint8).reshape(-1, 2)
    for i in list(range(0, n, 97)) + [n - 1]:
        assert "".join("%d%s" % (int(c), chr(int(o))) for c, o in runs[ro[i]:ro[i + 1]]) == cigars[i]


@pytest.mark.parametrize("dec_kernel", ["lane", "wave", "quad"])
def test_decode_large_launch_stores_pieces_together(aligner, dec_kernel, monkeypatch):
    """(dec_kernel: lane — the store form this test is named after — and wave: the decoder such a launch takes by itself,
    every wavefront a queue of pairs.)
    scrg_decode_edit_stream on more than 200 000 pairs in one launch — what the root of an N > 1 job decodes per step —
    stores the 64-byte pieces of the dense array by the wavefront together (edit_stream_decode_kernel.hip:
    write_whole_pieces), smaller launches lane by lane.  2 100 ragged pairs (empty reads and texts, error-free pairs among
    them) replicated 100 times through the offset arrays: every replica's runs must be those of scrg_compact_runs, and the
    small launch must agree with the large one."""
    monkeypatch.setenv("SCRG_DEC_KERNEL", dec_kernel)
    import torch
    import scrooge_amd
    dev = torch.device("cuda", 0)
    t, q = synth.make_pairs(1500, 1500, "ont", seed=91)
    t2, q2 = synth.make_pairs(600, 700, "pacbio15", seed=92)
    t, q = t + t2, q + q2
    q[3], t[8] = b"", b""
    q[12] = t[12][:1500]
    n0, R = len(t), 100
    tw, rw = (max(len(x) for x in t) + 31) // 32, (1500 + 31) // 32
    rows = np.zeros((n0, (tw + rw) * 32), dtype=np.uint8)
    for k in range(n0):
        rows[k, :len(t[k])] = np.frombuffer(t[k], dtype=np.uint8)
        rows[k, tw * 32: tw * 32 + len(q[k])] = np.frombuffer(q[k], dtype=np.uint8)
    cap = (2 * 1500 + 8 + 15) // 16 * 16
    idx = torch.arange(n0, dtype=torch.int64, device=dev)
    aligner.set_stream(0)
    try:
        seq = torch.zeros(n0 * (tw + rw) + scrooge_amd.api.SEQ_PAD_WORDS, dtype=torch.int64, device=dev)
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
        aligner.pack_planar(torch.from_numpy(rows).to(dev).view(-1), seq, bad)
        rl = torch.tensor([len(x) for x in q], device=dev)
        desc = torch.stack([idx * (tw + rw) * 32, torch.tensor([len(x) for x in t], device=dev), (idx * (tw + rw) + tw) * 32, rl,
                            idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
        runs = torch.zeros(n0 * cap * 2, dtype=torch.uint8, device=dev)
        ed = torch.empty(n0, dtype=torch.int64, device=dev)
        nr = torch.empty(n0, dtype=torch.int32, device=dev)
        st = torch.empty(n0, dtype=torch.int32, device=dev)
        aligner.align_device(n0, seq, desc, runs, ed, nr, st)
        assert int(st.max().item()) == 0
        stream = torch.zeros(n0 * 1024 + 64, dtype=torch.uint8, device=dev)
        s_off = torch.empty(n0, dtype=torch.int64, device=dev)
        s_len = torch.empty(n0, dtype=torch.int32, device=dev)
        tot = torch.empty(2, dtype=torch.int64, device=dev)
        aligner.encode_edit_stream(n0, desc, runs, nr, stream, s_off, s_len, tot)
        cnt = nr.to(torch.int64)
        total = int(cnt.sum().item())
        off = torch.cumsum(cnt, 0) - cnt
        want = torch.zeros(total * 2 + 8, dtype=torch.uint8, device=dev)
        aligner.compact_runs(n0, desc, runs, nr, off, want)
        nbad = torch.zeros(1, dtype=torch.int32, device=dev)
        small = torch.zeros(total * 2 + 8, dtype=torch.uint8, device=dev)
        aligner.decode_edit_stream(n0, stream, s_off, s_len, rl.contiguous(), 1, off, small, nr, nbad)
        assert int(nbad.item()) == 0 and torch.equal(small, want)
        n = n0 * R
        assert n > 200000
        cnt_rep = nr.repeat(R)
        c64 = cnt_rep.to(torch.int64)
        off_rep = torch.cumsum(c64, 0) - c64
        big = torch.zeros(R * total * 2 + 64, dtype=torch.uint8, device=dev)
        aligner.decode_edit_stream(n, stream, s_off.repeat(R), s_len.repeat(R), rl.repeat(R).contiguous(), 1, off_rep, big, cnt_rep, nbad)
        torch.cuda.synchronize()
        assert int(nbad.item()) == 0
        assert torch.equal(big[: R * total * 2].view(R, total * 2), want[: total * 2].unsqueeze(0).expand(R, -1))
        assert int(big[R * total * 2:].max().item()) == 0                 # nothing past the last pair's segment
    finally:
        aligner.use_own_stream()


@pytest.mark.parametrize("dec_kernel", ["lane", "wave", "quad"])
def test_decode_two_runs_per_step_fills_the_ring(aligner, dec_kernel, monkeypatch):
    """(dec_kernel: the lane-per-pair decoder, whose ring this is about, and the wavefront-per-pair one, for which the same
    streams are runs of up to 64 heads per chunk and edit runs that stay open over many chunks.)
    The decoder's worst case for its output ring (edit_stream_decode_kernel.hip: DEC_FLUSH_AT + 2 runs per step until the
    next look == DEC_RING, tied together by a static_assert): streams of "1 match, then an X" bytes commit TWO runs per step
    for hundreds of steps in a row ("1=1X1=1X..."), next to streams of "1 match, then an I / a D" and to ordinary ones, in both
    store forms (lane by lane: <= 200 000 pairs; 64-byte pieces by the wavefront together: more).  Expected runs: the same
    state machine on the host (scrg_edit_stream_to_runs_lane, held to the format's definition by tests/test_edit_stream.py)."""
    monkeypatch.setenv("SCRG_DEC_KERNEL", dec_kernel)
    import torch
    import scrooge_amd
    dev = torch.device("cuda", 0)
    api = scrooge_amd.api
    cases = []          # (stream bytes, read length)
    for k in (70, 129, 500, 2001):
        # no window ends until the very last byte: 2 runs per step for k steps in a row (the device decodes what the stream
        # says; the host decoder that checks the windows would refuse these)
        cases.append((bytes([0x41]) * k + b"\0", 2 * k))              # k x "1=1X"
        cases.append((bytes([0x81]) * k + b"\0", 2 * k))              # k x "1=1I"
        cases.append((bytes([0xC1]) * (k - 1) + b"\1", k))            # (k - 1) x "1=1D", then the last match
        cases.append((bytes([0x41, 0x81, 0xC1, 0x42]) * k + b"\0", 8 * k))    # mixed, 2 runs per step
        # the same alignments as an encoder writes them (a window end every 31 characters)
        for cig, rl_ in (("1=1X" * k, 2 * k), ("1=1I" * k, 2 * k), ("1=1D" * (k - 1) + "1=", k), ("1=1X1=1I1=1D2=1X" * k, 8 * k)):
            cases.append((api.cigar_to_edit_stream(cig), rl_))
    cases.append((b"", 0))
    cases.append((api.cigar_to_edit_stream("77="), 77))               # no edits: three window ends
    want_runs = []
    import re
    for st_, rl_ in cases:
        cig = api.edit_stream_to_cigar(st_, rl_, lane_form=True)
        want_runs.append(bytes(b for cnt, op in re.findall(r"(\d+)([=XID])", cig) for b in (int(cnt), ord(op))))
    assert want_runs[0] == bytes([1, ord("="), 1, ord("X")]) * 70 and len(want_runs[-1]) == 6
    n0 = len(cases)
    offs, blob = [], bytearray()
    for k_, (st_, _) in enumerate(cases):
        blob += bytes([0xC1] * (k_ % 5))                              # (streams start at any byte offset: the encoder's multiples of 4 are not required; the bytes between streams are not zeros)
        offs.append(len(blob))
        blob += st_
    stream = torch.tensor(list(blob) + [0] * 64, dtype=torch.uint8, device=dev)
    s_off = torch.tensor(offs, dtype=torch.int64, device=dev)
    s_len = torch.tensor([len(c[0]) for c in cases], dtype=torch.int32, device=dev)
    rl = torch.tensor([c[1] for c in cases], dtype=torch.int64, device=dev)
    cnt0 = torch.tensor([len(w) // 2 for w in want_runs], dtype=torch.int32, device=dev)
    want0 = torch.tensor(list(b"".join(want_runs)), dtype=torch.uint8, device=dev)
    total = int(want0.numel()) // 2
    aligner.set_stream(0)
    try:
        for R in (1, 40, 6000):                                      # 34, 1 360 and 204 000 pairs: both store forms
            cnt = cnt0.repeat(R)
            c64 = cnt.to(torch.int64)
            off = torch.cumsum(c64, 0) - c64
            got = torch.zeros(R * total * 2 + 64, dtype=torch.uint8, device=dev)
            counted = torch.zeros(R * n0, dtype=torch.int32, device=dev)
            nbad = torch.zeros(1, dtype=torch.int32, device=dev)
            aligner.decode_edit_stream(R * n0, stream, s_off.repeat(R), s_len.repeat(R), rl.repeat(R).contiguous(), 1, None, None, counted, nbad)
            torch.cuda.synchronize()
            assert int(nbad.item()) == 0 and torch.equal(counted, cnt), R
            aligner.decode_edit_stream(R * n0, stream, s_off.repeat(R), s_len.repeat(R), rl.repeat(R).contiguous(), 1, off, got, cnt, nbad)
            torch.cuda.synchronize()
            assert int(nbad.item()) == 0, R
            assert torch.equal(got[: R * total * 2].view(R, total * 2), want0.unsqueeze(0).expand(R, -1)), R
            assert int(got[R * total * 2:].max().item()) == 0
    finally:
        aligner.use_own_stream()


def test_decoders_agree_on_arbitrary_streams(aligner, monkeypatch):
    """All device decoders (and the two loops of each launch) on byte strings that mostly are NOT alignments (random bytes, long runs of one edit — 300 deletions
    in a row —, 0x3F stretches, streams that start at any byte offset): the verdict (clean or not), the run count and the runs of
    every clean stream must be those of the format's definition (tests/test_edit_stream.py: py_decode), nothing is written outside
    a pair's segment, and a segment that is one run short is reported and not overrun.  "auto" is the library's own choice, made on
    the device from a sample of the lengths (dec_sample_long): long streams, and short ones in a buffer sized for long ones."""
    import re
    import torch
    from tests.test_edit_stream import py_decode
    dev = torch.device("cuda", 0)
    rng = np.random.Generator(np.random.PCG64(99))
    aligner.set_stream(0)
    try:
        for decs, longest, pad in ((("quad", "wave", "lane", "plain", "auto"), 1200, 0), (("auto", "plain", "lane"), 40, 200)):
            _decoders_case(aligner, monkeypatch, torch, dev, rng, py_decode, re, decs, longest, pad)
    finally:
        aligner.use_own_stream()


def _decoders_case(aligner, monkeypatch, torch, dev, rng, py_decode, re, decs, longest, pad):
    streams, rls = [], []
    for k in range(1500):
        n = int(rng.integers(0, longest))
        kind = k % 5 if longest > 300 else k % 4
        if kind == 0:
            b = rng.integers(0, 256, n, dtype=np.uint8)
        elif kind == 1:
            b = rng.choice(np.array([0x40, 0x80, 0xC0, 0x00, 0x41, 0x3F], dtype=np.uint8), n, p=[.3, .2, .3, .05, .1, .05])
        elif kind == 2:
            b = rng.choice(np.array([0x40, 0x80, 0xC0], dtype=np.uint8), n, p=[.1, .1, .8])
        elif kind == 3:
            b = rng.integers(0, 64, n, dtype=np.uint8)
        else:
            b = np.frombuffer(bytes([0x41]) * int(rng.integers(0, 300)) + bytes([0xC0]) * int(rng.integers(250, 262)) + b"\x02", dtype=np.uint8)
        st = bytes(b) + (b"\0" if rng.integers(0, 4) else b"")
        placed = sum((x & 63) + (1 if (x >> 6) in (1, 2) else 0) for x in st)
        streams.append(st)
        rls.append(placed if rng.integers(0, 8) else placed + 1)
    want = [py_decode(s_, r_) for s_, r_ in zip(streams, rls)]
    n = len(streams)
    assert 100 < sum(w is not None for w in want) < n - 100
    offs, blob = [], bytearray()
    for k, st in enumerate(streams):
        blob += bytes([0xC1] * (k % 7 + pad))            # (any byte offset; what lies between the streams is not zeros)
        offs.append(len(blob))
        blob += st
    stream = torch.tensor(list(blob) + [0xC1] * 64, dtype=torch.uint8, device=dev)
    s_off = torch.tensor(offs, dtype=torch.int64, device=dev)
    s_len = torch.tensor([len(x) for x in streams], dtype=torch.int32, device=dev)
    rl = torch.tensor(rls, dtype=torch.int64, device=dev)
    runs_of = lambda c: [int(a) | (ord(o) << 8) for a, o in re.findall(r"(\d+)([=XID])", c)]
    if True:
        for dec in decs:
            if dec == "auto":
                monkeypatch.delenv("SCRG_DEC_KERNEL", raising=False)
            else:
                monkeypatch.setenv("SCRG_DEC_KERNEL", dec)
            counted = torch.zeros(n, dtype=torch.int32, device=dev)
            nbad = torch.zeros(1, dtype=torch.int32, device=dev)
            aligner.decode_edit_stream(n, stream, s_off, s_len, rl, 1, None, None, counted, nbad)
            torch.cuda.synchronize()
            ch = counted.cpu().numpy().astype(np.int64)
            assert int(nbad.item()) == sum(w is None for w in want), dec
            for k in range(n):
                assert (ch[k] == 0xffffffff - (1 << 32) or ch[k] == -1) == (want[k] is None), (dec, k)
                if want[k] is not None:
                    assert ch[k] == len(runs_of(want[k])), (dec, k)
            # the runs: every pair gets the segment its count asks for (a stream that is not clean: 40 runs), one in ten a run short
            seg = np.array([len(runs_of(w)) if w is not None else 40 for w in want], dtype=np.int64)
            short = (np.arange(n) % 10 == 3) & (seg > 0)
            seg_cap = seg - short
            off = np.cumsum(seg + 3) - (seg + 3)               # three guard runs behind every segment
            dense = torch.full((int(off[-1] + seg[-1] + 3) * 2 + 64,), 0xEE, dtype=torch.uint8, device=dev)
            nbad.zero_()
            aligner.decode_edit_stream(n, stream, s_off, s_len, rl, 1, torch.from_numpy(off).to(dev), dense,
                                       torch.from_numpy(seg_cap.astype(np.int32)).to(dev), nbad)
            torch.cuda.synchronize()
            dh = dense.cpu().numpy().view(np.uint16)
            assert int(nbad.item()) == sum(1 for k in range(n) if want[k] is None or short[k]), dec
            for k in range(n):
                assert (dh[off[k] + seg_cap[k]: off[k] + seg[k] + 3] == 0xEEEE).all(), (dec, k)      # nothing past the segment
                if want[k] is not None:
                    assert dh[off[k]: off[k] + seg_cap[k]].tolist() == runs_of(want[k])[: seg_cap[k]], (dec, k)


def _run_tool(args, timeout):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable] + args, cwd=root, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert lines, (out.stdout[-1500:], out.stderr[-3000:])
    j = json.loads(lines[-1])
    assert out.returncode == 0 and j["ok"], (j.get("checks"), out.stderr[-2000:])
    return j


def test_config4_full_size_one_gpu():
    """BASELINE configs[3] AT ITS SIZE — 1 M x 10 kb ONT-error pairs — on the one GPU there is, as the eight shards of 125 000 an
    8-GPU job aligns (tests/tools/config3_full.py): every shard through the real N > 1 step (edit-stream kernel + run counts,
    compaction into the wire buffer, the RCCL gather on a one-rank group) into one eight-slot receive buffer, ONE decode launch
    over all 1 M pairs.  Every pair is held to the size-independent properties (validateCigarString, src/tests.cu:27-169, as
    tensor arithmetic), edit distance == edit bytes of its stream, decoded runs == the runs kernel's bytes; 20 000 pairs
    spread over all eight shards run for run against the reference CPU path (src/genasm_cpu.cpp:411-438 via oracle/_ref).
    The same 1 M pairs then go through ONE scrg_align_pairs_multi call with eight logical devices and must give the same
    arrays."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    j = _run_tool([os.path.join(root, "tests", "tools", "config3_full.py"), "--multi"], timeout=1100)
    assert j["checks"]["properties_hold_for_every_pair"] and j["checks"]["run_for_run_with_the_reference"]
    assert j["checks"]["multi_runs_equal"] and j["reference_pairs"]["checked"] >= 20000
    assert 700 < j["mean_edit_distance"] < 1300 and j["total_runs"] > 1.9e9


def test_config5_full_size_one_gpu():
    """BASELINE configs[4] at >= 10 000 pairs: 10 240 x 50 kb PacBio-error (15 %) pairs, ~1 650 windows per pair
    (src/genasm_cpu.cpp:411-438), generated and packed on the GPU like bench.py's other_configs leg: every pair held to the
    size-independent properties, 1 500 pairs run for run against the reference CPU path, the edit-stream kernel decoded
    back to the same runs."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    j = _run_tool([os.path.join(root, "tests", "tools", "config3_full.py"), "--shards", "1", "--pairs", "10240", "--read-len", "50000",
                   "--profile", "pacbio15", "--ref-pairs", "1500", "--seed", "50"], timeout=900)
    assert j["checks"]["properties_hold_for_every_pair"] and j["checks"]["run_for_run_with_the_reference"]
    assert j["reference_pairs"]["checked"] >= 1500 and 5500 < j["mean_edit_distance"] < 11000
