"""Stress of the host entry points' pipeline (chunks, slots, two collect threads, growing result arrays, multi-device
lists): many calls of random sizes, read lengths, W/O, output selections and device lists; every result is compared with
the oracle.  usage: python tests/tools/host_stress.py [calls=150] [seed=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scrooge_amd
from scrooge_amd import synth
from oracle.pyoracle import Oracle



RC = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def run(calls=150, seed=1, verbose=True):
    rng = np.random.Generator(np.random.PCG64(seed))
    orc = Oracle()
    a = scrooge_amd.Aligner(0)
    pool = {}
    for L, prof, n in ((150, "illumina", 60000), (1000, "ont", 12000), (5000, "ont", 3000), (12000, "pacbio15", 600)):
        pool[L] = synth.make_pairs(n, L, prof, seed=seed * 7 + L)
    genome = synth.random_seq(3_000_000, rng)
    t0 = time.time()
    for it in range(calls):
        kind = int(rng.integers(0, 4))
        W, O = [(64, 33), (64, 33), (64, 33), (64, 2), (128, 65), (48, 24), (64, 40), (96, 49), (192, 97), (256, 129),
                (16, 0), (31, 0), (40, 0), (64, 0)][int(rng.integers(0, 14))]            # (O = 0: the reference's no-overlap build)
        outputs = int(rng.integers(0, 3))
        devices = [None, None, [0, 0], [0, 0, 0]][int(rng.integers(0, 4))]
        sort = int(rng.integers(0, 2))
        if kind < 3:
            # pairs of mixed lengths
            T, Q = [], []
            for L in pool:
                t, q = pool[L]
                k = int(rng.integers(0, min(len(t), [20000, 4000, 1200, 200][list(pool).index(L)]) + 1))
                s = int(rng.integers(0, len(t) - k + 1))
                T += t[s:s + k]; Q += q[s:s + k]
            if not T:
                continue
            perm = rng.permutation(len(T))
            T = [T[i] for i in perm]; Q = [Q[i] for i in perm]
            eds, cigars, _, _ = orc.align(T, Q, W=W, O=O, threads=16)
            if devices:
                r = a.align_pairs_multi(devices, T, Q, arrays=True, W=W, O=O, outputs=outputs, sort_by_length=sort)
            else:
                r = a.align_pairs(T, Q, arrays=True, W=W, O=O, outputs=outputs, sort_by_length=sort)
        else:
            # read mapping against the 3 Mbp genome
            nr = int(rng.integers(1, 30000))
            starts = rng.integers(0, len(genome) - 400, nr)
            reads = [genome[int(s):int(s) + 150] for s in starts]
            cands = [[int(s), max(0, int(s) - 2), int(rng.integers(0, len(genome) - 10))] for s in starts]
            # (two calls in three carry minus-strand candidates: the read's reverse complement is aligned — on the device from the one
            # packed copy of the read when the geometry is the default one, from a reverse-complemented row otherwise)
            revs = [[int(rng.integers(0, 2)) for _ in cs] for cs in cands] if rng.integers(0, 3) else None
            T = [genome[c:c + 400] for cs in cands for c in cs]
            Q = [(reads[i] if not (revs and revs[i][k]) else reads[i].translate(RC)[::-1]) for i, cs in enumerate(cands) for k in range(len(cs))]
            eds, cigars, _, _ = orc.align(T, Q, W=W, O=O, threads=16)
            kw_rev = {"reverse": revs} if revs else {}
            if revs and not devices:
                devices = [0]                  # (the binding passes strands through the device-list entry point)
            if devices:
                r = a.align_mapping_multi(devices, genome, reads, cands, arrays=True, W=W, O=O, outputs=outputs, sort_by_length=sort, **kw_rev)
            else:
                r = a.align_mapping(genome, reads, cands, arrays=True, W=W, O=O, outputs=outputs, sort_by_length=sort, **kw_rev)
        n = len(eds)
        assert (r["edit_distance"] == np.array(eds)).all(), (it, "edit distances")
        assert not r["status"].any()
        if outputs != 2:
            off = r["cigar_offset"]
            got = [r["cigar_text"][int(off[i]):int(off[i + 1]) - 1].decode() for i in range(n)]
            bad = [i for i in range(n) if got[i] != cigars[i]]
            assert not bad, (it, "text", bad[:3], got[bad[0]][:60], cigars[bad[0]][:60])
        if outputs != 1:
            ro = r["run_offset"].astype(np.int64)
            runs = r["runs"]
            for i in list(rng.integers(0, n, 200)) + [0, n - 1]:
                s = "".join("%d%s" % (int(c), chr(int(o))) for c, o in runs[ro[i]:ro[i + 1]])
                assert s == cigars[i], (it, "runs", i)
        if verbose and it % 10 == 0:
            print("call %d: %d pairs, kind %d W=%d O=%d outputs=%d devices=%s sort=%d ok (%.0f s)" % (it, n, kind, W, O, outputs, devices, sort, time.time() - t0), flush=True)
    if verbose:
        print("host stress ok: %d calls" % calls)


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 150, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
