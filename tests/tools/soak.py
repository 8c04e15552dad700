"""Randomised soak of the default path against the oracle: many pairs of mixed lengths and error
rates (0-30 %), unrelated pairs, homopolymers and ragged text ends, every CIGAR compared.
usage: python tests/tools/soak.py [n_pairs] [seed]"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import numpy as np
import scrooge_amd
from scrooge_amd import synth
from oracle.pyoracle import Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.Generator(np.random.PCG64(seed))
T, Q = [], []
t0 = time.time()
while len(T) < n:
    L = int(rng.choice([30, 150, 400, 1000, 3000])) + int(rng.integers(0, 64))
    err = float(rng.choice([0.0, 0.01, 0.05, 0.1, 0.15, 0.2, 0.3]))
    ratio = [(1, 1, 1), (23, 31, 46), (6, 50, 54), (90, 5, 5)][int(rng.integers(0, 4))]
    k = int(rng.integers(0, 10))
    if k == 0:                                   # unrelated
        T.append(synth.random_seq(L + int(rng.integers(0, 100)), rng)); Q.append(synth.random_seq(L, rng))
    elif k == 1:                                 # homopolymer / low complexity
        a = b"ACGT"[int(rng.integers(0, 4))]; T.append(bytes([a]) * (L + 20)); Q.append(bytes([a]) * int(L * rng.uniform(0.7, 1.3)))
    else:
        slack = float(rng.choice([0.0, 0.02, 0.15, 0.5]))   # 0: the text ends with (or before) the read
        t, q = synth.make_pair(L, err, ratio, rng, slack)
        T.append(synth.BASES[t].tobytes()); Q.append(synth.BASES[q].tobytes())
print("generated %d pairs in %.1fs" % (len(T), time.time() - t0))
t0 = time.time()
eds, cigars, _, _ = Oracle().align(T, Q, threads=16)
print("oracle %.1fs" % (time.time() - t0))
a = scrooge_amd.Aligner(0)
for kw in ({}, {"sort_by_length": 0}, {"lanes_per_pair": 8, "lds_rows": 7}):
    t0 = time.time()
    got = a.align_pairs(T, Q, **kw)
    bad = [i for i in range(len(T)) if got[i].edit_distance != eds[i] or got[i].cigar != cigars[i]]
    print(kw, "gpu %.1fs" % (time.time() - t0), "mismatches:", len(bad), bad[:5])
    assert not bad
# other window settings of the one-pair-per-lane kernels, on a slice of the same pairs
m = min(len(T), 20000)
for W, O in ((64, 63), (64, 40), (33, 2), (48, 24), (32, 17), (17, 9), (5, 2), (2, 1), (63, 32), (40, 9),
             (64, 2), (64, 20), (64, 1), (64, 32), (40, 5), (33, 1), (50, 18),         # this row: W-O > 31
             (128, 65), (96, 49), (256, 129), (200, 50), (130, 1),                       # W > 64: multi-word vectors
             (160, 81), (192, 97), (128, 20), (224, 113), (255, 128)):                   # 64 <= W-O <= 127: the table in parts
    e2, c2, _, _ = Oracle().align(T[:m], Q[:m], W=W, O=O, threads=16)
    got = a.align_pairs(T[:m], Q[:m], W=W, O=O)
    bad = [i for i in range(m) if got[i].edit_distance != e2[i] or got[i].cigar != c2[i]]
    print("W=%d O=%d lanes=%d" % (W, O, a.resolved_params(W=W, O=O).lanes_per_pair), "mismatches:", len(bad), bad[:5])
    assert not bad
print("soak ok")
# the align kernel's edit-stream output (scrg_align_device_edits) on the same pairs: every stream must be the canonical
# encoding of the oracle's CIGAR (long insertion runs and match stretches take the kernel's side path)
import torch
from tests.test_edit_stream import py_encode
dev = torch.device("cuda", 0)
a.set_stream(0)
for W, O in ((64, 33), (48, 24), (33, 2), (64, 63), (64, 2), (50, 18), (128, 65), (256, 100), (256, 129), (160, 81)):
    m = min(len(T), 12000)
    e2, c2, _, _ = Oracle().align(T[:m], Q[:m], W=W, O=O, threads=16)
    tw = (max(len(x) for x in T[:m]) + 31) // 32
    rw = (max(len(x) for x in Q[:m]) + 31) // 32
    rows = np.zeros((m, (tw + rw) * 32), dtype=np.uint8)
    for k in range(m):
        rows[k, :len(T[k])] = np.frombuffer(T[k], dtype=np.uint8)
        rows[k, tw * 32: tw * 32 + len(Q[k])] = np.frombuffer(Q[k], dtype=np.uint8)
    seq = torch.zeros(m * (tw + rw) + scrooge_amd.api.SEQ_PAD_WORDS, dtype=torch.int64, device=dev)
    nbad = torch.zeros(1, dtype=torch.int32, device=dev)
    a.pack_planar(torch.from_numpy(rows).to(dev).view(-1), seq, nbad)
    cap = (max(len(x) for x in T[:m]) + max(len(x) for x in Q[:m]) + 8 + 15) // 16 * 16
    idx = torch.arange(m, dtype=torch.int64, device=dev)
    desc = torch.stack([idx * (tw + rw) * 32, torch.tensor([len(x) for x in T[:m]], device=dev),
                        (idx * (tw + rw) + tw) * 32, torch.tensor([len(x) for x in Q[:m]], device=dev),
                        idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
    slices = torch.zeros(m * cap * 2, dtype=torch.uint8, device=dev)
    ed = torch.empty(m, dtype=torch.int64, device=dev)
    ln = torch.empty(m, dtype=torch.int32, device=dev)
    st = torch.empty(m, dtype=torch.int32, device=dev)
    a.align_device_edits(m, seq, desc, slices, ed, ln, st, W=W, O=O)
    torch.cuda.synchronize()
    sl, lh = slices.cpu().numpy(), ln.cpu().tolist()
    assert int(st.max()) == 0 and ed.cpu().tolist() == e2
    bad = [k for k in range(m) if sl[2 * k * cap: 2 * k * cap + lh[k]].tobytes() != py_encode(c2[k], W, O)]
    print("edit streams W=%d O=%d: %d pairs, %d bytes, mismatches: %d %s" % (W, O, m, sum(lh), len(bad), bad[:5]))
    assert not bad
    for k in range(0, m, 97):
        assert scrooge_amd.api.edit_stream_to_cigar(sl[2 * k * cap: 2 * k * cap + lh[k]].tobytes(), len(Q[k]), W=W, O=O) == c2[k]
a.use_own_stream()
print("edit-stream soak ok")
