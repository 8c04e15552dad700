"""The batch form of the property checker (tests/cigar_check.py: validate_batch — what the full-size GPU tests hold every
pair of a million to) against the per-pair form, on the CPU: clean oracle results pass, and of deliberately corrupted
results exactly those the per-pair checker rejects are reported."""
import random

import numpy as np
import torch

from scrooge_amd import synth
from tests.cigar_check import parse, validate, validate_batch


def _rows(texts, reads):
    n, tw, rw = len(texts), max(map(len, texts)), max(map(len, reads))
    rows = np.zeros((n, tw + rw), np.uint8)
    for k in range(n):
        rows[k, :len(texts[k])] = np.frombuffer(texts[k], np.uint8)
        rows[k, tw:tw + len(reads[k])] = np.frombuffer(reads[k], np.uint8)
    return torch.from_numpy(rows), tw


def _pack(cigars):
    runs, off, cnt = [], [], []
    for c in cigars:
        r = parse(c)
        off.append(len(runs) // 2)
        cnt.append(len(r))
        for a, b in r:
            runs += [a, ord(b)]
    return torch.tensor(runs + [0, 0], dtype=torch.uint8), torch.tensor(off), torch.tensor(cnt)


def test_batch_checker_agrees_with_the_per_pair_checker(oracle):
    t, q = synth.make_pairs(200, 700, "ont", seed=3)
    t += [b"ACGT", b"AAAA", b"ACGTACGT", b"acgtacgt"]
    q += [b"", b"AAAA", b"ACGAACG", b"ACGTACG"]
    eds, cigars, _, _ = oracle.align(t, q)
    rows, tw = _rows(t, q)
    tl, rl = torch.tensor([len(x) for x in t]), torch.tensor([len(x) for x in q])
    runs, off, cnt = _pack(cigars)
    assert all(validate(a, b, c, e) is None for a, b, c, e in zip(t, q, cigars, eds))
    assert validate_batch(torch, rows, 0, tl, tw, rl, runs, off, cnt, torch.tensor(eds), chunk_pairs=37).numel() == 0
    rnd = random.Random(1)
    want, cig2, ed2 = set(), list(cigars), list(eds)
    for k in rnd.sample(range(200), 60):
        r = parse(cig2[k])
        kind = rnd.randrange(5)
        if kind == 0:
            ed2[k] += 1
        elif kind == 1:
            i = rnd.randrange(len(r))
            r[i] = (r[i][0], {"=": "X", "X": "=", "I": "D", "D": "I"}[r[i][1]])
        elif kind == 2:
            i = rnd.randrange(len(r))
            r[i] = (r[i][0] + 1, r[i][1])
        elif kind == 3:
            r = r[:-1] if len(r) > 1 else r + [(1, "I")]
        else:
            r = r + [(5, "D")]                     # past the end of the text for most pairs, and one more edit for all
        cig2[k] = "".join("%d%s" % x for x in r)
        if validate(t[k], q[k], cig2[k], ed2[k]) is not None:
            want.add(k)
    runs, off, cnt = _pack(cig2)
    got = set(validate_batch(torch, rows, 0, tl, tw, rl, runs, off, cnt, torch.tensor(ed2), chunk_pairs=50).tolist())
    assert got == want and len(want) >= 50
