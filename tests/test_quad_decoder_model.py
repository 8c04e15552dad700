"""The formulation of decode_edits_quad_kernel (four stream bytes per lane, run slots filled by additions) held to the
format's definition on the CPU, through its lane-by-lane model tests/tools/quad_decoder_model.py: golden fixtures at every
W/O the reference was built with, random byte strings (most of them malformed), long edit runs, 0x3F stretches, segments at
every offset of the dense array, capacities one short."""
import glob
import json
import os
import re

import numpy as np
import pytest

from scrooge_amd import api
from tests.test_edit_stream import py_decode, GOLDEN
from tests.tools.quad_decoder_model import Model


def runs_of(cigar):
    return [int(c) | (ord(o) << 8) for c, o in re.findall(r"(\d+)([=XID])", cigar)]


def check(streams, read_lens, g0=0, expect_fast=None):
    """Decode the streams back to back into one dense array starting at run g0; compare with py_decode."""
    want = [py_decode(s, rl) for s, rl in zip(streams, read_lens)]
    m_cnt = Model(8)
    counts = []
    for s, rl, w in zip(streams, read_lens, want):
        n, clean = m_cnt.decode_pair(s, rl, 0, 0, store=False)
        assert clean == (w is not None), (s[:40], rl)
        if clean:
            assert n == len(runs_of(w))
        counts.append(n)
    total = sum(counts)
    m = Model(g0 + total + 16)
    at = g0
    for s, rl, w, n in zip(streams, read_lens, want, counts):
        n2, clean = m.decode_pair(s, rl, at, n, store=True)
        assert n2 == n and clean == (w is not None)
        if clean:
            assert m.dense[at:at + n].tolist() == runs_of(w), (s[:40], rl)
            assert m.written[at:at + n].all()
        at += n
    assert not m.written[:g0].any() and not m.written[at:].any()
    return m


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "pairs_*.json")))[:6], ids=os.path.basename)
def test_goldens(path):
    with open(path) as f:
        g = json.load(f)
    W, O = g.get("W", 64), g.get("O", 33)
    cases = g["cases"] if "cases" in g else g["pairs"]
    streams, rls = [], []
    for c in cases[:120]:
        cig = c["cigar"]
        streams.append(api.cigar_to_edit_stream(cig, W=W, O=O))
        rls.append(sum(int(n) for n, o in re.findall(r"(\d+)([=XID])", cig) if o != "D"))
    for g0 in (0, 5):
        check(streams, rls, g0)


def test_random_bytes_and_offsets():
    rng = np.random.Generator(np.random.PCG64(7))
    streams, rls = [], []
    for _ in range(300):
        n = int(rng.integers(0, 900))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            b = rng.integers(0, 256, n, dtype=np.uint8)
        elif kind == 1:          # few distinct bytes: long runs of the same edit, many joins
            b = rng.choice(np.array([0x40, 0x80, 0xC0, 0x00, 0x41, 0x3F], dtype=np.uint8), n, p=[.3, .2, .3, .05, .1, .05])
        elif kind == 2:          # edits only
            b = rng.choice(np.array([0x40, 0x80, 0xC0], dtype=np.uint8), n, p=[.1, .1, .8])
        else:
            b = rng.integers(0, 64, n, dtype=np.uint8)
        s = bytes(b) + (b"\0" if rng.integers(0, 4) else b"")
        w = None
        # the read length that makes the stream clean, if it is otherwise well formed
        placed = sum((x & 63) + (1 if (x >> 6) in (1, 2) else 0) for x in s)
        rls.append(placed if rng.integers(0, 8) else placed + 1)
        streams.append(s)
    for g0 in range(0, 9):
        check(streams[g0 * 30:(g0 + 1) * 30 + 30], rls[g0 * 30:(g0 + 1) * 30 + 30], g0)


@pytest.mark.parametrize("k", [3, 4, 5, 8, 63, 64, 250, 251, 254, 255, 256, 257, 300, 520, 1100])
def test_long_edit_runs(k):
    """k deletions in a row, at every phase of the 4-byte lanes and the 256-byte chunks: a run of 255 is the longest there is"""
    for lead in (0, 1, 2, 3, 5, 250, 253, 255, 256, 258):
        s = bytes([0x41]) * lead + bytes([0xC0]) * k + b"\x02" + b"\0"
        m = check([s], [2 * lead + 2])
        assert (py_decode(s, 2 * lead + 2) is None) == (k > 255)
    # the run continues over a chunk border after chunks without a long run
    s = bytes([0x41, 0x00]) * 126 + bytes([0x40]) * k + b"\0"
    check([s], [126 + k])


def test_more_bytes():
    cases = [(b"\x3f" * a + bytes([b]) + b"\0" * z, 63 * a + (b & 63) + (1 if (b >> 6) in (1, 2) else 0))
             for a in (0, 1, 2, 3, 4, 5, 63, 64, 65, 260) for b in (0x00, 0x05, 0x45, 0xC0, 0x3E) for z in (0, 1)]
    streams = [c[0] for c in cases] + [b"\x41" * 300 + c[0] for c in cases]
    rls = [c[1] for c in cases] + [600 + c[1] for c in cases]
    check(streams, rls, 3)
    # canonical long stretches (W-O > 63)
    for W, O in ((128, 1), (256, 1), (200, 50)):
        cig = "300=1X127=2D500=1I40="
        s = api.cigar_to_edit_stream(cig, W=W, O=O)
        check([s, s], [969, 969], 1)


def test_capacity_one_short_and_fast_path_is_the_common_one():
    cig = "5=1X20=2I7=1D" * 200
    s = api.cigar_to_edit_stream(cig)
    rl = sum(int(n) for n, o in re.findall(r"(\d+)([=XID])", cig) if o != "D")
    m = check([s], [rl], 6)
    assert m.slow_chunks == 0 and m.fast_chunks >= 3
    n = len(runs_of(py_decode(s, rl)))
    m2 = Model(n + 32)
    n2, clean = m2.decode_pair(s, rl, 7, n - 3, store=True)      # a segment three runs short: nothing past it
    assert n2 == n and m2.written[7:7 + n - 3].all() and not m2.written[7 + n - 3:].any() and not m2.written[:7].any()
