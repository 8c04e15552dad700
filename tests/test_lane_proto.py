"""The lane-per-pair window arithmetic of the HIP kernel (genasm_lane_kernel.hip: Myers/Hyyro difference
vectors per text column, two traceback words per column, column-synchronous traceback with one
count-leading-zeros per insertion run), restated in C (tests/proto/lane_proto.c) and checked against the
oracle on the CPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from scrooge_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def proto():
    so = os.path.join(HERE, "proto", "liblane_proto.so")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-fopenmp", "-fPIC", "-shared", "-w", "-o", so,
                           os.path.join(HERE, "proto", "lane_proto.c")])
    return C.CDLL(so)


class LS(C.Structure):
    _fields_ = [("windows", C.c_uint64), ("columns", C.c_uint64), ("tb_columns", C.c_uint64)]


CODE = np.zeros(256, np.uint8)
CODE[ord("C")], CODE[ord("G")], CODE[ord("T")] = 1, 2, 3


def _run(fn, t, q, head, tail, unit=2):
    tc, qc = CODE[np.frombuffer(t, np.uint8)], CODE[np.frombuffer(q, np.uint8)]
    cap = 2 * (len(t) + len(q)) + 8          # (runs; as bytes of an edit stream at W-O = 1: one per edit and one per window)
    runs = (C.c_uint8 * (2 * cap))()
    n, ed = C.c_size_t(), C.c_longlong()
    st = fn(tc.ctypes.data_as(C.c_void_p), C.c_size_t(len(tc)), qc.ctypes.data_as(C.c_void_p), C.c_size_t(len(qc)),
            *head, runs, C.c_size_t(cap), C.byref(n), C.byref(ed), *tail)
    assert st == 0
    return ed.value, bytes(runs[:unit * n.value])


def _cases(seed):
    T, Q = [], []
    for prof, L, n in [("ont", 2000, 20), ("pacbio15", 2000, 12), ("illumina", 300, 30)]:
        t, q = synth.make_pairs(n, L, prof, seed=seed + L)
        T, Q = T + t, Q + q
    rng = np.random.Generator(np.random.PCG64(seed))
    for _ in range(40):                       # unrelated sequences, ragged and empty inputs
        T.append(synth.random_seq(int(rng.integers(0, 400)), rng))
        Q.append(synth.random_seq(int(rng.integers(0, 400)), rng))
    for _ in range(20):                       # low-complexity sequences: long insertion/deletion runs, many ties
        a = bytes(rng.choice(np.frombuffer(b"AC", np.uint8), int(rng.integers(1, 300))))
        b = bytes(rng.choice(np.frombuffer(b"AC", np.uint8), int(rng.integers(1, 300))))
        T.append(a), Q.append(b)
    T += [b"", b"ACGT", b"A" * 200, b"A" * 10, b"ACGT" * 50]
    Q += [b"ACGT", b"", b"A" * 10, b"A" * 200, b"TGCA" * 50]
    return T, Q


@pytest.mark.parametrize("W,O", [(64, 33), (64, 40), (64, 50), (64, 63), (32, 17), (48, 24), (33, 2), (2, 1), (17, 9), (64, 2), (64, 20), (50, 1),
                                 (16, 0), (31, 0), (24, 0), (40, 0), (63, 0), (64, 0)])           # O = 0: a window's traceback takes all W characters
def test_lane_form_matches_oracle(proto, W, O):
    T, Q = _cases(W * 100 + O)
    ls = LS()
    for t, q in zip(T, Q):
        got = _run(proto.lane_align_codes, t, q, (C.c_int(W), C.c_int(O)), (C.byref(ls),))
        want = _run(proto.go_align_codes, t, q, (C.c_int(W), C.c_int(O)), (None,))
        assert got == want
    assert ls.windows > 1000


@pytest.mark.parametrize("W,O", [(64, 33), (64, 40), (64, 63), (32, 17), (48, 24), (2, 1), (17, 9), (33, 2), (16, 0), (31, 0)])
def test_lane_edit_stream_form_matches_oracle(proto, oracle, W, O):
    """The kernel's edit-stream traceback (pending matches as mbase + column, every window closed by its END byte) produces the
    canonical edit stream of the oracle's CIGAR (tests/test_edit_stream.py: py_encode)."""
    from tests.test_edit_stream import py_encode
    T, Q = _cases(W * 100 + O + 7)
    T += [b"ACGT" * 300 + b"T" + b"ACGT" * 40, b"A" * 700]          # stretches of several hundred matches, then an edit
    Q += [b"ACGT" * 300 + b"G" + b"ACGT" * 40, b"A" * 700]
    eds, cigars, _, _ = oracle.align(T, Q, W=W, O=O)
    ls = LS()
    for t, q, e, c in zip(T, Q, eds, cigars):
        ed, stream = _run(proto.lane_align_edits, t, q, (C.c_int(W), C.c_int(O)), (C.byref(ls),), unit=1)
        assert ed == e and stream == py_encode(c, W, O)
    assert ls.windows > 1000


@pytest.mark.parametrize("W,O", [(128, 65), (96, 49), (80, 41), (256, 129), (192, 97), (128, 20), (200, 50), (65, 1), (128, 64), (129, 1)])
def test_lane_multiword_form_matches_oracle(proto, W, O):
    """The same table with multi-word vectors (W > 64: genasm_lane_mw_kernel.hip), rows of (W-O)/64 + 1 words."""
    T, Q = _cases(W * 100 + O + 3)
    ls = LS()
    for t, q in zip(T, Q):
        got = _run(proto.lane_align_codes_mw, t, q, (C.c_int(W), C.c_int(O)), (C.byref(ls),))
        want = _run(proto.go_align_codes, t, q, (C.c_int(W), C.c_int(O)), (None,))
        assert got == want
    assert ls.windows > 300


class LSB(C.Structure):
    _fields_ = [("windows", C.c_uint64), ("columns", C.c_uint64), ("tb_columns", C.c_uint64), ("escapes", C.c_uint64)]


@pytest.mark.parametrize("W,O", [(64, 2), (64, 32), (64, 16), (64, 1), (63, 20), (40, 5), (33, 1), (128, 65), (96, 49), (80, 41),
                                 (112, 57), (128, 96), (100, 40), (65, 2)])
def test_lane_band_form_matches_oracle(proto, W, O):
    """32 <= W-O <= 63, a design that was prototyped and NOT adopted: the table keeps 32 rows around the diagonal per
    column, and a walk that leaves the band in a column it is alive in has the window redone on the full rows.  The form is
    exact (checked here), but at W-O close to W — the reference's O = 2 sweep point — the traceback reaches the end of the
    window, where the insertion-first rule parks the insertions the window is forced to make (j - i up to +26, down to
    -12 on ONT-error reads): several per cent of the windows leave any band of 32 rows, i.e. most rounds of a wavefront
    of 64 pairs would take the slow path.  genasm_lane_wide_kernel.hip keeps full rows and builds the table in two
    halves instead (lane_align_codes_mw with one-word rows is its restatement)."""
    T, Q = _cases(W * 100 + O + 11)
    rng = np.random.Generator(np.random.PCG64(W + O))
    for _ in range(30):                       # long gaps: the walk crosses the band's edges
        a = synth.random_seq(int(rng.integers(200, 600)), rng)
        cut, gap = int(rng.integers(10, 150)), int(rng.integers(10, 60))
        T += [a, a[:cut] + a[cut + gap:]]
        Q += [a[:cut] + a[cut + gap:], a]
    ls, related = LSB(), LSB()
    for k, (t, q) in enumerate(zip(T, Q)):
        st = related if k < 62 else ls        # the first 62 cases of _cases(): ont / pacbio15 / illumina pairs
        got = _run(proto.lane_align_codes_band, t, q, (C.c_int(W), C.c_int(O)), (C.byref(st),))
        want = _run(proto.go_align_codes, t, q, (C.c_int(W), C.c_int(O)), (None,))
        assert got == want, k
    assert related.windows > 300 and ls.escapes > 0
    if W - O <= W // 2:                       # the traceback stays away from the window's end: the band holds related reads
        assert related.escapes <= related.windows // 200, (related.escapes, related.windows)
    elif (W, O) in ((64, 2), (64, 1)):        # ... and here it does not
        assert related.escapes > related.windows // 300, (related.escapes, related.windows)


class LSB32(C.Structure):
    _fields_ = [("windows", C.c_uint64), ("banded", C.c_uint64), ("escapes", C.c_uint64), ("mismatching_safe_windows", C.c_uint64),
                ("hist", C.c_uint64 * 66)]


def test_band32_table_is_exact_where_it_claims_and_escapes_too_often(proto):
    """Round 6's step-change experiment on the default kernel, prototyped and NOT adopted: the window's table computed in a 32-row
    band around the main diagonal (one dword per vector: 10 instead of 19 instructions per column; tests/proto/lane_proto.c,
    lane_dc_band32).  The band's own D[0][0] <= 14 proves a window safe, and on every safe window the banded table gives the walk
    of the full one (checked here: 0 differences) — but 1.6 % of the WINDOWS of 10 % ONT-error reads are not safe (the window
    distance has a long tail: mean 7.3, 1 in 60 above 14), and a wavefront redoes its round on the full table when ANY of its 64
    lanes is not: 65 % of the rounds (PacBio 15 %: 7.4 % of the windows, 99 % of the rounds).  A band wide enough for 1 % of the
    rounds needs |i - j| <= 20: 41 rows, two dwords again."""
    ls = LSB32()
    T, Q = synth.make_pairs(120, 4000, "ont", seed=77)
    t2, q2 = synth.make_pairs(40, 4000, "pacbio15", seed=78)
    n_ont_windows = None
    for k, (t, q) in enumerate(zip(T + t2, Q + q2)):
        if k == 120:
            n_ont_windows, ont_escapes = ls.windows, ls.escapes
        got = _run(proto.lane_align_codes_band32, t, q, (C.c_int(64), C.c_int(33)), (C.byref(ls),))
        want = _run(proto.go_align_codes, t, q, (C.c_int(64), C.c_int(33)), (None,))
        assert got == want, k
    assert ls.mismatching_safe_windows == 0 and ls.banded > 15000
    p_ont = ont_escapes / n_ont_windows
    assert 0.008 < p_ont < 0.03                                  # ~1.6 % of the windows ...
    assert 1 - (1 - p_ont) ** 64 > 0.4                           # ... i.e. most rounds of a wavefront of 64 lanes
    p_pb = (ls.escapes - ont_escapes) / (ls.windows - n_ont_windows)
    assert p_pb > 0.04
