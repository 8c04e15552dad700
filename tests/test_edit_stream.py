"""CPU tests of the edit-stream transfer format (include/scrooge_amd.h, scrooge_amd/csrc/edit_stream.h): the
host encoder and the window replay of the decoder, against the reference's own CIGARs.

The decoder has to put the window breaks back (runs are flushed per window and never merged,
genasm_cpu.cpp:304-305, 400-403); these tests pin that to every committed golden fixture — all W/O the
reference was built with — and to fresh oracle output on random, ragged and degenerate pairs."""
import glob
import json
import os
import re

import numpy as np
import pytest

from scrooge_amd import api, synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def py_encode(cigar):
    """Independent restatement of the canonical encoding, straight from the format's definition."""
    code = {"X": 1, "I": 2, "D": 3}
    out = bytearray()
    pend = 0
    for cnt, op in re.findall(r"(\d+)([=XID])", cigar):
        c = int(cnt)
        if op == "=":
            pend += c
            continue
        out += b"\x3f" * (pend >> 6)
        out.append(code[op] << 6 | (pend & 63))
        out += bytes([code[op] << 6]) * (c - 1)
        pend = 0
    return bytes(out)


def round_trip(cigar, read_len, ed, W, O):
    s = api.cigar_to_edit_stream(cigar)
    assert s == py_encode(cigar)
    assert sum(1 for b in s if b >> 6) == ed          # one byte per edit (+ the long-match bytes, op 0)
    assert api.edit_stream_to_cigar(s, read_len, W=W, O=O) == cigar
    # the same through the state machine the GPU decoder runs in every lane (edit_stream.h: decode_lane_step)
    assert api.edit_stream_to_cigar(s, read_len, W=W, O=O, lane_form=True) == cigar


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "pairs_*.json"))), ids=os.path.basename)
def test_golden_pairs_round_trip(path):
    g = json.load(open(path))
    for c in g["cases"]:
        round_trip(c["cigar"], len(c["read"]), c["ed"], g["W"], g["O"])


def test_golden_mapping_round_trip(golden_mapping):
    g = golden_mapping
    k = 0
    for r, cands in zip(g["reads"], g["candidates"]):
        for _ in cands:
            round_trip(g["cigar"][k], len(r), g["ed"][k], 64, 33)
            k += 1
    assert k == len(g["cigar"])


@pytest.mark.parametrize("W,O", [(64, 33), (64, 1), (64, 63), (32, 17), (5, 2), (2, 1), (128, 65), (256, 20), (50, 18)])
def test_oracle_round_trip(oracle, W, O):
    rng = np.random.Generator(np.random.PCG64(100 * W + O))
    T, Q = [], []
    for L, err in ((0, 0.0), (1, 0.0), (63, 0.0), (64, 0.0), (65, 0.0), (300, 0.0), (300, 0.02), (1000, 0.1), (700, 0.3),
                   (2500, 0.15), (129, 0.05)):
        for slack in (0.0, 0.2):
            t, q = synth.make_pair(L, err, (23, 31, 46), rng, slack) if L else (np.zeros(5, np.uint8), np.zeros(0, np.uint8))
            T.append(synth.BASES[t].tobytes()); Q.append(synth.BASES[q].tobytes())
    T += [b"A" * 500, b"", b"ACGT" * 100, synth.random_seq(400, rng)]
    Q += [b"A" * 700, b"ACGTACGTAC" * 30, b"ACGT" * 100, synth.random_seq(400, rng)]      # text runs out; empty text; exact; unrelated
    eds, cigars, _, _ = oracle.align(T, Q, W=W, O=O)
    for q, e, c in zip(Q, eds, cigars):
        round_trip(c, len(q), e, W, O)


def test_long_match_stretches():
    # 64 q + r matches before an edit: q bytes 0x3F, then the edit byte carries r
    for p in (0, 1, 62, 63, 64, 65, 127, 128, 129, 1000):
        cig = ("%d=" % p if p else "") + "1X5="
        s = api.cigar_to_edit_stream(cig)
        assert s == b"\x3f" * (p >> 6) + bytes([1 << 6 | (p & 63)])
        # W-O = 31: the decoder restores the window breaks
        want = api.edit_stream_to_cigar(s, p + 6)
        assert api.edit_stream_to_cigar(s, p + 6, lane_form=True) == want
        runs = re.findall(r"(\d+)([=XID])", want)
        assert all(int(c) <= 31 for c, _ in runs)
        assert sum(int(c) for c, op in runs if op == "=") == p + 5 and [op for _, op in runs].count("X") == 1
    # an error-free read needs no bytes at all
    assert api.cigar_to_edit_stream("31=31=31=7=") == b""
    assert api.edit_stream_to_cigar(b"", 100) == "31=31=31=7="
    assert api.edit_stream_to_cigar(b"", 0) == ""


@pytest.mark.parametrize("lane_form", [False, True])
def test_malformed_streams_are_rejected(lane_form):
    with pytest.raises(api.ScroogeError):
        api.edit_stream_to_cigar(bytes([1 << 6 | 5]), 3, lane_form=lane_form)            # 5 matches + X in a read of 3
    with pytest.raises(api.ScroogeError):
        api.edit_stream_to_cigar(bytes([2 << 6, 2 << 6]), 1, lane_form=lane_form)        # two insertions, one base
    with pytest.raises(api.ScroogeError):
        api.edit_stream_to_cigar(bytes([0x3F]), 10, lane_form=lane_form)                  # 64 matches in a read of 10
    with pytest.raises(api.ScroogeError):
        api.edit_stream_to_cigar(bytes([0x3F, 0x3F]), 100, lane_form=lane_form)           # 128 matches in a read of 100
    with pytest.raises(api.ScroogeError):
        api.edit_stream_to_cigar(bytes([1 << 6]), 0, lane_form=lane_form)                 # an edit in an empty read
    assert api.edit_stream_to_cigar(bytes([3 << 6]), 2, lane_form=lane_form) == "1D2="    # a deletion uses no read base
    assert api.edit_stream_to_cigar(bytes([3 << 6] * 70), 2, lane_form=lane_form) == "31D31D8D2="     # deletions only: windows end on the text side
    with pytest.raises(ValueError):
        api.cigar_to_edit_stream("5M")


def test_c_example_runs_without_a_gpu():
    """examples/edit_stream_example.c: plain C against the C ABI (gcc, no HIP in the program), the host conversions only."""
    import subprocess
    import scrooge_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scrooge_amd.build_library()
    libdir = os.path.join(root, "scrooge_amd")
    exe = "/tmp/scrg_edit_stream_example"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "edit_stream_example.c"),
                           "-L" + libdir, "-lscrooge_amd", "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    assert lines[0] == "runs   31=9=1X21=6=1D24=8="
    assert lines[1] == "stream 2 bytes: 68 db"            # X after 40 matches: 1 << 6 | 40; D after 27: 3 << 6 | 27
    assert lines[2] == "decoded 31=9=1X21=6=1D24=8="
    assert lines[3].startswith("W32/O17 15=") and lines[-1] == "ok"


@pytest.mark.parametrize("W,O", [(64, 33), (64, 1), (5, 2), (2, 1), (256, 1), (128, 65)])
def test_lane_state_machine_equals_replay_on_arbitrary_streams(W, O):
    """Fuzz: ANY byte string is either rejected by both decoders or gives the same runs in both — non-canonical
    streams included (op-0 bytes of any length, runs of them, edits after the read is used up)."""
    import ctypes as C
    lib = api.load_library()
    rng = np.random.Generator(np.random.PCG64(7 * W + O))
    p = api.Params()
    lib.scrg_params_default(C.byref(p))
    p.W, p.O = W, O
    accepted = 0
    for it in range(3000):
        n = int(rng.integers(0, 40))
        kind = it % 3
        if kind == 0:
            raw = rng.integers(0, 256, n, dtype=np.uint8)
        elif kind == 1:              # mostly small match counts: many windows end inside edits
            raw = (rng.integers(0, 4, n, dtype=np.uint8) << 6 | rng.integers(0, 3, n, dtype=np.uint8)).astype(np.uint8)
        else:                        # long stretches: op-0 bytes in a row
            raw = np.where(rng.random(n) < 0.6, rng.integers(0, 64, n), rng.integers(64, 256, n)).astype(np.uint8)
        s = raw.tobytes()
        # a read length that the stream fits exactly (most of the time), or a wrong one
        used = sum((b & 63) + (1 if (b >> 6) in (0, 1, 2) else 0) for b in s)
        read_len = used + (int(rng.integers(0, 50)) if it % 5 else -int(rng.integers(1, 3)))
        if read_len < 0:
            read_len = 0
        buf = (C.c_uint8 * max(1, n)).from_buffer_copy(s or b"\0")
        out = []
        for fn in (lib.scrg_edit_stream_to_runs, lib.scrg_edit_stream_to_runs_lane):
            cnt = C.c_uint64(0)
            runs = (C.c_uint8 * (2 * (2 * n + read_len + 8)))()
            st = fn(C.byref(p), read_len, buf, n, runs, 2 * n + read_len + 8, C.byref(cnt))
            out.append((st, cnt.value, bytes(runs[: 2 * cnt.value]) if st == api.SCRG_OK else b""))
        assert out[0] == out[1], (it, s.hex(), read_len, out)
        accepted += out[0][0] == api.SCRG_OK
    assert accepted > 500
