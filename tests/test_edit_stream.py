"""CPU tests of the edit-stream transfer format (include/scrooge_amd.h, scrooge_amd/csrc/edit_stream.h): the
host encoder (which replays the window loop to place the window-end bytes) and the decoders, against the reference's own CIGARs.

What comes out must have the reference's window breaks (runs are flushed per window and never merged,
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


def py_encode(cigar, W, O):
    """Independent restatement of the canonical encoding, straight from the format's definition (include/scrooge_amd.h):
    one byte per edit, one per window end; the window loop of genasm_cpu.cpp:307-310 places the ends."""
    code = {"X": 1, "I": 2, "D": 3}
    L = W - O
    out = bytearray()
    st = {"pend": 0}

    def emit(c):
        out.extend(b"\x3f" * (st["pend"] // 63))
        out.append(c << 6 | st["pend"] % 63)
        st["pend"] = 0

    i = j = 0
    open_ = False
    for cnt, op in re.findall(r"(\d+)([=XID])", cigar):
        c = int(cnt)
        while c:
            if op == "=":
                t = min(c, L - i, L - j)
                st["pend"] += t
                i += t; j += t; c -= t
            else:
                emit(code[op])
                j += op != "D"
                i += op != "I"
                c -= 1
            open_ = True
            if i == L or j == L:
                emit(0)
                i = j = 0
                open_ = False
    if open_:
        emit(0)
    return bytes(out)


def py_decode(s, read_len):
    """The decoder from the format's definition, without a look at the window geometry (what the device checks):
    -> CIGAR text, or None for a stream that is not the alignment of a read of this length."""
    runs = []
    pend = placed = 0
    cur = None
    for b in s:
        op, ln = b >> 6, b & 63
        if b == 0x3F:
            pend += 63
            continue
        t = pend + ln
        pend = 0
        if t:
            runs.append(["=", t])
            placed += t
            cur = None
        if op == 0:
            cur = None
            continue
        c = " XID"[op]
        if cur == c:
            runs[-1][1] += 1
        else:
            runs.append([c, 1])
            cur = c
        placed += c != "D"
    if pend or placed != read_len or (s and (s[-1] >> 6 or s[-1] == 0x3F)) or any(n > 255 for _, n in runs):
        return None
    return "".join("%d%s" % (n, c) for c, n in runs)


def round_trip(cigar, read_len, ed, W, O):
    s = api.cigar_to_edit_stream(cigar, W=W, O=O)
    assert s == py_encode(cigar, W, O)
    assert sum(1 for b in s if b >> 6) == ed          # one byte per edit (+ the window ends and long-match bytes, op 0)
    assert py_decode(s, read_len) == cigar
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
    # W-O = 31: every window ends with its END byte carrying the matches since the last edit; no byte 0x3F ever
    for p in (0, 1, 29, 30, 31, 32, 62, 63, 64, 65, 127, 128, 129, 1000):
        cig = ("%d=" % p if p else "") + "1X5="                                   # (not cut at the windows: the encoder does that)
        s = api.cigar_to_edit_stream(cig)
        assert 0x3F not in s and s == py_encode(cig, 64, 33)
        full, rest = divmod(p, 31)
        assert s[:full] == bytes([31]) * full and s[full] == 1 << 6 | rest       # whole windows of matches, then the X after the rest
        want = api.edit_stream_to_cigar(s, p + 6)
        assert want == api.edit_stream_to_cigar(s, p + 6, lane_form=True) == py_decode(s, p + 6)
        runs = re.findall(r"(\d+)([=XID])", want)
        assert all(0 < int(c) <= 31 for c, _ in runs) and want.startswith("31=" * full)
        assert sum(int(c) for c, op in runs if op == "=") == p + 5 and [op for _, op in runs].count("X") == 1
    # W-O = 127 / 255: 63 matches and nothing else are a byte 0x3F; 63 q + r matches = q such bytes, then the byte carries r
    for W, O in ((128, 1), (256, 1)):
        L = W - O
        for p in (62, 63, 64, 125, 126, 127, 189, 250, 254):
            if p >= L:
                continue
            cig = "%d=1X" % p + ("%d=" % (L - p - 1) if L - p - 1 else "")          # one full window
            s = api.cigar_to_edit_stream(cig, W=W, O=O)
            q2, r2 = divmod(L - p - 1, 63)
            assert s == b"\x3f" * (p // 63) + bytes([1 << 6 | p % 63]) + b"\x3f" * q2 + bytes([r2])
            for lane_form in (False, True):
                assert api.edit_stream_to_cigar(s, L, W=W, O=O, lane_form=lane_form) == cig
    # an error-free read: one END byte per window
    assert api.cigar_to_edit_stream("31=31=31=7=") == bytes([31, 31, 31, 7])
    assert api.edit_stream_to_cigar(bytes([31, 31, 31, 7]), 100) == "31=31=31=7="
    assert api.cigar_to_edit_stream("") == b"" and api.edit_stream_to_cigar(b"", 0) == ""
    # a run the caller did not cut at the windows is cut by the encoder
    assert api.cigar_to_edit_stream("100=") == bytes([31, 31, 31, 7])


@pytest.mark.parametrize("lane_form", [False, True])
def test_malformed_streams_are_rejected(lane_form):
    bad = [(bytes([1 << 6 | 5, 0]), 3),              # 5 matches + X in a read of 3
           (bytes([2 << 6, 2 << 6, 0]), 1),          # two insertions, one base
           (bytes([0x3F, 0]), 10),                   # 63 matches in a read of 10
           (bytes([1 << 6, 0]), 0),                  # an edit in an empty read
           (bytes([1 << 6 | 5]), 6),                 # no window end at the end
           (bytes([5, 0x3F]), 68),                   # a stretch nobody closes
           (b"", 5)]                                 # no bytes, but a read
    for s, rl in bad:
        with pytest.raises(api.ScroogeError):
            api.edit_stream_to_cigar(s, rl, lane_form=lane_form)
    assert api.edit_stream_to_cigar(bytes([3 << 6, 2]), 2, lane_form=lane_form) == "1D2="    # a deletion uses no read base
    # deletions only: windows end on the text side
    s = bytes([3 << 6] * 31 + [0] + [3 << 6] * 31 + [0] + [3 << 6] * 8 + [2])
    assert api.edit_stream_to_cigar(s, 2, lane_form=lane_form) == "31D31D8D2="
    # 256 deletions in a row without a window end: no scrg_run can hold that count
    with pytest.raises(api.ScroogeError):
        api.edit_stream_to_cigar(bytes([3 << 6] * 256 + [2]), 2, W=256, O=1, lane_form=lane_form)
    with pytest.raises(ValueError):
        api.cigar_to_edit_stream("5M")


def test_window_ends_are_held_to_the_window_loop():
    """scrg_edit_stream_to_runs (the host decoder, which is given W and O) rejects a stream whose window ends are not where
    the reference's loop puts them (genasm_cpu.cpp:307-310); the lane form — the device's — decodes what the stream says."""
    ok = bytes([31, 31, 31, 7])
    assert api.edit_stream_to_cigar(ok, 100) == "31=31=31=7="
    for s, lenient in ((bytes([30, 32, 31, 7]), "30=32=31=7="),         # a window of 30, one of 32
                       (bytes([31, 31, 31, 3, 4]), "31=31=31=3=4="),    # a window that ends for no reason
                       (bytes([31, 31, 31, 7, 0]), "31=31=31=7="),      # an empty window
                       (bytes([62, 31, 7]), "62=31=7=")):               # two windows in one
        with pytest.raises(api.ScroogeError):
            api.edit_stream_to_cigar(s, 100)
        assert api.edit_stream_to_cigar(s, 100, lane_form=True) == lenient == py_decode(s, 100)


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
    assert lines[1] == "stream 6 bytes: 1f 49 15 c6 18 08"        # END(31) | X after 9, END(21) | D after 6, END(24) | END(8)
    assert lines[2] == "decoded 31=9=1X21=6=1D24=8="
    assert lines[3].startswith("W32/O17 15=") and lines[-1] == "ok"


@pytest.mark.parametrize("W,O", [(64, 33), (64, 1), (5, 2), (2, 1), (256, 1), (128, 65)])
def test_lane_state_machine_on_arbitrary_streams(W, O):
    """Fuzz: ANY byte string is decoded by the lane form (the device's state machine) exactly as the format's definition
    says (py_decode) — non-canonical streams included (0x3F bytes anywhere, empty windows, windows of any size) — and
    whatever the host decoder accepts for W/O, the lane form accepts with the same runs."""
    import ctypes as C
    lib = api.load_library()
    rng = np.random.Generator(np.random.PCG64(7 * W + O))
    p = api.Params()
    lib.scrg_params_default(C.byref(p))
    p.W, p.O = W, O
    accepted = strict = 0
    for it in range(3000):
        n = int(rng.integers(0, 40))
        kind = it % 4
        if kind == 0:
            raw = rng.integers(0, 256, n, dtype=np.uint8)
        elif kind == 1:              # mostly small match counts: many merges
            raw = (rng.integers(0, 4, n, dtype=np.uint8) << 6 | rng.integers(0, 3, n, dtype=np.uint8)).astype(np.uint8)
        elif kind == 2:              # long stretches: op-0 bytes in a row
            raw = np.where(rng.random(n) < 0.6, rng.integers(0, 64, n), rng.integers(64, 256, n)).astype(np.uint8)
        else:                        # a canonical stream of a random alignment
            ops = rng.choice(list("====XID"), size=int(rng.integers(0, 300)))
            cig = "".join("1" + o for o in ops).rstrip("1D")            # (the window loop ends with the read: no deletion after its last character)
            while cig.endswith("D"):
                cig = cig[:-2]
            raw = np.frombuffer(py_encode(cig, W, O), dtype=np.uint8)
            n = len(raw)
        s = raw.tobytes()
        if kind != 3 and n and it % 2:
            s = s[:-1] + bytes([s[-1] & 63 if s[-1] != 0x3F else 0])          # ends with a window end
        used = sum((b & 63) + (1 if (b >> 6) in (1, 2) else 0) for b in s)
        read_len = used + (0 if it % 5 else int(rng.integers(1, 3)))
        buf = (C.c_uint8 * max(1, n)).from_buffer_copy(s or b"\0")
        out = []
        for fn in (lib.scrg_edit_stream_to_runs, lib.scrg_edit_stream_to_runs_lane):
            cnt = C.c_uint64(0)
            runs = (C.c_uint8 * (2 * (2 * n + 8)))()
            st = fn(C.byref(p), read_len, buf, n, runs, 2 * n + 8, C.byref(cnt))
            out.append("".join("%d%s" % (runs[2 * k], chr(runs[2 * k + 1])) for k in range(cnt.value)) if st == api.SCRG_OK else None)
        want = py_decode(s, read_len)
        assert out[1] == want, (it, s.hex(), read_len, out, want)
        if out[0] is not None:
            assert out[0] == out[1], (it, s.hex(), read_len, out)
            strict += 1
        if kind == 3 and read_len == used:
            assert out[0] is not None, (it, s.hex(), read_len)
        accepted += out[1] is not None
    assert accepted > 500 and strict > 300
