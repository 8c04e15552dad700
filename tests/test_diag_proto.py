"""The diagonal-major window arithmetic of the HIP kernel (genasm_kernels.hip: carry-chain rows, band of 32
diagonals, count-leading-zeros traceback), restated in C (tests/proto/diag_proto.c) and checked against the
oracle on the CPU.  The C file includes the oracle's source for the windows the diagonal form does not cover."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from scrooge_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def proto():
    so = os.path.join(HERE, "proto", "libdiag_proto.so")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-fopenmp", "-fPIC", "-shared", "-w", "-o", so,
                           os.path.join(HERE, "proto", "diag_proto.c")])
    return C.CDLL(so)


class PS(C.Structure):
    _fields_ = [("diag", C.c_uint64), ("fallback", C.c_uint64), ("rows", C.c_uint64)]


CODE = np.zeros(256, np.uint8)
CODE[ord("C")], CODE[ord("G")], CODE[ord("T")] = 1, 2, 3


def _run(fn, t, q, head, tail):
    tc, qc = CODE[np.frombuffer(t, np.uint8)], CODE[np.frombuffer(q, np.uint8)]
    cap = len(t) + len(q) + 8
    runs = (C.c_uint8 * (2 * cap))()
    n, ed = C.c_size_t(), C.c_longlong()
    st = fn(tc.ctypes.data_as(C.c_void_p), C.c_size_t(len(tc)), qc.ctypes.data_as(C.c_void_p), C.c_size_t(len(qc)),
            *head, runs, C.c_size_t(cap), C.byref(n), C.byref(ed), *tail)
    assert st == 0
    return ed.value, bytes(runs[:2 * n.value])


@pytest.mark.parametrize("O,max_rows", [(33, 15), (33, 13), (40, 6), (50, 15)])
def test_diagonal_form_matches_oracle(proto, O, max_rows):
    T, Q = [], []
    for prof, L, n in [("ont", 2000, 25), ("pacbio15", 2000, 15), ("illumina", 300, 30)]:
        t, q = synth.make_pairs(n, L, prof, seed=O * 100 + L + max_rows)
        T, Q = T + t, Q + q
    rng = np.random.Generator(np.random.PCG64(O + max_rows))
    for _ in range(10):                       # unrelated sequences: every window falls back
        T.append(synth.random_seq(int(rng.integers(0, 400)), rng))
        Q.append(synth.random_seq(int(rng.integers(0, 400)), rng))
    n_diag = n_fb = 0
    for t, q in zip(T, Q):
        ps = PS()
        got = _run(proto.proto_align_codes, t, q, (C.c_int(O), C.c_int(max_rows)), (C.byref(ps),))
        want = _run(proto.go_align_codes, t, q, (C.c_int(64), C.c_int(O)), (None,))
        assert got == want
        n_diag += ps.diag
        n_fb += ps.fallback
    assert n_diag > 1000 and n_fb > 10        # both kinds of window were exercised
