"""Checks scripts/proto/diag_proto.c against the oracle on seeded pairs (CPU only)."""
import ctypes as C, sys, subprocess, os
import numpy as np
sys.path.insert(0, ".")
from scrooge_amd import synth
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libdiag_proto.so")
subprocess.check_call(["gcc", "-O2", "-std=c11", "-fopenmp", "-fPIC", "-shared", "-w", "-o", so, os.path.join(here, "diag_proto.c")])
lib = C.CDLL(so)
class PS(C.Structure):
    _fields_ = [("diag", C.c_uint64), ("fb", C.c_uint64), ("rows", C.c_uint64)]
code = np.zeros(256, np.uint8); code[ord("C")] = 1; code[ord("G")] = 2; code[ord("T")] = 3
def run(fn, t, q, *extra):
    tc = code[np.frombuffer(t, np.uint8)]; qc = code[np.frombuffer(q, np.uint8)]
    cap = len(t) + len(q) + 8
    runs = (C.c_uint8 * (2 * cap))(); n = C.c_size_t(); ed = C.c_longlong()
    st = fn(tc.ctypes.data_as(C.c_void_p), C.c_size_t(len(tc)), qc.ctypes.data_as(C.c_void_p), C.c_size_t(len(qc)), *extra,
            runs, C.c_size_t(cap), C.byref(n), C.byref(ed), *tail)
    assert st == 0
    return ed.value, bytes(runs[:2 * n.value])
tot = PS(); bad = 0; cnt = 0
for O in (33, 40, 50):
    for prof, L, N, mr in [("ont", 3000, 60, 13), ("pacbio15", 3000, 40, 13), ("illumina", 300, 100, 13), ("ont", 2000, 40, 15), ("pacbio15", 2000, 40, 6)]:
        T, Q = synth.make_pairs(N, L, prof, seed=O * 100 + L + mr)
        rng = np.random.Generator(np.random.PCG64(O + L))
        for _ in range(10):
            T.append(synth.random_seq(int(rng.integers(0, 400)), rng)); Q.append(synth.random_seq(int(rng.integers(0, 400)), rng))
        for t, q in zip(T, Q):
            ps = PS(); tail = (C.byref(ps),)
            a = run(lib.proto_align_codes, t, q, C.c_int(O), C.c_int(mr))
            tail = (None,)
            b = run(lib.go_align_codes, t, q, C.c_int(64), C.c_int(O))
            cnt += 1
            if a != b:
                bad += 1
                if bad < 4: print("MISMATCH", O, prof, len(t), len(q), a[0], b[0])
            tot.diag += ps.diag; tot.fb += ps.fb; tot.rows += ps.rows
print("pairs", cnt, "bad", bad, "diag windows", tot.diag, "fallback", tot.fb, "rows/diag-window", tot.rows / max(1, tot.diag + 0.0))
