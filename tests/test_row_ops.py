"""The multi-word bit rows of the W > 64 kernels (scrooge_amd/csrc/row_ops.h, compiled here for the host with g++) and
the oracle's 256-bit vector against the known answers of the reference's own bitvector tests
(src/bitvector_test.cu:22-132: a 65-bit vector of 32-bit elements — shifts carried across elements, or / and / not,
has_one_at, single_one_at — and insert_bits on a 64-bit one), then against Python integers on random rows.

The kernels keep a row mirrored (word 0 most significant, position c = bit 63 - c % 64 of word c / 64): a reference vector
of `bits` bits is embedded top-aligned, reference bit i <-> position bits-1-i, and the reference's `<< n`, which drops
what leaves bit bits-1, is row_shl(n)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BITS = 65
MASK = (1 << BITS) - 1


def ref_bv(*elements):
    """bv{e0, e1, e2} of the reference's tests: 32-bit elements, least significant first."""
    return sum(int(e) << (32 * k) for k, e in enumerate(elements))


# src/bitvector_test.cu:25-33
A = ref_bv(0x89ABCDEF, 0x01234567, 0x1)
SHIFTS = [(0, ref_bv(0x89ABCDEF, 0x01234567, 0x1)), (4, ref_bv(0x9ABCDEF0, 0x12345678, 0x0)),
          (8, ref_bv(0xABCDEF00, 0x23456789, 0x1)), (12, ref_bv(0xBCDEF000, 0x3456789A, 0x0)),
          (16, ref_bv(0xCDEF0000, 0x456789AB, 0x1)), (20, ref_bv(0xDEF00000, 0x56789ABC, 0x0)),
          (24, ref_bv(0xEF000000, 0x6789ABCD, 0x1)), (65, 0)]
# :41-54 (or), :62-72 (and), :80-84 (not)
ONES, ZEROS = MASK, 0
OR_CASES = [(ref_bv(0xFFFF0000, 0xFFFF0000, 0x0), ref_bv(0x0000FFFF, 0x0000FFFF, 0x1), ONES),
            (ref_bv(0x0F0F0F0F, 0x0F0F0F0F, 0x0), ref_bv(0xF0F0F0F0, 0xF0F0F0F0, 0x1), ONES),
            (ref_bv(0x0A0A0A0A, 0x0A0A0A0A, 0x0), ref_bv(0x05050505, 0x05050505, 0x0), ref_bv(0x0F0F0F0F, 0x0F0F0F0F, 0x0)),
            (ONES, ZEROS, ONES)]
AND_CASES = [(ref_bv(0xFFFF0000, 0xFFFF0000, 0x0), ref_bv(0x5555BBBB, 0x5555BBBB, 0x1), ref_bv(0x55550000, 0x55550000, 0x0)),
             (ref_bv(0x5555BBBB, 0x5555BBBB, 0x1), ref_bv(0xBBBB5555, 0xBBBB5555, 0x1), ref_bv(0x11111111, 0x11111111, 0x1)),
             (ONES, ZEROS, ZEROS)]
NOT_CASES = [(ONES, ZEROS), (ref_bv(0xFFFF0000, 0xFFFF0000, 0x0), ref_bv(0x0000FFFF, 0x0000FFFF, 0x1))]
# :97-101
EVERY_FOURTH = ref_bv(0x11111111, 0x11111111, 0x1)
# :109-113
SINGLE_ONES = [(0, ref_bv(0x1)), (1, ref_bv(0x2)), (2, ref_bv(0x4)), (34, ref_bv(0x0, 0x4)), (BITS - 1, ref_bv(0x0, 0x0, 0x1))]


# ---------------------------------------------------------------- the kernels' rows (row_ops.h on the host)
@pytest.fixture(scope="module")
def rows():
    so = os.path.join(HERE, "proto", "librow_ops_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas",
                           "-I", os.path.join(ROOT, "scrooge_amd", "csrc"), "-o", so, os.path.join(HERE, "proto", "row_ops_host.cpp")])
    lib = C.CDLL(so)
    lib.row_op.restype = C.c_uint32
    lib.row_op.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
    return lib


SHL, SHL1_IN, SHR1, BIT, TOP, BV_SHL1, CLZ, TEST, ANY, POP = range(10)


def to_words(v, rw):
    return np.array([(v >> (64 * (rw - 1 - k))) & (2 ** 64 - 1) for k in range(rw)], np.uint64)


def from_words(w):
    v = 0
    for x in w:
        v = (v << 64) | int(x)
    return v


def op(lib, rw, code, v, s=0):
    a = to_words(v, rw)
    out = np.zeros(rw, np.uint64)
    r = lib.row_op(rw, code, a.ctypes.data, s, out.ctypes.data)
    assert r != 0xffffffff
    return from_words(out), r


def embed(v, rw):
    return v << (64 * rw - BITS)


@pytest.mark.parametrize("rw", [2, 3, 4])
def test_rows_shift_known_answers(rows, rw):
    for n, want in SHIFTS:
        assert op(rows, rw, SHL, embed(A, rw), n)[0] == embed(want, rw), n
        if n <= 24:                        # the row-sweep kernel's vector only ever shifts by one
            v = embed(A, rw)
            for _ in range(n):
                v = op(rows, rw, BV_SHL1, v)[0]
            assert v == embed(want, rw), n
            v = embed(A, rw)
            for _ in range(n):
                v = op(rows, rw, SHL1_IN, v, 0)[0]
            assert v == embed(want, rw), n


@pytest.mark.parametrize("rw", [2, 3, 4])
def test_rows_has_one_at_known_answers(rows, rw):
    for i in range(BITS):
        c = BITS - 1 - i
        assert op(rows, rw, TEST, embed(ONES, rw), c)[1] == 1
        assert op(rows, rw, TEST, embed(ZEROS, rw), c)[1] == 0
        assert op(rows, rw, TEST, embed(EVERY_FOURTH, rw), c)[1] == (1 if i % 4 == 0 else 0)


@pytest.mark.parametrize("rw", [2, 3, 4])
def test_rows_single_one_at_known_answers(rows, rw):
    for i, want in SINGLE_ONES:
        assert op(rows, rw, BIT, 0, BITS - 1 - i)[0] == embed(want, rw)


@pytest.mark.parametrize("rw", [1, 2, 3, 4])
def test_rows_against_integers(rows, rw):
    rng = np.random.Generator(np.random.PCG64(1234 + rw))
    nb = 64 * rw
    full = (1 << nb) - 1
    for it in range(150):
        v = from_words(rng.integers(0, 2 ** 64, rw, dtype=np.uint64))
        if it % 3 == 0:                    # sparse rows: long zero prefixes for row_clz
            v >>= int(rng.integers(0, nb + 1))
        if it % 50 == 0:
            v = 0
        s = int(rng.integers(0, nb))
        assert op(rows, rw, SHL, v, s)[0] == (v << s) & full
        assert op(rows, rw, SHL1_IN, v, 1)[0] == ((v << 1) | 1) & full
        assert op(rows, rw, SHL1_IN, v, 0)[0] == (v << 1) & full
        assert op(rows, rw, BV_SHL1, v)[0] == (v << 1) & full
        assert op(rows, rw, SHR1, v)[0] == v >> 1
        assert op(rows, rw, BIT, 0, s)[0] == 1 << (nb - 1 - s)
        t = int(rng.integers(0, nb + 1))
        assert op(rows, rw, TOP, 0, t)[0] == full ^ (full >> t)
        assert op(rows, rw, CLZ, v)[1] == nb - v.bit_length()
        assert op(rows, rw, TEST, v, s)[1] == (v >> (nb - 1 - s)) & 1
        assert op(rows, rw, ANY, v)[1] == (1 if v else 0)
        assert op(rows, rw, POP, v)[1] == bin(v).count("1")


# ---------------------------------------------------------------- the oracle's 256-bit vector
@pytest.fixture(scope="module")
def bv256():
    from oracle import pyoracle
    lib = C.CDLL(pyoracle.build())
    lib.go_bv256_op.restype = C.c_int
    lib.go_bv256_op.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]

    def call(code, a, b=0, s=0):
        wa = np.array([(a >> (64 * k)) & (2 ** 64 - 1) for k in range(4)], np.uint64)
        wb = np.array([(b >> (64 * k)) & (2 ** 64 - 1) for k in range(4)], np.uint64)
        out = np.zeros(4, np.uint64)
        r = lib.go_bv256_op(code, wa.ctypes.data, wb.ctypes.data, s, out.ctypes.data)
        assert r >= 0
        return sum(int(x) << (64 * k) for k, x in enumerate(out)), r
    return call


def test_oracle_vector_known_answers(bv256):
    # the oracle's vector is wider than the 65 bits of the reference's test type: what the reference truncates lies
    # above bit 64 there and is masked here (the algorithm never looks above bit m-1 <= W-1)
    for n, want in SHIFTS:
        assert bv256(0, A, s=n)[0] & MASK == want, n
    for a, b, want in OR_CASES:
        assert bv256(1, a, b)[0] == want
    for a, b, want in AND_CASES:
        assert bv256(2, a, b)[0] == want
    for a, want in NOT_CASES:              # not: all ones with the ones of `a` cleared (the oracle's only use: pattern masks)
        v = ONES
        for i in range(BITS):
            if (a >> i) & 1:
                v = bv256(3, v, s=i)[0]
        assert v == want
    for i in range(BITS):
        assert bv256(1, ONES, 0, i)[1] == 0          # returns bit_is_zero(a, s) = !has_one_at
        assert bv256(1, ZEROS, 0, i)[1] == 1
        assert bv256(1, EVERY_FOURTH, 0, i)[1] == (0 if i % 4 == 0 else 1)
    for i, want in SINGLE_ONES:
        assert bv256(0, 1, s=i)[0] == want
    # src/bitvector_test.cu:121-129 (insert_bits)
    v = bv256(4, 0, 0xFF, 32)[0]
    v = bv256(4, v, 0xAA, 0)[0]
    assert v == 0xFF000000AA


def test_oracle_vector_against_integers(bv256):
    rng = np.random.Generator(np.random.PCG64(99))
    full = (1 << 256) - 1
    for _ in range(300):
        a = sum(int(x) << (64 * k) for k, x in enumerate(rng.integers(0, 2 ** 64, 4, dtype=np.uint64)))
        b = sum(int(x) << (64 * k) for k, x in enumerate(rng.integers(0, 2 ** 64, 4, dtype=np.uint64)))
        s = int(rng.integers(0, 256))
        assert bv256(0, a, s=s)[0] == (a << s) & full
        assert bv256(1, a, b)[0] == a | b
        assert bv256(2, a, b)[0] == a & b
        assert bv256(3, a, s=s) == (a & ~(1 << s), 1 - ((a >> s) & 1))
