"""How a host call is cut up (scrg_host_plan: no GPU involved): the issue order is what the reference's callers do for load
balance — longest read first (src/tests.cu:375-377), stable —, chunks are whole groups of 64 pairs that tile the issue
order, and chunk k belongs to device k mod N: the in-process counterpart of scrooge_amd.distributed.shard_plan."""
import ctypes as C

import numpy as np
import pytest

from scrooge_amd import api


def plan(read_lens, text_lens=None, n_devices=1, sort=1):
    lib = api.load_library()
    n = len(read_lens)
    rl = (C.c_uint64 * max(n, 1))(*read_lens)
    tl = (C.c_uint64 * max(n, 1))(*(text_lens if text_lens is not None else [0] * n))
    order = (C.c_uint32 * max(n, 1))()
    chunks = (C.c_uint64 * (n // 64 + 8))()
    nc = C.c_uint64(0)
    p = api.Params()
    lib.scrg_params_default(C.byref(p))
    p.sort_by_length = sort
    st = lib.scrg_host_plan(C.byref(p), n_devices, n, tl, rl, order, chunks, n // 64 + 8, C.byref(nc))
    assert st == api.SCRG_OK
    return list(order[:n]), list(chunks[: nc.value + 1])


@pytest.mark.parametrize("n_devices", [1, 2, 8])
def test_issue_order_and_chunks(n_devices):
    rng = np.random.Generator(np.random.PCG64(n_devices))
    lens = rng.integers(0, 12000, 7001).tolist()
    order, chunks = plan(lens, [int(1.15 * x) for x in lens], n_devices)
    assert sorted(order) == list(range(len(lens)))                                  # a permutation
    got = [lens[i] for i in order]
    assert got == sorted(lens, reverse=True)                                        # longest read first
    for a, b in zip(order, order[1:]):                                              # stable: ties keep the caller's order
        if lens[a] == lens[b]:
            assert a < b
    assert chunks[0] == 0 and chunks[-1] == len(lens) and all(x < y for x, y in zip(chunks, chunks[1:]))
    assert all((y - x) % 64 == 0 for x, y in zip(chunks[:-2], chunks[1:-1]))        # whole groups of 64, except the last chunk
    # enough chunks for every device and stream, none beyond 32 MB of packed sequence
    assert len(chunks) - 1 >= min(4 * n_devices, (len(lens) + 511) // 512)
    for x, y in zip(chunks, chunks[1:]):
        words = sum((lens[i] + 31) // 32 + (int(1.15 * lens[i]) + 31) // 32 for i in order[x:y])
        assert words <= (4 << 20) + 2 * 64 * 800
    # every device gets about the same number of bases: chunk k goes to device k mod N, longest reads dealt first
    per_dev = [0] * n_devices
    for k, (x, y) in enumerate(zip(chunks, chunks[1:])):
        per_dev[k % n_devices] += sum(lens[i] for i in order[x:y])
    if len(chunks) - 1 >= 4 * n_devices:
        assert max(per_dev) < 1.6 * (sum(per_dev) / n_devices)


def test_sorted_batches_keep_their_order():
    lens = [5000] * 1000 + [150] * 3000
    order, chunks = plan(lens)
    assert order == list(range(4000))                       # already longest first: results need no permutation
    order, _ = plan(list(reversed(lens)), sort=0)
    assert order == list(range(4000))                       # sort_by_length = 0: issue order = caller order
    order, chunks = plan([])
    assert order == [] and chunks == [0]


def test_root_share_plan_of_the_bench():
    """bench.py --root-share: what rank 0 aligns when it also decodes every rank's CIGARs (N > 1, root 0): whole groups of
    64 pairs, the step never smaller than N x --pairs, rank 0 a pure collector from N = 6 on."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    assert bench.root_share_plan(2, 100000, "auto") == (80000, 120000)
    assert bench.root_share_plan(4, 100000, "auto") == (40000, 120000)
    assert bench.root_share_plan(8, 125000, "auto") == (0, 142912)
    assert bench.root_share_plan(2, 6400, "0.3") == (1920, 10880)
    for world in (2, 3, 4, 6, 8):
        for nominal in (6400, 100000, 125000, 99999):
            for mode in ("auto", "0", "0.5", "1"):
                n0, n = bench.root_share_plan(world, nominal, mode)
                assert n0 % 64 == 0 and n % 64 == 0 and n0 + (world - 1) * n >= world * nominal
                assert n0 + (world - 1) * n < world * nominal + 64 * world


def test_host_packer_matches_the_layout():
    """scrg_pack_planar_host — the packer of the host entry points (AVX2 or scalar, chosen at run time) — against the definition of the planar layout (include/scrooge_amd.h: bit k of the low dword of word w = bit 0 of
    base 32 w + k's code, high dword = bit 1; A0 C1 G2 T3, lower case too; zero past the end) for lengths around every
    block size, with a stride, and its verdict on bytes that are not bases (the reference asserts, src/genasm_cpu.cpp:487-489)."""
    import ctypes as C
    import numpy as np
    import scrooge_amd
    from scrooge_amd import api
    scrooge_amd.build_library()
    lib = api.load_library()
    rng = np.random.Generator(np.random.PCG64(11))
    code = {65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}
    letters = np.frombuffer(b"ACGTacgt", np.uint8)
    for n in [0, 1, 31, 32, 33, 63, 64, 65, 95, 127, 128, 129, 191, 200, 1000, 4097]:
        for stride in (1, 3):
            seq = letters[rng.integers(0, 8, n)]
            n_words = (n + 31) // 32 + 2
            out = np.full(n_words * stride + 1, 0x5555555555555555, dtype=np.uint64)
            buf = seq.tobytes() + b"#"             # (one byte past the end must not be looked at)
            st = lib.scrg_pack_planar_host(C.c_char_p(buf), n, out.ctypes.data_as(C.c_void_p), stride, n_words)
            assert st == api.SCRG_OK, (n, stride)
            want = np.zeros(n_words, dtype=np.uint64)
            for k in range(n):
                c = code[int(seq[k])]
                want[k // 32] |= np.uint64((c & 1) << (k % 32)) | np.uint64((c >> 1) << (32 + k % 32))
            assert (out[0: n_words * stride: stride] == want).all(), (n, stride)
            if stride > 1:
                assert (out[1: n_words * stride: stride] == 0x5555555555555555).all()      # only its own words are written
        for pos in sorted(set([0, n // 2, n - 1])) if n else []:
            bad = bytearray(letters[rng.integers(0, 8, n)].tobytes())
            for b in (ord("N"), 0, 0x20, 0xC1, ord("B"), ord("@"), ord("U")):
                bad[pos] = b
                out = np.zeros((n + 31) // 32 + 1, dtype=np.uint64)
                assert lib.scrg_pack_planar_host(C.c_char_p(bytes(bad)), n, out.ctypes.data_as(C.c_void_p), 1, len(out)) == api.SCRG_ERR_BAD_BASE, (n, pos, b)
