"""ctypes loaders for the two CPU checkers (TEST INFRASTRUCTURE ONLY).

* ``Oracle``    — oracle/liboracle.so, this repo's C restatement of
  /root/reference/src/genasm_cpu.cpp:178-460 (travels to the GPU box).
* ``Reference`` — oracle/_ref/libgenasm_ref.so, the unmodified reference CPU
  path compiled by oracle/Makefile (prebuilt file travels; sources do not).
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


class GoStats(C.Structure):
    _fields_ = [("windows", C.c_uint64), ("dc_cells", C.c_uint64),
                ("tb_steps", C.c_uint64), ("runs", C.c_uint64),
                ("text_used", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def _digest():
    import hashlib
    h = hashlib.sha256()
    for f in ("genasm_oracle.c", "genasm_oracle_core.inc", "genasm_oracle.h", "Makefile", "ref_driver.cpp", "ref_score_driver.cpp"):
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    return h.hexdigest()


def is_stale():
    """Decided by the CONTENT of the sources (a digest kept next to the library), never by modification times: the
    snapshot a GPU box gets keeps contents, not the order of time stamps (same rule as scrooge_amd.build_library)."""
    so = os.path.join(HERE, "liboracle.so")
    if not os.path.exists(so):
        return True
    try:
        return open(so + ".sources.sha256").read().strip() != _digest()
    except OSError:
        return True


def build(force=False, allow_compile=True):
    """Compile liboracle.so (and _ref when /root/reference is present) if its sources changed.

    allow_compile=False never forks a compiler: a stale or missing library is an error.  Callers that initialise a
    GPU (bench.py, anything under rocprofv3) build BEFORE the first HIP call and pass False afterwards — a compiler
    must never be forked from a process that holds the device."""
    so = os.path.join(HERE, "liboracle.so")
    if force or is_stale():
        if not allow_compile:
            raise RuntimeError("oracle/liboracle.so is missing or older than its sources and building is not allowed here "
                               "(run `python -c 'from oracle.pyoracle import build; build()'` first)")
        import fcntl
        with open(os.path.join(HERE, ".build.lock"), "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            if force or is_stale():
                subprocess.check_call(["make", "-C", HERE, "--no-print-directory", "-B"], stdout=subprocess.DEVNULL)
                with open(so + ".sources.sha256", "w") as fh:
                    fh.write(_digest() + "\n")
    return so


def _as_bytes_list(seqs):
    return [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]


def _marshal(texts, reads):
    texts = _as_bytes_list(texts)
    reads = _as_bytes_list(reads)
    n = len(texts)
    assert n == len(reads)
    tp = (C.c_char_p * n)(*texts)
    rp = (C.c_char_p * n)(*reads)
    tl = (C.c_uint64 * n)(*[len(t) for t in texts])
    rl = (C.c_uint64 * n)(*[len(r) for r in reads])
    bufs = [C.create_string_buffer(4 * len(r) + 1) for r in reads]
    cp = (C.c_char_p * n)(*[C.cast(b, C.c_char_p) for b in bufs])
    eds = (C.c_longlong * n)()
    return n, tp, tl, rp, rl, bufs, cp, eds


def _align_rows(fn, rows, text_off, text_len, read_off, read_len, threads, knobs, with_stats, text_lens=None, read_lens=None):
    """fn: the *_var entry point; text_lens / read_lens: per-row lengths (uint64 arrays) or None = the slot sizes."""
    import numpy as np
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n, stride = rows.shape
    assert text_off + text_len <= stride and read_off + read_len <= stride
    if text_lens is not None:
        text_lens = np.ascontiguousarray(text_lens, dtype=np.uint64)
        assert text_lens.shape == (n,)
    if read_lens is not None:
        read_lens = np.ascontiguousarray(read_lens, dtype=np.uint64)
        assert read_lens.shape == (n,)
    eds = np.zeros(n, dtype=np.int64)
    off = np.zeros(n + 1, dtype=np.uint64)
    cap = n * (2 * int(read_len) + 8)
    runs = np.empty(2 * max(cap, 1), dtype=np.uint8)          # (pages are touched only as far as runs are written)
    st = GoStats()
    ns = C.c_longlong(0)
    vp = C.c_void_p
    args = [C.c_size_t(n), vp(rows.ctypes.data), C.c_uint64(stride), C.c_uint64(text_off), C.c_uint64(text_len),
            C.c_uint64(read_off), C.c_uint64(read_len),
            vp(text_lens.ctypes.data if text_lens is not None else None),
            vp(read_lens.ctypes.data if read_lens is not None else None)] + [C.c_int(k) for k in knobs] + \
           [C.c_int(int(threads)), vp(eds.ctypes.data), vp(off.ctypes.data), vp(runs.ctypes.data), C.c_uint64(cap)] + \
           ([C.byref(st)] if with_stats else []) + [C.byref(ns)]
    fn.restype = C.c_int
    fn.argtypes = None
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError("checker status %d" % rc)
    total = int(off[n])
    return eds, off, runs[: 2 * total].reshape(-1, 2), st.as_dict(), ns.value


class Oracle:
    def __init__(self, allow_compile=True):
        self.lib = C.CDLL(build(allow_compile=allow_compile))
        self.lib.go_align_batch_ascii.restype = C.c_int
        self.lib.go_align_batch_ascii.argtypes = [
            C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64),
            C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.c_int, C.c_int, C.c_int,
            C.POINTER(C.c_char_p), C.POINTER(C.c_longlong), C.POINTER(GoStats),
            C.POINTER(C.c_longlong)]

    def align_rows(self, rows, text_off, text_len, read_off, read_len, W=64, O=33, threads=1, text_lens=None, read_lens=None):
        """A whole batch held in one 2-D uint8 array (a text slot and a read slot per row; text_lens / read_lens: every
        row's own lengths, None = the slot sizes), results as arrays:
        -> (edit distances int64 [n], run offsets uint64 [n + 1], runs uint8 [total, 2] = (count, op), stats dict, kernel_ns)"""
        return _align_rows(self.lib.go_align_batch_rows_var, rows, text_off, text_len, read_off, read_len, threads, (int(W), int(O)), True,
                           text_lens, read_lens)

    def align(self, texts, reads, W=64, O=33, threads=1):
        """-> (edit_distances, cigars, stats dict, kernel_ns)"""
        n, tp, tl, rp, rl, bufs, cp, eds = _marshal(texts, reads)
        st = GoStats()
        ns = C.c_longlong(0)
        rc = self.lib.go_align_batch_ascii(n, tp, tl, rp, rl, W, O, threads, cp, eds,
                                           C.byref(st), C.byref(ns))
        if rc != 0:
            raise ValueError("oracle status %d" % rc)
        return [int(e) for e in eds], [b.value.decode() for b in bufs], st.as_dict(), ns.value


class Reference:
    """The real reference CPU path at its default knobs (W=64, K=64, O=33)."""
    PATH = os.path.join(HERE, "_ref", "libgenasm_ref.so")

    @classmethod
    def path_for(cls, W=64, O=33):
        if (W, O) == (64, 33):
            return cls.PATH
        return os.path.join(HERE, "_ref", "libgenasm_ref_w%d_o%d.so" % (W, O))

    @classmethod
    def available(cls, W=64, O=33):
        return os.path.exists(cls.path_for(W, O))

    def __init__(self, W=64, O=33):
        """W/O other than the defaults load a build of the same reference sources with
        -DCLI_KNOBS -DCLI_W -DCLI_K=W -DCLI_O (oracle/Makefile)."""
        self.lib = C.CDLL(self.path_for(W, O))
        self.lib.ref_align_pairs.restype = C.c_int
        self.lib.ref_align_pairs.argtypes = [
            C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64),
            C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.c_int,
            C.POINTER(C.c_char_p), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        self.lib.ref_align_mapping.restype = C.c_int
        self.lib.ref_align_mapping.argtypes = [
            C.c_char_p, C.c_uint64, C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64),
            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int,
            C.POINTER(C.c_char_p), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]

    def align_rows(self, rows, text_off, text_len, read_off, read_len, threads=1, text_lens=None, read_lens=None):
        """Like Oracle.align_rows, through the reference itself -> (edit distances, run offsets, runs, kernel_ns)"""
        e, off, runs, _, ns = _align_rows(self.lib.ref_align_rows_var, rows, text_off, text_len, read_off, read_len, threads, (), False,
                                          text_lens, read_lens)
        return e, off, runs, ns

    def align(self, texts, reads, threads=1):
        """-> (edit_distances, cigars, kernel_ns)"""
        n, tp, tl, rp, rl, bufs, cp, eds = _marshal(texts, reads)
        ns = C.c_longlong(0)
        rc = self.lib.ref_align_pairs(n, tp, tl, rp, rl, threads, cp, eds, C.byref(ns))
        if rc != 0:
            raise RuntimeError("reference driver status %d" % rc)
        return [int(e) for e in eds], [b.value.decode() for b in bufs], ns.value

    def align_mapping(self, genome, reads, candidates, threads=1):
        """candidates: list (per read) of lists of start_in_reference."""
        genome = genome.encode() if isinstance(genome, str) else bytes(genome)
        reads = _as_bytes_list(reads)
        nr = len(reads)
        offs = [0]
        starts = []
        for c in candidates:
            starts.extend(int(x) for x in c)
            offs.append(len(starts))
        npairs = len(starts)
        rp = (C.c_char_p * nr)(*reads)
        rl = (C.c_uint64 * nr)(*[len(r) for r in reads])
        co = (C.c_uint64 * (nr + 1))(*offs)
        cs = (C.c_uint64 * max(npairs, 1))(*starts)
        bufs = []
        for r, c in zip(reads, candidates):
            bufs.extend(C.create_string_buffer(4 * len(r) + 1) for _ in c)
        cp = (C.c_char_p * max(npairs, 1))(*[C.cast(b, C.c_char_p) for b in bufs])
        eds = (C.c_longlong * max(npairs, 1))()
        ns = C.c_longlong(0)
        rc = self.lib.ref_align_mapping(genome, len(genome), nr, rp, rl, co, cs, threads,
                                        cp, eds, C.byref(ns))
        if rc != 0:
            raise RuntimeError("reference driver status %d" % rc)
        return [int(e) for e in eds[:npairs]], [b.value.decode() for b in bufs], ns.value
