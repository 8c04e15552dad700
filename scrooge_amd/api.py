"""ctypes binding of include/scrooge_amd.h."""
import ctypes as C
import os
import subprocess
import sys
from collections import namedtuple

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# same two fields as the reference's Alignment_t (src/util.hpp:38-41)
Alignment = namedtuple("Alignment", ["cigar", "edit_distance"])

SCRG_OK = 0
SCRG_ERR_INVALID_ARG = 1
SCRG_ERR_BAD_BASE = 2
SCRG_ERR_NO_DEVICE = 3
SCRG_ERR_HIP = 4
SCRG_ERR_OOM = 5
SCRG_ERR_CIGAR_OVERFLOW = 6
SCRG_ABI_VERSION = 7          # include/scrooge_amd.h (tests/test_abi.py holds the two equal)
SEQ_PAD_WORDS = 4
GROUP = 64                     # rows per group of the lane-interleaved layout
SEQ_PAD_WORDS_GROUPS = 2 * GROUP + 2


class ScroogeError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("scrooge_amd status %d: %s" % (status, message))
        self.status = status


class Params(C.Structure):
    _fields_ = [("W", C.c_int32), ("O", C.c_int32), ("lanes_per_pair", C.c_int32),
                ("lds_rows", C.c_int32), ("waves_per_cu", C.c_int32),
                ("sort_by_length", C.c_int32), ("text_stride_words", C.c_int32), ("read_stride_words", C.c_int32),
                ("outputs", C.c_int32), ("reserved", C.c_int32 * 2), ("stranded", C.c_int32)]


READ_REVCOMP = 1 << 63            # include/scrooge_amd.h: SCRG_READ_REVCOMP


class PairDesc(C.Structure):
    _fields_ = [("text_off", C.c_uint64), ("text_len", C.c_uint64), ("read_off", C.c_uint64),
                ("read_len", C.c_uint64), ("cigar_off", C.c_uint64), ("cigar_cap", C.c_uint64)]


class Run(C.Structure):
    _fields_ = [("count", C.c_uint8), ("op", C.c_char)]


class Result(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64),
                ("edit_distance", C.POINTER(C.c_int64)),
                ("pair_status", C.POINTER(C.c_uint32)),
                ("run_offset", C.POINTER(C.c_uint64)),
                ("runs", C.POINTER(Run)),
                ("cigar_offset", C.POINTER(C.c_uint64)),
                ("cigar_text", C.POINTER(C.c_char)),
                ("kernel_ns", C.c_int64), ("pack_ns", C.c_int64), ("total_ns", C.c_int64)]


def library_path(variant=None):
    """The shipped library, or — variant="select" — the TEST build of the same sources (-DSCRG_SELECT: ab_libs/lib_select.so),
    in which scrg_params.reserved[0] selects between formulations that give identical results; the parity tests that compare
    formulations load it next to the shipped one.  SCRG_LIB names another build of the shipped library (kernel A/B)."""
    if variant == "select":
        return os.path.join(HERE, "..", "ab_libs", "lib_select.so")
    if variant is not None:
        raise ValueError("unknown library variant %r" % (variant,))
    return os.environ.get("SCRG_LIB") or os.path.join(HERE, "libscrooge_amd.so")


def _source_digest(srcs):
    import hashlib
    h = hashlib.sha256()
    for f in sorted(srcs):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build_library(force=False, variant=None):
    """Compile the HIP kernels + host code for gfx950 (hipcc cross-compiles without a GPU); variant="select": the test build
    (library_path).

    Whether the built library is current is decided by the CONTENT of the sources (a digest kept next to the
    library), not by modification times: a copy of the tree (the snapshot a GPU box gets) keeps the contents but
    not necessarily the order of the time stamps, and a needless rebuild costs a minute and needs a compiler."""
    so = library_path(variant)
    if variant is None and os.environ.get("SCRG_LIB"):
        return so
    src_dir = os.path.join(HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if not f.startswith(".")] + \
        [os.path.join(HERE, "..", "include", f) for f in ("scrooge_amd.h", "scrooge_amd_io.h", "scrooge_amd_device.hpp")]
    stamp = so + ".sources.sha256"
    digest = _source_digest(srcs)

    def is_stale():
        if not os.path.exists(so):
            return True
        try:
            return open(stamp).read().strip() != digest
        except OSError:
            return True

    if force or is_stale():
        # several ranks of one job may get here at once: serialise, and re-check under the lock
        import fcntl
        with open(os.path.join(HERE, ".build.lock"), "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            if force or is_stale():
                # (make echoes the compiler command: keep it off stdout, which callers such as bench.py reserve for their
                # own output; -B because make's own idea of freshness is the time stamps)
                cmd = ["make", "-C", src_dir, "--no-print-directory", "-B"] + (["VARIANT=%s" % variant] if variant else [])
                subprocess.check_call(cmd, stdout=sys.stderr)
                with open(stamp, "w") as fh:
                    fh.write(digest + "\n")
    return so


_VARIANT_LIBS = {}


def load_library(variant=None):
    """Load libscrooge_amd.so (variant="select": the test build, a second handle next to it); raises (never falls back) when
    it is missing."""
    global _LIB
    if variant is None and _LIB is not None:
        return _LIB
    if variant is not None and variant in _VARIANT_LIBS:
        return _VARIANT_LIBS[variant]
    so = library_path(variant)
    # PyTorch-ROCm ships its own libamdhip64.so; whichever HIP runtime is loaded first serves the whole
    # process.  If this library pulled in /opt/rocm's copy first, a later `import torch` finds no GPU.
    # So when torch is installed, let it load its runtime before us (SCRG_NO_TORCH_PRELOAD=1 skips this).
    if "torch" not in sys.modules and not os.environ.get("SCRG_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.exists(so):
        raise ScroogeError(SCRG_ERR_NO_DEVICE,
                           "%s not built; run scrooge_amd.build_library() (needs hipcc)" % so)
    lib = C.CDLL(so)
    vp, u64, i32p = C.c_void_p, C.c_uint64, C.POINTER(C.c_int32)
    sigs = {
        "scrg_params_default": (None, [C.POINTER(Params)]),
        "scrg_params_resolve": (C.c_int32, [C.POINTER(Params), C.POINTER(Params)]),
        "scrg_ctx_create": (C.c_int32, [C.c_int, C.POINTER(vp)]),
        "scrg_ctx_destroy": (None, [vp]),
        "scrg_ctx_set_stream": (C.c_int32, [vp, vp]),
        "scrg_ctx_use_own_stream": (C.c_int32, [vp]),
        "scrg_stream_create": (C.c_int32, [C.c_int, C.c_int, C.POINTER(vp)]),
        "scrg_stream_destroy": (C.c_int32, [vp]),
        "scrg_last_error": (C.c_char_p, [vp]),
        "scrg_status_string": (C.c_char_p, [C.c_int32]),
        "scrg_set_log": (None, [C.c_int]),
        "scrg_get_log": (C.c_int, []),
        "scrg_device_count": (C.c_int, []),
        "scrg_build_flags": (C.c_int, []),
        "scrg_abi_version": (C.c_int, []),
        "scrg_result_free": (None, [C.POINTER(Result)]),
        "scrg_result_pool_trim": (None, []),
        "scrg_align_pairs": (C.c_int32, [vp, C.POINTER(Params), u64, C.POINTER(C.c_char_p),
                                         C.POINTER(u64), C.POINTER(C.c_char_p), C.POINTER(u64),
                                         C.POINTER(C.POINTER(Result))]),
        "scrg_align_mapping": (C.c_int32, [vp, C.POINTER(Params), C.c_char_p, u64, u64,
                                           C.POINTER(C.c_char_p), C.POINTER(u64), C.POINTER(u64),
                                           C.POINTER(u64), C.POINTER(C.POINTER(Result))]),
        "scrg_align_pairs_multi": (C.c_int32, [i32p, C.c_int32, C.POINTER(Params), u64, C.POINTER(C.c_char_p), C.POINTER(u64),
                                               C.POINTER(C.c_char_p), C.POINTER(u64), C.POINTER(C.POINTER(Result))]),
        "scrg_align_mapping_multi": (C.c_int32, [i32p, C.c_int32, C.POINTER(Params), C.c_char_p, u64, u64, C.POINTER(C.c_char_p),
                                                 C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), vp, C.POINTER(C.POINTER(Result))]),
        "scrg_host_plan": (C.c_int32, [C.POINTER(Params), C.c_int32, u64, vp, vp, vp, vp, u64, C.POINTER(u64)]),
        "scrg_multi_release": (None, []),
        "scrg_multi_last_error": (C.c_char_p, []),
        "scrg_genome_set": (C.c_int32, [vp, C.c_char_p, u64]),
        "scrg_genome_clear": (None, [vp]),
        "scrg_align_mapping_resident": (C.c_int32, [vp, vp, u64, vp, vp, vp, vp, vp, vp]),
        "scrg_pack_planar": (C.c_int32, [vp, vp, u64, vp, vp]),
        "scrg_pack_planar_host": (C.c_int32, [vp, u64, vp, u64, u64]),
        "scrg_pack_planar_groups": (C.c_int32, [vp, vp, u64, u64, vp, vp]),
        "scrg_compact_runs_packed": (C.c_int32, [vp, vp, u64, vp, vp, vp, vp, vp]),
        "scrg_unpack_runs": (C.c_int32, [vp, u64, vp, vp]),
        "scrg_encode_edit_stream": (C.c_int32, [vp, C.POINTER(Params), u64, vp, vp, vp, vp, u64, vp, vp, vp]),
        "scrg_decode_edit_stream": (C.c_int32, [vp, vp, u64, vp, u64, vp, vp, vp, u64, vp, vp, u64, vp, vp]),
        "scrg_edit_stream_to_runs": (C.c_int32, [C.POINTER(Params), u64, vp, u64, vp, u64, C.POINTER(u64)]),
        "scrg_edit_stream_to_runs_lane": (C.c_int32, [C.POINTER(Params), u64, vp, u64, vp, u64, C.POINTER(u64)]),
        "scrg_runs_to_edit_stream": (C.c_int32, [C.POINTER(Params), vp, u64, vp, u64, C.POINTER(u64)]),
        "scrg_align_device": (C.c_int32, [vp, C.POINTER(Params), u64, vp, vp, vp, vp, vp, vp]),
        "scrg_compact_runs": (C.c_int32, [vp, u64, vp, vp, vp, vp, vp]),
        "scrg_align_device_edits": (C.c_int32, [vp, C.POINTER(Params), u64, vp, vp, vp, vp, vp, vp, vp]),
        "scrg_ascii_to_twobit": (C.c_int32, [vp, u64, vp, vp, vp, vp, vp, vp]),
        "scrg_query_launch": (C.c_int32, [vp, C.POINTER(Params), i32p, i32p, i32p, i32p]),
        "scrg_last_kernel_ms": (C.c_int32, [vp, C.POINTER(C.c_float)]),
        "scrg_debug_stats": (C.c_int32, [vp, C.POINTER(C.c_uint64)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)   # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    # the signatures above are those of interface version SCRG_ABI_VERSION (include/scrooge_amd.h): a library built from other
    # sources (SCRG_LIB, a stale copy) must say so instead of taking shifted arguments
    if lib.scrg_abi_version() != SCRG_ABI_VERSION:
        raise ScroogeError(SCRG_ERR_INVALID_ARG, "%s has interface version %d, this binding speaks version %d"
                           % (so, lib.scrg_abi_version(), SCRG_ABI_VERSION))
    if variant is None:
        _LIB = lib
    else:
        _VARIANT_LIBS[variant] = lib
    return lib


EXPORTED_SYMBOLS = [
    "scrg_params_default", "scrg_params_resolve", "scrg_ctx_create", "scrg_ctx_destroy", "scrg_ctx_set_stream",
    "scrg_ctx_use_own_stream", "scrg_stream_create", "scrg_stream_destroy",
    "scrg_last_error", "scrg_status_string", "scrg_set_log", "scrg_get_log", "scrg_device_count", "scrg_build_flags", "scrg_abi_version",
    "scrg_result_free", "scrg_result_pool_trim", "scrg_align_pairs", "scrg_align_mapping", "scrg_align_pairs_multi", "scrg_align_mapping_multi",
    "scrg_host_plan", "scrg_multi_release", "scrg_multi_last_error", "scrg_genome_set", "scrg_genome_clear",
    "scrg_align_mapping_resident", "scrg_pack_planar", "scrg_pack_planar_host", "scrg_pack_planar_groups",
    "scrg_align_device", "scrg_align_device_edits", "scrg_compact_runs", "scrg_compact_runs_packed", "scrg_unpack_runs",
    "scrg_encode_edit_stream", "scrg_decode_edit_stream", "scrg_edit_stream_to_runs", "scrg_edit_stream_to_runs_lane", "scrg_runs_to_edit_stream", "scrg_ascii_to_twobit", "scrg_query_launch",
    "scrg_last_kernel_ms", "scrg_debug_stats"]


def edit_stream_to_cigar(stream, read_len, W=64, O=33, lane_form=False):
    """Host-side decoder of ONE pair's edit stream (bytes, format version 2: window ends on the wire) -> the CIGAR text the
    aligner returns (scrg_edit_stream_to_runs: no GPU involved; it also holds the window ends to the window loop of W/O.
    lane_form=True: through the state machine the device decoder runs in every lane, scrg_edit_stream_to_runs_lane, which
    like the device does not look at the window geometry).  Raises ScroogeError for a malformed stream."""
    lib = load_library()
    fn = lib.scrg_edit_stream_to_runs_lane if lane_form else lib.scrg_edit_stream_to_runs
    p = Params()
    lib.scrg_params_default(C.byref(p))
    p.W, p.O = int(W), int(O)
    buf = (C.c_uint8 * max(1, len(stream))).from_buffer_copy(bytes(stream) or b"\0")
    n = C.c_uint64(0)
    st = fn(C.byref(p), int(read_len), buf, len(stream), None, 0, C.byref(n))
    if st not in (SCRG_OK, SCRG_ERR_CIGAR_OVERFLOW):
        raise ScroogeError(st, "malformed edit stream")
    runs = (C.c_uint8 * (2 * max(1, n.value)))()
    st = fn(C.byref(p), int(read_len), buf, len(stream), runs, n.value, C.byref(n))
    if st != SCRG_OK:
        raise ScroogeError(st, "malformed edit stream")
    return "".join("%d%s" % (runs[2 * k], chr(runs[2 * k + 1])) for k in range(n.value))


def cigar_to_edit_stream(cigar, W=64, O=33):
    """Host-side encoder: CIGAR text (=, X, I, D runs) -> canonical edit stream bytes (scrg_runs_to_edit_stream; the window
    loop of W/O places the window-end bytes)."""
    import re
    lib = load_library()
    p = Params()
    lib.scrg_params_default(C.byref(p))
    p.W, p.O = int(W), int(O)
    items = re.findall(r"(\d+)([=XID])", cigar)
    if "".join(a + b for a, b in items) != cigar:
        raise ValueError("not a CIGAR of =, X, I, D runs: %r" % cigar[:40])
    raw = bytearray()
    for cnt, op in items:
        c = int(cnt)
        while c > 0:                       # scrg_run counts are one byte
            raw += bytes((min(c, 255), ord(op)))
            c -= 255
    runs = (C.c_uint8 * max(1, len(raw))).from_buffer_copy(bytes(raw) or b"\0")
    n = C.c_uint64(0)
    lib.scrg_runs_to_edit_stream(C.byref(p), runs, len(raw) // 2, None, 0, C.byref(n))
    out = (C.c_uint8 * max(1, n.value))()
    st = lib.scrg_runs_to_edit_stream(C.byref(p), runs, len(raw) // 2, out, n.value, C.byref(n))
    if st != SCRG_OK:
        raise ScroogeError(st, "bad runs")
    return bytes(out[: n.value])


def create_stream(device=0, priority=0):
    """A raw HIP stream handle of the given priority (-1 high, 0 normal, 1 low); wrap it with
    torch.cuda.ExternalStream(handle) to use it from torch."""
    lib = load_library()
    h = C.c_void_p()
    st = lib.scrg_stream_create(int(device), int(priority), C.byref(h))
    if st != SCRG_OK:
        raise ScroogeError(st, lib.scrg_status_string(st).decode())
    return h.value


def _bytes_list(seqs):
    return [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]


def _ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else C.c_void_p(t.data_ptr())


class Aligner:
    """One handle per GPU (scrg_ctx).

    ``align_pairs`` / ``align_mapping`` mirror the two overloads of
    genasm_gpu::align_all (src/genasm_gpu.hpp:7-8) and return a list of
    ``Alignment(cigar, edit_distance)`` in the reference's result order.
    """

    def __init__(self, device=0, variant=None, **params):
        self.lib = load_library(variant)
        h = C.c_void_p()
        st = self.lib.scrg_ctx_create(int(device), C.byref(h))
        if st != SCRG_OK:
            raise ScroogeError(st, self.lib.scrg_status_string(st).decode())
        self.h = h
        self.device = int(device)
        self.params = self.make_params(**params)
        self.last_timing = {}

    def make_params(self, **kw):
        p = Params()
        self.lib.scrg_params_default(C.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise TypeError("unknown parameter %r" % k)
            setattr(p, k, int(v))
        return p

    def close(self):
        if getattr(self, "h", None):
            self.lib.scrg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st, allow=()):
        if st != SCRG_OK and st not in allow:
            raise ScroogeError(st, (self.lib.scrg_last_error(self.h) or b"").decode() or
                               self.lib.scrg_status_string(st).decode())

    def _params(self, kw):
        if not kw:
            return self.params
        p = Params()
        C.memmove(C.byref(p), C.byref(self.params), C.sizeof(Params))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise TypeError("unknown parameter %r" % k)
            setattr(p, k, int(v))
        return p

    def _collect(self, res_p, st):
        try:
            r = res_p.contents
            n = int(r.n_pairs)
            text = C.string_at(r.cigar_text, int(r.cigar_offset[n])) if n else b""
            out = []
            for i in range(n):
                a, b = int(r.cigar_offset[i]), int(r.cigar_offset[i + 1])
                out.append(Alignment(text[a:b - 1].decode(), int(r.edit_distance[i])))
            self.last_timing = {"kernel_ns": int(r.kernel_ns), "pack_ns": int(r.pack_ns),
                                "total_ns": int(r.total_ns)}
            self.last_status = [int(r.pair_status[i]) for i in range(n)]
        finally:
            self.lib.scrg_result_free(res_p)
        return out

    def _collect_arrays(self, res_p, st):
        """The result as numpy arrays (copies): no per-pair Python objects — for batches of millions of pairs."""
        import numpy as np
        try:
            r = res_p.contents
            n = int(r.n_pairs)

            def arr(ptr, count, dtype):
                if count == 0:
                    return np.zeros(0, dtype=dtype)
                return np.ctypeslib.as_array(ptr, shape=(count,)).view(dtype).copy()

            run_offset = arr(r.run_offset, n + 1, np.uint64)
            cigar_offset = arr(r.cigar_offset, n + 1, np.uint64)
            total_runs = int(run_offset[n]) if n else 0
            out = {"edit_distance": arr(r.edit_distance, n, np.int64),
                   "status": arr(r.pair_status, n, np.uint32),
                   "run_offset": run_offset,
                   # (views of the library's arrays copied with numpy: ctypes.string_at takes no more than 2 GB, a batch of a
                   # million 10 kb pairs has 4.3 GB of runs)
                   "runs": arr(C.cast(r.runs, C.POINTER(C.c_uint8)), 2 * total_runs, np.uint8).reshape(-1, 2)
                           if total_runs else np.zeros((0, 2), np.uint8),      # columns: count, op
                   "cigar_offset": cigar_offset,
                   "cigar_text": arr(C.cast(r.cigar_text, C.POINTER(C.c_uint8)), int(cigar_offset[n]), np.uint8).tobytes() if n else b""}
            self.last_timing = {"kernel_ns": int(r.kernel_ns), "pack_ns": int(r.pack_ns),
                                "total_ns": int(r.total_ns)}
        finally:
            self.lib.scrg_result_free(res_p)
        return out

    # -- genasm_gpu::align_all(texts, queries)  (src/genasm_gpu.cu:982-1065) --------
    def _finish(self, res, st, arrays, strict):
        """Collect (and free) the library's result; a pair that overflowed its CIGAR slice is an error unless
        strict=False, in which case the truncated result is returned and `last_status` / `status` says which pairs
        (the C++ shim throws in the same case, scrooge_amd.hpp)."""
        out = self._collect_arrays(res, st) if arrays else self._collect(res, st)
        if st == SCRG_ERR_CIGAR_OVERFLOW and strict:
            raise ScroogeError(st, (self.lib.scrg_last_error(self.h) or b"").decode() or
                               "at least one pair overflowed its CIGAR slice")
        return out

    def align_pairs(self, texts, queries, arrays=False, strict=True, **kw):
        texts, queries = _bytes_list(texts), _bytes_list(queries)
        if len(texts) != len(queries):
            raise ValueError("texts and queries differ in length")   # reference: assert, genasm_cpu.cpp:559
        n = len(texts)
        tp = (C.c_char_p * max(n, 1))(*texts)
        qp = (C.c_char_p * max(n, 1))(*queries)
        tl = (C.c_uint64 * max(n, 1))(*[len(t) for t in texts])
        ql = (C.c_uint64 * max(n, 1))(*[len(q) for q in queries])
        res = C.POINTER(Result)()
        st = self.lib.scrg_align_pairs(self.h, C.byref(self._params(kw)), n, tp, tl, qp, ql,
                                       C.byref(res))
        self._check(st, allow=(SCRG_ERR_CIGAR_OVERFLOW,))
        return self._finish(res, st, arrays, strict)

    def align_pairs_rows(self, rows, text_off, text_lens, read_off, read_lens, strict=True, devices=None, **kw):
        """scrg_align_pairs on sequences that sit in ONE 2-D uint8 numpy array (row p: the text of pair p at byte
        text_off, its read at byte read_off — bench.py's staging layout), lengths as integers or per-row arrays: the
        pointer arrays are built with numpy, so a batch of 100 k x 10 kb pairs costs no per-pair Python objects.
        devices: a device list -> scrg_align_pairs_multi (a device may be listed more than once).
        -> the result as numpy arrays (see _collect_arrays); `last_timing` has the library's own clock."""
        import numpy as np
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        n, stride = rows.shape
        base = rows.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(stride)
        tp = (base + np.uint64(text_off)).astype(np.uint64)
        qp = (base + np.uint64(read_off)).astype(np.uint64)
        tl = np.ascontiguousarray(np.broadcast_to(np.asarray(text_lens, dtype=np.uint64), (n,)))
        ql = np.ascontiguousarray(np.broadcast_to(np.asarray(read_lens, dtype=np.uint64), (n,)))
        pp, up = C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)
        res = C.POINTER(Result)()
        if devices is not None:
            dv = (C.c_int32 * len(devices))(*devices)
            st = self.lib.scrg_align_pairs_multi(dv, len(devices), C.byref(self._params(kw)), n, C.cast(tp.ctypes.data, pp),
                                                 C.cast(tl.ctypes.data, up), C.cast(qp.ctypes.data, pp), C.cast(ql.ctypes.data, up), C.byref(res))
            if st not in (SCRG_OK, SCRG_ERR_CIGAR_OVERFLOW):
                raise ScroogeError(st, (self.lib.scrg_multi_last_error() or b"").decode())
            return self._finish(res, st, True, strict)
        st = self.lib.scrg_align_pairs(self.h, C.byref(self._params(kw)), n, C.cast(tp.ctypes.data, pp), C.cast(tl.ctypes.data, up),
                                       C.cast(qp.ctypes.data, pp), C.cast(ql.ctypes.data, up), C.byref(res))
        self._check(st, allow=(SCRG_ERR_CIGAR_OVERFLOW,))
        return self._finish(res, st, True, strict)

    def align_mapping_rows(self, genome, read_rows, read_lens, cand_offsets, cand_start, strict=True, **kw):
        """scrg_align_mapping (genome: bytes / uint8 array) or scrg_align_mapping_resident (genome=None) with the reads in
        one 2-D uint8 numpy array (one read per row) and the candidates as numpy arrays (cand_offsets: n_reads + 1)."""
        import numpy as np
        read_rows = np.ascontiguousarray(read_rows, dtype=np.uint8)
        nr, stride = read_rows.shape
        rp = (read_rows.ctypes.data + np.arange(nr, dtype=np.uint64) * np.uint64(stride)).astype(np.uint64)
        rl = np.ascontiguousarray(np.broadcast_to(np.asarray(read_lens, dtype=np.uint64), (nr,)))
        co = np.ascontiguousarray(cand_offsets, dtype=np.uint64)
        cs = np.ascontiguousarray(cand_start, dtype=np.uint64)
        assert co.shape == (nr + 1,) and cs.shape == (int(co[nr]),)
        pp, up = C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)
        res = C.POINTER(Result)()
        if genome is None:
            st = self.lib.scrg_align_mapping_resident(self.h, C.byref(self._params(kw)), nr, C.c_void_p(rp.ctypes.data), C.c_void_p(rl.ctypes.data),
                                                      C.c_void_p(co.ctypes.data), C.c_void_p(cs.ctypes.data), None, C.byref(res))
        else:
            g = np.ascontiguousarray(np.frombuffer(genome, dtype=np.uint8) if isinstance(genome, (bytes, bytearray)) else genome, dtype=np.uint8)
            st = self.lib.scrg_align_mapping(self.h, C.byref(self._params(kw)), C.cast(g.ctypes.data, C.c_char_p), g.size, nr,
                                             C.cast(rp.ctypes.data, pp), C.cast(rl.ctypes.data, up), C.cast(co.ctypes.data, up),
                                             C.cast(cs.ctypes.data, up), C.byref(res))
        self._check(st, allow=(SCRG_ERR_CIGAR_OVERFLOW,))
        return self._finish(res, st, True, strict)

    def set_genome_array(self, genome_u8):
        """set_genome for a uint8 numpy array (no copy into a bytes object)."""
        import numpy as np
        g = np.ascontiguousarray(genome_u8, dtype=np.uint8)
        self._check(self.lib.scrg_genome_set(self.h, C.cast(g.ctypes.data, C.c_char_p), g.size))

    def align_pairs_multi(self, devices, texts, queries, arrays=False, strict=True, **kw):
        """scrg_align_pairs_multi: the same call spread over several GPUs (a device may be listed more than once)."""
        texts, queries = _bytes_list(texts), _bytes_list(queries)
        n = len(texts)
        tp = (C.c_char_p * max(n, 1))(*texts)
        qp = (C.c_char_p * max(n, 1))(*queries)
        tl = (C.c_uint64 * max(n, 1))(*[len(t) for t in texts])
        ql = (C.c_uint64 * max(n, 1))(*[len(q) for q in queries])
        dv = (C.c_int32 * len(devices))(*devices)
        res = C.POINTER(Result)()
        st = self.lib.scrg_align_pairs_multi(dv, len(devices), C.byref(self._params(kw)), n, tp, tl, qp, ql, C.byref(res))
        if st not in (SCRG_OK, SCRG_ERR_CIGAR_OVERFLOW):
            raise ScroogeError(st, (self.lib.scrg_multi_last_error() or b"").decode())
        return self._finish(res, st, arrays, strict)

    def align_mapping_multi(self, devices, genome, reads, candidates, reverse=None, arrays=False, strict=True, **kw):
        """scrg_align_mapping_multi: read mapping spread over several GPUs; reverse: optional list (per read) of 0/1 lists."""
        genome = genome.encode() if isinstance(genome, str) else bytes(genome)
        reads = _bytes_list(reads)
        nr = len(reads)
        offs, starts, rev = [0], [], []
        for k, c in enumerate(candidates):
            starts.extend(int(x) for x in c)
            rev.extend(int(x) for x in (reverse[k] if reverse is not None else [0] * len(c)))
            offs.append(len(starts))
        rp = (C.c_char_p * max(nr, 1))(*reads)
        rl = (C.c_uint64 * max(nr, 1))(*[len(r) for r in reads])
        co = (C.c_uint64 * (nr + 1))(*offs)
        cs = (C.c_uint64 * max(len(starts), 1))(*starts)
        cr = (C.c_uint8 * max(len(rev), 1))(*rev)
        dv = (C.c_int32 * len(devices))(*devices)
        res = C.POINTER(Result)()
        st = self.lib.scrg_align_mapping_multi(dv, len(devices), C.byref(self._params(kw)), genome, len(genome), nr, rp, rl, co, cs,
                                               C.cast(cr, C.c_void_p) if reverse is not None else None, C.byref(res))
        if st not in (SCRG_OK, SCRG_ERR_CIGAR_OVERFLOW):
            raise ScroogeError(st, (self.lib.scrg_multi_last_error() or b"").decode())
        return self._finish(res, st, arrays, strict)

    # -- genasm_gpu::align_all(genome, reads)  (src/genasm_gpu.cu:890-980) ----------
    def set_genome(self, genome):
        """Stage, transfer and pack a genome once; align_mapping(None, reads, candidates) then aligns batches
        against it without touching it again (scrg_genome_set / scrg_align_mapping_resident)."""
        genome = genome.encode() if isinstance(genome, str) else bytes(genome)
        self._check(self.lib.scrg_genome_set(self.h, genome, len(genome)))

    def clear_genome(self):
        self.lib.scrg_genome_clear(self.h)

    def align_mapping(self, genome, reads, candidates, arrays=False, strict=True, **kw):
        """candidates[r] = list of start_in_reference for read r (forward strand).  genome=None: the genome
        left resident by set_genome()."""
        if genome is not None:
            genome = genome.encode() if isinstance(genome, str) else bytes(genome)
        reads = _bytes_list(reads)
        nr = len(reads)
        if len(candidates) != nr:
            raise ValueError("one candidate list per read expected")
        offs, starts = [0], []
        for c in candidates:
            starts.extend(int(x) for x in c)
            offs.append(len(starts))
        rp = (C.c_char_p * max(nr, 1))(*reads)
        rl = (C.c_uint64 * max(nr, 1))(*[len(r) for r in reads])
        co = (C.c_uint64 * (nr + 1))(*offs)
        cs = (C.c_uint64 * max(len(starts), 1))(*starts)
        res = C.POINTER(Result)()
        if genome is None:
            st = self.lib.scrg_align_mapping_resident(self.h, C.byref(self._params(kw)), nr, rp, rl, co, cs, None,
                                                      C.byref(res))
        else:
            st = self.lib.scrg_align_mapping(self.h, C.byref(self._params(kw)), genome, len(genome),
                                             nr, rp, rl, co, cs, C.byref(res))
        self._check(st, allow=(SCRG_ERR_CIGAR_OVERFLOW,))
        return self._finish(res, st, arrays, strict)

    # -- device-pointer layer (torch tensors as plain device memory) -----------
    def set_stream(self, stream_handle):
        """stream_handle: a hipStream_t as an integer (0 = the device's null stream)."""
        self._check(self.lib.scrg_ctx_set_stream(self.h, C.c_void_p(stream_handle)))
        self._stream = int(stream_handle)

    def use_own_stream(self):
        self._check(self.lib.scrg_ctx_use_own_stream(self.h))
        self._stream = None

    def restore_stream(self, saved):
        """saved: what `stream_setting` returned — the handle enqueues where it did then."""
        if saved is None:
            self.use_own_stream()
        else:
            self.set_stream(saved)

    @property
    def stream_setting(self):
        """None: the handle's own stream; else the hipStream_t (integer) given to set_stream."""
        return getattr(self, "_stream", None)

    def pack_planar(self, ascii_u8, planar_u64, bad_u32):
        n_words = ascii_u8.numel() // 32
        self._check(self.lib.scrg_pack_planar(self.h, _ptr(ascii_u8), n_words, _ptr(planar_u64),
                                              _ptr(bad_u32)))

    def pack_planar_groups(self, ascii_u8, n_rows, words_per_row, planar_u64, bad_u32):
        """ASCII rows -> planar 2-bit in the lane-interleaved layout (scrg_pack_planar_groups): word w of row r at
        ((r // 64) * words_per_row + w) * 64 + r % 64; align with text_stride_words = read_stride_words = 64."""
        self._check(self.lib.scrg_pack_planar_groups(self.h, _ptr(ascii_u8), int(n_rows), int(words_per_row),
                                                     _ptr(planar_u64), _ptr(bad_u32)))

    def align_device(self, n_pairs, seq, pairs, runs, ed, n_runs, status, **kw):
        self._check(self.lib.scrg_align_device(self.h, C.byref(self._params(kw)), int(n_pairs),
                                               _ptr(seq), _ptr(pairs), _ptr(runs), _ptr(ed),
                                               _ptr(n_runs), _ptr(status)))

    def align_device_edits(self, n_pairs, seq, pairs, streams_u8, ed, stream_len, status, n_runs=None, **kw):
        """Like align_device, but the pairs' slices receive EDIT STREAMS (one byte per edit and per window end) and stream_len their
        lengths in bytes: the one-pair-per-lane kernels only (lanes_per_pair = 1, the default for every W/O).
        n_runs (optional int32 tensor): the run count of every alignment, for a receiver that decodes in one pass."""
        self._check(self.lib.scrg_align_device_edits(self.h, C.byref(self._params(kw)), int(n_pairs),
                                                     _ptr(seq), _ptr(pairs), _ptr(streams_u8), _ptr(ed),
                                                     _ptr(stream_len), _ptr(status), _ptr(n_runs)))

    def compact_runs(self, n_pairs, pairs, runs, n_runs, dense_off, dense):
        self._check(self.lib.scrg_compact_runs(self.h, int(n_pairs), _ptr(pairs), _ptr(runs),
                                               _ptr(n_runs), _ptr(dense_off), _ptr(dense)))

    def compact_runs_packed(self, n_pairs, pairs, runs, n_runs, dense_off, packed_u8, **kw):
        """Like compact_runs, one byte per run (op << 6 | count; W-O <= 63): the transfer format of the RCCL gather."""
        self._check(self.lib.scrg_compact_runs_packed(self.h, C.byref(self._params(kw)), int(n_pairs), _ptr(pairs),
                                                      _ptr(runs), _ptr(n_runs), _ptr(dense_off), _ptr(packed_u8)))

    def unpack_runs(self, n_runs, packed_u8, runs_u8):
        """Restores scrg_run pairs (2 bytes each) from packed runs."""
        self._check(self.lib.scrg_unpack_runs(self.h, int(n_runs), _ptr(packed_u8), _ptr(runs_u8)))

    def encode_edit_stream(self, n_pairs, pairs, runs, n_runs, stream_u8, stream_off_i64, stream_len_i32, total_i64, **kw):
        """Runs -> edit stream (one byte per edit and per window, scrooge_amd.h): the transfer format of the RCCL gather.
        total_i64[0] = bytes of stream_u8 used, total_i64[1] = pairs that did not fit.  W / O (kw) place the window ends."""
        self._check(self.lib.scrg_encode_edit_stream(self.h, C.byref(self._params(kw)), int(n_pairs), _ptr(pairs), _ptr(runs), _ptr(n_runs),
                                                     _ptr(stream_u8), int(stream_u8.numel()), _ptr(stream_off_i64),
                                                     _ptr(stream_len_i32), _ptr(total_i64)))

    def decode_edit_stream(self, n_pairs, stream_u8, stream_off_i64, stream_len_i32, read_len_i64, read_len_stride,
                           dense_off_i64, dense_u8, n_runs_i32, bad_i32, **kw):
        """Edit stream -> scrg_run pairs, window by window as the stream says; dense_u8 None: count only."""
        self._check(self.lib.scrg_decode_edit_stream(self.h, C.byref(self._params(kw)), int(n_pairs), _ptr(stream_u8),
                                                     int(stream_u8.numel()), _ptr(stream_off_i64), _ptr(stream_len_i32), _ptr(read_len_i64),
                                                     int(read_len_stride),
                                                     _ptr(dense_off_i64) if dense_off_i64 is not None else None,
                                                     _ptr(dense_u8) if dense_u8 is not None else None,
                                                     int(dense_u8.numel() * dense_u8.element_size() // 2) if dense_u8 is not None else 0,      # capacity in runs (2 bytes each), whatever the tensor's element type
                                                     _ptr(n_runs_i32), _ptr(bad_i32)))

    def ascii_to_twobit(self, count, lens, ascii_off, ascii, twobit_off, twobit, bad):
        self._check(self.lib.scrg_ascii_to_twobit(self.h, int(count), _ptr(lens), _ptr(ascii_off),
                                                  _ptr(ascii), _ptr(twobit_off), _ptr(twobit),
                                                  _ptr(bad)))

    def last_kernel_ms(self):
        ms = C.c_float(0)
        self._check(self.lib.scrg_last_kernel_ms(self.h, C.byref(ms)))
        return float(ms.value)

    def debug_stats(self):
        out = (C.c_uint64 * 12)()
        self._check(self.lib.scrg_debug_stats(self.h, out))
        return {"rounds": int(out[0]), "dc_steps": int(out[1]), "tb_macro_steps": int(out[2]),
                "cycles_fetch": int(out[3]), "cycles_setup": int(out[4]), "cycles_dc": int(out[5]),
                "cycles_tb": int(out[6]), "cycles_tb_loop": int(out[7]), "diag_rounds": int(out[8]),
                "diag_fallbacks": int(out[9]), "cycles_diag_dc": int(out[10]), "cycles_diag_tb": int(out[11])}

    def debug_stats_lane(self):
        """The same counters as read by the one-pair-per-lane kernel (scrooge_amd.h: scrg_debug_stats): window rounds,
        rounds with a short text window, shader cycles per phase summed over wavefronts, wavefront life times on the
        100 MHz wall clock."""
        out = (C.c_uint64 * 12)()
        self._check(self.lib.scrg_debug_stats(self.h, out))
        o = [int(x) for x in out]
        big = 1 << 62
        return {"rounds": o[0], "short_text_rounds": o[1], "cycles_pass1": o[2], "cycles_fetch": o[3], "cycles_setup": o[4],
                "cycles_table": o[5], "cycles_traceback": o[6], "life_ticks_sum": o[7], "last_start": o[8],
                "first_start": big - o[9], "last_end": o[10], "first_end": big - o[11]}

    def resolved_params(self, **kw):
        """The parameters a launch with these overrides will really use (defaults filled in)."""
        out = Params()
        st = self.lib.scrg_params_resolve(C.byref(self._params(kw)), C.byref(out))
        if st != SCRG_OK:
            raise ScroogeError(st, "invalid parameters")
        return out

    def query_launch(self, **kw):
        a, b, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.lib.scrg_query_launch(self.h, C.byref(self._params(kw)), C.byref(a),
                                               C.byref(b), C.byref(c), C.byref(d)))
        return {"n_waves": a.value, "pairs_per_wave": b.value, "lds_bytes": c.value,
                "n_cus": d.value}
