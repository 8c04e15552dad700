"""ctypes binding of include/scrooge_amd_io.h: the reference's read-mapping front door
(FASTA + FASTQ + MAF/PAF, src/util.cpp:45-336), affine re-scoring
(src/cpu_baseline.cpp:694-725) and the CIGAR validator (src/tests.cu:27-169)."""
import ctypes as C

from . import api

SCRG_ERR_IO = 16
SCRG_ERR_FORMAT = 17


class JobOptions(C.Structure):
    _fields_ = [("reverse_strand", C.c_int32), ("sort_by_length", C.c_int32), ("inflation", C.c_int32),
                ("left_extend", C.c_int32), ("read_length_cap", C.c_int64)]


_bound = False


def _lib():
    global _bound
    lib = api.load_library()
    if _bound:
        return lib
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    sig = {
        "scrg_job_options_default": (None, [C.POINTER(JobOptions)]),
        "scrg_job_load": (C.c_int32, [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(JobOptions), C.POINTER(vp),
                                      C.c_char_p, C.c_size_t]),
        "scrg_job_free": (None, [vp]),
        "scrg_job_counts": (None, [vp, u64p, u64p, u64p, u64p]),
        "scrg_job_arrays": (None, [vp, C.POINTER(C.c_char_p), C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(u64p),
                                   C.POINTER(u64p), C.POINTER(u64p), C.POINTER(C.POINTER(C.c_uint8))]),
        "scrg_job_read_name": (C.c_char_p, [vp, C.c_uint64]),
        "scrg_job_pair_chromosome": (C.c_char_p, [vp, C.c_uint64, u64p, u64p]),
        "scrg_align_mapping_stranded": (C.c_int32, [vp, C.POINTER(api.Params), C.c_char_p, C.c_uint64, C.c_uint64,
                                                    C.POINTER(C.c_char_p), u64p, u64p, u64p,
                                                    C.POINTER(C.c_uint8), C.POINTER(C.POINTER(api.Result))]),
        "scrg_job_align": (C.c_int32, [vp, C.POINTER(api.Params), vp, C.POINTER(C.POINTER(api.Result))]),
        "scrg_job_write": (C.c_int32, [vp, C.POINTER(api.Result), C.c_char_p, C.c_int]),
        "scrg_affine_score": (C.c_int32, [C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                          C.POINTER(C.c_int64)]),
        "scrg_validate_alignment": (C.c_int32, [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_char_p,
                                                C.c_int64, C.POINTER(C.c_int32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _bound = True
    return lib


IO_SYMBOLS = ["scrg_job_options_default", "scrg_job_load", "scrg_job_free", "scrg_job_counts", "scrg_job_arrays",
              "scrg_job_read_name", "scrg_job_pair_chromosome", "scrg_align_mapping_stranded", "scrg_job_align",
              "scrg_job_write", "scrg_affine_score", "scrg_validate_alignment"]


def affine_score(cigar, match=2, mismatch=4, gap_open=4, gap_extend=2):
    """Defaults are the costs of the reference's accuracy study (scripts/profile.py, 2,4,4,2)."""
    out = C.c_int64(0)
    st = _lib().scrg_affine_score(cigar.encode(), match, mismatch, gap_open, gap_extend, C.byref(out))
    if st != 0:
        raise api.ScroogeError(st, "malformed CIGAR")
    return int(out.value)


def validate_alignment(text, read, cigar, edit_distance):
    """0 if consistent, else the reason code of scrg_validate_alignment."""
    text = text.encode() if isinstance(text, str) else bytes(text)
    read = read.encode() if isinstance(read, str) else bytes(read)
    why = C.c_int32(0)
    st = _lib().scrg_validate_alignment(text, len(text), read, len(read), cigar.encode(), int(edit_distance),
                                        C.byref(why))
    return 0 if st == 0 else int(why.value)


class Job:
    """A read-mapping job loaded from genome FASTA + reads FASTQ + seeds (.maf/.paf)."""

    def __init__(self, genome_fasta, reads_fastq, seeds, **options):
        self.lib = _lib()
        o = JobOptions()
        self.lib.scrg_job_options_default(C.byref(o))
        for k, v in options.items():
            if not hasattr(o, k):
                raise TypeError("unknown option %r" % k)
            setattr(o, k, int(v))
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        st = self.lib.scrg_job_load(str(genome_fasta).encode(), str(reads_fastq).encode(), str(seeds).encode(),
                                    C.byref(o), C.byref(h), err, len(err))
        if st != 0:
            raise api.ScroogeError(st, err.value.decode())
        self.h = h
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.lib.scrg_job_counts(h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        self.n_reads, self.n_pairs, self.genome_len, self.n_chromosomes = a.value, b.value, c.value, d.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.scrg_job_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def views(self):
        """-> (genome bytes, [read bytes], [[(start_in_reference, reverse)]...], [names])"""
        g = C.c_char_p()
        reads = C.POINTER(C.c_char_p)()
        lens, offs, starts = (C.POINTER(C.c_uint64)() for _ in range(3))
        rev = C.POINTER(C.c_uint8)()
        self.lib.scrg_job_arrays(self.h, C.byref(g), C.byref(reads), C.byref(lens), C.byref(offs), C.byref(starts),
                                 C.byref(rev))
        genome = C.string_at(g, self.genome_len)
        rs, cands, names = [], [], []
        for r in range(self.n_reads):
            rs.append(C.string_at(reads[r], lens[r]))
            cands.append([(int(starts[k]), bool(rev[k])) for k in range(offs[r], offs[r + 1])])
            names.append(self.lib.scrg_job_read_name(self.h, r).decode())
        return genome, rs, cands, names

    def pair_chromosome(self, k):
        s, ln = C.c_uint64(), C.c_uint64()
        nm = self.lib.scrg_job_pair_chromosome(self.h, k, C.byref(s), C.byref(ln))
        return nm.decode(), int(s.value), int(ln.value)

    def align(self, aligner, out_path=None, fmt="paf", **params):
        """Aligns every pair on `aligner`'s GPU; optionally writes PAF/SAM.  -> [Alignment]"""
        res = C.POINTER(api.Result)()
        st = self.lib.scrg_job_align(aligner.h, C.byref(aligner._params(params)), self.h, C.byref(res))
        aligner._check(st, allow=(api.SCRG_ERR_CIGAR_OVERFLOW,))
        try:
            if out_path is not None:
                w = self.lib.scrg_job_write(self.h, res, str(out_path).encode(), 1 if fmt == "sam" else 0)
                if w != 0:
                    raise api.ScroogeError(w, "could not write %s" % out_path)
            r = res.contents
            n = int(r.n_pairs)
            text = C.string_at(r.cigar_text, int(r.cigar_offset[n])) if n else b""
            out = [api.Alignment(text[int(r.cigar_offset[i]):int(r.cigar_offset[i + 1]) - 1].decode(),
                                 int(r.edit_distance[i])) for i in range(n)]
            aligner.last_timing = {"kernel_ns": int(r.kernel_ns), "pack_ns": int(r.pack_ns),
                                   "total_ns": int(r.total_ns)}
        finally:
            self.lib.scrg_result_free(res)
        return out
