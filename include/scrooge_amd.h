/*
 * scrooge_amd.h — C ABI of the MI355X-native GenASM/Scrooge aligner.
 *
 * This is the drop-in boundary for Scrooge's GPU library surface
 * (reference: src/genasm_gpu.hpp:5-10, implemented in src/genasm_gpu.cu:890-1065):
 *
 *   genasm_gpu::align_all(vector<string>& texts, vector<string>& queries, long long* ns)
 *       -> scrg_align_pairs()
 *   genasm_gpu::align_all(Genome_t&, vector<Read_t>&, long long* ns)
 *       -> scrg_align_mapping()
 *   __global__ genasm_gpu::ascii_to_twobit_strings(count, lens, ascii, twobit)
 *       -> scrg_ascii_to_twobit()          (same byte layout, src/genasm_gpu.cu:631-685; the kernel itself, with the
 *                                           reference's signature, is in include/compat/genasm_gpu.hpp for hipcc-compiled
 *                                           callers: both are made of include/scrooge_amd_device.hpp)
 *   genasm_gpu::enabled_algorithm_log
 *       -> scrg_set_log() / scrg_get_log()
 *
 * include/scrooge_amd.hpp rebuilds the reference's C++ signatures on top of
 * these entry points.  Everything here is plain pointers and sizes; the
 * library owns no torch/STL types at the boundary.  Unlike the reference
 * (exit()/assert on every error) every entry point returns a
 * status code.
 *
 * Algorithm contract: results (edit distance and CIGAR) are bit-identical to
 * the reference CPU path src/genasm_cpu.cpp:178-438 at the same W and O with
 * K = W; CIGAR runs are flushed per window and never merged across windows
 * (src/genasm_cpu.cpp:304-305, 400-403).
 */
#ifndef SCROOGE_AMD_H
#define SCROOGE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t scrg_status;
enum {
    SCRG_OK = 0,
    SCRG_ERR_INVALID_ARG = 1,   /* null pointer, W/O out of range, ...              */
    SCRG_ERR_BAD_BASE = 2,      /* non-ACGT/acgt input (reference: assert, genasm_cpu.cpp:487-489) */
    SCRG_ERR_NO_DEVICE = 3,     /* no usable gfx950 device / HIP runtime failure at init */
    SCRG_ERR_HIP = 4,           /* a HIP call failed; see scrg_last_error()         */
    SCRG_ERR_OOM = 5,           /* host or device allocation failed                 */
    SCRG_ERR_CIGAR_OVERFLOW = 6 /* a pair produced more runs than its arena slice   */
};

/* Tunables.  Zero-initialise and call scrg_params_default(). */
typedef struct scrg_params {
    int32_t W;               /* window length, 2..256; reference default 64 (genasm_cpu.cpp:7).
                                W > 64 uses multi-word vectors (src/bitvector.hpp:45-48)           */
    int32_t O;               /* window overlap, 1 <= O < W; reference default 33 (genasm_cpu.cpp:9).
                                W <= 64 with W-O <= 31 (e.g. the defaults) keeps the traceback table in 62 registers;
                                32 <= W-O <= 63 with W <= 128 in 128 registers, built in two halves of 32 columns;
                                beyond that the table rows are 128 to 256 bits wide and kept in HBM                */
    int32_t lanes_per_pair;  /* 1 = one pair per lane, 64 pairs per wavefront (the default for every W);
                                64 = one pair per wavefront (lane = text column); 4/8/16/32 pack 64/lanes
                                pairs into one wavefront (GenASM rows; for W > 64 only 32 and 64 exist).
                                0 = default                                                          */
    int32_t lds_rows;        /* rows of the R table kept in LDS per pair (rest spills to HBM);
                                0 = default                                                        */
    int32_t waves_per_cu;    /* persistent wavefronts launched per CU; 0 = default (as many as fit, at most one per 64
                                pairs).  A batch of MIXED read lengths issued longest first balances better with
                                fewer wavefronts than groups of 64 pairs — the work queue then hands the short pairs
                                at the end of the order to the lanes that finish first: 4 (one per SIMD) is +12 % on
                                100 k pairs of 2 .. 20 kb (bench.py other_configs); equal lengths: leave 0           */
    int32_t sort_by_length;  /* host entry points: issue pairs longest-read-first (the reference's
                                callers do this themselves, src/tests.cu:375-377); results keep
                                input order either way.  Default 1                                 */
    int32_t text_stride_words; /* device-pointer entry points: distance, in 64-bit words, between consecutive
                                words of one text in d_seq.  0 or 1 = contiguous; 64 = the lane-interleaved
                                layout of scrg_pack_planar_groups() (see there).  Only lanes_per_pair = 1
                                accepts a stride other than 1                                        */
    int32_t read_stride_words; /* the same for the reads                                             */
    int32_t outputs;         /* host entry points: what the result holds — SCRG_OUT_ALL (0, default): runs and text;
                                SCRG_OUT_TEXT: the CIGAR text only (runs stays empty, run_offset all zero);
                                SCRG_OUT_RUNS: the runs only (cigar_text stays empty, cigar_offset all zero).
                                What is not asked for does not cross PCIe                                   */
    int32_t reserved[2];     /* must be 0: the shipped library rejects every bit of both with SCRG_ERR_INVALID_ARG, so an
                                uninitialised struct cannot silently change anything.  (Other BUILDS of the same sources
                                give them a meaning — the test build, -DSCRG_SELECT, selects between formulations that give
                                identical results; profiling builds add counters and ablations: see scrg_debug_stats.)   */
    int32_t stranded;        /* device-pointer entry points (ABI 7): 1 = bit 63 of scrg_pair_desc.read_off (SCRG_READ_REVCOMP) marks
                                a pair whose read is aligned as its REVERSE COMPLEMENT, taken from the one packed copy of the read
                                (a read-mapping candidate on the minus strand; the reference drops those, src/tests.cu:346-355).
                                Served by the one-pair-per-lane kernels (lanes_per_pair = 1, the default, at every W and O);
                                SCRG_ERR_INVALID_ARG for the GenASM-row mappings.  0 (default): bit 63 must be clear      */
} scrg_params;
#define SCRG_READ_REVCOMP (1ull << 63)

enum { SCRG_OUT_ALL = 0, SCRG_OUT_TEXT = 1, SCRG_OUT_RUNS = 2 };

void scrg_params_default(scrg_params *p);
/* Fills in every 0 ("default") field for the given W/O and validates; the values a launch will use. */
scrg_status scrg_params_resolve(const scrg_params *in, scrg_params *out);

/* One CIGAR run, layout-compatible with the reference's CigarEntry_t
 * (src/util.hpp:43-46). */
typedef struct scrg_run {
    uint8_t count;
    char    op;     /* one of '=', 'X', 'I', 'D' */
} scrg_run;

/* Per-device handle: owns a HIP stream, the work counter and scratch.  One
 * handle may be used by one thread at a time; separate handles are independent
 * (the reference's GPU path is not re-entrant at all, genasm_gpu.cu:582). */
typedef struct scrg_ctx scrg_ctx;

scrg_status scrg_ctx_create(int device, scrg_ctx **out);
void        scrg_ctx_destroy(scrg_ctx *ctx);
/* Enqueue on a caller-owned hipStream_t (e.g. torch's current stream) instead of
 * the handle's own.  NULL is a valid value: the device's default (null) stream.
 * scrg_ctx_use_own_stream() goes back to the handle's private stream. */
scrg_status scrg_ctx_set_stream(scrg_ctx *ctx, void *hip_stream);
scrg_status scrg_ctx_use_own_stream(scrg_ctx *ctx);

/* A HIP stream (hipStream_t) of the given priority: -1 high, 0 normal, 1 low.  Streams of different
 * priorities never share a hardware queue, which is what lets launches of two handles overlap
 * (INTEGRATION.md §4b); streams of one priority may be multiplexed onto one queue. */
scrg_status scrg_stream_create(int device, int priority, void **stream);
scrg_status scrg_stream_destroy(void *stream);
const char *scrg_last_error(const scrg_ctx *ctx);
const char *scrg_status_string(scrg_status s);
/* mirrors genasm_gpu::enabled_algorithm_log (src/genasm_gpu.hpp:6) */
void        scrg_set_log(int enabled);
int         scrg_get_log(void);
int         scrg_device_count(void);
/* How this library was built: 0 for the shipped build; SCRG_BUILD_SELECT for the test build, whose scrg_params.reserved[0]
 * selects between formulations that give identical results (-DSCRG_SELECT); SCRG_BUILD_STATS if the kernels also carry their
 * profiling counters and scheduling switches (-DSCRG_STATS), SCRG_BUILD_ABLATE if also the ablation switches
 * (-DSCRG_ABLATE: results wrong by design).  See scrg_debug_stats. */
enum { SCRG_BUILD_STATS = 1, SCRG_BUILD_ABLATE = 2, SCRG_BUILD_SELECT = 4 };
int         scrg_build_flags(void);
/* The version of THIS interface.  It goes up whenever an entry point changes its arguments under the same name (version 5:
 * scrg_decode_edit_stream takes the capacity of its output array; version 6: scrg_encode_edit_stream and
 * scrg_runs_to_edit_stream take the parameters, and the edit stream itself carries the window ends) — such a change still links against code compiled with the
 * older header and would shift every later argument.  scrg_abi_version() is what the loaded library was built with; a binding
 * compares it with the SCRG_ABI_VERSION it was compiled against before anything else (include/scrooge_amd.hpp throws,
 * scrooge_amd/api.py raises). */
#define SCRG_ABI_VERSION 7
int         scrg_abi_version(void);

/* ---------------------------------------------------------------------------
 * Host-pointer entry points (the drop-in path).
 * ------------------------------------------------------------------------- */

/* Library-owned result of one batch; release with scrg_result_free(). */
typedef struct scrg_result {
    uint64_t  n_pairs;
    int64_t  *edit_distance;  /* [n_pairs]                                              */
    uint32_t *pair_status;    /* [n_pairs] SCRG_OK or SCRG_ERR_CIGAR_OVERFLOW            */
    uint64_t *run_offset;     /* [n_pairs+1] into runs                                   */
    scrg_run *runs;           /* all runs, pair after pair                               */
    uint64_t *cigar_offset;   /* [n_pairs+1] into cigar_text (each CIGAR NUL-terminated) */
    char     *cigar_text;     /* "%d%c" rendering, as genasm_cpu.cpp:387-403             */
    int64_t   kernel_ns;      /* align kernel only (reference: core_algorithm_ns): the sum over
                                 the launches of the call's chunks, which overlap         */
    int64_t   pack_ns;        /* host threads: ASCII -> 2 bits per base into pinned memory */
    int64_t   total_ns;       /* whole call                                              */
} scrg_result;

/* (The large arrays are recycled by the library: up to 2 GB of them are kept for the next call of similar size
 * instead of being returned to the allocator, so that a stream of batches does not fault fresh pages every time.
 * scrg_result_pool_trim() returns the parked arrays to the allocator; destroying the last handle does the same.
 * Recycled arrays are NOT zeroed: every element a result publishes is written by the call that publishes it.) */
void scrg_result_free(scrg_result *r);
void scrg_result_pool_trim(void);

/* Unstructured pairwise alignment: queries[i] is consumed completely against a
 * prefix of texts[i] (reference: genasm_gpu.cu:982-1065; returns all n results,
 * unlike the CPU overload's double increment at genasm_cpu.cpp:600-605). */
scrg_status scrg_align_pairs(scrg_ctx *ctx, const scrg_params *params, uint64_t n_pairs,
                             const char *const *texts, const uint64_t *text_lens,
                             const char *const *queries, const uint64_t *query_lens,
                             scrg_result **out);

/* Read-mapping alignment: read r is aligned against the genome suffix starting
 * at cand_start[c] for every c in [cand_offsets[r], cand_offsets[r+1])
 * (reference: genasm_gpu.cu:890-980 / genasm_cpu.cpp:495-555, :512-514 for the
 * suffix semantics).  Genome and reads are packed and stored once.  Results are
 * in read-major, candidate-minor order. */
scrg_status scrg_align_mapping(scrg_ctx *ctx, const scrg_params *params,
                               const char *genome, uint64_t genome_len,
                               uint64_t n_reads, const char *const *reads, const uint64_t *read_lens,
                               const uint64_t *cand_offsets, const uint64_t *cand_start,
                               scrg_result **out);

/* Many read batches against one reference: scrg_genome_set() packs the genome ONCE (host threads), transfers it and
 * keeps it in the handle's HBM; scrg_align_mapping_resident() then aligns batches against it without touching
 * the genome again (scrg_align_mapping stages its genome on every call, as the reference re-converts it,
 * genasm_cpu.cpp:508 — and leaves it resident afterwards).  cand_reverse may be NULL (all forward) or hold one 0/1 per
 * candidate (1 = align the reverse complement of the read, as scrg_align_mapping_stranded in scrooge_amd_io.h).  The
 * genome stays resident until another one is set or scrg_genome_clear() is called; pairwise calls on the same handle do
 * not disturb it. */
scrg_status scrg_genome_set(scrg_ctx *ctx, const char *genome, uint64_t genome_len);
void        scrg_genome_clear(scrg_ctx *ctx);
scrg_status scrg_align_mapping_resident(scrg_ctx *ctx, const scrg_params *params,
                                        uint64_t n_reads, const char *const *reads, const uint64_t *read_lens,
                                        const uint64_t *cand_offsets, const uint64_t *cand_start,
                                        const uint8_t *cand_reverse, scrg_result **out);

/* Several GPUs, one call, one process (the reference is single-GPU: GPU_ID 0, src/genasm_gpu.cu:67).  The batch is cut
 * into chunks in issue order (longest read first) and chunk k goes to devices[k mod n_devices]; every device has its own
 * host thread, streams and buffers, brings its chunks' results back itself (no inter-GPU traffic) and the result is
 * assembled in caller order exactly as by the single-device calls.  A device may be listed more than once (more chunks
 * in flight on it).  No handle: per-device state is created on first use and kept for later calls —
 * scrg_multi_release() frees it.  The mapping call stages the genome on every listed device.  Errors: status code and
 * scrg_multi_last_error() (per thread). */
scrg_status scrg_align_pairs_multi(const int32_t *devices, int32_t n_devices, const scrg_params *params, uint64_t n_pairs,
                                   const char *const *texts, const uint64_t *text_lens,
                                   const char *const *queries, const uint64_t *query_lens,
                                   scrg_result **out);
scrg_status scrg_align_mapping_multi(const int32_t *devices, int32_t n_devices, const scrg_params *params,
                                     const char *genome, uint64_t genome_len,
                                     uint64_t n_reads, const char *const *reads, const uint64_t *read_lens,
                                     const uint64_t *cand_offsets, const uint64_t *cand_start,
                                     const uint8_t *cand_reverse, scrg_result **out);
/* How a host call with these sequence lengths is cut up (no GPU involved): issue_order[k] = caller index of the k-th
 * pair issued (longest read first, stable — src/tests.cu:375-377; NULL to skip), chunk_first[0 .. *n_chunks] = issue index of
 * every chunk's first pair (chunk_cap entries available; NULL to skip).  Chunk k is processed by devices[k mod n_devices];
 * chunks are whole groups of 64 pairs except the last.  text_lens may be NULL. */
scrg_status scrg_host_plan(const scrg_params *params, int32_t n_devices, uint64_t n_pairs, const uint64_t *text_lens,
                           const uint64_t *read_lens, uint32_t *issue_order, uint64_t *chunk_first, uint64_t chunk_cap,
                           uint64_t *n_chunks);
void        scrg_multi_release(void);
const char *scrg_multi_last_error(void);

/* ---------------------------------------------------------------------------
 * Device-pointer entry points (inputs/outputs already resident in HBM; this is
 * what bench.py times).  All pointers below are device pointers valid on the
 * handle's device; work is enqueued on the handle's stream and NOT synchronised.
 * ------------------------------------------------------------------------- */

/* Sequence storage used by the align kernel: "planar 2-bit" — one uint64_t per
 * 32 bases; bit k of the low word is bit 0 of base k's code, bit k of the high
 * word is bit 1 (A=0 C=1 G=2 T=3, genasm_cpu.cpp:87-90).  Sequence arrays need
 * SCRG_SEQ_PAD_WORDS readable words past the last base. */
#define SCRG_SEQ_PAD_WORDS 4

/* ASCII -> planar.  d_ascii holds n_words*32 bytes; bytes equal to 0 are
 * padding and encode as A.  *d_bad_count is incremented for every other
 * non-ACGTacgt byte. */
scrg_status scrg_pack_planar(scrg_ctx *ctx, const char *d_ascii, uint64_t n_words,
                             uint64_t *d_planar, uint32_t *d_bad_count);

/* ASCII -> planar in the LANE-INTERLEAVED layout the one-pair-per-lane kernel reads best.  d_ascii holds
 * n_rows rows of words_per_row*32 bytes each (zero padded, every sequence starting at a multiple of 32
 * bytes within its row).  Rows are taken in groups of 64 (the pairs one wavefront aligns side by side):
 * word w of row r is stored at d_planar[((r / 64) * words_per_row + w) * 64 + r % 64], so the 64 lanes of a
 * wavefront, which walk their sequences at the same pace, read 512 contiguous bytes per load instead of 64
 * separate cache lines.  A sequence that starts at byte 32*k of row r has
 *     offset_in_bases = 32 * (((r / 64) * words_per_row + k) * 64 + r % 64)
 * and a word stride of 64 (scrg_params.text_stride_words / read_stride_words).  d_planar needs
 * ceil(n_rows / 64) * 64 * words_per_row words plus SCRG_SEQ_PAD_WORDS_STRIDED(64) readable words of padding. */
scrg_status scrg_pack_planar_groups(scrg_ctx *ctx, const char *d_ascii, uint64_t n_rows, uint64_t words_per_row,
                                    uint64_t *d_planar, uint32_t *d_bad_count);
/* The same packing of ONE sequence on the host, with the packer the host entry points use on their threads (AVX2 or scalar
 * code, chosen at run time; no GPU, no handle): word w of the sequence goes to planar[w * stride_words] (stride 0 = 1), the
 * words past the sequence up to n_words are zeroed.  SCRG_ERR_BAD_BASE if a byte is not one of ACGTacgt (the reference asserts,
 * src/genasm_cpu.cpp:487-489).  A caller that stages its own device arrays can use it; the CPU tests hold it to the layout above. */
scrg_status scrg_pack_planar_host(const char *ascii, uint64_t n_bases, uint64_t *planar, uint64_t stride_words, uint64_t n_words);
#define SCRG_SEQ_PAD_WORDS_STRIDED(stride) (2 * (stride) + 2)

/* One alignment problem.  Offsets are in bases from the start of d_seq: base k of a sequence with offset
 * `off` and word stride s lives in word (off / 32) + ((off % 32 + k) / 32) * s, bit (off + k) % 32 of each
 * plane (a strided sequence therefore starts at a multiple of 32). */
typedef struct scrg_pair_desc {
    uint64_t text_off;
    uint64_t text_len;
    uint64_t read_off;    /* (| SCRG_READ_REVCOMP with scrg_params.stranded: the read's reverse complement is aligned) */
    uint64_t read_len;
    uint64_t cigar_off;   /* first run of this pair's slice, in scrg_run units; multiple of 16 */
    uint64_t cigar_cap;   /* slice capacity in runs; multiple of 16 (runs leave the GPU in
                             aligned 32-byte pieces), d_runs itself 32-byte aligned */
} scrg_pair_desc;

/* Aligns n_pairs problems.  Outputs: d_edit_distance[n], d_n_runs[n],
 * d_pair_status[n], runs in d_runs at each pair's slice. */
scrg_status scrg_align_device(scrg_ctx *ctx, const scrg_params *params, uint64_t n_pairs,
                              const uint64_t *d_seq, const scrg_pair_desc *d_pairs,
                              scrg_run *d_runs, int64_t *d_edit_distance,
                              uint32_t *d_n_runs, uint32_t *d_pair_status);

/* The same alignment delivered as EDIT STREAMS (format below, "Edit stream") straight from the align kernel: pair p's
 * stream goes to its slice — bytes [2 * cigar_off, 2 * cigar_off + 2 * cigar_cap) of d_streams, i.e. the very slice
 * scrg_align_device would fill with runs, so descriptors and buffers can be shared —, d_stream_len[p] is its length
 * in bytes (the bytes up to the next multiple of 4 are zero), d_pair_status[p] is 1 if it did not fit (a stream is
 * never longer than edit distance + number of windows + read_len / 63 bytes; a slice sized like the reference's,
 * cigar_cap = 2 x read length runs, always fits).  The kernel does less
 * work than for runs (it visits edits, not run boundaries) and writes a third of the bytes.
 * scrg_compact_runs with d_n_runs[p] = (d_stream_len[p] + 3) / 4 * 2 and even d_dense_offset gathers the slices.
 * d_n_runs may be NULL; otherwise d_n_runs[p] = the number of runs of the same alignment (what scrg_align_device
 * reports): sent along with the streams it lets the receiver lay out the decoded runs by a prefix sum and restore them
 * with ONE scrg_decode_edit_stream pass (no counting pass).
 * One-pair-per-lane kernels only: lanes_per_pair = 1 (the default for every W and O); SCRG_ERR_INVALID_ARG for the
 * GenASM-row mappings (use scrg_align_device + scrg_encode_edit_stream there).  d_streams 32-byte aligned. */
scrg_status scrg_align_device_edits(scrg_ctx *ctx, const scrg_params *params, uint64_t n_pairs,
                                    const uint64_t *d_seq, const scrg_pair_desc *d_pairs,
                                    uint8_t *d_streams, int64_t *d_edit_distance,
                                    uint32_t *d_stream_len, uint32_t *d_pair_status, uint32_t *d_n_runs);

/* Gathers every pair's runs from its slice into one dense array:
 * d_dense[d_dense_offset[p] + k] = d_runs[d_pairs[p].cigar_off + k]. */
scrg_status scrg_compact_runs(scrg_ctx *ctx, uint64_t n_pairs, const scrg_pair_desc *d_pairs,
                              const scrg_run *d_runs, const uint32_t *d_n_runs,
                              const uint64_t *d_dense_offset, scrg_run *d_dense);

/* The same gather into ONE BYTE per run — op in bits 7..6 (0 '=', 1 'X', 2 'I', 3 'D'), count in bits 5..0 — for
 * transfers (the RCCL gather of CIGARs to one GPU moves half the bytes).  A run never spans windows, so its count
 * is at most W-O; valid for W-O <= 63, i.e. every W <= 64 (SCRG_ERR_INVALID_ARG otherwise).  d_dense_offset is in
 * runs (= bytes).  scrg_unpack_runs restores scrg_run pairs bit for bit; both buffers 4-byte aligned. */
scrg_status scrg_compact_runs_packed(scrg_ctx *ctx, const scrg_params *params, uint64_t n_pairs,
                                     const scrg_pair_desc *d_pairs, const scrg_run *d_runs, const uint32_t *d_n_runs,
                                     const uint64_t *d_dense_offset, uint8_t *d_packed);
scrg_status scrg_unpack_runs(scrg_ctx *ctx, uint64_t n_runs, const uint8_t *d_packed, scrg_run *d_runs);

/* ---- Edit stream: the compact transfer format for CIGARs (multi-GPU gather, D2H) ----
 * The runs of a pair carry the alignment operations AND the places where a window ended (runs are flushed per window,
 * genasm_cpu.cpp:304-305, 400-403).  One byte per EDIT and one per WINDOW carries both (format version 2, ABI 6):
 *     byte = op << 6 | len     op 1 'X', 2 'I', 3 'D': `len` matches, then that edit;
 *                              op 0, len <= 62        : `len` matches, then the window ends (every window of a pair has
 *                                                       its END byte, the last one too);
 *                              op 0, len == 63 (0x3F) : 63 matches and nothing else (only W-O > 63 has such stretches)
 * in alignment order.  Canonical form: P matches before an edit or a window end = P / 63 bytes 0x3F, then the byte with
 * len = P % 63.  A 10 kb read at 10 % error and W-O = 31 is ~1.3 KB (2140 runs = 4.3 KB as scrg_run, 2.1 KB packed).
 * Valid for every W (run counts are <= W-O <= 255).  (Version 1 — ABI <= 5 — sent the edits only and the receiver replayed
 * the window loop: a quarter fewer bytes, three times the decoding work.)
 *
 * scrg_encode_edit_stream: for every pair, the stream of its runs (as scrg_align_device left them in d_runs; W-O of
 *   `params` places the window ends).
 *   Streams are placed in d_stream back to back in no particular order, each starting at a multiple of 4:
 *   d_stream_off[p] (bytes; ~0 if the pair did not fit into stream_cap) and d_stream_len[p] say where.
 *   d_total[0] = bytes of d_stream used (the amount to transfer), d_total[1] = pairs that did not fit.
 *   stream_cap >= sum over the pairs of (edit distance + 2 * (read_len + edit distance) / (W-O) + read_len / 63 + 8) always
 *   suffices (the read_len / 63 term is the 0x3F bytes, one per 63 matches in a row: only W-O > 63 has any).
 * scrg_decode_edit_stream: the inverse (streams of 64 bytes and more on average: one pair per wavefront, four stream bytes per
 *   lane side by side, runs staged in LDS and stored in aligned 16-byte units; shorter ones: one pair per lane, streams read in
 *   aligned 16-byte blocks, runs written in aligned 64-byte pieces — the same runs either way; the launch follows
 *   stream_bytes / n_pairs, the kernel then looks at a sample of d_stream_len itself, so a buffer sized for the worst case
 *   costs nothing).  d_stream holds stream_bytes bytes (16-byte aligned, readable up to the next
 *   multiple of 16): a pair whose stream is not inside [0, stream_bytes) — offsets and lengths may come off a wire —
 *   is counted as bad, never read.  Read lengths are taken from d_read_len[p * read_len_stride] (stride 1: a plain
 *   array; 6: &d_pairs[0].read_len; 0: one length for all).  `params` is not looked at beyond its validity (the window
 *   ends are in the stream).
 *   With d_dense == NULL it only counts: d_n_runs[p] = runs of pair p.  Otherwise d_n_runs[p] is an input, the size
 *   of pair p's segment at d_dense + d_dense_offset[p] (d_dense 16-byte aligned, room for dense_capacity runs), and the
 *   runs are written there — bit for bit the runs the align kernel produced.  Counts and offsets
 *   may come off a wire like the streams: a pair whose segment [d_dense_offset[p], + d_n_runs[p]) does not lie inside
 *   [0, dense_capacity) is counted as bad and nothing of it is written.  (scrg_align_device_edits can deliver the
 *   run counts along with the streams, so that the receiver sizes the dense array by a prefix sum and decodes in ONE
 *   pass.)  *d_bad_count is incremented for every pair whose stream is not an alignment of a read of that length
 *   (it does not end with a window end, places another number of read characters, holds a run longer than 255),
 *   or whose run count differs from d_n_runs[p]. */
scrg_status scrg_encode_edit_stream(scrg_ctx *ctx, const scrg_params *params, uint64_t n_pairs,
                                    const scrg_pair_desc *d_pairs,
                                    const scrg_run *d_runs, const uint32_t *d_n_runs,
                                    uint8_t *d_stream, uint64_t stream_cap,
                                    uint64_t *d_stream_off, uint32_t *d_stream_len, uint64_t *d_total);
scrg_status scrg_decode_edit_stream(scrg_ctx *ctx, const scrg_params *params, uint64_t n_pairs,
                                    const uint8_t *d_stream, uint64_t stream_bytes,
                                    const uint64_t *d_stream_off, const uint32_t *d_stream_len,
                                    const uint64_t *d_read_len, uint64_t read_len_stride,
                                    const uint64_t *d_dense_offset, scrg_run *d_dense, uint64_t dense_capacity,
                                    uint32_t *d_n_runs, uint32_t *d_bad_count);
/* The same two conversions for ONE pair on the host (no GPU, no handle): what a receiver without a GPU, or a
 * test, uses.  scrg_edit_stream_to_runs returns SCRG_ERR_INVALID_ARG for a malformed stream — it also checks that
 * every window ends where the reference's window loop of W/O ends it (genasm_cpu.cpp:307-310) — and
 * SCRG_ERR_CIGAR_OVERFLOW if runs_cap is too small (*n_runs is the number needed either way; runs may be NULL
 * with runs_cap 0 to ask for it).  scrg_runs_to_edit_stream likewise with stream_cap / *n_bytes; it replays the
 * window loop of W/O to place the window ends (SCRG_ERR_INVALID_ARG for an operation other than = X I D). */
scrg_status scrg_edit_stream_to_runs(const scrg_params *params, uint64_t read_len,
                                     const uint8_t *stream, uint64_t n_bytes,
                                     scrg_run *runs, uint64_t runs_cap, uint64_t *n_runs);
scrg_status scrg_runs_to_edit_stream(const scrg_params *params, const scrg_run *runs, uint64_t n_runs,
                                     uint8_t *stream, uint64_t stream_cap, uint64_t *n_bytes);
/* scrg_edit_stream_to_runs through the state machine the device decoder runs in every lane (same code, compiled for the
 * host): same arguments, same runs; like the device it does not look at the window geometry.  Lets a host without a
 * GPU — and the CPU tests — check that form too. */
scrg_status scrg_edit_stream_to_runs_lane(const scrg_params *params, uint64_t read_len,
                                          const uint8_t *stream, uint64_t n_bytes,
                                          scrg_run *runs, uint64_t runs_cap, uint64_t *n_runs);

/* Reference-layout 2-bit packer, mirrors the exported kernel
 * genasm_gpu::ascii_to_twobit_strings (src/genasm_gpu.cu:631-685): 4 bases per
 * byte, first base in bits 7..6, last byte zero-padded.  Each string occupies
 * ceil(len/4) bytes at d_twobit + d_twobit_off[s]. */
scrg_status scrg_ascii_to_twobit(scrg_ctx *ctx, uint64_t count, const uint64_t *d_lens,
                                 const uint64_t *d_ascii_off, const char *d_ascii,
                                 const uint64_t *d_twobit_off, uint8_t *d_twobit,
                                 uint32_t *d_bad_count);

/* Launch geometry for params on this device (for the bench's roofline line): persistent wavefronts, pairs per
 * wavefront, LDS bytes per wavefront.  It describes the kernel a launch that FILLS the GPU takes.  One exception is
 * decided per launch, by its size: with the default table (W <= 64, W-O <= 31, runs output) and waves_per_cu left at 0, a
 * launch of at most one wavefront per SIMD (n_pairs <= 64 x 4 x CUs) runs as workgroups of 512 threads — four producer and
 * four consumer wavefronts, 58 432 bytes of LDS per workgroup, two workgroups per CU.  A caller that sizes co-resident
 * work from these numbers sets waves_per_cu explicitly: the launch then has exactly the geometry reported here. */
scrg_status scrg_query_launch(scrg_ctx *ctx, const scrg_params *params,
                              int32_t *n_waves, int32_t *pairs_per_wave, int32_t *lds_bytes,
                              int32_t *n_cus);

/* Times the most recent scrg_align_device launch with HIP events recorded on
 * the launch stream (milliseconds); blocks until that launch finished. */
scrg_status scrg_last_kernel_ms(scrg_ctx *ctx, float *ms);

/* scrg_params.reserved[] in the SHIPPED library: both must be 0; scrg_params_resolve() and every entry point REJECT anything
 * else, and none of the code below exists in the shipped kernels or in the shipped kernel selection.
 *
 * Test build only (-DSCRG_SELECT: ab_libs/lib_select.so, scrooge_amd.build_library("select"); scrg_build_flags() tells):
 * reserved[0] selects between formulations that give identical results, for the parity tests that compare them —
 * 32 (lanes_per_pair = 8: GenASM rows only, no diagonal-major path), 256 (32 <= W-O <= 127: the kernel that keeps the
 * window table in HBM instead of the one that keeps it in registers), 512 / 1024 (W <= 64, W-O <= 31, runs output: always /
 * never the variant of the default kernel that splits a window's work over two wavefronts; by default a launch that cannot
 * put two wavefronts on every SIMD takes it).
 *
 * Profiling builds only (scripts/ab.sh; they include the selections above): with -DSCRG_STATS the align kernels accumulate
 * twelve counters per launch when reserved[1] != 0, read back by scrg_debug_stats (blocks on the stream):
 *   lanes_per_pair = 1 (genasm_lane_kernel): [0] window rounds (one window of each of a wavefront's 64 pairs),
 *     [1] rounds that took a short-window variant, [2..6] shader cycles summed over wavefronts: traceback pass 1,
 *     queue/fetch, window setup, table, traceback (both passes + CIGAR flush), [7] wavefront life times and
 *     [8..11] latest start, 2^62 - earliest start, latest end, 2^62 - earliest end on the 100 MHz wall clock.
 *   lanes_per_pair >= 4 (genasm_align_kernel): {window rounds, DC sweep steps, TB macro-steps, shader cycles for
 *     fetch / window setup / DC / TB / TB loop, rounds on the diagonal-major path, rounds that fell back from it,
 *     DC / TB cycles of the diagonal-major rounds (not included in the former)};
 * and reserved[0] also accepts the scheduling switches 1 (one pair per lane: no wavefront priority rotation) and
 * 64 / 128 (workgroups of one / two wavefronts instead of four).  With -DSCRG_ABLATE also 2, 4, 8, 16 (skip the table,
 * a traceback pass, the stores: results wrong by design).  None of this code exists in the shipped kernels; there
 * scrg_debug_stats returns SCRG_ERR_INVALID_ARG. */
scrg_status scrg_debug_stats(scrg_ctx *ctx, uint64_t out[12]);

#ifdef __cplusplus
}
#endif
#endif /* SCROOGE_AMD_H */
