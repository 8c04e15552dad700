/*
 * genasm_oracle.c — TEST INFRASTRUCTURE ONLY (see genasm_oracle.h).
 *
 * Plain-C restatement of the windowed GenASM aligner in the reference's
 * src/genasm_cpu.cpp.  It is written from the algorithm's definition, one
 * unsigned integer per bitvector (uint64_t for W <= 64, src/bitvector.hpp:42-44; unsigned __int128 up to
 * W = 128, four 64-bit words up to W = 256; the core lives in genasm_oracle_core.inc), and keeps the
 * full (W+1) x (W+1) table of centre entries ("SENE" storage,
 * genasm_cpu.cpp:63-78); the reference's three storage/termination toggles do
 * not change results, so one variant is enough for a checker.
 *
 * Bit conventions (genasm_cpu.cpp:178-198, 210-288):
 *   bit b of a pattern mask for base c is 0 iff pattern[m-1-b] == c;
 *   bit b of R[d][i] is 0 iff the last b+1 pattern characters match a text
 *   substring starting at i with at most d edits.
 */
#include "genasm_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GO_MAXW 256

typedef struct run_sink {
    go_run *runs;
    size_t cap;
    size_t n;
    int overflow;
} run_sink;

static void sink_push(run_sink *o, char op, unsigned count)
{
    if (o->n < o->cap) {
        o->runs[o->n].count = (uint8_t)count;
        o->runs[o->n].op = op;
    } else {
        o->overflow = 1;
    }
    o->n++;
}


#define GO_ONES() (~(GO_BV)0)
#define GO_ZERO() ((GO_BV)0)
#define GO_SHL(v, s) ((v) << (s))
#define GO_OR(a, b) ((a) | (b))
#define GO_AND(a, b) ((a) & (b))
#define GO_CLRBIT(v, b) ((v) & ~((GO_BV)1 << (b)))
#define GO_BIT0(v, b) ((((v) >> (b)) & 1u) == 0)

#define GO_BV uint64_t
#define GO_BV_BITS 64u
#define GO_BV_MAXW 64
#define GO_NAME(x) x##_64
#include "genasm_oracle_core.inc"
#undef GO_BV
#undef GO_BV_BITS
#undef GO_BV_MAXW
#undef GO_NAME

#define GO_BV unsigned __int128
#define GO_BV_BITS 128u
#define GO_BV_MAXW 128
#define GO_NAME(x) x##_128
#include "genasm_oracle_core.inc"
#undef GO_BV
#undef GO_BV_BITS
#undef GO_BV_MAXW
#undef GO_NAME
#undef GO_ONES
#undef GO_ZERO
#undef GO_SHL
#undef GO_OR
#undef GO_AND
#undef GO_CLRBIT
#undef GO_BIT0

/* 256 bits as four 64-bit words, w[0] least significant (the reference: N/32 elements of 32 bits,
 * src/bitvector.hpp:45-48, shifts carried across elements, :124-139). */
typedef struct go_bv256 {
    uint64_t w[4];
} go_bv256;

static inline go_bv256 bv256_fill(uint64_t x)
{
    go_bv256 r = { { x, x, x, x } };
    return r;
}
static inline go_bv256 bv256_shl(go_bv256 v, unsigned s) /* s < 256 */
{
    go_bv256 r;
    const unsigned ws = s / 64u, bs = s % 64u;
    for (int k = 3; k >= 0; k--) {
        uint64_t x = 0;
        if (k >= (int)ws) {
            x = v.w[k - (int)ws] << bs;
            if (bs && k - (int)ws - 1 >= 0)
                x |= v.w[k - (int)ws - 1] >> (64u - bs);
        }
        r.w[k] = x;
    }
    return r;
}
static inline go_bv256 bv256_or(go_bv256 a, go_bv256 b)
{
    for (int k = 0; k < 4; k++) a.w[k] |= b.w[k];
    return a;
}
static inline go_bv256 bv256_and(go_bv256 a, go_bv256 b)
{
    for (int k = 0; k < 4; k++) a.w[k] &= b.w[k];
    return a;
}
static inline go_bv256 bv256_clrbit(go_bv256 v, unsigned b)
{
    v.w[b / 64u] &= ~(1ull << (b % 64u));
    return v;
}
#define GO_ONES() bv256_fill(~0ull)
#define GO_ZERO() bv256_fill(0ull)
#define GO_SHL(v, s) bv256_shl((v), (s))
#define GO_OR(a, b) bv256_or((a), (b))
#define GO_AND(a, b) bv256_and((a), (b))
#define GO_CLRBIT(v, b) bv256_clrbit((v), (b))
#define GO_BIT0(v, b) ((((v).w[(b) / 64u] >> ((b) % 64u)) & 1u) == 0)
#define GO_BV go_bv256
#define GO_BV_BITS 256u
#define GO_BV_MAXW 256
#define GO_NAME(x) x##_256
#include "genasm_oracle_core.inc"
#undef GO_BV
#undef GO_BV_BITS
#undef GO_BV_MAXW
#undef GO_NAME

/* Test hook over the 4-word vector above (tests/test_row_ops.py pins it to the known answers of the reference's
 * bitvector tests, src/bitvector_test.cu:22-132).  op: 0 shl(a, s)  1 or  2 and  3 clrbit(a, s)  4 pattern-mask style
 * insertion of the 32 bits `b[0]` at bit s (src/bitvector.hpp insert_bits: or into place); returns bit_is_zero(a, s). */
int go_bv256_op(int op, const uint64_t a[4], const uint64_t b[4], unsigned s, uint64_t out[4])
{
    go_bv256 x, y, r;
    memcpy(x.w, a, sizeof x.w);
    memcpy(y.w, b, sizeof y.w);
    r = x;
    switch (op) {
    case 0: r = s >= 256u ? bv256_fill(0) : bv256_shl(x, s); break;
    case 1: r = bv256_or(x, y); break;
    case 2: r = bv256_and(x, y); break;
    case 3: r = bv256_clrbit(x, s); break;
    case 4: {
        go_bv256 piece = { { y.w[0] & 0xffffffffull, 0, 0, 0 } };
        r = bv256_or(x, bv256_shl(piece, s));
        break;
    }
    default: return -1;
    }
    memcpy(out, r.w, sizeof r.w);
    return (int)((x.w[(s % 256u) / 64u] >> (s % 64u)) & 1u) == 0;
}

int go_align_codes(const uint8_t *text, size_t text_len,
                   const uint8_t *read, size_t read_len,
                   int W, int O,
                   go_run *runs, size_t cap, size_t *n_runs,
                   long long *edit_distance, go_stats *stats)
{
    /* one 64-bit word per bitvector up to W = 64 (src/bitvector.hpp:42-44), wider types beyond */
    if (W <= 64)
        return go_align_codes_64(text, text_len, read, read_len, W, O, runs, cap, n_runs, edit_distance, stats);
    if (W <= 128)
        return go_align_codes_128(text, text_len, read, read_len, W, O, runs, cap, n_runs, edit_distance, stats);
    return go_align_codes_256(text, text_len, read, read_len, W, O, runs, cap, n_runs, edit_distance, stats);
}

/* ASCII -> 0..3, genasm_cpu.cpp:462-493 (upper and lower case ACGT only). */
static int encode_bases(const char *src, size_t len, uint8_t *dst)
{
    for (size_t k = 0; k < len; k++) {
        switch (src[k]) {
        case 'A': case 'a': dst[k] = 0; break;
        case 'C': case 'c': dst[k] = 1; break;
        case 'G': case 'g': dst[k] = 2; break;
        case 'T': case 't': dst[k] = 3; break;
        default: return GO_ERR_BAD_BASE;
        }
    }
    return GO_OK;
}

static int render_cigar(const go_run *runs, size_t n, char *dst, size_t cap)
{
    size_t pos = 0;
    for (size_t k = 0; k < n; k++) {
        int w = snprintf(dst + pos, cap - pos, "%u%c", (unsigned)runs[k].count, runs[k].op);
        if (w < 0 || (size_t)w >= cap - pos)
            return GO_ERR_CAPACITY;
        pos += (size_t)w;
    }
    if (pos >= cap)
        return GO_ERR_CAPACITY;
    dst[pos] = '\0';
    return GO_OK;
}

int go_align_ascii(const char *text, size_t text_len,
                   const char *read, size_t read_len,
                   int W, int O,
                   char *cigar, size_t cigar_cap,
                   long long *edit_distance, go_stats *stats)
{
    uint8_t *t = (uint8_t *)malloc(text_len + 1);
    uint8_t *r = (uint8_t *)malloc(read_len + 1);
    /* every traceback step consumes a read or a text character */
    size_t cap = read_len + text_len + 2;
    go_run *runs = (go_run *)malloc(cap * sizeof(go_run));
    int rc = GO_ERR_CAPACITY;
    if (t && r && runs) {
        rc = encode_bases(text, text_len, t);
        if (rc == GO_OK)
            rc = encode_bases(read, read_len, r);
        size_t n = 0;
        if (rc == GO_OK)
            rc = go_align_codes(t, text_len, r, read_len, W, O, runs, cap, &n, edit_distance, stats);
        if (rc == GO_OK)
            rc = render_cigar(runs, n, cigar, cigar_cap);
    }
    free(t);
    free(r);
    free(runs);
    return rc;
}

static long long now_ns(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 1000000000LL + ts.tv_nsec;
}

int go_align_batch_ascii(size_t n_pairs,
                         const char *const *texts, const uint64_t *text_lens,
                         const char *const *reads, const uint64_t *read_lens,
                         int W, int O, int threads,
                         char *const *cigars, long long *edit_distances,
                         go_stats *total_stats, long long *kernel_ns)
{
    if (W < 2 || W > GO_MAXW || O < 0 || O >= W)
        return GO_ERR_PARAMS;
    if (threads < 1)
        threads = 1;

    /* untimed staging, as genasm_cpu.cpp:569-586 */
    uint8_t **tc = (uint8_t **)calloc(n_pairs ? n_pairs : 1, sizeof(*tc));
    uint8_t **rc_ = (uint8_t **)calloc(n_pairs ? n_pairs : 1, sizeof(*rc_));
    go_run **rr = (go_run **)calloc(n_pairs ? n_pairs : 1, sizeof(*rr));
    size_t *nr = (size_t *)calloc(n_pairs ? n_pairs : 1, sizeof(*nr));
    int status = GO_OK;
    if (!tc || !rc_ || !rr || !nr)
        status = GO_ERR_CAPACITY;
    for (size_t p = 0; p < n_pairs && status == GO_OK; p++) {
        tc[p] = (uint8_t *)malloc(text_lens[p] + 1);
        rc_[p] = (uint8_t *)malloc(read_lens[p] + 1);
        /* 2*read_len covers every case short of adversarial deletion-only
         * windows; those report GO_ERR_CAPACITY instead of overrunning */
        rr[p] = (go_run *)malloc((2 * read_lens[p] + 8) * sizeof(go_run));
        if (!tc[p] || !rc_[p] || !rr[p]) { status = GO_ERR_CAPACITY; break; }
        status = encode_bases(texts[p], text_lens[p], tc[p]);
        if (status == GO_OK)
            status = encode_bases(reads[p], read_lens[p], rc_[p]);
    }

    go_stats sum;
    memset(&sum, 0, sizeof(sum));
    long long t0 = now_ns();
    if (status == GO_OK) {
        int first_err = GO_OK;
        #pragma omp parallel num_threads(threads)
        {
            go_stats local;
            memset(&local, 0, sizeof(local));
            #pragma omp for schedule(dynamic)
            for (long long p = 0; p < (long long)n_pairs; p++) {
                int e = go_align_codes(tc[p], text_lens[p], rc_[p], read_lens[p], W, O,
                                       rr[p], 2 * read_lens[p] + 8, &nr[p],
                                       &edit_distances[p], &local);
                if (e != GO_OK) {
                    #pragma omp critical
                    if (first_err == GO_OK) first_err = e;
                }
            }
            #pragma omp critical
            {
                sum.windows += local.windows;
                sum.dc_cells += local.dc_cells;
                sum.tb_steps += local.tb_steps;
                sum.runs += local.runs;
                sum.text_used += local.text_used;
            }
        }
        status = first_err;
    }
    long long t1 = now_ns();

    for (size_t p = 0; p < n_pairs && status == GO_OK; p++)
        status = render_cigar(rr[p], nr[p], cigars[p], 4 * read_lens[p] + 1);

    for (size_t p = 0; p < n_pairs; p++) {
        if (tc) free(tc[p]);
        if (rc_) free(rc_[p]);
        if (rr) free(rr[p]);
    }
    free(tc); free(rc_); free(rr); free(nr);
    if (total_stats)
        *total_stats = sum;
    if (kernel_ns)
        *kernel_ns = t1 - t0;
    return status;
}

/* The same batch for inputs that sit in ONE array of fixed-size rows (the staging layout of bench.py: a text slot and a
 * read slot per row; text_lens / read_lens, if not NULL, give every row's own lengths <= text_len / read_len, the slot
 * sizes) and results as arrays: edit distances, run offsets and the runs themselves as {count, op} byte pairs
 * (the layout of the reference's CigarEntry_t, src/util.hpp:43-46) — a full-size batch is compared array against array,
 * with no per-pair objects on the caller's side.  runs_cap counts runs; GO_ERR_CAPACITY if they do not fit. */
int go_align_batch_rows_var(size_t n_pairs, const char *rows, uint64_t row_stride,
                            uint64_t text_off, uint64_t text_len, uint64_t read_off, uint64_t read_len,
                            const uint64_t *text_lens, const uint64_t *read_lens,
                            int W, int O, int threads,
                            long long *edit_distances, uint64_t *run_offsets /* n_pairs + 1 */,
                            uint8_t *runs_out, uint64_t runs_cap,
                            go_stats *total_stats, long long *kernel_ns)
{
    if (W < 2 || W > GO_MAXW || O < 0 || O >= W)
        return GO_ERR_PARAMS;
    if (threads < 1)
        threads = 1;
    const size_t cap1 = 2 * read_len + 8;
    uint8_t *tc = (uint8_t *)malloc((n_pairs ? n_pairs : 1) * (text_len + 1));
    uint8_t *rc_ = (uint8_t *)malloc((n_pairs ? n_pairs : 1) * (read_len + 1));
    go_run *rr = (go_run *)malloc((n_pairs ? n_pairs : 1) * cap1 * sizeof(go_run));
    size_t *nr = (size_t *)calloc(n_pairs ? n_pairs : 1, sizeof(*nr));
    int status = (tc && rc_ && rr && nr) ? GO_OK : GO_ERR_CAPACITY;
    if (status == GO_OK) {
        int first_err = GO_OK;
        #pragma omp parallel for num_threads(threads) schedule(static)
        for (long long p = 0; p < (long long)n_pairs; p++) {
            const uint64_t tl = text_lens ? text_lens[p] : text_len, rl = read_lens ? read_lens[p] : read_len;
            int e = (tl <= text_len && rl <= read_len) ? GO_OK : GO_ERR_PARAMS;
            if (e == GO_OK)
                e = encode_bases(rows + (size_t)p * row_stride + text_off, tl, tc + (size_t)p * (text_len + 1));
            if (e == GO_OK)
                e = encode_bases(rows + (size_t)p * row_stride + read_off, rl, rc_ + (size_t)p * (read_len + 1));
            if (e != GO_OK) {
                #pragma omp critical
                if (first_err == GO_OK) first_err = e;
            }
        }
        status = first_err;
    }
    go_stats sum;
    memset(&sum, 0, sizeof(sum));
    long long t0 = now_ns();
    if (status == GO_OK) {
        int first_err = GO_OK;
        #pragma omp parallel num_threads(threads)
        {
            go_stats local;
            memset(&local, 0, sizeof(local));
            #pragma omp for schedule(dynamic)
            for (long long p = 0; p < (long long)n_pairs; p++) {
                const uint64_t tl = text_lens ? text_lens[p] : text_len, rl = read_lens ? read_lens[p] : read_len;
                int e = go_align_codes(tc + (size_t)p * (text_len + 1), tl, rc_ + (size_t)p * (read_len + 1), rl, W, O,
                                       rr + (size_t)p * cap1, cap1, &nr[p], &edit_distances[p], &local);
                if (e != GO_OK) {
                    #pragma omp critical
                    if (first_err == GO_OK) first_err = e;
                }
            }
            #pragma omp critical
            {
                sum.windows += local.windows;
                sum.dc_cells += local.dc_cells;
                sum.tb_steps += local.tb_steps;
                sum.runs += local.runs;
                sum.text_used += local.text_used;
            }
        }
        status = first_err;
    }
    long long t1 = now_ns();
    if (status == GO_OK) {
        uint64_t acc = 0;
        for (size_t p = 0; p < n_pairs; p++) {
            run_offsets[p] = acc;
            acc += nr[p];
        }
        run_offsets[n_pairs] = acc;
        if (acc > runs_cap)
            status = GO_ERR_CAPACITY;
    }
    if (status == GO_OK) {
        #pragma omp parallel for num_threads(threads) schedule(static)
        for (long long p = 0; p < (long long)n_pairs; p++) {
            uint8_t *o = runs_out + 2 * run_offsets[p];
            const go_run *r = rr + (size_t)p * cap1;
            for (size_t k = 0; k < nr[p]; k++) {
                o[2 * k] = (uint8_t)r[k].count;
                o[2 * k + 1] = (uint8_t)r[k].op;
            }
        }
    }
    free(tc); free(rc_); free(rr); free(nr);
    if (total_stats)
        *total_stats = sum;
    if (kernel_ns)
        *kernel_ns = t1 - t0;
    return status;
}

/* fixed lengths: every row holds a text of text_len and a read of read_len characters */
int go_align_batch_rows(size_t n_pairs, const char *rows, uint64_t row_stride,
                        uint64_t text_off, uint64_t text_len, uint64_t read_off, uint64_t read_len,
                        int W, int O, int threads,
                        long long *edit_distances, uint64_t *run_offsets /* n_pairs + 1 */,
                        uint8_t *runs_out, uint64_t runs_cap,
                        go_stats *total_stats, long long *kernel_ns)
{
    return go_align_batch_rows_var(n_pairs, rows, row_stride, text_off, text_len, read_off, read_len, NULL, NULL, W, O, threads,
                                   edit_distances, run_offsets, runs_out, runs_cap, total_stats, kernel_ns);
}
