/*
 * genasm_oracle.c — TEST INFRASTRUCTURE ONLY (see genasm_oracle.h).
 *
 * Plain-C restatement of the windowed GenASM aligner in the reference's
 * src/genasm_cpu.cpp.  It is written from the algorithm's definition, one
 * uint64_t per bitvector (W <= 64, src/bitvector.hpp:42-44), and keeps the
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

#define GO_MAXW 64

typedef struct go_scratch {
    uint64_t R[GO_MAXW + 1][GO_MAXW + 1]; /* R[d][i] */
} go_scratch;

static inline uint64_t shl64(uint64_t v, unsigned s)
{
    /* bitvector::operator<< yields zero once the shift reaches the width
     * (src/bitvector.hpp:116-122). */
    return s >= 64 ? 0 : v << s;
}

static inline int bit_is_zero(uint64_t v, unsigned b)
{
    return ((v >> b) & 1u) == 0; /* has_zero_at, src/bitvector.hpp:163-175 */
}

/* Pattern masks, genasm_cpu.cpp:178-198. */
static void pattern_masks(const uint8_t *pattern, size_t m, uint64_t pm[4])
{
    pm[0] = pm[1] = pm[2] = pm[3] = ~(uint64_t)0;
    for (size_t b = 0; b < m; b++)
        pm[pattern[m - 1 - b]] &= ~((uint64_t)1 << b);
}

/*
 * Distance calculation, genasm_cpu.cpp:210-288: rows d = 0..K, columns
 * i = n..0, stops at the first row whose column 0 has bit m-1 clear
 * (EARLY_TERMINATION, :278-283).  Returns that row, or -1 if none exists
 * (cannot happen while K >= m, since m insertions always work).
 */
static int distance_sweep(const uint8_t *text, size_t n,
                          const uint8_t *pattern, size_t m, int K,
                          go_scratch *s, go_stats *st)
{
    uint64_t pm[4];
    pattern_masks(pattern, m, pm);

    for (int d = 0; d <= K; d++) {
        uint64_t *row = s->R[d];
        const uint64_t *up = d ? s->R[d - 1] : NULL;

        /* column n: nothing of the text left, only insertions (:239-245) */
        row[n] = d ? shl64(~(uint64_t)0, (unsigned)d) : ~(uint64_t)0;

        for (size_t i = n; i-- > 0;) {
            uint64_t match = (row[i + 1] << 1) | pm[text[i]];
            if (d == 0) {
                row[i] = match; /* :232-238 */
            } else {
                uint64_t sub = up[i + 1] << 1;
                uint64_t ins = up[i] << 1;
                uint64_t del = up[i + 1];
                row[i] = match & sub & ins & del; /* :246-252 */
            }
        }
        if (st)
            st->dc_cells += n + 1;
        if (bit_is_zero(row[0], (unsigned)(m - 1)))
            return d;
    }
    return -1;
}

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

/*
 * Traceback of one window, genasm_cpu.cpp:290-409.  Walks from (i,j,d) =
 * (0,0,window distance); stops when the pattern is used up or either index
 * reaches W-O (:307-310).  Edit preference: insertion, deletion,
 * substitution, match (:346-370); the last pattern character has its own
 * rule (:336-343).  Runs are flushed per window and never merged with the
 * next window's (:304-305, 400-403).
 */
static int traceback(const go_scratch *s, size_t n, size_t m, int dist,
                     size_t limit, size_t *text_used, size_t *pattern_used,
                     run_sink *out, go_stats *st)
{
    size_t i = 0, j = 0;
    int d = dist;
    char cur = 0;
    unsigned cur_len = 0;

    while (j < m && i < limit && j < limit) {
        int room = d > 0;
        int text_left = i < n;
        int ins, del, sub;

        if (j + 1 < m) {
            unsigned bj = (unsigned)(m - 1 - j);      /* TB_BIT(j), :59 */
            unsigned bj1 = bj - 1;                    /* TB_BIT(j+1) */
            ins = room && bit_is_zero(s->R[d - 1][i], bj1);
            del = room && text_left && bit_is_zero(s->R[d - 1][i + 1], bj);
            sub = room && text_left && bit_is_zero(s->R[d - 1][i + 1], bj1);
        } else {
            ins = room;
            del = 0;
            sub = room && text_left;
        }

        char op;
        if (ins)      { op = 'I'; j++; d--; }
        else if (del) { op = 'D'; i++; d--; }
        else if (sub) { op = 'X'; i++; j++; d--; }
        else          { op = '='; i++; j++; }

        if (op == cur) {
            cur_len++;
        } else {
            if (cur_len)
                sink_push(out, cur, cur_len);
            cur = op;
            cur_len = 1;
        }
        if (st)
            st->tb_steps++;
    }
    if (cur_len)
        sink_push(out, cur, cur_len);

    *text_used = i;
    *pattern_used = j;
    return dist - d;
}

int go_align_codes(const uint8_t *text, size_t text_len,
                   const uint8_t *read, size_t read_len,
                   int W, int O,
                   go_run *runs, size_t cap, size_t *n_runs,
                   long long *edit_distance, go_stats *stats)
{
    if (W < 2 || W > GO_MAXW || O < 0 || O >= W)
        return GO_ERR_PARAMS;

    /* per-thread table, like the per-thread R of genasm_cpu.cpp:444 */
    static _Thread_local go_scratch scratch;
    go_scratch *s = &scratch;

    run_sink out = { runs, cap, 0, 0 };
    size_t ti = 0, ri = 0;
    long long total = 0;
    const size_t limit = (size_t)(W - O);

    /* window loop, genasm_cpu.cpp:411-438 */
    while (ri < read_len) {
        size_t n = text_len - ti < (size_t)W ? text_len - ti : (size_t)W;
        size_t m = read_len - ri < (size_t)W ? read_len - ri : (size_t)W;

        int dist = distance_sweep(text + ti, n, read + ri, m, W, s, stats);
        if (dist < 0) /* unreachable with K == W */
            return GO_ERR_PARAMS;
        if (stats)
            stats->windows++;

        size_t tu, pu;
        total += traceback(s, n, m, dist, limit, &tu, &pu, &out, stats);
        ti += tu;
        ri += pu;
    }

    if (n_runs)
        *n_runs = out.n;
    if (edit_distance)
        *edit_distance = total;
    if (stats) {
        stats->runs += out.n;
        stats->text_used += ti;
    }
    return out.overflow ? GO_ERR_CAPACITY : GO_OK;
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
