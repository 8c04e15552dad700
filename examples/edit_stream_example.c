/* edit_stream_example.c — what the receiving side of a multi-GPU job (or anything that stores CIGARs compactly)
 * does with an edit stream, in plain C and without a GPU: one byte per edit and per window on the wire, the reference's per-window
 * CIGAR runs (src/genasm_cpu.cpp:304-305, 400-403) restored bit for bit from it.
 *
 *   gcc -std=c11 -Iinclude examples/edit_stream_example.c -Lscrooge_amd -lscrooge_amd -Wl,-rpath,$PWD/scrooge_amd
 */
#include <stdio.h>
#include <string.h>

#include "scrooge_amd.h"

static void show_runs(const char *what, const scrg_run *r, uint64_t n)
{
    printf("%s ", what);
    for (uint64_t k = 0; k < n; k++) printf("%u%c", (unsigned)r[k].count, r[k].op);
    printf("\n");
}

int main(void)
{
    /* the runs of one alignment as the aligner delivers them for W = 64, O = 33 (a window consumes at most 31
     * characters of the read or of the text, whichever comes first): 100 bp read, one substitution, one deletion */
    const scrg_run runs[] = { {31, '='}, {9, '='}, {1, 'X'}, {21, '='}, {6, '='}, {1, 'D'}, {24, '='}, {8, '='} };
    const uint64_t n_runs = sizeof runs / sizeof runs[0], read_len = 31 + 9 + 1 + 21 + 6 + 24 + 8;
    show_runs("runs  ", runs, n_runs);

    scrg_params p;
    scrg_params_default(&p);                                              /* W = 64, O = 33 */
    uint8_t stream[32];
    uint64_t n_bytes = 0;
    if (scrg_runs_to_edit_stream(&p, runs, n_runs, stream, sizeof stream, &n_bytes) != SCRG_OK) return 1;
    /* one byte per edit (X after 9 matches: 1 << 6 | 9; D after 6: 3 << 6 | 6) and one per window end (the matches before it) */
    printf("stream %llu bytes:", (unsigned long long)n_bytes);
    for (uint64_t k = 0; k < n_bytes; k++) printf(" %02x", stream[k]);
    printf("\n");

    scrg_run back[32];
    uint64_t n_back = 0;
    if (scrg_edit_stream_to_runs(&p, read_len, stream, n_bytes, back, 32, &n_back) != SCRG_OK) return 2;
    show_runs("decoded", back, n_back);
    if (n_back != n_runs || memcmp(back, runs, sizeof runs) != 0) return 3;

    /* other window settings put the breaks elsewhere: the encoder cuts the same alignment at THEIR windows ... */
    scrg_params q = p;
    q.W = 32;
    q.O = 17;
    uint8_t stream_q[32];
    uint64_t n_bytes_q = 0;
    if (scrg_runs_to_edit_stream(&q, runs, n_runs, stream_q, sizeof stream_q, &n_bytes_q) != SCRG_OK) return 4;
    if (scrg_edit_stream_to_runs(&q, read_len, stream_q, n_bytes_q, back, 32, &n_back) != SCRG_OK) return 4;
    show_runs("W32/O17", back, n_back);
    /* ... and the host decoder, which is told W and O, refuses a stream whose windows are not theirs */
    if (scrg_edit_stream_to_runs(&q, read_len, stream, n_bytes, back, 32, &n_back) != SCRG_ERR_INVALID_ARG) return 6;

    /* a stream that is not an alignment of a read of this length is refused */
    if (scrg_edit_stream_to_runs(&p, read_len - 50, stream, n_bytes, back, 32, &n_back) != SCRG_ERR_INVALID_ARG) return 5;
    printf("ok\n");
    return 0;
}
