/*
 * genasm_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the Scrooge/GenASM windowed aligner, used as the parity
 * checker for the HIP path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may call into this; the product library
 * (scrooge_amd/csrc) never links or loads it.
 *
 * Parity status: PINNED.  The restatement is checked against
 *   (1) the reference's own known answers (src/tests.cu:246 edit distances),
 *   (2) outputs of the unmodified reference CPU path compiled into
 *       oracle/_ref/ (see oracle/Makefile, oracle/ref_driver.cpp), and
 *   (3) the committed fixtures in tests/golden/ generated from (2).
 */
#ifndef GENASM_ORACLE_H
#define GENASM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One CIGAR run; same two-byte shape as the reference's CigarEntry_t
 * (src/util.hpp:43-46): a count that never exceeds 2*(W-O) and an op in "=XID". */
typedef struct go_run {
    uint8_t count;
    char    op;
} go_run;

/* Work counters the bench uses to turn pairs/s into algorithmic ops
 * (SURVEY.md §8d): dc_cells = sum over windows of (n_w+1)*(d_w+1). */
typedef struct go_stats {
    uint64_t windows;
    uint64_t dc_cells;
    uint64_t tb_steps;
    uint64_t runs;
    uint64_t text_used;
} go_stats;

enum {
    GO_OK = 0,
    GO_ERR_BAD_BASE = 1,   /* non-ACGT input (reference: assert(false), genasm_cpu.cpp:487-489) */
    GO_ERR_PARAMS = 2,     /* W/O outside the supported range */
    GO_ERR_CAPACITY = 3    /* caller-provided run/char buffer too small */
};

/* Align one pair given 0..3 base codes.  W in [2,256], 0 <= O < W.  K == W as in
 * the reference defaults (genasm_cpu.cpp:7-9).  Writes up to cap runs. */
int go_align_codes(const uint8_t *text, size_t text_len,
                   const uint8_t *read, size_t read_len,
                   int W, int O,
                   go_run *runs, size_t cap, size_t *n_runs,
                   long long *edit_distance, go_stats *stats);

/* ASCII front end: converts like ascii_to_zero_based_string
 * (genasm_cpu.cpp:462-493) and renders the CIGAR as "%d%c" text like
 * genasm_tb's sprintf (genasm_cpu.cpp:387-403).  cigar must hold
 * 4*read_len+1 bytes (same bound as genasm_cpu.cpp:520). */
int go_align_ascii(const char *text, size_t text_len,
                   const char *read, size_t read_len,
                   int W, int O,
                   char *cigar, size_t cigar_cap,
                   long long *edit_distance, go_stats *stats);

/* Batch over pairs with an OpenMP dynamic loop (genasm_cpu.cpp:440-460).
 * cigars[i] must hold 4*read_lens[i]+1 bytes.  Returns the first non-OK
 * status; total_stats may be NULL.  kernel_ns covers the pair loop only,
 * conversion excluded, like genasm_cpu.cpp:532-534. */
int go_align_batch_ascii(size_t n_pairs,
                         const char *const *texts, const uint64_t *text_lens,
                         const char *const *reads, const uint64_t *read_lens,
                         int W, int O, int threads,
                         char *const *cigars, long long *edit_distances,
                         go_stats *total_stats, long long *kernel_ns);

/* Test hook over the 256-bit vector type (see genasm_oracle.c). */
int go_bv256_op(int op, const uint64_t a[4], const uint64_t b[4], unsigned s, uint64_t out[4]);

/* The same for inputs in one array of fixed-size rows (text slot + read slot per row) with results as arrays: run
 * offsets [n_pairs + 1] and the runs as {count, op} byte pairs.  runs_cap counts runs. */
int go_align_batch_rows(size_t n_pairs, const char *rows, uint64_t row_stride,
                        uint64_t text_off, uint64_t text_len, uint64_t read_off, uint64_t read_len,
                        int W, int O, int threads,
                        long long *edit_distances, uint64_t *run_offsets,
                        uint8_t *runs_out, uint64_t runs_cap,
                        go_stats *total_stats, long long *kernel_ns);
/* the same with every row's own lengths (NULL = the slot sizes text_len / read_len) */
int go_align_batch_rows_var(size_t n_pairs, const char *rows, uint64_t row_stride,
                        uint64_t text_off, uint64_t text_len, uint64_t read_off, uint64_t read_len,
                            const uint64_t *text_lens, const uint64_t *read_lens,
                        int W, int O, int threads,
                        long long *edit_distances, uint64_t *run_offsets,
                        uint8_t *runs_out, uint64_t runs_cap,
                        go_stats *total_stats, long long *kernel_ns);

#ifdef __cplusplus
}
#endif
#endif
