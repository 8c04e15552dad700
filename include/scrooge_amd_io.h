/*
 * scrooge_amd_io.h — the callers and data formats either side of the hot path
 * (SURVEY.md §8f): the reference's read-mapping front door and its result checks.
 *
 *   read_genome / read_fasta                  src/util.cpp:45-108
 *   read_fastq                                src/util.cpp:110-155
 *   read_maf / read_paf                       src/util.cpp:175-276
 *   left_extend_locations, get_global_seeds   src/util.cpp:284-301
 *   read_fastq_and_seed_locations             src/util.cpp:303-336
 *   the perf-test preparation (strand filter, length cap, inflation, sort)
 *                                             src/tests.cu:346-377
 *   get_alignment_score (affine re-scoring)   src/cpu_baseline.cpp:694-725
 *   validateCigarString                       src/tests.cu:27-169
 *
 * Host-side C++ behind a C ABI, in the same shared library as the aligner.
 * Beyond the reference: reverse-strand candidates can be aligned (the read is
 * reverse-complemented while staging) instead of dropped, and results can be
 * written as PAF with a cg:Z: tag or as SAM.
 */
#ifndef SCROOGE_AMD_IO_H
#define SCROOGE_AMD_IO_H

#include "scrooge_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    SCRG_ERR_IO = 16,          /* file missing / unreadable                         */
    SCRG_ERR_FORMAT = 17       /* seed names an unknown read or chromosome, bad line */
};

typedef struct scrg_job scrg_job;   /* genome + reads + candidate locations */

typedef struct scrg_job_options {
    int32_t reverse_strand;    /* 0: drop '-' candidates (reference, src/tests.cu:346-355);
                                  1: align the reverse-complemented read against them        */
    int32_t sort_by_length;    /* 1: order reads longest first (src/tests.cu:375-377)        */
    int32_t inflation;         /* >1: replicate the read set (src/tests.cu:366-372)          */
    int32_t left_extend;       /* 1: left_extend_locations() before use (src/util.cpp:284)   */
    int64_t read_length_cap;   /* >=0: truncate reads (src/tests.cu:360-364); -1: off        */
} scrg_job_options;

void scrg_job_options_default(scrg_job_options *o);

/* Loads FASTA + FASTQ + seeds (.maf or .paf).  err (may be NULL) receives a message. */
scrg_status scrg_job_load(const char *genome_fasta, const char *reads_fastq, const char *seeds_path,
                          const scrg_job_options *opt, scrg_job **out, char *err, size_t err_len);
void scrg_job_free(scrg_job *job);

/* Sizes: reads, (read, candidate) pairs, genome bases, chromosomes. */
void scrg_job_counts(const scrg_job *job, uint64_t *n_reads, uint64_t *n_pairs, uint64_t *genome_len,
                     uint64_t *n_chromosomes);
/* Borrowed views in the shape scrg_align_mapping() takes; cand_reverse[k] != 0 marks a
 * reverse-strand candidate (its cand_start is on the forward genome). */
void scrg_job_arrays(const scrg_job *job, const char **genome, const char *const **reads,
                     const uint64_t **read_lens, const uint64_t **cand_offsets, const uint64_t **cand_start,
                     const uint8_t **cand_reverse);
const char *scrg_job_read_name(const scrg_job *job, uint64_t read);
/* chromosome name and 0-based start inside it for pair k */
const char *scrg_job_pair_chromosome(const scrg_job *job, uint64_t pair, uint64_t *start_in_chromosome,
                                     uint64_t *chromosome_len);

/* scrg_align_mapping() with a per-candidate strand flag (NULL = all forward). */
scrg_status scrg_align_mapping_stranded(scrg_ctx *ctx, const scrg_params *params,
                                        const char *genome, uint64_t genome_len,
                                        uint64_t n_reads, const char *const *reads, const uint64_t *read_lens,
                                        const uint64_t *cand_offsets, const uint64_t *cand_start,
                                        const uint8_t *cand_reverse, scrg_result **out);
/* Convenience: align every pair of a loaded job. */
scrg_status scrg_job_align(scrg_ctx *ctx, const scrg_params *params, const scrg_job *job, scrg_result **out);

/* One line per pair.  format 0: PAF (12 columns + NM:i + cg:Z:), 1: SAM. */
scrg_status scrg_job_write(const scrg_job *job, const scrg_result *res, const char *path, int format);

/* Affine re-scoring of a CIGAR exactly as get_alignment_score (src/cpu_baseline.cpp:694-725):
 * +match per '=', -mismatch per 'X', -(open + extend*len) per maximal I/D stretch
 * (consecutive I and D runs share one opening). */
scrg_status scrg_affine_score(const char *cigar, int64_t match_bonus, int64_t mismatch_cost,
                              int64_t gap_open_cost, int64_t gap_extend_cost, int64_t *score);

/* validateCigarString (src/tests.cu:27-169).  Returns SCRG_OK, or SCRG_ERR_FORMAT and a
 * reason code in *why: 1 malformed, 2 zero-length run, 3 read not consumed exactly,
 * 4 runs past the text, 5 '='/'X' disagrees with the sequences, 6 edit count != edit_distance. */
scrg_status scrg_validate_alignment(const char *text, uint64_t text_len, const char *read, uint64_t read_len,
                                    const char *cigar, int64_t edit_distance, int32_t *why);

#ifdef __cplusplus
}
#endif
#endif
