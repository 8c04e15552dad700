/*
 * ref_driver.cpp — TEST INFRASTRUCTURE ONLY.
 *
 * Thin extern "C" wrapper around the UNMODIFIED reference CPU path
 * (/root/reference/src/genasm_cpu.cpp), compiled where it lies by
 * oracle/Makefile into oracle/_ref/libgenasm_ref.so.  No reference source is
 * copied into this repository; this file only calls the reference's public
 * entry points (src/genasm_cpu.hpp:6-7) through its own headers.
 *
 * The reference's unstructured overload drops every odd-indexed result
 * (src/genasm_cpu.cpp:600-605 increments pair_idx twice per iteration), so
 * this wrapper interleaves an empty dummy pair after every real pair: real
 * pairs land on even indices and all of them come back.  The timed kernel
 * region (:589-591) still covers exactly the real work (an empty read costs
 * zero windows).
 */
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "genasm_cpu.hpp"

extern "C" {

int ref_align_pairs(size_t n_pairs,
                    const char *const *texts, const uint64_t *text_lens,
                    const char *const *reads, const uint64_t *read_lens,
                    int threads,
                    char *const *cigars, long long *edit_distances,
                    long long *kernel_ns)
{
    genasm_cpu::enabled_algorithm_log = false;
    std::vector<std::string> t, q;
    t.reserve(2 * n_pairs);
    q.reserve(2 * n_pairs);
    for (size_t p = 0; p < n_pairs; p++) {
        t.emplace_back(texts[p], text_lens[p]);
        q.emplace_back(reads[p], read_lens[p]);
        t.emplace_back();
        q.emplace_back();
    }
    long long ns = 0;
    std::vector<Alignment_t> res = genasm_cpu::align_all(t, q, threads, &ns);
    if (res.size() != n_pairs)
        return 1;
    for (size_t p = 0; p < n_pairs; p++) {
        std::memcpy(cigars[p], res[p].cigar.c_str(), res[p].cigar.size() + 1);
        edit_distances[p] = res[p].edit_distance;
    }
    if (kernel_ns)
        *kernel_ns = ns;
    return 0;
}

/* The same call for inputs that sit in ONE array of fixed-size rows (bench.py's staging layout: a text slot and a read
 * slot per row), with the reference's CIGAR strings turned into arrays here: run offsets [n_pairs + 1] and the runs as
 * {count, op} byte pairs (CigarEntry_t, src/util.hpp:43-46), so that a full-size batch is compared array against array.
 * text_lens / read_lens, if not null, give every row's own lengths (<= text_len / read_len, the slot sizes; else 4).
 * Returns 2 if a CIGAR does not parse or has a count above 255, 3 if runs_cap (in runs) is too small. */
int ref_align_rows_var(size_t n_pairs, const char *rows, uint64_t row_stride,
                       uint64_t text_off, uint64_t text_len, uint64_t read_off, uint64_t read_len,
                       const uint64_t *text_lens, const uint64_t *read_lens,
                       int threads,
                       long long *edit_distances, uint64_t *run_offsets, uint8_t *runs_out, uint64_t runs_cap,
                       long long *kernel_ns)
{
    genasm_cpu::enabled_algorithm_log = false;
    std::vector<std::string> t, q;
    t.reserve(2 * n_pairs);
    q.reserve(2 * n_pairs);
    for (size_t p = 0; p < n_pairs; p++) {
        const uint64_t tl = text_lens ? text_lens[p] : text_len, rl = read_lens ? read_lens[p] : read_len;
        if (tl > text_len || rl > read_len) return 4;
        t.emplace_back(rows + p * row_stride + text_off, tl);
        q.emplace_back(rows + p * row_stride + read_off, rl);
        t.emplace_back();               // (the dummy pair that defeats the reference's double increment, see above)
        q.emplace_back();
    }
    long long ns = 0;
    std::vector<Alignment_t> res = genasm_cpu::align_all(t, q, threads, &ns);
    if (res.size() != n_pairs)
        return 1;
    uint64_t acc = 0;
    for (size_t p = 0; p < n_pairs; p++) {
        run_offsets[p] = acc;
        for (char c : res[p].cigar)
            if (c < '0' || c > '9') acc++;
        edit_distances[p] = res[p].edit_distance;
    }
    run_offsets[n_pairs] = acc;
    if (acc > runs_cap)
        return 3;
    int bad = 0;
    #pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static) reduction(|:bad)
    for (long long p = 0; p < (long long)n_pairs; p++) {
        uint8_t *o = runs_out + 2 * run_offsets[p];
        unsigned cnt = 0;
        bool digits = false;
        for (char c : res[p].cigar) {
            if (c >= '0' && c <= '9') {
                cnt = cnt * 10 + (unsigned)(c - '0');
                digits = true;
            } else {
                if (!digits || cnt == 0 || cnt > 255 || (c != '=' && c != 'X' && c != 'I' && c != 'D')) bad = 1;
                *o++ = (uint8_t)cnt;
                *o++ = (uint8_t)c;
                cnt = 0;
                digits = false;
            }
        }
        if (digits) bad = 1;
    }
    if (kernel_ns)
        *kernel_ns = ns;
    return bad ? 2 : 0;
}

/* fixed lengths: every row holds a text of text_len and a read of read_len characters */
int ref_align_rows(size_t n_pairs, const char *rows, uint64_t row_stride,
                   uint64_t text_off, uint64_t text_len, uint64_t read_off, uint64_t read_len,
                   int threads,
                   long long *edit_distances, uint64_t *run_offsets, uint8_t *runs_out, uint64_t runs_cap,
                   long long *kernel_ns)
{
    return ref_align_rows_var(n_pairs, rows, row_stride, text_off, text_len, read_off, read_len, nullptr, nullptr, threads,
                              edit_distances, run_offsets, runs_out, runs_cap, kernel_ns);
}

/* Read-mapping overload (src/genasm_cpu.cpp:495-555): candidate k of read r
 * aligns the read against the genome suffix starting at cand_starts[...]. */
int ref_align_mapping(const char *genome, uint64_t genome_len,
                      size_t n_reads,
                      const char *const *reads, const uint64_t *read_lens,
                      const uint64_t *cand_offsets, /* n_reads+1 */
                      const uint64_t *cand_starts,
                      int threads,
                      char *const *cigars, long long *edit_distances,
                      long long *kernel_ns)
{
    genasm_cpu::enabled_algorithm_log = false;
    Genome_t g;
    g.content.assign(genome, genome_len);
    std::vector<Read_t> rs(n_reads);
    for (size_t r = 0; r < n_reads; r++) {
        rs[r].content.assign(reads[r], read_lens[r]);
        for (uint64_t c = cand_offsets[r]; c < cand_offsets[r + 1]; c++) {
            CandidateLocation_t loc;
            loc.start_in_reference = (long long)cand_starts[c];
            loc.start_in_chromosome = (long long)cand_starts[c];
            loc.start_of_aligned_region = 0;
            loc.size_of_aligned_region = 0;
            loc.strand = true;
            rs[r].locations.push_back(loc);
        }
    }
    long long ns = 0;
    std::vector<Alignment_t> res = genasm_cpu::align_all(g, rs, threads, &ns);
    if (res.size() != cand_offsets[n_reads])
        return 1;
    for (size_t p = 0; p < res.size(); p++) {
        std::memcpy(cigars[p], res[p].cigar.c_str(), res[p].cigar.size() + 1);
        edit_distances[p] = res[p].edit_distance;
    }
    if (kernel_ns)
        *kernel_ns = ns;
    return 0;
}

}

/* ---- the reference's read-mapping front door (src/util.cpp), for tests of scrg_job_load ---- */
#include <algorithm>
#include <fstream>

extern "C" int ref_dump_job(const char *genome_fa, const char *fastq, const char *seeds, const char *out_path)
{
    /* same preparation as the reference's perf tests, src/tests.cu:339-355: load, attach seeds,
     * drop reverse-strand candidates; one line per read: name, sequence, forward start_in_reference list */
    try {
        Genome_t genome = read_genome(genome_fa);
        std::vector<Read_t> reads;
        read_fastq_and_seed_locations(genome, fastq, seeds, reads);
        std::ofstream o(out_path);
        o << "genome\t" << genome.content << "\n";
        for (Read_t &r : reads) {
            o << r.description << "\t" << r.content << "\t";
            for (CandidateLocation_t &l : r.locations)
                if (l.strand) o << l.start_in_reference << ",";
            o << "\n";
        }
    } catch (const std::exception &e) {
        return 1;
    }
    return 0;
}
