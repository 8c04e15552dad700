// scrg_io.cpp — file formats and result checks either side of the hot path
// (include/scrooge_amd_io.h).  Restates the behaviour of the reference's
// src/util.cpp readers and src/tests.cu / src/cpu_baseline.cpp checks; written
// against the formats, not translated from the reference sources.

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/scrooge_amd_io.h"

namespace {

bool slurp(const std::string& path, std::string& out, std::string& err)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        err = "could not read file \"" + path + "\"";
        return false;
    }
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    if (!f || n < 0) {                     // a directory or an unseekable path
        err = "could not read file \"" + path + "\"";
        return false;
    }
    f.seekg(0, std::ios::beg);
    out.resize((size_t)n);
    if (n) f.read(&out[0], n);
    if (!f) {
        err = "could not read file \"" + path + "\"";
        return false;
    }
    return true;
}

struct Chromosome {
    std::string name;
    uint64_t start, len;
};

struct Candidate {
    std::string read_name, chromosome;
    long long start_in_chromosome = 0;
    long long start_of_aligned_region = 0;
    long long size_of_aligned_region = 0;
    bool forward = true;
};

// FASTA (src/util.cpp:45-92): the description is the whole header line; content skips
// newlines, carriage returns and blanks up to the next '>'.
void parse_fasta(const std::string& raw, std::string& content, std::vector<Chromosome>& chroms)
{
    size_t i = 0;
    const size_t n = raw.size();
    while (i < n) {
        while (i < n && raw[i] != '>') i++;
        if (i >= n) break;
        i++;
        Chromosome c;
        while (i < n && raw[i] != '\n' && raw[i] != '\r') c.name.push_back(raw[i++]);
        c.start = content.size();
        while (i < n && raw[i] != '>') {
            const char ch = raw[i++];
            if (ch != '\n' && ch != '\r' && ch != ' ') content.push_back(ch);
        }
        c.len = content.size() - c.start;
        chroms.push_back(c);
    }
}

struct FastqRead {
    std::string name, seq;
};

// FASTQ: '@' header (blanks and CRs removed from the name, src/util.cpp:134-141), one sequence
// line, '+' line, quality line.  Unlike the reference's scan for the next '@' character
// (src/util.cpp:123-130), records are consumed as four lines, so an '@' inside a quality
// string cannot start a bogus record; on files without such characters both agree.
void parse_fastq(const std::string& raw, std::vector<FastqRead>& reads)
{
    size_t i = 0;
    const size_t n = raw.size();
    auto next_line = [&](size_t& b, size_t& e) {
        b = i;
        while (i < n && raw[i] != '\n') i++;
        e = i;
        if (i < n) i++;
        while (e > b && raw[e - 1] == '\r') e--;
    };
    while (i < n) {
        size_t b, e;
        next_line(b, e);
        if (e == b || raw[b] != '@') continue;
        FastqRead r;
        for (size_t k = b + 1; k < e; k++)
            if (raw[k] != ' ' && raw[k] != '\r') r.name.push_back(raw[k]);
        next_line(b, e);
        r.seq.assign(raw, b, e - b);
        const size_t save = i;
        next_line(b, e);
        if (e > b && raw[b] == '+') next_line(b, e);   // quality line
        else i = save;                                    // sequence-only record
        reads.push_back(std::move(r));
    }
}

// MAF as written by PBSIM (src/util.cpp:175-236): an 'a' line opens a block, 's' lines carry
// "src start size strand srcSize text"; the line whose src is "ref" gives the reference start.
void parse_maf(const std::string& raw, std::vector<Candidate>& out)
{
    std::istringstream in(raw);
    std::string line;
    while (std::getline(in, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        if (line.empty() || line[0] != 'a') continue;
        Candidate c;
        while (std::getline(in, line)) {
            while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
            if (line.empty()) break;
            if (line[0] != 's') continue;
            std::istringstream ls(line.substr(1));
            std::string src, strand, text;
            long long start = 0, size = 0, src_size = 0;
            ls >> src >> start >> size >> strand >> src_size >> text;
            if (src == "ref") {
                c.start_in_chromosome = start;
                c.chromosome = "ref";
            } else {
                c.read_name = src;
                c.forward = (strand == "+");
                c.start_of_aligned_region = start;
                c.size_of_aligned_region = size;
            }
        }
        out.push_back(c);
    }
}

// PAF (src/util.cpp:238-276): qname qlen qstart qend strand tname tlen tstart tend matches alnlen ...
bool parse_paf(const std::string& raw, std::vector<Candidate>& out, std::string& err)
{
    std::istringstream in(raw);
    std::string line;
    size_t lineno = 0;
    while (std::getline(in, line)) {
        lineno++;
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        if (line.empty()) continue;
        std::vector<std::string> f;
        size_t b = 0;
        while (b <= line.size() && f.size() < 12) {
            size_t e = line.find('\t', b);
            if (e == std::string::npos) e = line.size();
            f.push_back(line.substr(b, e - b));
            b = e + 1;
        }
        if (f.size() < 9) {
            err = "PAF line " + std::to_string(lineno) + " has fewer than 9 columns";
            return false;
        }
        Candidate c;
        c.read_name = f[0];
        const long long qstart = atoll(f[2].c_str()), qend = atoll(f[3].c_str());
        c.forward = (f[4] == "+");
        c.chromosome = f[5];
        c.start_in_chromosome = atoll(f[7].c_str());
        c.start_of_aligned_region = qstart;
        c.size_of_aligned_region = qend - qstart;
        out.push_back(c);
    }
    return true;
}

bool ends_with(const std::string& s, const char* suf)
{
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

void set_err(char* err, size_t n, const std::string& msg)
{
    if (err && n) {
        snprintf(err, n, "%s", msg.c_str());
    }
}

}  // namespace

struct scrg_job {
    std::string genome;
    std::vector<Chromosome> chroms;
    std::vector<std::string> read_names, read_seqs;
    std::vector<const char*> read_ptrs;
    std::vector<uint64_t> read_lens, cand_offsets, cand_start, cand_chrom, cand_start_in_chrom;
    std::vector<uint8_t> cand_reverse;
    std::vector<uint64_t> pair_read;   // read index of every pair
};

// The entry points below never let a C++ exception cross the C boundary: allocation failures become
// SCRG_ERR_OOM, anything else SCRG_ERR_INVALID_ARG.
template <typename F> static scrg_status guarded(F&& f)
{
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return SCRG_ERR_OOM;
    } catch (...) {
        return SCRG_ERR_INVALID_ARG;
    }
}

extern "C" {

void scrg_job_options_default(scrg_job_options* o)
{
    if (!o) return;
    o->reverse_strand = 0;
    o->sort_by_length = 1;
    o->inflation = 1;
    o->left_extend = 1;
    o->read_length_cap = -1;
}

static scrg_status job_load_impl(const char* genome_fasta, const char* reads_fastq, const char* seeds_path,
                          const scrg_job_options* opt_in, scrg_job** out, char* err, size_t err_len)
{
    if (!genome_fasta || !reads_fastq || !seeds_path || !out) return SCRG_ERR_INVALID_ARG;
    *out = nullptr;
    scrg_job_options opt;
    scrg_job_options_default(&opt);
    if (opt_in) opt = *opt_in;
    std::string raw, msg;
    std::unique_ptr<scrg_job> job_owner(new scrg_job());       // released to the caller on success only
    scrg_job* job = job_owner.get();
    auto fail = [&](scrg_status s, const std::string& m) {
        set_err(err, err_len, m);
        return s;
    };

    if (!slurp(genome_fasta, raw, msg)) return fail(SCRG_ERR_IO, msg);
    parse_fasta(raw, job->genome, job->chroms);
    std::map<std::string, size_t> chrom_idx;
    // seeds name chromosomes by the full header line (as the reference's map does) or, as PAF
    // writers do, by its first word
    for (size_t k = 0; k < job->chroms.size(); k++) {
        const std::string& nm = job->chroms[k].name;
        chrom_idx.emplace(nm.substr(0, nm.find_first_of(" \t")), k);
    }
    for (size_t k = 0; k < job->chroms.size(); k++) chrom_idx[job->chroms[k].name] = k;

    std::vector<FastqRead> reads;
    if (!slurp(reads_fastq, raw, msg)) return fail(SCRG_ERR_IO, msg);
    parse_fastq(raw, reads);

    std::vector<Candidate> cands;
    if (!slurp(seeds_path, raw, msg)) return fail(SCRG_ERR_IO, msg);
    const std::string sp(seeds_path);
    if (ends_with(sp, ".paf")) {
        if (!parse_paf(raw, cands, msg)) return fail(SCRG_ERR_FORMAT, msg);
    } else if (ends_with(sp, ".maf")) {
        parse_maf(raw, cands);
    } else {
        return fail(SCRG_ERR_FORMAT, "unknown seed file ending (want .maf or .paf)");   // src/util.cpp:312-314
    }

    std::map<std::string, size_t> read_idx;
    for (size_t k = 0; k < reads.size(); k++) read_idx[reads[k].name] = k;
    std::vector<std::vector<size_t>> per_read(reads.size());
    for (size_t k = 0; k < cands.size(); k++) {
        auto it = read_idx.find(cands[k].read_name);
        if (it == read_idx.end())   // src/util.cpp:326-329 exits here
            return fail(SCRG_ERR_FORMAT, "candidate location specified unknown read \"" + cands[k].read_name + "\"");
        per_read[it->second].push_back(k);
    }

    // read order: optional inflation, then longest first (stable), as src/tests.cu:366-377
    std::vector<size_t> order;
    const int infl = opt.inflation > 1 ? opt.inflation : 1;
    for (int rep = 0; rep < infl; rep++)
        for (size_t k = 0; k < reads.size(); k++) order.push_back(k);
    auto capped_len = [&](size_t k) {
        const size_t L = reads[k].seq.size();
        return (opt.read_length_cap >= 0 && (size_t)opt.read_length_cap < L) ? (size_t)opt.read_length_cap : L;
    };
    if (opt.sort_by_length)
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return capped_len(a) > capped_len(b); });

    job->cand_offsets.push_back(0);
    for (size_t k : order) {
        const size_t L = capped_len(k);
        const size_t full_len = reads[k].seq.size();
        job->read_names.push_back(reads[k].name);
        job->read_seqs.push_back(reads[k].seq.substr(0, L));
        job->read_lens.push_back(L);
        for (size_t ci : per_read[k]) {
            const Candidate& c = cands[ci];
            if (!c.forward && !opt.reverse_strand) continue;   // src/tests.cu:346-355
            size_t chrom = 0;
            if (job->chroms.size() > 1) {   // get_global_seeds, src/util.cpp:292-301
                auto it = chrom_idx.find(c.chromosome);
                if (it == chrom_idx.end())
                    return fail(SCRG_ERR_FORMAT, "candidate location names unknown chromosome \"" + c.chromosome + "\"");
                chrom = it->second;
            } else if (job->chroms.empty()) {
                return fail(SCRG_ERR_FORMAT, "genome FASTA holds no sequence");
            }
            long long start = c.start_in_chromosome;
            if (opt.left_extend) {
                // left_extend_locations (src/util.cpp:284-290): move the start back by the unaligned
                // read prefix.  On the reverse strand the prefix of the reverse-complemented read is
                // the unaligned SUFFIX of the original read.
                long long lead = c.forward ? c.start_of_aligned_region
                                           : (long long)full_len - (c.start_of_aligned_region + c.size_of_aligned_region);
                if (lead < 0) lead = 0;
                start = std::max(0ll, start - lead);
            }
            uint64_t s = (uint64_t)std::max(0ll, start);
            if (s > job->chroms[chrom].len) s = job->chroms[chrom].len;
            job->cand_chrom.push_back(chrom);
            job->cand_start_in_chrom.push_back(s);
            job->cand_start.push_back(job->chroms[chrom].start + s);
            job->cand_reverse.push_back(c.forward ? 0 : 1);
            job->pair_read.push_back(job->read_names.size() - 1);
        }
        job->cand_offsets.push_back(job->cand_start.size());
    }
    for (const std::string& s : job->read_seqs) job->read_ptrs.push_back(s.data());
    *out = job_owner.release();
    return SCRG_OK;
}

scrg_status scrg_job_load(const char* genome_fasta, const char* reads_fastq, const char* seeds_path,
                          const scrg_job_options* opt_in, scrg_job** out, char* err, size_t err_len)
{
    const scrg_status st = guarded([&] { return job_load_impl(genome_fasta, reads_fastq, seeds_path, opt_in, out, err, err_len); });
    if (st == SCRG_ERR_OOM) set_err(err, err_len, "out of memory while loading the job");
    return st;
}

void scrg_job_free(scrg_job* job) { delete job; }

void scrg_job_counts(const scrg_job* job, uint64_t* n_reads, uint64_t* n_pairs, uint64_t* genome_len,
                     uint64_t* n_chromosomes)
{
    if (!job) return;
    if (n_reads) *n_reads = job->read_lens.size();
    if (n_pairs) *n_pairs = job->cand_start.size();
    if (genome_len) *genome_len = job->genome.size();
    if (n_chromosomes) *n_chromosomes = job->chroms.size();
}

void scrg_job_arrays(const scrg_job* job, const char** genome, const char* const** reads, const uint64_t** read_lens,
                     const uint64_t** cand_offsets, const uint64_t** cand_start, const uint8_t** cand_reverse)
{
    if (!job) return;
    if (genome) *genome = job->genome.data();
    if (reads) *reads = job->read_ptrs.data();
    if (read_lens) *read_lens = job->read_lens.data();
    if (cand_offsets) *cand_offsets = job->cand_offsets.data();
    if (cand_start) *cand_start = job->cand_start.data();
    if (cand_reverse) *cand_reverse = job->cand_reverse.data();
}

const char* scrg_job_read_name(const scrg_job* job, uint64_t read)
{
    return (job && read < job->read_names.size()) ? job->read_names[read].c_str() : "";
}

const char* scrg_job_pair_chromosome(const scrg_job* job, uint64_t pair, uint64_t* start_in_chromosome,
                                     uint64_t* chromosome_len)
{
    if (!job || pair >= job->cand_start.size()) return "";
    const Chromosome& c = job->chroms[job->cand_chrom[pair]];
    if (start_in_chromosome) *start_in_chromosome = job->cand_start_in_chrom[pair];
    if (chromosome_len) *chromosome_len = c.len;
    return c.name.c_str();
}

scrg_status scrg_job_align(scrg_ctx* ctx, const scrg_params* params, const scrg_job* job, scrg_result** out)
{
    if (!job) return SCRG_ERR_INVALID_ARG;
    return scrg_align_mapping_stranded(ctx, params, job->genome.data(), job->genome.size(), job->read_lens.size(),
                                       job->read_ptrs.data(), job->read_lens.data(), job->cand_offsets.data(),
                                       job->cand_start.data(), job->cand_reverse.data(), out);
}

static scrg_status job_write_impl(const scrg_job* job, const scrg_result* res, const char* path, int format);
scrg_status scrg_job_write(const scrg_job* job, const scrg_result* res, const char* path, int format)
{
    return guarded([&] { return job_write_impl(job, res, path, format); });
}
static scrg_status job_write_impl(const scrg_job* job, const scrg_result* res, const char* path, int format)
{
    if (!job || !res || !path) return SCRG_ERR_INVALID_ARG;
    if (res->n_pairs != job->cand_start.size()) return SCRG_ERR_INVALID_ARG;
    FILE* f = fopen(path, "w");
    if (!f) return SCRG_ERR_IO;
    if (format == 1) {
        fprintf(f, "@HD\tVN:1.6\tSO:unknown\n");
        for (const Chromosome& c : job->chroms) {
            std::string nm = c.name.substr(0, c.name.find_first_of(" \t"));
            fprintf(f, "@SQ\tSN:%s\tLN:%llu\n", nm.c_str(), (unsigned long long)c.len);
        }
        fprintf(f, "@PG\tID:scrooge_amd\tPN:scrooge_amd\n");
    }
    for (uint64_t k = 0; k < res->n_pairs; k++) {
        const uint64_t r = job->pair_read[k];
        const Chromosome& c = job->chroms[job->cand_chrom[k]];
        const std::string chrom = c.name.substr(0, c.name.find_first_of(" \t"));
        const char* cigar = res->cigar_text + res->cigar_offset[k];
        // text bases consumed, matches and alignment columns from the runs
        uint64_t tcons = 0, matches = 0, cols = 0;
        for (uint64_t q = res->run_offset[k]; q < res->run_offset[k + 1]; q++) {
            const unsigned n = res->runs[q].count;
            const char op = res->runs[q].op;
            if (op != 'I') tcons += n;
            if (op == '=') matches += n;
            cols += n;
        }
        const uint64_t ts = job->cand_start_in_chrom[k];
        const bool rev = job->cand_reverse[k] != 0;
        if (format == 1) {
            // SAM uses M/I/D/=/X; '=' and 'X' are valid SAM operators, so the CIGAR is kept verbatim
            const std::string& seq = job->read_seqs[r];
            std::string s = seq;
            if (rev) {
                std::reverse(s.begin(), s.end());
                for (char& ch : s) {
                    switch (ch) {
                    case 'A': ch = 'T'; break; case 'C': ch = 'G'; break; case 'G': ch = 'C'; break; case 'T': ch = 'A'; break;
                    case 'a': ch = 't'; break; case 'c': ch = 'g'; break; case 'g': ch = 'c'; break; case 't': ch = 'a'; break;
                    default: break;
                    }
                }
            }
            fprintf(f, "%s\t%d\t%s\t%llu\t255\t%s\t*\t0\t0\t%s\t*\tNM:i:%lld\n", job->read_names[r].c_str(), rev ? 16 : 0,
                    chrom.c_str(), (unsigned long long)(ts + 1), *cigar ? cigar : "*", s.empty() ? "*" : s.c_str(),
                    (long long)res->edit_distance[k]);
        } else {
            fprintf(f, "%s\t%llu\t0\t%llu\t%c\t%s\t%llu\t%llu\t%llu\t%llu\t%llu\t255\tNM:i:%lld\tcg:Z:%s\n",
                    job->read_names[r].c_str(), (unsigned long long)job->read_lens[r],
                    (unsigned long long)job->read_lens[r], rev ? '-' : '+', chrom.c_str(), (unsigned long long)c.len,
                    (unsigned long long)ts, (unsigned long long)(ts + tcons), (unsigned long long)matches,
                    (unsigned long long)cols, (long long)res->edit_distance[k], cigar);
        }
    }
    fclose(f);
    return SCRG_OK;
}

scrg_status scrg_affine_score(const char* cigar, int64_t match_bonus, int64_t mismatch_cost, int64_t gap_open_cost,
                              int64_t gap_extend_cost, int64_t* score)
{
    if (!cigar || !score) return SCRG_ERR_INVALID_ARG;
    int64_t s = 0;
    bool in_gap = false;
    const char* p = cigar;
    while (*p) {
        if (*p < '0' || *p > '9') return SCRG_ERR_FORMAT;
        int64_t n = 0;
        while (*p >= '0' && *p <= '9') n = n * 10 + (*p++ - '0');
        const char op = *p++;
        if (op == '=') {
            s += n * match_bonus;
            in_gap = false;
        } else if (op == 'X') {
            s -= n * mismatch_cost;
            in_gap = false;
        } else if (op == 'I' || op == 'D') {
            if (!in_gap) s -= gap_open_cost;
            s -= n * gap_extend_cost;
            in_gap = true;
        } else {
            return SCRG_ERR_FORMAT;
        }
    }
    *score = s;
    return SCRG_OK;
}

scrg_status scrg_validate_alignment(const char* text, uint64_t text_len, const char* read, uint64_t read_len,
                                    const char* cigar, int64_t edit_distance, int32_t* why)
{
    if (!cigar || (text_len && !text) || (read_len && !read)) return SCRG_ERR_INVALID_ARG;
    auto bad = [&](int32_t code) {
        if (why) *why = code;
        return (scrg_status)SCRG_ERR_FORMAT;
    };
    auto up = [](char c) { return (char)((c >= 'a' && c <= 'z') ? c - 32 : c); };
    uint64_t i = 0, j = 0;
    int64_t edits = 0;
    const char* p = cigar;
    while (*p) {
        if (*p < '0' || *p > '9') return bad(1);
        uint64_t n = 0;
        while (*p >= '0' && *p <= '9') n = n * 10 + (uint64_t)(*p++ - '0');
        const char op = *p++;
        if (n == 0) return bad(2);
        if (op == '=' || op == 'X') {
            if (j + n > read_len) return bad(3);
            if (i + n > text_len) return bad(4);
            for (uint64_t k = 0; k < n; k++)
                if ((up(text[i + k]) == up(read[j + k])) != (op == '=')) return bad(5);
            i += n;
            j += n;
        } else if (op == 'I') {
            if (j + n > read_len) return bad(3);
            j += n;
        } else if (op == 'D') {
            if (i + n > text_len) return bad(4);
            i += n;
        } else {
            return bad(1);
        }
        if (op != '=') edits += (int64_t)n;
    }
    if (j != read_len) return bad(3);
    if (edits != edit_distance) return bad(6);
    if (why) *why = 0;
    return SCRG_OK;
}

}  // extern "C"
