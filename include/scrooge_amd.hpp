// scrooge_amd.hpp — header-only C++ shim that rebuilds the reference's library
// surface (src/genasm_gpu.hpp:5-10, data types src/util.hpp:11-46) on top of the
// C ABI in scrooge_amd.h, so code written against Scrooge's GPU interface
// compiles against this library by swapping the include and the namespace:
//
//     #include "genasm_gpu.hpp"                 ->  #include "scrooge_amd.hpp"
//     genasm_gpu::align_all(genome, reads)      ->  scrooge_amd::align_all(genome, reads)
//     genasm_gpu::align_all(texts, queries)     ->  scrooge_amd::align_all(texts, queries)
//     genasm_gpu::enabled_algorithm_log = false ->  scrooge_amd::enabled_algorithm_log = false
//
// Callers that must compile UNCHANGED include the headers of the reference's own names instead:
// include/compat/genasm_gpu.hpp and include/compat/genasm_cpu.hpp declare `namespace genasm_gpu` / `genasm_cpu`
// with exactly the reference's overloads on top of this file (tests/test_reference_callers.py compiles the
// reference's src/library_example.cu against them as it is).
//
// Semantics kept from the reference: result k belongs to the k-th (read,
// location) in nested order / the k-th string pair; texts[i] is the target,
// queries[i] is consumed completely; only Genome_t::content, Read_t::content and
// CandidateLocation_t::start_in_reference are read (src/genasm_cpu.cpp:508-517);
// *core_algorithm_ns is the align kernel's device time.  Differences: errors
// throw std::runtime_error instead of exit()/assert, and the pairwise overload
// returns all N results (the CPU overload drops odd ones, genasm_cpu.cpp:600-605).
#pragma once

#include <algorithm>
#include <cstdint>
#include <exception>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "scrooge_amd.h"

// The reference declares these types at global scope in src/util.hpp.  They are
// re-declared here (same member names and order) only when util.hpp has not been
// included, so a translation unit may include both headers.
#ifndef SCROOGE_AMD_NO_REFERENCE_TYPES
#if !defined(SEED_FILE_MAF)   // util.hpp's first macro: its types are already visible if set
typedef struct Genome {
    std::map<std::string, long long> chromosome_starts;
    std::string content;
} Genome_t;

typedef struct CandidateLocation {
    std::string read_description;
    std::string chromosome;
    long long start_in_chromosome;
    long long start_in_reference;
    long long start_of_aligned_region;
    long long size_of_aligned_region;
    bool strand;
} CandidateLocation_t;

typedef struct Read {
    std::string description;
    std::string content;
    std::vector<CandidateLocation_t> locations;
} Read_t;

typedef struct Alignment {
    std::string cigar;
    long long edit_distance;
} Alignment_t;

typedef struct CigarEntry {
    uint8_t edit_count;
    char edit_type;
} CigarEntry_t;
#endif
#endif

namespace scrooge_amd {

namespace detail {
// scrg_result -> the reference's result type, one std::string per CIGAR (src/util.hpp:38-41).  For a batch of long reads this is
// hundreds of MB of small allocations and copies — more host time than the library call itself when done on one thread — so
// the strings of a large result are filled by several threads, each a range of pairs of about equal bytes (results often come
// longest first).
inline std::vector<Alignment_t> to_alignments(const scrg_result* r)
{
    const uint64_t n = r->n_pairs;
    std::vector<Alignment_t> out(n);
    auto fill = [&](uint64_t a, uint64_t e) {
        for (uint64_t i = a; i < e; i++) {
            const char* b = r->cigar_text + r->cigar_offset[i];
            out[i].cigar.assign(b, (size_t)(r->cigar_offset[i + 1] - r->cigar_offset[i] - 1));
            out[i].edit_distance = (long long)r->edit_distance[i];
        }
    };
    const uint64_t PER_PAIR = 64;                                            // what a pair costs besides its bytes
    const uint64_t work = n ? r->cigar_offset[n] + PER_PAIR * n : 0;
    unsigned nt = std::thread::hardware_concurrency();
    nt = std::min(nt ? nt : 1u, 16u);
    if (work < (8u << 20) || nt <= 1) {
        fill(0, n);
        return out;
    }
    // The strings are ALLOCATED here, on the calling thread, and only filled by the others: memory from another thread's
    // malloc arena is slow to give back for the thread that later destroys the vector (glibc: 50 ms instead of 13 for 100 k
    // CIGARs of 4 kB), and it is the caller who does.
    for (uint64_t i = 0; i < n; i++) out[i].cigar.reserve((size_t)(r->cigar_offset[i + 1] - r->cigar_offset[i] - 1));
    std::vector<uint64_t> cut(nt + 1, n);
    cut[0] = 0;
    for (unsigned k = 1; k < nt; k++) {                                      // first pair whose start is past k / nt of the work
        const uint64_t target = work / nt * k;
        uint64_t lo = cut[k - 1], hi = n;
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2;
            if (r->cigar_offset[mid] + PER_PAIR * mid < target) lo = mid + 1;
            else hi = mid;
        }
        cut[k] = lo;
    }
    std::vector<std::exception_ptr> err(nt);
    std::vector<std::thread> th;
    auto guarded = [&](unsigned k) {
        try {
            fill(cut[k], cut[k + 1]);
        } catch (...) {
            err[k] = std::current_exception();
        }
    };
    for (unsigned k = 1; k < nt; k++) {
        try {
            th.emplace_back(guarded, k);
        } catch (...) {                                                      // no more threads: this one does the range itself
            guarded(k);
        }
    }
    guarded(0);
    for (std::thread& t : th) t.join();
    for (const std::exception_ptr& e : err)
        if (e) std::rethrow_exception(e);
    return out;
}
}  // namespace detail

// One GPU, explicitly: a handle per (thread, device), for callers that place work themselves or keep a genome resident.
// (The free functions below, the reference's surface, use every visible GPU.)
namespace detail {
// the loaded library must speak the interface this header declares (scrg_abi_version, scrooge_amd.h)
inline void check_abi()
{
    if (scrg_abi_version() != SCRG_ABI_VERSION)
        throw std::runtime_error("scrooge_amd: libscrooge_amd.so has interface version " + std::to_string(scrg_abi_version()) +
                                 ", this program was compiled against version " + std::to_string(SCRG_ABI_VERSION));
}
}  // namespace detail

class Handle {
public:
    explicit Handle(int device = 0)
    {
        detail::check_abi();
        scrg_status s = scrg_ctx_create(device, &ctx_);
        if (s != SCRG_OK) throw std::runtime_error(std::string("scrooge_amd: ") + scrg_status_string(s));
        scrg_params_default(&params_);
        params_.outputs = SCRG_OUT_TEXT;      // Alignment_t carries the CIGAR as text: the runs need not cross PCIe
    }
    ~Handle() { scrg_ctx_destroy(ctx_); }
    Handle(const Handle&) = delete;
    Handle& operator=(const Handle&) = delete;

    scrg_ctx* ctx() const { return ctx_; }
    scrg_params& params() { return params_; }

    std::vector<Alignment_t> align_all(std::vector<std::string>& texts, std::vector<std::string>& queries,
                                       long long* core_algorithm_ns = nullptr)
    {
        if (texts.size() != queries.size())   // reference: assert, genasm_cpu.cpp:559
            throw std::invalid_argument("scrooge_amd::align_all: texts and queries differ in size");
        const size_t n = texts.size();
        std::vector<const char*> tp(n), qp(n);
        std::vector<uint64_t> tl(n), ql(n);
        for (size_t i = 0; i < n; i++) {
            tp[i] = texts[i].data();
            tl[i] = texts[i].size();
            qp[i] = queries[i].data();
            ql[i] = queries[i].size();
        }
        scrg_result* r = nullptr;
        scrg_status s = scrg_align_pairs(ctx_, &params_, n, tp.data(), tl.data(), qp.data(), ql.data(), &r);
        return collect(s, r, core_algorithm_ns);
    }

    std::vector<Alignment_t> align_all(Genome_t& reference, std::vector<Read_t>& reads,
                                       long long* core_algorithm_ns = nullptr)
    {
        const size_t nr = reads.size();
        std::vector<const char*> rp(nr);
        std::vector<uint64_t> rl(nr), off(nr + 1, 0), starts;
        for (size_t r = 0; r < nr; r++) {
            rp[r] = reads[r].content.data();
            rl[r] = reads[r].content.size();
            for (const CandidateLocation_t& loc : reads[r].locations) {
                if (loc.start_in_reference < 0)
                    throw std::invalid_argument("scrooge_amd::align_all: negative start_in_reference");
                starts.push_back((uint64_t)loc.start_in_reference);
            }
            off[r + 1] = starts.size();
        }
        scrg_result* res = nullptr;
        scrg_status s = scrg_align_mapping(ctx_, &params_, reference.content.data(), reference.content.size(), nr,
                                           rp.data(), rl.data(), off.data(), starts.data(), &res);
        return collect(s, res, core_algorithm_ns);
    }

    // Many read batches against one reference: set_genome() stages and packs it once and keeps it in HBM,
    // align_all(reads) then aligns batches against it without touching it again (scrg_genome_set /
    // scrg_align_mapping_resident; the two-argument overload above re-stages the genome on every call, as the
    // reference re-converts it, src/genasm_cpu.cpp:508).
    void set_genome(const Genome_t& reference)
    {
        scrg_status s = scrg_genome_set(ctx_, reference.content.data(), reference.content.size());
        if (s != SCRG_OK)
            throw std::runtime_error(std::string("scrooge_amd: ") + scrg_status_string(s) + " (" + scrg_last_error(ctx_) + ")");
    }
    void clear_genome() { scrg_genome_clear(ctx_); }

    std::vector<Alignment_t> align_all(std::vector<Read_t>& reads, long long* core_algorithm_ns = nullptr)
    {
        const size_t nr = reads.size();
        std::vector<const char*> rp(nr);
        std::vector<uint64_t> rl(nr), off(nr + 1, 0), starts;
        for (size_t r = 0; r < nr; r++) {
            rp[r] = reads[r].content.data();
            rl[r] = reads[r].content.size();
            for (const CandidateLocation_t& loc : reads[r].locations) {
                if (loc.start_in_reference < 0)
                    throw std::invalid_argument("scrooge_amd::align_all: negative start_in_reference");
                starts.push_back((uint64_t)loc.start_in_reference);
            }
            off[r + 1] = starts.size();
        }
        scrg_result* res = nullptr;
        scrg_status s = scrg_align_mapping_resident(ctx_, &params_, nr, rp.data(), rl.data(), off.data(), starts.data(),
                                                    nullptr, &res);
        return collect(s, res, core_algorithm_ns);
    }

private:
    std::vector<Alignment_t> collect(scrg_status s, scrg_result* r, long long* ns)
    {
        if (s != SCRG_OK && s != SCRG_ERR_CIGAR_OVERFLOW) {
            std::string msg = std::string("scrooge_amd: ") + scrg_status_string(s) + " (" + scrg_last_error(ctx_) + ")";
            scrg_result_free(r);
            throw std::runtime_error(msg);
        }
        std::vector<Alignment_t> out;
        try {
            out = detail::to_alignments(r);
        } catch (...) {
            scrg_result_free(r);
            throw;
        }
        if (ns) *ns = (long long)r->kernel_ns;
        const bool overflowed = (s == SCRG_ERR_CIGAR_OVERFLOW);
        scrg_result_free(r);
        if (overflowed) throw std::runtime_error("scrooge_amd: a pair overflowed its CIGAR slice");
        return out;
    }

    scrg_ctx* ctx_ = nullptr;
    scrg_params params_;
};

inline Handle& default_handle()
{
    thread_local Handle h(0);
    return h;
}

namespace detail {
// the free functions (the reference's surface) use EVERY visible GPU: scrg_align_pairs_multi / scrg_align_mapping_multi
inline std::vector<int32_t> all_devices()
{
    check_abi();
    const int n = scrg_device_count();
    if (n <= 0) throw std::runtime_error(std::string("scrooge_amd: ") + scrg_status_string(SCRG_ERR_NO_DEVICE));
    std::vector<int32_t> d((size_t)n);
    for (int k = 0; k < n; k++) d[(size_t)k] = k;
    return d;
}
inline std::vector<Alignment_t> collect_multi(scrg_status s, scrg_result* r, long long* ns)
{
    if (s != SCRG_OK && s != SCRG_ERR_CIGAR_OVERFLOW) {
        const std::string msg = std::string("scrooge_amd: ") + scrg_status_string(s) + " (" + scrg_multi_last_error() + ")";
        scrg_result_free(r);
        throw std::runtime_error(msg);
    }
    std::vector<Alignment_t> out;
    try {
        out = to_alignments(r);
    } catch (...) {
        scrg_result_free(r);
        throw;
    }
    if (ns) *ns = (long long)r->kernel_ns;
    const bool overflowed = (s == SCRG_ERR_CIGAR_OVERFLOW);
    scrg_result_free(r);
    if (overflowed) throw std::runtime_error("scrooge_amd: a pair overflowed its CIGAR slice");
    return out;
}
inline scrg_params text_params()
{
    scrg_params p;
    scrg_params_default(&p);
    p.outputs = SCRG_OUT_TEXT;             // Alignment_t carries the CIGAR as text: the runs need not cross PCIe
    return p;
}
}  // namespace detail

// The reference exports an assignable `extern bool enabled_algorithm_log` per namespace (src/genasm_gpu.hpp:6,
// src/genasm_cpu.hpp:5) and its callers write `genasm_gpu::enabled_algorithm_log = verbose;`
// (src/library_example.cu:91-92, src/tests.cu:791-792).  The switch itself lives inside the shared library
// (scrg_set_log), so the name here is an object that forwards: assignable from and convertible to bool.
// (The call form enabled_algorithm_log(false) of earlier versions of this header still works.)
struct LogSwitch {
    LogSwitch& operator=(bool on)
    {
        scrg_set_log(on ? 1 : 0);
        return *this;
    }
    void operator()(bool on) const { scrg_set_log(on ? 1 : 0); }
    operator bool() const { return scrg_get_log() != 0; }
    LogSwitch() = default;
    LogSwitch(const LogSwitch&) = delete;
};
inline LogSwitch enabled_algorithm_log;

// src/genasm_gpu.hpp:7 — on every visible GPU
inline std::vector<Alignment_t> align_all(Genome_t& reference, std::vector<Read_t>& reads,
                                          long long* core_algorithm_ns = nullptr)
{
    const size_t nr = reads.size();
    std::vector<const char*> rp(nr);
    std::vector<uint64_t> rl(nr), off(nr + 1, 0), starts;
    for (size_t r = 0; r < nr; r++) {
        rp[r] = reads[r].content.data();
        rl[r] = reads[r].content.size();
        for (const CandidateLocation_t& loc : reads[r].locations) {
            if (loc.start_in_reference < 0)
                throw std::invalid_argument("scrooge_amd::align_all: negative start_in_reference");
            starts.push_back((uint64_t)loc.start_in_reference);
        }
        off[r + 1] = starts.size();
    }
    const std::vector<int32_t> dev = detail::all_devices();
    const scrg_params p = detail::text_params();
    scrg_result* res = nullptr;
    const scrg_status s = scrg_align_mapping_multi(dev.data(), (int32_t)dev.size(), &p, reference.content.data(), reference.content.size(), nr,
                                                   rp.data(), rl.data(), off.data(), starts.data(), nullptr, &res);
    return detail::collect_multi(s, res, core_algorithm_ns);
}

// src/genasm_gpu.hpp:8 — on every visible GPU
inline std::vector<Alignment_t> align_all(std::vector<std::string>& texts, std::vector<std::string>& queries,
                                          long long* core_algorithm_ns = nullptr)
{
    if (texts.size() != queries.size())   // reference: assert, genasm_cpu.cpp:559
        throw std::invalid_argument("scrooge_amd::align_all: texts and queries differ in size");
    const size_t n = texts.size();
    std::vector<const char*> tp(n), qp(n);
    std::vector<uint64_t> tl(n), ql(n);
    for (size_t i = 0; i < n; i++) {
        tp[i] = texts[i].data();
        tl[i] = texts[i].size();
        qp[i] = queries[i].data();
        ql[i] = queries[i].size();
    }
    const std::vector<int32_t> dev = detail::all_devices();
    const scrg_params p = detail::text_params();
    scrg_result* res = nullptr;
    const scrg_status s = scrg_align_pairs_multi(dev.data(), (int32_t)dev.size(), &p, n, tp.data(), tl.data(), qp.data(), ql.data(), &res);
    return detail::collect_multi(s, res, core_algorithm_ns);
}

// Same argument list as the CPU overloads (src/genasm_cpu.hpp:6-7); `threads` is accepted and ignored.
inline std::vector<Alignment_t> align_all(Genome_t& reference, std::vector<Read_t>& reads, int /*threads*/,
                                          long long* core_algorithm_ns)
{
    return align_all(reference, reads, core_algorithm_ns);
}
inline std::vector<Alignment_t> align_all(std::vector<std::string>& texts, std::vector<std::string>& queries,
                                          int /*threads*/, long long* core_algorithm_ns = nullptr)
{
    return align_all(texts, queries, core_algorithm_ns);
}

}  // namespace scrooge_amd

#ifdef SCROOGE_AMD_AS_GENASM_GPU
// (kept for callers of earlier versions of this header; include/compat/genasm_gpu.hpp is the drop-in form: an alias
// also exposes the `threads` overloads, which makes align_all(texts, queries, NULL) ambiguous)
namespace genasm_gpu = scrooge_amd;
#endif
