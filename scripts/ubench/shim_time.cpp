// shim_time.cpp — wall time of the reference-shaped call itself, scrooge_amd::align_all(texts, queries) -> vector<Alignment_t>
// (include/scrooge_amd.hpp), next to the library call underneath it (scrg_align_pairs, text output): what the conversion of the
// result into one std::string per CIGAR costs.
// build: g++ -O2 -std=c++17 -Iinclude scripts/ubench/shim_time.cpp -Lscrooge_amd -lscrooge_amd -Wl,-rpath,$PWD/scrooge_amd -pthread -o scripts/ubench/shim_time
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "scrooge_amd.hpp"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 20000, L = argc > 2 ? (size_t)atol(argv[2]) : 10000;
    std::mt19937_64 g(1);
    std::vector<std::string> texts(n), reads(n);
    const size_t distinct = std::min<size_t>(n, 512);
    for (size_t i = 0; i < distinct; i++) {
        std::string t(L + L * 15 / 100, 'A');
        for (char& c : t) c = "ACGT"[g() & 3];
        std::string q;                                     // ~10 % edits: substitutions, insertions, deletions
        for (size_t k = 0; q.size() < L && k < t.size(); k++) {
            const unsigned u = (unsigned)(g() % 100);
            if (u < 3) q.push_back("ACGT"[g() & 3]);                  // substitution (or a match by chance)
            else if (u < 6) { q.push_back("ACGT"[g() & 3]); k--; }    // insertion
            else if (u < 10) continue;                                // deletion
            else q.push_back(t[k]);
        }
        texts[i] = t;
        reads[i] = q;
    }
    for (size_t i = distinct; i < n; i++) { texts[i] = texts[i % distinct]; reads[i] = reads[i % distinct]; }
    scrooge_amd::Handle h(0);
    for (int rep = 0; rep < 4; rep++) {
        double t0 = now();
        std::vector<Alignment_t> a = h.align_all(texts, reads);
        const double t_shim = now() - t0;
        size_t bytes = 0;
        for (const Alignment_t& x : a) bytes += x.cigar.size();
        // the library call alone, same input
        std::vector<const char*> tp(n), qp(n);
        std::vector<uint64_t> tl(n), ql(n);
        for (size_t i = 0; i < n; i++) { tp[i] = texts[i].data(); tl[i] = texts[i].size(); qp[i] = reads[i].data(); ql[i] = reads[i].size(); }
        scrg_result* r = nullptr;
        t0 = now();
        if (scrg_align_pairs(h.ctx(), &h.params(), n, tp.data(), tl.data(), qp.data(), ql.data(), &r) != SCRG_OK) return 1;
        const double t_lib = now() - t0;
        t0 = now();
        { std::vector<Alignment_t> tmp = std::move(a); }               // what the caller pays to let go of the strings
        const double t_free = now() - t0;
        scrg_result_free(r);
        printf("call %d: %zu pairs x %zu: align_all %.2f ms  (scrg_align_pairs alone %.2f ms; %.1f MB of CIGAR text; destroying the vector %.2f ms)\n",
               rep, n, L, t_shim * 1e3, t_lib * 1e3, bytes / 1048576.0, t_free * 1e3);
    }
    return 0;
}
