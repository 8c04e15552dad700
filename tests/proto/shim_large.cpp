// shim_large.cpp — the C++ shim's conversion of a LARGE result (more than 8 MB of CIGAR text: filled by several threads,
// include/scrooge_amd.hpp detail::to_alignments) against the C ABI's own arrays, pair by pair.  Built and run by
// tests/test_cpp_shim.py::test_shim_converts_large_results_in_parallel.
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "scrooge_amd.hpp"

int main()
{
    if (scrg_device_count() <= 0) { fprintf(stderr, "no usable HIP device\n"); return 2; }
    const size_t n = 3000, distinct = 300;
    std::mt19937_64 g(5);
    std::vector<std::string> texts(n), reads(n);
    for (size_t i = 0; i < distinct; i++) {
        const size_t L = 2000 + (size_t)(g() % 12000);            // ragged lengths, not sorted
        std::string t(L + L * 15 / 100, 'A');
        for (char& c : t) c = "ACGT"[g() & 3];
        std::string q;
        for (size_t k = 0; q.size() < L && k < t.size(); k++) {
            const unsigned u = (unsigned)(g() % 100);
            if (u < 3) q.push_back("ACGT"[g() & 3]);
            else if (u < 6) { q.push_back("ACGT"[g() & 3]); k--; }
            else if (u < 10) continue;
            else q.push_back(t[k]);
        }
        texts[i] = t;
        reads[i] = q;
    }
    for (size_t i = distinct; i < n; i++) { texts[i] = texts[(i * 7) % distinct]; reads[i] = reads[(i * 7) % distinct]; }
    reads[17].clear();                                            // an empty read: an empty CIGAR in the middle
    scrooge_amd::Handle h(0);
    const std::vector<Alignment_t> a = h.align_all(texts, reads);
    std::vector<const char*> tp(n), qp(n);
    std::vector<uint64_t> tl(n), ql(n);
    for (size_t i = 0; i < n; i++) { tp[i] = texts[i].data(); tl[i] = texts[i].size(); qp[i] = reads[i].data(); ql[i] = reads[i].size(); }
    scrg_result* r = nullptr;
    if (scrg_align_pairs(h.ctx(), &h.params(), n, tp.data(), tl.data(), qp.data(), ql.data(), &r) != SCRG_OK) return 1;
    size_t bad = a.size() == n ? 0 : 1, bytes = 0;
    for (size_t i = 0; i < n && i < a.size(); i++) {
        const size_t len = (size_t)(r->cigar_offset[i + 1] - r->cigar_offset[i] - 1);
        bytes += len;
        if (a[i].cigar.size() != len || memcmp(a[i].cigar.data(), r->cigar_text + r->cigar_offset[i], len) != 0 ||
            a[i].edit_distance != (long long)r->edit_distance[i])
            bad++;
    }
    scrg_result_free(r);
    printf("pairs=%zu text_mb=%zu mismatches=%zu\n", n, bytes >> 20, bad);
    return bad ? 1 : 0;
}
