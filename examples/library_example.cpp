// Usage example for the C++ shim (include/scrooge_amd.hpp); covers the same four call
// shapes as the reference's src/library_example.cu:11-88 (pairwise and read-mapping
// interface, with and without the kernel-time out-parameter).
//
// Build (one line):
//   g++ -std=c++17 -Iinclude examples/library_example.cpp -Lscrooge_amd -lscrooge_amd
//       -Wl,-rpath,$PWD/scrooge_amd -o /tmp/library_example
#include <cstdio>
#include <iostream>

#include "scrooge_amd.hpp"

static void show(const char* what, const std::vector<Alignment_t>& alns, long long ns = -1)
{
    for (const Alignment_t& a : alns)
        std::cout << what << " cigar=" << a.cigar << " edit_distance=" << a.edit_distance << "\n";
    if (ns >= 0) std::cout << what << " kernel_ns>0=" << (ns > 0) << "\n";
}

int main()
{
    scrooge_amd::enabled_algorithm_log = false;
    try {
        // pairwise interface: queries[i] against texts[i]
        std::vector<std::string> texts = {"ACGTACGT", "AAAACCCCGGGGTTTT"};
        std::vector<std::string> queries = {"ACGTACG", "AAAAGGGGAAAATTTT"};
        show("pairwise", scrooge_amd::align_all(texts, queries));
        long long ns = 0;
        std::vector<Alignment_t> timed = scrooge_amd::align_all(texts, queries, &ns);
        show("pairwise_timed", timed, ns);

        // read-mapping interface: reads with candidate locations in one reference
        Genome_t genome;
        genome.content = "TTTTACGTACGTTTTTAAAACCCCGGGGTTTT";
        Read_t r1;
        r1.content = "ACGTACG";
        CandidateLocation_t loc{};
        loc.start_in_reference = 4;
        loc.strand = true;
        r1.locations.push_back(loc);
        loc.start_in_reference = 0;
        r1.locations.push_back(loc);
        Read_t r2;
        r2.content = "AAAACCCCGGGG";
        loc.start_in_reference = 16;
        r2.locations.push_back(loc);
        std::vector<Read_t> reads = {r1, r2};
        show("mapping", scrooge_amd::align_all(genome, reads));
        ns = 0;
        timed = scrooge_amd::align_all(genome, reads, &ns);
        show("mapping_timed", timed, ns);

        // many batches against one reference: the genome is staged and packed once
        scrooge_amd::Handle& h = scrooge_amd::default_handle();
        h.set_genome(genome);
        show("resident", h.align_all(reads));
        std::vector<Read_t> second = {r2};
        show("resident", h.align_all(second));
        h.clear_genome();
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 2;
    }
    return 0;
}
