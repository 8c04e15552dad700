// compat/genasm_cpu.hpp — the reference's CPU interface (src/genasm_cpu.hpp:4-8) forwarded to the SAME GPU path:
// `threads` is accepted and ignored, results are bit-identical to src/genasm_cpu.cpp by construction (that file is
// the oracle of this repository).  One observable difference: the pairwise overload returns all N alignments; the
// reference's CPU overload drops every odd one (src/genasm_cpu.cpp:600-605 increments the pair index twice).
// genasm_cpu::enabled_algorithm_log and genasm_gpu::enabled_algorithm_log are the same switch here.
#pragma once

#include <util.hpp>      // by include path, not relative to this file: inside the reference tree this is the reference's own util.hpp
#include "scrooge_amd.hpp"

namespace genasm_cpu {
using scrooge_amd::enabled_algorithm_log;

inline std::vector<Alignment_t> align_all(Genome_t& reference, std::vector<Read_t>& reads, int threads = 1,
                                          long long* core_algorithm_ns = NULL)
{
    (void)threads;
    return scrooge_amd::align_all(reference, reads, core_algorithm_ns);
}
inline std::vector<Alignment_t> align_all(std::vector<std::string>& texts, std::vector<std::string>& queries, int threads,
                                          long long* core_algorithm_ns = NULL)
{
    (void)threads;
    return scrooge_amd::align_all(texts, queries, core_algorithm_ns);
}
}  // namespace genasm_cpu
