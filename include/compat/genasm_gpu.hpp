// compat/genasm_gpu.hpp — a header of the reference's name (src/genasm_gpu.hpp:5-10) in front of this library, so
// that a caller written against Scrooge's GPU interface compiles UNCHANGED with a plain C++ compiler:
//
//     g++ -std=c++17 -I<repo>/include/compat -I<repo>/include caller.cpp -L<repo>/scrooge_amd -lscrooge_amd
//
//   extern bool enabled_algorithm_log            -> an assignable switch object (scrooge_amd::LogSwitch)
//   align_all(Genome_t&, vector<Read_t>&, ns)    -> scrg_align_mapping   (every visible device, scrooge_amd.hpp)
//   align_all(texts, queries, ns)                -> scrg_align_pairs
//   __global__ ascii_to_twobit_strings           -> not offered to host compilers (a kernel cannot be launched from
//                                                   g++); the same packing is scrg_ascii_to_twobit() in scrooge_amd.h
// Exactly the reference's two overloads are declared here (no `threads` variants), so that calls such as
// align_all(texts, queries, NULL) resolve as they do against the reference.
#pragma once

#include <util.hpp>      // by include path, not relative to this file: inside the reference tree this is the reference's own util.hpp
#include "scrooge_amd.hpp"

namespace genasm_gpu {
using scrooge_amd::enabled_algorithm_log;

inline std::vector<Alignment_t> align_all(Genome_t& reference, std::vector<Read_t>& reads, long long* core_algorithm_ns = NULL)
{
    return scrooge_amd::align_all(reference, reads, core_algorithm_ns);
}
inline std::vector<Alignment_t> align_all(std::vector<std::string>& texts, std::vector<std::string>& queries,
                                          long long* core_algorithm_ns = NULL)
{
    return scrooge_amd::align_all(texts, queries, core_algorithm_ns);
}
}  // namespace genasm_gpu
