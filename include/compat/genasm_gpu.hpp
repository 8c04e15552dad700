// compat/genasm_gpu.hpp — a header of the reference's name (src/genasm_gpu.hpp:5-10) in front of this library, so
// that a caller written against Scrooge's GPU interface compiles UNCHANGED with a plain C++ compiler:
//
//     g++ -std=c++17 -I<repo>/include/compat -I<repo>/include caller.cpp -L<repo>/scrooge_amd -lscrooge_amd
//
//   extern bool enabled_algorithm_log            -> an assignable switch object (scrooge_amd::LogSwitch)
//   align_all(Genome_t&, vector<Read_t>&, ns)    -> scrg_align_mapping   (every visible device, scrooge_amd.hpp)
//   align_all(texts, queries, ns)                -> scrg_align_pairs
//   __global__ ascii_to_twobit_strings           -> under hipcc: the kernel itself, with the reference's signature and launch
//                                                   shape (src/genasm_gpu.hpp:9, launched by src/tests.cu:626), made of the same
//                                                   device function as the library's packer (scrooge_amd_device.hpp); a host
//                                                   compiler cannot launch a kernel: there the same packing is
//                                                   scrg_ascii_to_twobit() in scrooge_amd.h
// Exactly the reference's two overloads are declared here (no `threads` variants), so that calls such as
// align_all(texts, queries, NULL) resolve as they do against the reference.
#pragma once

#include <util.hpp>      // by include path, not relative to this file: inside the reference tree this is the reference's own util.hpp
#include "scrooge_amd.hpp"
#if defined(__HIPCC__)
#include <cassert>
#include "scrooge_amd_device.hpp"
#endif

namespace genasm_gpu {
using scrooge_amd::enabled_algorithm_log;

#if defined(__HIPCC__)
// The reference's third export (src/genasm_gpu.hpp:9, src/genasm_gpu.cu:680-685): workgroup b packs strings b, b + gridDim.x, …,
// its threads the bytes of a string side by side; twobit_strings[i] receives ceil(string_lengths[i] / 4) bytes, 4 bases per
// byte, first base in bits 7..6, the last byte zero-padded (:640-669).  An invalid base asserts, as the reference does (:636).
// `static`: the header may be included by several translation units of a caller (a kernel cannot be `inline`); each
// compiles its own copy of these few instructions.
static __global__ void ascii_to_twobit_strings(int count, long long* string_lengths, char** ascii_strings, char** twobit_strings)
{
    for (int i = (int)blockIdx.x; i < count; i += (int)gridDim.x) {
        const uint32_t bad = scrooge_amd::device::twobit_string(string_lengths[i], ascii_strings[i], twobit_strings[i]);
        assert(bad == 0 && "ascii_to_twobit_strings: invalid character");
        (void)bad;
    }
}
#endif

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
