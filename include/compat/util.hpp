// compat/util.hpp — the data types of the reference's src/util.hpp:11-55 for callers that are compiled OUTSIDE the
// reference tree (inside it, the reference's own util.hpp is found first and this file is not used: scrooge_amd.hpp
// keys on SEED_FILE_MAF and does not declare the types twice).  Types and the measure_ns timer only; the file readers
// of src/util.cpp have their counterpart behind the C ABI (include/scrooge_amd_io.h: scrg_job_load).
#pragma once

#include <chrono>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "scrooge_amd.hpp"      // Genome_t, CandidateLocation_t, Read_t, Alignment_t, CigarEntry_t

#ifndef SEED_FILE_MAF
#define SEED_FILE_MAF 0
#define SEED_FILE_PAF 1

typedef struct Sequence {
    std::string description;
    std::string content;
} Sequence_t;

// wall time of a callable in nanoseconds (src/util.hpp:48-55)
template <typename Callable> long long measure_ns(Callable target)
{
    const auto t0 = std::chrono::high_resolution_clock::now();
    target();
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::high_resolution_clock::now() - t0).count();
}
#endif
