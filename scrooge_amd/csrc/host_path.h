// host_path.h — launch interface of host_path_kernels.hip (the pipelined host-pointer path, scrg_host.cpp).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/scrooge_amd.h"

namespace scrg {

struct HostDescArgs {
    uint64_t n;                   // pairs of the chunk
    scrg_pair_desc* desc;         // out
    const uint32_t* read_len;     // [n]
    const uint32_t* text_len;     // [n]   pairwise
    const uint64_t* start;        // [n]   mapping: start_in_reference; null = pairwise
    const uint32_t* row;          // [n]   mapping: the read row of the pair; null = row i
    uint64_t genome_len;
    uint64_t read_base, read_words;   // first word and words per row of the read region (lane-interleaved groups of 64 rows)
    uint64_t text_base, text_words;   // the same for the texts (pairwise)
    uint64_t cap;                 // runs per slice, a multiple of 16
    uint32_t linear;              // 1: rows are contiguous (word stride 1: the GenASM-row kernels); 0: lane-interleaved groups of 64 rows
};

hipError_t launch_build_desc(const HostDescArgs& a, hipStream_t s);
size_t host_scan_temp_bytes(uint64_t n);
// (wire: 3 n uint32 — edit distance | run count, bit 31 = the slice overflowed | text length — what goes back to the host)
hipError_t launch_result_layout(uint64_t n, const scrg_pair_desc* pairs, const uint16_t* runs, const uint32_t* n_runs, const int64_t* ed,
                                const uint32_t* status, uint64_t* cnt64, uint64_t* len64, uint64_t* run_off, uint64_t* text_off,
                                uint64_t* totals, uint32_t* wire, void* temp, size_t temp_bytes, int want_text, int n_cus, hipStream_t s);
hipError_t launch_render_text(uint64_t n, const uint16_t* dense, const uint64_t* run_off, const uint64_t* cnt64, const uint64_t* text_off,
                              uint8_t* text, int n_cus, hipStream_t s);

}  // namespace scrg
