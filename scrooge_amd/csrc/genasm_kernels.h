// genasm_kernels.h — launch interface between the C-ABI host code
// (scrg_api.cpp) and the gfx950 kernels (genasm_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/scrooge_amd.h"

namespace scrg {

// rows of the HBM spill area per pair slot: K+1 = 65 rows of R can exist, plus
// the rows lanes ahead of the slot leader run past the final row
constexpr int SPILL_ROWS = 72;
// (genasm_kernel_multiword.hip, 64 < W <= 256, sizes its spill area as W+1 rows of stored_row_dwords())

struct AlignArgs {
    const uint64_t* seq;          // planar 2-bit words
    const scrg_pair_desc* pairs;
    uint16_t* runs;               // scrg_run as count | op << 8
    int64_t* ed;
    uint32_t* n_runs;
    uint32_t* status;
    uint32_t* run_count;          // edit-stream output only, may be null: the number of runs the same alignment has (what
                                  //   scrg_align_device would report in n_runs), so that a receiver can size the decoded array
    uint32_t* counter;            // work queue head, zeroed before launch
    uint32_t* spill;              // grid * slots * SPILL_ROWS * 32 words
    uint32_t n_pairs;
    int32_t W;
    int32_t tb_limit;             // W - O
    int32_t lds_rows;
    uint32_t text_stride;         // words between consecutive words of a text / a read in seq (1 = contiguous;
    uint32_t read_stride;         //   64 = lane-interleaved groups, scrg_pack_planar_groups; lane kernel only)
    uint64_t* stats;              // profiling builds (-DSCRG_STATS) only: counters, may be null; never read by the shipped kernels
    int32_t debug;                // params.reserved[0] (always 0 in the shipped build): see SCRG_SEL / SCRG_SW / SCRG_ABL below
    uint32_t stranded;            // params.stranded: bit 63 of a pair's read_off = align the read's reverse complement (the one-pair-per-lane kernels)
};

// scrg_params.reserved[0] / reserved[1].  The SHIPPED library accepts neither: scrg_params_resolve() rejects every bit.
// Everything they can do is experiment and test plumbing and exists only in other builds of the same sources (scripts/ab.sh;
// scrooge_amd.build_library(variant=...) for the test build):
//   -DSCRG_SELECT  the test build (ab_libs/lib_select.so): reserved[0] selects between formulations that give IDENTICAL results,
//                  which the parity tests compare — 32 (lanes_per_pair = 8: no diagonal-major path), 256 (the kernel with the
//                  window table in HBM where one that keeps it in registers would serve), 512 / 1024 (the default kernel as two
//                  wavefronts per 64 pairs / as one, whatever the launch size);
//   -DSCRG_STATS   (implies SELECT) the kernels' counters (reserved[1] != 0 -> scrg_debug_stats: window rounds, shader cycles
//                  per part, wavefront life times) and the scheduling switches 1 (one pair per lane: no wavefront priority
//                  rotation), 64 / 128 (workgroups of one / two wavefronts) — results intact;
//   -DSCRG_ABLATE  (implies STATS) the ablation switches 2, 4, 8, 16 and 1 for the GenASM-row kernel: skip the table, the
//                  runs, the walk, the stores — results are WRONG by design (bench.py --ablate).
// In the shipped build SCRG_TIMING / SCRG_SW / SCRG_ABL / SCRG_SEL are compile-time false: no counter, no clock read, no exit
// atomic and no switch test is left in the kernels or in the kernel selection.
#if defined(SCRG_ABLATE) && !defined(SCRG_STATS)
#define SCRG_STATS 1
#endif
#if defined(SCRG_STATS) && !defined(SCRG_SELECT)
#define SCRG_SELECT 1
#endif
constexpr int32_t SCRG_SWITCH_NO_DIAG = 32;       // lanes_per_pair = 8: GenASM rows only (no diagonal-major path)
constexpr int32_t SCRG_SWITCH_MW_TABLE = 256;     // genasm_lane_mw_kernel (table in HBM) where genasm_lane_wide_kernel / genasm_lane_parts_kernel (table in registers) would serve
constexpr int32_t SCRG_SWITCH_SPLIT = 512;        // W <= 64, W-O <= 31, runs output: genasm_lane_split_kernel (a window's work on two wavefronts) whatever the launch size
constexpr int32_t SCRG_SWITCH_NO_SPLIT = 1024;    // ... genasm_lane_kernel whatever the launch size (default: by launch size, scrg_api.cpp)
constexpr int32_t SCRG_SAFE_SWITCHES = SCRG_SWITCH_NO_DIAG | SCRG_SWITCH_MW_TABLE | SCRG_SWITCH_SPLIT | SCRG_SWITCH_NO_SPLIT;
#ifdef SCRG_SELECT
#define SCRG_SEL(flags, bit) (((flags) & (bit)) != 0)
constexpr bool SCRG_HAVE_SELECT = true;
#else
#define SCRG_SEL(flags, bit) false
constexpr bool SCRG_HAVE_SELECT = false;
#endif
#ifdef SCRG_STATS
#define SCRG_TIMING(args) ((args).stats != nullptr)
#define SCRG_SW(args, bit) (((args).debug & (bit)) != 0)
constexpr bool SCRG_HAVE_STATS = true;
#else
#define SCRG_TIMING(args) false
#define SCRG_SW(args, bit) false
constexpr bool SCRG_HAVE_STATS = false;
#endif
#ifdef SCRG_ABLATE
#define SCRG_ABL(args, bit) (((args).debug & (bit)) != 0)
constexpr int32_t SCRG_ALLOWED_SWITCHES = 0x1ff | SCRG_SAFE_SWITCHES;
#elif defined(SCRG_STATS)
#define SCRG_ABL(args, bit) false
constexpr int32_t SCRG_ALLOWED_SWITCHES = SCRG_SAFE_SWITCHES | 1 | 64 | 128;
#elif defined(SCRG_SELECT)
#define SCRG_ABL(args, bit) false
constexpr int32_t SCRG_ALLOWED_SWITCHES = SCRG_SAFE_SWITCHES;
#else
#define SCRG_ABL(args, bit) false
constexpr int32_t SCRG_ALLOWED_SWITCHES = 0;
#endif

hipError_t launch_align(int lanes_per_pair, const AlignArgs& a, int grid, size_t lds_bytes, hipStream_t s);
hipError_t launch_align_multiword(int lanes_per_pair, const AlignArgs& a, int grid, size_t lds_bytes, hipStream_t s);
// lanes_per_pair = 1: one pair per lane, 64 pairs per wavefront (genasm_lane_kernel.hip; W <= 64, W-O <= 31)
// (edits: the pairs' slices receive edit streams instead of runs, n_runs their lengths in bytes — scrg_align_device_edits)
hipError_t launch_align_lane(const AlignArgs& a, int grid, size_t lds_bytes, hipStream_t s, bool edits = false);
// the same alignment (runs output) with a window's work split over a producer and a consumer wavefront (genasm_lane_kernel.hip:
// genasm_lane_split_kernel): for launches that cannot fill the SIMDs.  grid counts producer wavefronts; 8 resident per CU.
hipError_t launch_align_lane_split(const AlignArgs& a, int grid, hipStream_t s);
constexpr int LANE_SPLIT_PRODUCERS_PER_CU = 8;

// dwords of one stored row of R (the part the traceback can reach):
//   W <= 64: the high dword of columns 0..31, or whole entries of all 64 columns when W-O > 31;
//   W  > 64: words 0..SW-1 of columns 0..64*SW-1 with SW = (W-O)/64 + 1.
#ifdef __HIPCC__
#define SCRG_HD __host__ __device__
#else
#define SCRG_HD
#endif
// genasm_lane_mw_kernel (W-O > 31 or W > 64): LDS per wavefront = CIGAR ring + one length byte for each of the W-O columns;
// its table — two rows of (W-O)/64 + 1 64-bit words for each of the W-O columns, for 64 lanes — is a slab of HBM
SCRG_HD inline unsigned lane_mw_len_bytes(int tb_limit) { return (((unsigned)tb_limit + 3u) & ~3u) + 4u; }
SCRG_HD inline unsigned lane_mw_lds_bytes(int tb_limit) { return 64u * (68u + lane_mw_len_bytes(tb_limit)); }
SCRG_HD inline size_t lane_mw_table_bytes(int tb_limit) { return (size_t)tb_limit * 2u * ((unsigned)tb_limit / 64u + 1u) * 64u * 8u; }
hipError_t launch_align_lane_mw(const AlignArgs& a, int grid, size_t lds_bytes, hipStream_t s, bool edits = false);
// genasm_lane_wide_kernel (one pair per lane, W <= 128, 32 <= W-O <= 63): the table in registers, built in two halves.
// LDS per wavefront and lane: CIGAR ring, 32 insertion-run lengths, the window's Eq words for the four bases and "no match".
SCRG_HD inline bool lane_wide_serves(int W, int tb_limit) { return W <= 128 && tb_limit >= 32 && tb_limit <= 63; }
SCRG_HD inline unsigned lane_wide_lds_bytes(int W) { return 64u * (68u + 36u + (W <= 64 ? 40u : 80u)); }
hipError_t launch_align_lane_wide(const AlignArgs& a, int grid, size_t lds_bytes, hipStream_t s, bool edits = false);
// genasm_lane_parts_kernel (one pair per lane, W <= 256, 64 <= W-O <= 127 — or W > 128 with any W-O <= 63 —: two-word table rows): the table in registers, built
// in parts of 16 columns from checkpoints of the difference vectors.  LDS per wavefront and lane: CIGAR ring, 16 insertion-run
// lengths, the window's Eq words for the four bases and "no match" (NW words each, slots of 16 or 32 bytes), the window's text.
// HBM per wavefront: 8 checkpoints of 4 NW dwords per lane.
// (Round 5: also W > 128 with W-O <= 63 — 1 to 4 parts, rows in the first word only — which the two-halves kernel, W <= 128, does
// not reach; genasm_lane_mw_kernel is left with W-O >= 128: table rows of three and four words.)
SCRG_HD inline bool lane_parts_serves(int W, int tb_limit)
{
    return W > 64 && W <= 256 && ((tb_limit >= 64 && tb_limit <= 127) || (W > 128 && tb_limit >= 1 && tb_limit <= 63));      // (W = 64, O = 0: genasm_lane_mw_kernel)
}
SCRG_HD inline unsigned lane_parts_lds_bytes(int W)
{
    const unsigned nw = ((unsigned)W + 63u) / 64u, slot = nw == 2u ? 16u : 32u;
    return 64u * (68u + 20u + 5u * slot + 16u * nw);
}
SCRG_HD inline size_t lane_parts_checkpoint_bytes(int W) { return (size_t)8u * 4u * (((unsigned)W + 63u) / 64u) * 64u * 4u; }
hipError_t launch_align_lane_parts(const AlignArgs& a, int grid, size_t lds_bytes, hipStream_t s, bool edits = false);

SCRG_HD inline unsigned stored_row_dwords(int W, int tb_limit)
{
    if (W <= 64) return tb_limit > 31 ? 128u : 32u;
    const unsigned sw = (unsigned)tb_limit / 64u + 1u;
    return 64u * sw * sw * 2u;
}

// dwords of LDS per pair slot for the table R: lds_rows stored rows + 1 (slots land on distinct banks).
// The G = 8 kernel's diagonal-major path (W = 64, W-O <= 31) keeps 16 rows in its own compacted
// layout — 8 rows x 32 diagonals, 8 rows x 16, 4 dwords where idle lanes park their stores — and needs
// 397 dwords whatever lds_rows says.
constexpr unsigned DIAG_WIDE_ROWS = 8;                    // rows 0..7: all 32 diagonals, 32 dwords each
constexpr unsigned DIAG_LATE_BASE = 128;                  // rows r >= 8: diagonal x at 128 + 16 r + x (x = 8..23), i.e. from dword 264
constexpr unsigned DIAG_PARK_DWORD = 392;                 // lanes without a late diagonal park their four stores here
constexpr unsigned DIAG_SLOT_DWORDS = DIAG_PARK_DWORD + 4 + 1;     // 397
SCRG_HD inline unsigned slot_stride_dwords(int W, int tb_limit, int lanes_per_pair, int lds_rows)
{
    unsigned s = (unsigned)lds_rows * stored_row_dwords(W, tb_limit) + 1u;
    if (W == 64 && lanes_per_pair == 8 && tb_limit <= 31 && s < DIAG_SLOT_DWORDS) s = DIAG_SLOT_DWORDS;
    return s;
}
hipError_t launch_pack_planar(const char* d_ascii, uint64_t n_words, uint64_t* d_planar, uint32_t* d_bad,
                              int n_cus, hipStream_t s);
hipError_t launch_pack_planar_groups(const char* d_ascii, uint64_t n_rows, uint64_t words_per_row, uint64_t* d_planar,
                                     uint32_t* d_bad, int n_cus, hipStream_t s);
hipError_t launch_ascii_to_twobit(uint64_t count, const uint64_t* d_lens, const uint64_t* d_ascii_off,
                                  const char* d_ascii, const uint64_t* d_twobit_off, uint8_t* d_twobit,
                                  uint32_t* d_bad, uint64_t max_len, hipStream_t s);
hipError_t launch_compact_runs(uint64_t n_pairs, const scrg_pair_desc* d_pairs, const uint16_t* d_runs,
                               const uint32_t* d_n_runs, const uint64_t* d_dense_off, uint16_t* d_dense,
                               int n_cus, hipStream_t s);

hipError_t launch_compact_runs_packed(uint64_t n_pairs, const scrg_pair_desc* d_pairs, const uint16_t* d_runs,
                                      const uint32_t* d_n_runs, const uint64_t* d_dense_off, uint8_t* d_dense,
                                      int n_cus, hipStream_t s);
hipError_t launch_unpack_runs(uint64_t n_runs, const uint8_t* d_packed, uint16_t* d_runs, int n_cus, hipStream_t s);

}  // namespace scrg
