// host_path_kernels.hip — the small kernels of the pipelined host-pointer path (scrg_host.cpp): problem descriptors
// built on the device from a few bytes per pair, per-pair CIGAR text lengths, offsets by prefix sum (hipCUB), and the
// "%d%c" rendering of the runs (src/genasm_cpu.cpp:387-403 produces that text on the host, one sprintf per run).
// All HBM-bound helpers: the work of the path is genasm_lane_kernel.
#include <hipcub/hipcub.hpp>

#include "host_path.h"

namespace scrg {

// ---------------------------------------------------------------------------------------------------------------
// descriptors.  Pair i of a chunk (issue order).  Reads live in lane-interleaved groups of 64 rows (stride 64 words):
// row r starts at word read_base + (r / 64) * read_words * 64 + r % 64.
//   pairwise: the text of pair i is row i of a second such region (text_base, text_words);
//   mapping : the text is the packed genome at word 0 of the sequence array, from base start[i] to its end
//             (src/genasm_cpu.cpp:512-514); the read row of pair i is row[i] (candidates of one read share a row).
// Every pair gets a slice of `cap` runs (a multiple of 16) at i * cap.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_desc_kernel(HostDescArgs a)
{
    __builtin_amdgcn_s_setprio(3);      // (a helper between align launches: it goes first, see compact_runs_kernel)
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    scrg_pair_desc d;
    const uint32_t row_word = a.row ? a.row[i] : 0u;
    const uint64_t row = a.row ? row_word & 0x7fffffffu : i;              // (bit 31 of a mapping pair's row: the read's reverse complement is aligned)
    d.read_off = a.linear ? 32ull * (a.read_base + row * a.read_words) : 32ull * (a.read_base + (row >> 6) * a.read_words * 64ull + (row & 63ull));
    if (row_word & 0x80000000u) d.read_off |= SCRG_READ_REVCOMP;
    d.read_len = a.read_len[i];
    if (a.start) {
        const uint64_t st = a.start[i];
        d.text_off = st;
        d.text_len = a.genome_len - st;
    } else {
        d.text_off = a.linear ? 32ull * (a.text_base + i * a.text_words) : 32ull * (a.text_base + (i >> 6) * a.text_words * 64ull + (i & 63ull));
        d.text_len = a.text_len[i];
    }
    d.cigar_off = i * a.cap;
    d.cigar_cap = a.cap;
    a.desc[i] = d;
}

// ---------------------------------------------------------------------------------------------------------------
// per pair: run count (capped at the slice) and the length of its CIGAR text incl. the terminating NUL, as uint64 for
// the scans.  Long alignments: one wavefront per pair over the pair's slice; short ones: one pair per lane.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t run_chars(uint32_t run)        // digits of the count + the op character
{
    const uint32_t c = run & 0xffu;
    return 2u + (c >= 10u ? 1u : 0u) + (c >= 100u ? 1u : 0u);
}

__global__ __launch_bounds__(256) void text_len_kernel(uint64_t n, const scrg_pair_desc* __restrict__ pairs,
                                                       const uint16_t* __restrict__ runs, const uint32_t* __restrict__ n_runs,
                                                       uint64_t* __restrict__ cnt64, uint64_t* __restrict__ len64, int want_text,
                                                       uint32_t split)
{
    __builtin_amdgcn_s_setprio(3);      // (a helper between align launches: it goes first, see compact_runs_kernel)
    // pairs are taken 64 at a time; `split` wavefronts share a group of 64 (each takes every split-th pair of it), so
    // that a batch of few, long alignments still fills the GPU
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave_all = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t wave = wave_all / split, n_waves = (((uint64_t)gridDim.x * blockDim.x) >> 6) / split;
    const uint32_t sub = (uint32_t)(wave_all % split);
    if (wave >= n_waves) return;                       // (the grid is rounded up to whole workgroups)
    for (uint64_t g0 = wave * 64; g0 < n; g0 += n_waves * 64) {
        const uint64_t mine = g0 + lane;
        uint64_t my_cnt = 0, my_src = 0;
        if (mine < n) {
            const uint64_t cap = pairs[mine].cigar_cap;
            my_cnt = n_runs[mine];
            if (my_cnt > cap) my_cnt = cap;
            my_src = pairs[mine].cigar_off;
            if (sub == 0) cnt64[mine] = my_cnt;
        }
        if (!want_text) continue;
        // short alignments: the lane's own; the long ones of the group: the whole wavefront, one after the other
        const bool big_me = my_cnt > 24;
        if (sub == 0 && !big_me) {
            uint64_t chars = 1;
            const uint16_t* const s = runs + my_src;
            for (uint64_t k = 0; k < my_cnt; k++) chars += run_chars(s[k]);
            if (mine < n) len64[mine] = chars;
        }
        uint64_t big = __ballot(big_me);
        for (uint32_t ord = 0; big != 0; ord++) {
            const uint32_t q = (uint32_t)__builtin_ctzll(big);
            big &= big - 1;
            if (ord % split != sub) continue;
            const uint64_t cnt = __shfl(my_cnt, (int)q, 64), src = __shfl(my_src, (int)q, 64);
            const uint32_t* const s32 = reinterpret_cast<const uint32_t*>(runs + src);       // slices are 32-byte aligned
            uint32_t chars = 0;
            for (uint64_t k = lane; 2 * k < cnt; k += 64) {
                const uint32_t w = s32[k];
                chars += run_chars(w & 0xffffu) + (2 * k + 1 < cnt ? run_chars(w >> 16) : 0u);
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) chars += __shfl_xor((int)chars, d, 64);
            if (lane == 0) len64[g0 + q] = (uint64_t)chars + 1u;
        }
    }
}

// totals[0] = all runs, totals[1] = all text bytes of the chunk (the offsets are exclusive prefix sums); and what of the
// per-pair results travels to the host: 12 bytes per pair instead of the 28 of the four arrays the caller gets — the edit
// distance as 32 bits, the run count with the "slice overflowed" flag in bit 31, the text length; the host makes the
// offsets from the counts again (scrg_host.cpp, stage 3).  For read mapping that is a tenth of all the bytes that come back.
__global__ __launch_bounds__(256) void wire_totals_kernel(uint64_t n, const int64_t* __restrict__ ed, const uint32_t* __restrict__ status,
                                                          const uint64_t* __restrict__ cnt64, const uint64_t* __restrict__ run_off,
                                                          const uint64_t* __restrict__ len64, const uint64_t* __restrict__ text_off,
                                                          uint64_t* __restrict__ totals, uint32_t* __restrict__ wire, int want_text)
{
    __builtin_amdgcn_s_setprio(3);      // (a helper between align launches: it goes first, see compact_runs_kernel)
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        totals[0] = n ? run_off[n - 1] + cnt64[n - 1] : 0;
        totals[1] = (n && want_text) ? text_off[n - 1] + len64[n - 1] : 0;
    }
    if (i >= n) return;
    wire[i] = (uint32_t)ed[i];
    wire[n + i] = (uint32_t)cnt64[i] | (status[i] ? 0x80000000u : 0u);
    if (want_text) wire[2 * n + i] = (uint32_t)len64[i];
}

// ---------------------------------------------------------------------------------------------------------------
// "%d%c" text of every pair from its DENSE runs (after the compaction), NUL-terminated, at text_off[p].
// One wavefront per pair for long alignments: a tile of 512 runs (8 per lane, one 16-byte load) is rendered into LDS at
// the positions a wavefront scan of the lanes' character counts gives, then leaves as contiguous bytes.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t put_run(uint8_t* o, uint32_t run)
{
    const uint32_t c = run & 0xffu, op = run >> 8;
    uint32_t k = 0;
    if (c >= 100u) o[k++] = (uint8_t)('0' + c / 100u);
    if (c >= 10u) o[k++] = (uint8_t)('0' + (c / 10u) % 10u);
    o[k++] = (uint8_t)('0' + c % 10u);
    o[k++] = (uint8_t)op;
    return k;
}

constexpr uint32_t TEXT_TILE_RUNS = 512;
constexpr uint32_t TEXT_TILE_BYTES = TEXT_TILE_RUNS * 4;

__global__ __launch_bounds__(256) void render_text_kernel(uint64_t n, const uint16_t* __restrict__ dense,
                                                          const uint64_t* __restrict__ run_off, const uint64_t* __restrict__ cnt64,
                                                          const uint64_t* __restrict__ text_off, uint8_t* __restrict__ text, uint32_t split)
{
    __builtin_amdgcn_s_setprio(3);      // (a helper between align launches: it goes first, see compact_runs_kernel)
    __shared__ __attribute__((aligned(16))) uint8_t tile_all[4][TEXT_TILE_BYTES];
    const uint32_t lane = threadIdx.x & 63u;
    uint8_t* const tile = tile_all[threadIdx.x >> 6];
    const uint64_t wave_all = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t wave = wave_all / split, n_waves = (((uint64_t)gridDim.x * blockDim.x) >> 6) / split;
    const uint32_t sub = (uint32_t)(wave_all % split);
    if (wave >= n_waves) return;                       // (the grid is rounded up to whole workgroups)
    for (uint64_t g0 = wave * 64; g0 < n; g0 += n_waves * 64) {
        const uint64_t mine = g0 + lane;
        uint64_t my_cnt = 0, my_src = 0, my_dst = 0;
        if (mine < n) {
            my_cnt = cnt64[mine];
            my_src = run_off[mine];
            my_dst = text_off[mine];
        }
        // short alignments: one pair per lane, straight to memory; the long ones of the group: the whole wavefront
        const bool big_me = my_cnt > 24;
        if (mine < n && sub == 0 && !big_me) {
            uint8_t* o = text + my_dst;
            const uint16_t* const s = dense + my_src;
            for (uint64_t k = 0; k < my_cnt; k++) o += put_run(o, s[k]);
            *o = 0;
        }
        uint64_t big = __ballot(big_me);
        for (uint32_t ord = 0; big != 0; ord++) {
            const uint32_t q = (uint32_t)__builtin_ctzll(big);
            big &= big - 1;
            if (ord % split != sub) continue;
            const uint64_t cnt = __shfl(my_cnt, (int)q, 64), src = __shfl(my_src, (int)q, 64);
            uint64_t dst = __shfl(my_dst, (int)q, 64);
            const uint16_t* const s = dense + src;
            for (uint64_t t0 = 0; t0 < cnt; t0 += TEXT_TILE_RUNS) {
                // my eight runs of this tile (the dense array is only 2-byte aligned: plain loads)
                uint32_t r[8], chars = 0;
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) {
                    const uint64_t idx = t0 + 8u * lane + k;
                    r[k] = idx < cnt ? s[idx] : 0u;
                    chars += idx < cnt ? run_chars(r[k]) : 0u;
                }
                uint32_t incl = chars;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t v = (uint32_t)__shfl_up((int)incl, d, 64);
                    if (lane >= (uint32_t)d) incl += v;
                }
                const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
                uint8_t* o = tile + (incl - chars);
#pragma unroll
                for (uint32_t k = 0; k < 8; k++)
                    if (t0 + 8u * lane + k < cnt) o += put_run(o, r[k]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                uint8_t* const out = text + dst;
                for (uint32_t b = lane; b < total; b += 64) out[b] = tile[b];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                dst += total;
            }
            if (lane == 0) text[dst] = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------
// workgroups of four wavefronts for kernels that take pairs in groups of 64: about sixteen wavefronts per CU; a batch
// with fewer groups than that lets `split` wavefronts share a group
static uint64_t group_grid(uint64_t n, int n_cus, uint32_t* split)
{
    const uint64_t groups = (n + 63) / 64, target_waves = (uint64_t)n_cus * 16;
    uint64_t sp = 1;
    while (sp < 64 && groups * sp * 2 <= target_waves) sp *= 2;       // a power of two
    uint64_t waves = groups * sp;
    if (waves > target_waves * 2) waves = target_waves * 2;
    *split = (uint32_t)sp;
    return (waves + 3) / 4;
}

hipError_t launch_build_desc(const HostDescArgs& a, hipStream_t s)
{
    if (a.n == 0) return hipSuccess;
    hipLaunchKernelGGL(build_desc_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

size_t host_scan_temp_bytes(uint64_t n)
{
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)n, (hipStream_t)0);
    return bytes;
}

// cnt64 / len64 -> run_off / text_off (exclusive sums) and the two totals
hipError_t launch_result_layout(uint64_t n, const scrg_pair_desc* pairs, const uint16_t* runs, const uint32_t* n_runs, const int64_t* ed,
                                const uint32_t* status, uint64_t* cnt64, uint64_t* len64, uint64_t* run_off, uint64_t* text_off,
                                uint64_t* totals, uint32_t* wire, void* temp, size_t temp_bytes, int want_text, int n_cus, hipStream_t s)
{
    if (n == 0) return hipMemsetAsync(totals, 0, 16, s);
    uint32_t split = 1;
    const uint64_t blocks = group_grid(n, n_cus, &split);
    hipLaunchKernelGGL(text_len_kernel, dim3((unsigned)blocks), dim3(256), 0, s, n, pairs, runs, n_runs, cnt64, len64, want_text, split);
    size_t tb = temp_bytes;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(temp, tb, cnt64, run_off, (int)n, s);
    if (e != hipSuccess) return e;
    if (want_text) {
        tb = temp_bytes;
        e = hipcub::DeviceScan::ExclusiveSum(temp, tb, len64, text_off, (int)n, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(wire_totals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, ed, status, cnt64, run_off, len64, text_off, totals,
                       wire, want_text);
    return hipGetLastError();
}

hipError_t launch_render_text(uint64_t n, const uint16_t* dense, const uint64_t* run_off, const uint64_t* cnt64, const uint64_t* text_off,
                              uint8_t* text, int n_cus, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    uint32_t split = 1;
    const uint64_t blocks = group_grid(n, n_cus, &split);
    hipLaunchKernelGGL(render_text_kernel, dim3((unsigned)blocks), dim3(256), 0, s, n, dense, run_off, cnt64, text_off, text, split);
    return hipGetLastError();
}

}  // namespace scrg
