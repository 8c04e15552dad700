// seq_kernels.hip — the HBM-bound helper kernels around the aligner (gfx950): ASCII -> planar 2-bit packing,
// the reference's 4-bases-per-byte layout (src/genasm_gpu.cu:631-685) and the run compaction pass.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "genasm_kernels.h"
#include "../../include/scrooge_amd_device.hpp"

namespace scrg {

// ----------------------------------------------------------------------------
// ASCII -> planar 2-bit.  One thread per 32 bases: two 16-byte loads, one
// 8-byte store; a wave reads 2 KiB contiguous and writes 512 B contiguous.
// ----------------------------------------------------------------------------
__device__ __forceinline__ void pack4(uint32_t v, uint32_t& lo, uint32_t& hi, uint32_t& bad)
{
    // per byte: x = (c>>1)&3 gives A0 C1 T2 G3; code = x ^ (x>>1) gives A0 C1 G2 T3
    const uint32_t b1 = (v >> 1) & 0x01010101u;
    const uint32_t b2 = (v >> 2) & 0x01010101u;
    const uint32_t l = b1 ^ b2;
    const uint32_t h = b2;
    // gather the four byte-lsbs into a nibble
    lo = ((l * 0x01020408u) >> 24) & 0xfu;
    hi = ((h * 0x01020408u) >> 24) & 0xfu;
    // validity, all four bytes at once: the upper-cased byte must be the letter its code stands for
    // ('A' + {0, 2, 6, 19} for codes 0..3), or the byte is 0 (padding)
    const uint32_t u = v & 0xdfdfdfdfu;
    const uint32_t expect = 0x41414141u + 2u * l + 6u * h + 11u * (l & h);
    const uint32_t diff = u ^ expect;
    const uint32_t nz_diff = (((diff & 0x7f7f7f7fu) + 0x7f7f7f7fu) | diff) & 0x80808080u;   // bit 7 of every non-zero byte
    const uint32_t nz_raw = (((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v) & 0x80808080u;
    bad += (uint32_t)__popc(nz_diff & nz_raw);
}

__global__ __launch_bounds__(256) void pack_planar_kernel(const uint4* __restrict__ ascii, uint64_t n_words,
                                                          uint64_t* __restrict__ planar, uint32_t* __restrict__ bad_count)
{
    __builtin_amdgcn_s_setprio(3);      // (a helper between align launches: it goes first, see compact_runs_kernel)
    uint32_t bad = 0;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words;
         w += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 q0 = ascii[2 * w], q1 = ascii[2 * w + 1];
        const uint32_t v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t l, h;
            pack4(v[k], l, h, bad);
            lo |= l << (4 * k);
            hi |= h << (4 * k);
        }
        planar[w] = ((uint64_t)hi << 32) | lo;
    }
    if (bad) atomicAdd(bad_count, bad);
}

// The same packing into the lane-interleaved layout (scrg_pack_planar_groups): consecutive threads produce
// consecutive OUTPUT words, i.e. the same word of 64 consecutive rows — 512 B contiguous per wave on the
// write side.  A thread takes TWO consecutive words of its row: 64 contiguous bytes of text, a whole
// memory line (round 6: with one word — a 32-byte piece — per thread every line was asked for twice, by
// different wavefronts at different times: 0.77 ms per 100 k x 21.5 kb batch).
__global__ __launch_bounds__(256) void pack_planar_groups_kernel(const uint4* __restrict__ ascii, uint64_t n_rows,
                                                                 uint64_t words_per_row, uint64_t* __restrict__ planar,
                                                                 uint32_t* __restrict__ bad_count)
{
    __builtin_amdgcn_s_setprio(3);      // (a helper between align launches: it goes first, see compact_runs_kernel)
    uint32_t bad = 0;
    const uint64_t pairs_per_row = (words_per_row + 1) / 2;
    const uint64_t n_out = ((n_rows + 63) / 64) * 64 * pairs_per_row;
    for (uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t lane = o & 63, gp = o >> 6;                 // gp = group * pairs_per_row + word pair
        const uint64_t group = gp / pairs_per_row, w = 2 * (gp % pairs_per_row);
        const uint64_t row = group * 64 + lane;
        const bool second = w + 1 < words_per_row;
        uint64_t out0 = 0, out1 = 0;
        if (row < n_rows) {
            const uint64_t q = row * words_per_row + w;
            const uint4 q0 = ascii[2 * q], q1 = ascii[2 * q + 1];
            uint4 q2 = make_uint4(0, 0, 0, 0), q3 = q2;
            if (second) { q2 = ascii[2 * q + 2]; q3 = ascii[2 * q + 3]; }
            const uint32_t v[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            uint32_t lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t l, h;
                pack4(v[k], l, h, bad);
                lo0 |= l << (4 * k);
                hi0 |= h << (4 * k);
                pack4(v[8 + k], l, h, bad);
                lo1 |= l << (4 * k);
                hi1 |= h << (4 * k);
            }
            out0 = ((uint64_t)hi0 << 32) | lo0;
            out1 = ((uint64_t)hi1 << 32) | lo1;
        }
        const uint64_t at = (group * words_per_row + w) * 64 + lane;
        planar[at] = out0;
        if (second) planar[at + 64] = out1;
    }
    if (bad) atomicAdd(bad_count, bad);
}

// ----------------------------------------------------------------------------
// Reference-layout packer (src/genasm_gpu.cu:631-685): 4 bases per byte, the
// first base of each quad in bits 7..6; one thread per output byte, strings
// concatenated.  (The reference launches every block over the whole buffer;
// here each output byte is produced exactly once.)
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ascii_to_twobit_kernel(uint64_t count, const uint64_t* __restrict__ lens,
                                                              const uint64_t* __restrict__ ascii_off,
                                                              const char* __restrict__ ascii,
                                                              const uint64_t* __restrict__ twobit_off,
                                                              uint8_t* __restrict__ twobit,
                                                              uint32_t* __restrict__ bad_count)
{
    // grid.y strides over strings, grid.x*block over bytes of a string
    uint32_t bad = 0;
    for (uint64_t s = blockIdx.y; s < count; s += gridDim.y) {
        const uint64_t len = lens[s];
        const uint64_t nbytes = (len + 3) / 4;
        const char* src = ascii + ascii_off[s];
        uint8_t* dst = twobit + twobit_off[s];
        for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nbytes;
             b += (uint64_t)gridDim.x * blockDim.x)
            dst[b] = scrooge_amd::device::twobit_quad(src, len, b, &bad);      // (include/scrooge_amd_device.hpp: shared with the reference-named kernel)
    }
    if (bad) atomicAdd(bad_count, bad);
}

// ----------------------------------------------------------------------------
// Run compaction: one wavefront per pair copies its runs from the pair's
// arena slice into the dense output.
// ----------------------------------------------------------------------------
struct __attribute__((aligned(4))) Dwords4 {
    uint32_t x, y, z, w;      // 16 bytes that are only known to be dword aligned
};

__global__ __launch_bounds__(256) void compact_runs_kernel(uint64_t n_pairs, const scrg_pair_desc* __restrict__ pairs,
                                                           const uint16_t* __restrict__ runs,
                                                           const uint32_t* __restrict__ n_runs,
                                                           const uint64_t* __restrict__ dense_off,
                                                           uint16_t* __restrict__ dense, uint32_t split)
{
    // Long alignments: one wavefront per pair.  The slice starts 32-byte aligned, the destination at any run (2-byte)
    // boundary: an odd destination run index means every output dword straddles two source dwords
    // (v_alignbit by 16).  The bulk moves 16 bytes per lane with 16-byte aligned stores.
    // Pairs are taken 64 at a time; `split` wavefronts share a group of 64 (each takes every split-th long alignment), so
    // that a batch of few, long alignments still fills the GPU.
    // This kernel sits between two align launches of its stream and shares the SIMDs with the align wavefronts of the other
    // streams (which rotate their priorities 0..3): it goes first.  Its instructions are ~1 % of a step's; at priority 0 it
    // took 1.2 ms of the stream's chain while overlapped, and the headline 57.2 -> 59-60 M pairs/s with it in front
    // (scripts/r06_chain_probe.sh, profiles/r06_chain_probe.txt).
    __builtin_amdgcn_s_setprio(3);
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave_all = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t wave = wave_all / split, n_waves = (((uint64_t)gridDim.x * blockDim.x) >> 6) / split;
    const uint32_t sub = (uint32_t)(wave_all % split);
    for (uint64_t g0 = wave * 64; g0 < n_pairs; g0 += n_waves * 64) {
      const uint64_t mine = g0 + lane;
      uint64_t my_cnt = 0, my_src = 0, my_dst = 0;
      if (mine < n_pairs) {
          const uint64_t cap = pairs[mine].cigar_cap;
          my_cnt = n_runs[mine];
          if (my_cnt > cap) my_cnt = cap;
          my_src = pairs[mine].cigar_off;
          my_dst = dense_off[mine];
      }
      // alignments of up to 16 runs (150 bp reads have ~3): every lane copies its own pair; the longer ones of the group
      // (a read-mapping batch: the candidates at the wrong locus, ~100 runs) are copied by the whole wavefront, one after
      // the other, dealt to the `split` wavefronts that share the group
      const bool big_me = my_cnt > 16;
      if (sub == 0 && !big_me) {
          const uint16_t* const s = runs + my_src;
          uint16_t* const d = dense + my_dst;
          for (uint64_t k = 0; k < my_cnt; k++) d[k] = s[k];
      }
      uint64_t big = __ballot(big_me);
      for (uint32_t ord = 0; big != 0; ord++) {
        const uint32_t q = (uint32_t)__builtin_ctzll(big);
        big &= big - 1;
        if (ord % split != sub) continue;
        uint64_t cnt = __shfl(my_cnt, (int)q, 64);
        if (cnt == 0) continue;
        const uint64_t src_off = __shfl(my_src, (int)q, 64), dst_off = __shfl(my_dst, (int)q, 64);
        const uint16_t* const s16 = runs + src_off;
        const uint32_t* const s32 = reinterpret_cast<const uint32_t*>(s16);
        uint16_t* const d16 = dense + dst_off;
        const uint32_t odd = (uint32_t)(dst_off & 1u);          // the first run goes out alone, the rest is dword aligned
        if (odd && lane == 0) d16[0] = s16[0];
        const uint64_t rem = cnt - odd;
        const uint64_t nd = rem >> 1;                                  // whole output dwords
        uint32_t* const d32 = reinterpret_cast<uint32_t*>(d16 + odd);
        // output dword q = source runs (odd + 2q, odd + 2q + 1)
        auto out_dword = [&](uint64_t q) -> uint32_t {
            return odd ? __builtin_amdgcn_alignbit(s32[q + 1], s32[q], 16) : s32[q];
        };
        uint64_t head = (uint64_t)((0u - (uint32_t)(reinterpret_cast<uintptr_t>(d32) >> 2)) & 3u);   // dwords up to 16-byte alignment
        if (head > nd) head = nd;
        if (lane < head) d32[lane] = out_dword(lane);
        const uint64_t groups = (nd - head) >> 2;
        for (uint64_t g = lane; g < groups; g += 64) {
            const uint64_t q = head + 4 * g;
            const Dwords4 a = *reinterpret_cast<const Dwords4*>(s32 + q);
            uint4 o;
            if (odd) {
                const uint32_t e = s32[q + 4];
                o.x = __builtin_amdgcn_alignbit(a.y, a.x, 16);
                o.y = __builtin_amdgcn_alignbit(a.z, a.y, 16);
                o.z = __builtin_amdgcn_alignbit(a.w, a.z, 16);
                o.w = __builtin_amdgcn_alignbit(e, a.w, 16);
            } else {
                o.x = a.x; o.y = a.y; o.z = a.z; o.w = a.w;
            }
            *reinterpret_cast<uint4*>(d32 + q) = o;
        }
        const uint64_t done = head + 4 * groups;
        if (done + lane < nd) d32[done + lane] = out_dword(done + lane);          // up to 3 dwords
        if ((rem & 1u) && lane == 63) d16[cnt - 1] = s16[cnt - 1];
      }
    }
}
// ----------------------------------------------------------------------------
// Packed runs for the RCCL gather: one byte per run, op in bits 7..6 (0 '=', 1 'X', 2 'I', 3 'D'), count in
// bits 5..0 (a run never spans windows, so its count is at most W-O <= 63 for every W <= 64).  Halves the
// bytes a rank sends to rank 0; unpack_runs_kernel restores scrg_run pairs bit for bit.
// ----------------------------------------------------------------------------
__device__ __forceinline__ uint32_t run_to_byte(uint32_t run16)
{
    // op char in bits 15..8: '=' 0x3D, 'X' 0x58, 'I' 0x49, 'D' 0x44 -> index (bit4, bit2) = 3, 2, 0, 1 -> code 0, 1, 2, 3
    const uint32_t op = run16 >> 8;
    const uint32_t idx = ((op >> 3) & 2u) | ((op >> 2) & 1u);
    // idx 0 'I' -> 2, idx 1 'D' -> 3, idx 2 'X' -> 1, idx 3 '=' -> 0: 2-bit fields 0b00'01'11'10
    const uint32_t code = (0x1Eu >> (2u * idx)) & 3u;
    return (code << 6) | (run16 & 63u);
}
__device__ __forceinline__ uint32_t byte_to_run(uint32_t b)
{
    const uint32_t chars = 0x4449583Du;                           // code 0 '=', 1 'X', 2 'I', 3 'D'
    return (((chars >> (8u * (b >> 6))) & 0xffu) << 8) | (b & 63u);
}

__global__ __launch_bounds__(256) void compact_runs_packed_kernel(uint64_t n_pairs, const scrg_pair_desc* __restrict__ pairs,
                                                                  const uint16_t* __restrict__ runs,
                                                                  const uint32_t* __restrict__ n_runs,
                                                                  const uint64_t* __restrict__ dense_off,
                                                                  uint8_t* __restrict__ dense)
{
    __builtin_amdgcn_s_setprio(3);      // (a helper between align launches: it goes first, see compact_runs_kernel)
    // one wavefront per pair; a lane converts four runs (8 bytes in, 4 bytes out) per iteration once the
    // destination is dword aligned
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t p = wave; p < n_pairs; p += n_waves) {
        const uint64_t cap = pairs[p].cigar_cap;
        uint64_t cnt = n_runs[p];
        if (cnt > cap) cnt = cap;
        if (cnt == 0) continue;
        const uint16_t* const s16 = runs + pairs[p].cigar_off;
        uint8_t* const d8 = dense + dense_off[p];
        uint64_t head = (uint64_t)((0u - (uint32_t)reinterpret_cast<uintptr_t>(d8)) & 3u);      // bytes up to dword alignment
        if (head > cnt) head = cnt;
        if (lane < head) d8[lane] = (uint8_t)run_to_byte(s16[lane]);
        const uint64_t quads = (cnt - head) >> 2;
        uint32_t* const d32 = reinterpret_cast<uint32_t*>(d8 + head);
        for (uint64_t q = lane; q < quads; q += 64) {
            const uint16_t* const s = s16 + head + 4 * q;
            d32[q] = run_to_byte(s[0]) | (run_to_byte(s[1]) << 8) | (run_to_byte(s[2]) << 16) | (run_to_byte(s[3]) << 24);
        }
        const uint64_t done = head + 4 * quads;
        if (done + lane < cnt) d8[done + lane] = (uint8_t)run_to_byte(s16[done + lane]);        // up to 3 runs
    }
}

__global__ __launch_bounds__(256) void unpack_runs_kernel(uint64_t n, const uint8_t* __restrict__ packed, uint16_t* __restrict__ runs)
{
    __builtin_amdgcn_s_setprio(3);      // (a helper between align launches: it goes first, see compact_runs_kernel)
    // n runs; a thread restores four at a time where both sides are aligned (callers pass 4-byte aligned buffers)
    const uint64_t quads = n >> 2;
    const uint32_t* const p32 = reinterpret_cast<const uint32_t*>(packed);
    uint2* const r64 = reinterpret_cast<uint2*>(runs);
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = p32[q];
        uint2 o;
        o.x = byte_to_run(v & 0xffu) | (byte_to_run((v >> 8) & 0xffu) << 16);
        o.y = byte_to_run((v >> 16) & 0xffu) | (byte_to_run(v >> 24) << 16);
        r64[q] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3u)) runs[4 * quads + threadIdx.x] = (uint16_t)byte_to_run(packed[4 * quads + threadIdx.x]);
}

hipError_t launch_pack_planar(const char* d_ascii, uint64_t n_words, uint64_t* d_planar, uint32_t* d_bad,
                              int n_cus, hipStream_t s)
{
    if (n_words == 0) return hipSuccess;
    uint64_t blocks = (n_words + 255) / 256;
    const uint64_t cap = (uint64_t)n_cus * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(pack_planar_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                       reinterpret_cast<const uint4*>(d_ascii), n_words, d_planar, d_bad);
    return hipGetLastError();
}

hipError_t launch_pack_planar_groups(const char* d_ascii, uint64_t n_rows, uint64_t words_per_row, uint64_t* d_planar,
                                     uint32_t* d_bad, int n_cus, hipStream_t s)
{
    const uint64_t n_out = ((n_rows + 63) / 64) * 64 * ((words_per_row + 1) / 2);      // (a thread takes two words of a row)
    if (n_out == 0) return hipSuccess;
    uint64_t blocks = (n_out + 255) / 256;
    const uint64_t cap = (uint64_t)n_cus * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(pack_planar_groups_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                       reinterpret_cast<const uint4*>(d_ascii), n_rows, words_per_row, d_planar, d_bad);
    return hipGetLastError();
}

hipError_t launch_ascii_to_twobit(uint64_t count, const uint64_t* d_lens, const uint64_t* d_ascii_off,
                                  const char* d_ascii, const uint64_t* d_twobit_off, uint8_t* d_twobit,
                                  uint32_t* d_bad, uint64_t max_len, hipStream_t s)
{
    if (count == 0) return hipSuccess;
    uint64_t bx = ((max_len + 3) / 4 + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    uint64_t by = count < 4096 ? count : 4096;
    hipLaunchKernelGGL(ascii_to_twobit_kernel, dim3((unsigned)bx, (unsigned)by), dim3(256), 0, s,
                       count, d_lens, d_ascii_off, d_ascii, d_twobit_off, d_twobit, d_bad);
    return hipGetLastError();
}

hipError_t launch_compact_runs(uint64_t n_pairs, const scrg_pair_desc* d_pairs, const uint16_t* d_runs,
                               const uint32_t* d_n_runs, const uint64_t* d_dense_off, uint16_t* d_dense,
                               int n_cus, hipStream_t s)
{
    if (n_pairs == 0) return hipSuccess;
    // groups of 64 pairs; up to 8 wavefronts share a group while there are fewer groups than the GPU has room for
    const uint64_t groups = (n_pairs + 63) / 64, room = (uint64_t)n_cus * 32;
    uint32_t split = 1;
    while (split < 8 && groups * split * 2 <= room) split *= 2;
    uint64_t blocks = (groups * split + 3) / 4;                  // four wavefronts per workgroup
    const uint64_t cap = (uint64_t)n_cus * 8;
    if (blocks > cap) blocks = cap;
    if (split == 8 && (blocks & 1)) blocks++;                    // (the wavefronts of a launch: a multiple of `split`)
    hipLaunchKernelGGL(compact_runs_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                       n_pairs, d_pairs, d_runs, d_n_runs, d_dense_off, d_dense, split);
    return hipGetLastError();
}

hipError_t launch_compact_runs_packed(uint64_t n_pairs, const scrg_pair_desc* d_pairs, const uint16_t* d_runs,
                                      const uint32_t* d_n_runs, const uint64_t* d_dense_off, uint8_t* d_dense,
                                      int n_cus, hipStream_t s)
{
    if (n_pairs == 0) return hipSuccess;
    uint64_t blocks = (n_pairs + 3) / 4;
    const uint64_t cap = (uint64_t)n_cus * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(compact_runs_packed_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                       n_pairs, d_pairs, d_runs, d_n_runs, d_dense_off, d_dense);
    return hipGetLastError();
}

hipError_t launch_unpack_runs(uint64_t n_runs, const uint8_t* d_packed, uint16_t* d_runs, int n_cus, hipStream_t s)
{
    if (n_runs == 0) return hipSuccess;
    uint64_t blocks = ((n_runs >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    const uint64_t cap = (uint64_t)n_cus * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(unpack_runs_kernel, dim3((unsigned)blocks), dim3(256), 0, s, n_runs, d_packed, d_runs);
    return hipGetLastError();
}

}  // namespace scrg
