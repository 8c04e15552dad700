// edit_stream_kernels.hip — CIGAR runs -> edit stream on the GPU (format and rationale: edit_stream.h).
//
// encode_edits_kernel: one pair per lane walks its runs (its slice of d_runs, as an align kernel left them, read in 16-byte blocks) twice with
// encode_runs() of edit_stream.h — the code the host conversion scrg_runs_to_edit_stream runs —: once to size the
// stream, once to write it, whole dwords at a time; between the walks ONE atomic per wavefront on the stream cursor
// reserves the bytes of its 64 pairs.  The window loop has to be replayed to place the window-end bytes (a run list
// does not say where a window ended when the runs on both sides differ), which is serial per pair: this kernel serves
// the mappings that only produce runs (the GenASM-row kernels); the one-pair-per-lane align kernels write edit streams
// themselves (scrg_align_device_edits), window ends included, as they go.  8.4 ms per 100 k x 10 kb pairs (the lanes of a
// wavefront are at different places of their window loops all the time; format 1 needed no replay and its encoder, 64 lanes per
// pair, took 0.39 ms) — next to the 12 ms and more that the mappings it serves take to align such a batch.
// (The way back, streams -> runs, is edit_stream_decode_kernel.hip.)
#include "edit_stream.h"

namespace scrg {

__global__ __launch_bounds__(256) void encode_edits_kernel(uint64_t n_pairs, uint32_t L, const scrg_pair_desc* __restrict__ pairs,
                                                           const uint16_t* __restrict__ runs,
                                                           const uint32_t* __restrict__ n_runs, uint8_t* __restrict__ stream,
                                                           uint64_t stream_cap, uint64_t* __restrict__ off,
                                                           uint32_t* __restrict__ len, uint64_t* __restrict__ total)
{
    __builtin_amdgcn_s_setprio(3);      // (a helper between align launches: it goes first, see compact_runs_kernel)
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = p < n_pairs;
    const uint16_t* my = runs;
    uint64_t cnt = 0;
    if (valid) {
        const uint64_t cap = pairs[p].cigar_cap;
        cnt = n_runs[p];
        if (cnt > cap) cnt = cap;
        my = runs + pairs[p].cigar_off;
    }
    // run r of my pair (count in the low byte, the letter in the high one).  encode_runs asks for the runs in order, each once:
    // they come from aligned 16-byte blocks of eight, the block after the current one asked for a block ahead (a lane's
    // loads are dependent on nothing but r, but one load per run — 2 140 per pair and walk — made the kernel wait for memory
    // all the time: 9.0 -> 8.4 ms per 100 k x 10 kb pairs).  Slices are 32-byte aligned and a multiple of 16 runs long.
    const uint64_t blocks = (cnt + 7u) >> 3;
    uint4 cur = make_uint4(0, 0, 0, 0), nxt = cur;
    auto load_block = [&](uint64_t blk) -> uint4 {
        return blk < blocks ? *reinterpret_cast<const uint4*>(my + 8u * blk) : make_uint4(0, 0, 0, 0);
    };
    auto rewind = [&]() { cur = load_block(0); nxt = load_block(1); };
    auto get = [&](uint64_t r) -> uint32_t {
        const uint32_t k = (uint32_t)r & 7u;
        if (k == 0u && r != 0u) {
            cur = nxt;
            nxt = load_block((r >> 3) + 1u);
        }
        const uint32_t w = k < 4u ? (k < 2u ? cur.x : cur.y) : (k < 6u ? cur.z : cur.w);
        return (k & 1u) ? w >> 16 : w & 0xffffu;
    };
    // ---- walk 1: the size
    rewind();
    uint64_t bytes = encode_runs(cnt, L, get, [](uint8_t) {});
    const bool bad_op = bytes == ~0ull;                                         // a letter that is not = X I D: reported like a pair that did not fit
    if (bad_op) bytes = 0;
    // ---- one reservation per wavefront (streams start at multiples of 4)
    const uint64_t mine = (bytes + 3u) & ~3ull;
    uint64_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t v = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += v;
    }
    const uint64_t all = __shfl(incl, 63, 64);
    uint64_t base = 0;
    if (lane == 0 && all) base = atomicAdd(reinterpret_cast<unsigned long long*>(total), (unsigned long long)all);
    base = __shfl(base, 0, 64);
    if (!valid) return;
    const uint64_t at = base + incl - mine;
    // (`mine`, not `bytes`, is what walk 2 stores — whole dwords — and what is subtracted: with a capacity that is not a multiple of 4 and
    // bytes in (cap - 3, cap] the difference stream_cap - mine would wrap and the pair "fit" anywhere)
    const bool fits = !bad_op && mine <= stream_cap && at <= stream_cap - mine;
    off[p] = fits ? at : ~0ull;
    len[p] = bytes > 0xffffffffull ? 0xffffffffu : (uint32_t)bytes;
    if (!fits) atomicAdd(reinterpret_cast<unsigned long long*>(total + 1), 1ull);
    if (!fits || bytes == 0) return;
    // ---- walk 2: the bytes, as whole dwords (the reservation is a multiple of 4: the last dword's padding is zeros)
    uint32_t* const d32 = reinterpret_cast<uint32_t*>(stream + at);
    uint32_t w = 0;
    uint64_t k = 0;
    rewind();
    (void)encode_runs(cnt, L, get, [&](uint8_t b) {
        w |= (uint32_t)b << (8u * ((uint32_t)k & 3u));
        if (((uint32_t)k & 3u) == 3u) {
            d32[k >> 2] = w;
            w = 0;
        }
        k++;
    });
    if ((uint32_t)k & 3u) d32[k >> 2] = w;
}

hipError_t launch_encode_edits(uint64_t n_pairs, uint32_t W, uint32_t O, const scrg_pair_desc* d_pairs, const uint16_t* d_runs,
                               const uint32_t* d_n_runs, uint8_t* d_stream, uint64_t stream_cap, uint64_t* d_off,
                               uint32_t* d_len, uint64_t* d_total, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(d_total, 0, 2 * sizeof(uint64_t), s);
    if (e != hipSuccess || n_pairs == 0) return e;
    const uint64_t blocks = (n_pairs + 255) / 256;
    hipLaunchKernelGGL(encode_edits_kernel, dim3((unsigned)blocks), dim3(256), 0, s, n_pairs, W - O, d_pairs, d_runs, d_n_runs,
                       d_stream, stream_cap, d_off, d_len, d_total);
    return hipGetLastError();
}

}  // namespace scrg
