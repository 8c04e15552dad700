// edit_stream_kernels.hip — CIGAR runs <-> edit stream on the GPU (format and rationale: edit_stream.h).
//
// encode_edits_kernel: a workgroup takes a tile of 32 pairs, each of its four wavefronts 8 of them, one after the
// other.  A pair's runs (its slice of d_runs, as the align kernel left them) are cut into 64 contiguous segments,
// one per lane; a lane walks its segment in registers, five 16-byte loads in flight.  Walk 1 sizes every lane's
// share — the only cross-lane quantity is the number of matches pending when a lane's segment begins (one segmented
// scan over the wavefront) —, ONE atomic on the stream cursor reserves the bytes of the whole tile (one per pair on
// the same address bounded the kernel at 1.3 ms per 100 k pairs), walk 2 (the slice is 4 KB for a 10 kb read and
// still in cache) writes the bytes into LDS, from where they leave as whole dwords.  2 bytes per run in, ~1 byte
// per edit out; 0.39 ms per 100 k x 10 kb pairs.  (The one-pair-per-lane align kernel writes edit streams itself,
// scrg_align_device_edits; this kernel serves the configurations that only produce runs.)
// (The way back, streams -> runs with the window breaks restored, is edit_stream_decode_kernel.hip.)
#include "edit_stream.h"

namespace scrg {

namespace {

struct SegVal { uint32_t f, v; };

// inclusive segmented sum over the wavefront: element = (f: this lane holds an edit, v: matches after its last
// edit, or all its matches if it has none); result v = matches pending after this lane
__device__ __forceinline__ SegVal seg_scan(uint32_t lane, uint32_t f, uint32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t pf = __shfl_up(f, d, 64);
        const uint32_t pv = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d && !f) { v += pv; f = pf; }
    }
    return SegVal{f, v};
}
__device__ __forceinline__ uint32_t incl_scan(uint32_t lane, uint32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t p = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += p;
    }
    return v;
}

constexpr uint32_t ENC_LDS_BYTES = 4096;          // per wavefront; longer streams are written straight to memory

__device__ __forceinline__ uint32_t run_of(const uint4& q, uint32_t r)
{
    const uint32_t w = r < 4 ? (r < 2 ? q.x : q.y) : (r < 6 ? q.z : q.w);
    return (r & 1u) ? (w >> 16) : (w & 0xffffu);
}

// A lane's walk over its segment: up to five 16-byte loads are issued together (40 runs, the whole segment of a
// pair with up to 2560 runs), then the runs are handed to per_run(run) one by one.  Runs past the end of the
// segment (only the pair's last lane has any) are replaced by "0 matches", which every consumer ignores.
template <typename F>
__device__ __forceinline__ void walk_segment(const uint16_t* __restrict__ my, uint32_t mine, F&& per_run)
{
    constexpr uint32_t NONE = ((uint32_t)'=' << 8) | ((uint32_t)'=' << 24);
    for (uint32_t g0 = 0; g0 < mine; g0 += 40) {
        uint4 q[5];
#pragma unroll
        for (uint32_t j = 0; j < 5; j++) {
            q[j] = make_uint4(NONE, NONE, NONE, NONE);
            if (g0 + 8 * j < mine) q[j] = *reinterpret_cast<const uint4*>(my + g0 + 8 * j);
        }
#pragma unroll
        for (uint32_t j = 0; j < 5; j++) {
            const uint32_t valid = mine > g0 + 8 * j ? mine - g0 - 8 * j : 0u;
            if (valid > 0 && valid < 8) {
                uint32_t w[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
                for (uint32_t i = 0; i < 4; i++)
                    w[i] = 2 * i + 1 < valid ? w[i] : (2 * i < valid ? ((w[i] & 0xffffu) | ((uint32_t)'=' << 24)) : NONE);
                q[j] = make_uint4(w[0], w[1], w[2], w[3]);
            }
#pragma unroll
            for (uint32_t r = 0; r < 8; r++) per_run(run_of(q[j], r));
        }
    }
}

}  // namespace

constexpr uint32_t ENC_TP = 8;                    // pairs per wavefront and tile
constexpr uint32_t ENC_TILE = 4 * ENC_TP;         // pairs per workgroup

__global__ __launch_bounds__(256) void encode_edits_kernel(uint64_t n_pairs, const scrg_pair_desc* __restrict__ pairs,
                                                           const uint16_t* __restrict__ runs,
                                                           const uint32_t* __restrict__ n_runs, uint8_t* __restrict__ stream,
                                                           uint64_t stream_cap, uint64_t* __restrict__ off,
                                                           uint32_t* __restrict__ len, uint64_t* __restrict__ total)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage_all[4 * ENC_LDS_BYTES];
    __shared__ uint2 state_all[4][ENC_TP][64];            // per lane: matches pending before its segment, its byte offset
    __shared__ uint64_t size_all[ENC_TILE];               // bytes of every pair of the tile
    __shared__ uint64_t offs_all[ENC_TILE];               // where they start within the tile's reservation
    __shared__ uint64_t tile_base;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint8_t* const lds = stage_all + w * ENC_LDS_BYTES;
    const uint64_t p0 = (uint64_t)blockIdx.x * ENC_TILE + w * ENC_TP;

    struct Seg { const uint16_t* my; uint32_t mine; };
    // lane l owns runs [l * seg, l * seg + mine) of the pair; seg is a multiple of 8 runs = one 16-byte load (slices
    // are 32-byte aligned and a multiple of 16 runs long, so a load never leaves the slice)
    auto segment = [&](uint64_t p) -> Seg {
        const uint64_t cap = pairs[p].cigar_cap;
        uint64_t cnt = n_runs[p];
        if (cnt > cap) cnt = cap;
        const uint64_t seg = (((cnt + 63) >> 6) + 7) & ~7ull;
        const uint64_t first = seg * lane;
        const uint32_t mine = first < cnt ? (uint32_t)(cnt - first < seg ? cnt - first : seg) : 0u;
        return Seg{runs + pairs[p].cigar_off + first, mine};
    };

    // ---- walk 1, every pair of the tile: matches before the first edit of the segment (lead), after the last
    // (pend), bytes of everything but the first edit's long-match prefix (that needs the matches pending from the
    // lanes before: one segmented scan)
    for (uint32_t k = 0; k < ENC_TP; k++) {
        const uint64_t p = p0 + k;
        uint64_t bytes = 0;
        if (p < n_pairs) {
            const Seg sg = segment(p);
            uint32_t pend = 0, has = 0, lead = 0, nb = 0;
            // branch-free: nb counts the first edit's long-match bytes from the lane's own matches (lead >> 6) and is
            // corrected below once the matches pending from the lanes before are known
            walk_segment(sg.my, sg.mine, [&](uint32_t run) {
                const uint32_t n = run & 0xffu;
                const bool e = (run >> 8) != (uint32_t)'=' && n != 0u;       // (a run of count 0 is no run: scrg_runs_to_edit_stream skips it too)
                const uint32_t t = n + (pend >> 6);
                nb += e ? t : 0u;
                lead = (e && !has) ? pend : lead;
                has |= e ? 1u : 0u;
                pend = e ? 0u : pend + n;
            });
            const SegVal after = seg_scan(lane, has, pend);
            uint32_t before = __shfl_up(after.v, 1, 64);
            if (lane == 0) before = 0;
            const uint32_t bytes_mine = has ? nb - (lead >> 6) + ((before + lead) >> 6) : 0u;
            const uint32_t incl = incl_scan(lane, bytes_mine);
            state_all[w][k][lane] = make_uint2(before, incl - bytes_mine);
            bytes = __shfl(incl, 63, 64);
        }
        if (lane == 0) size_all[w * ENC_TP + k] = bytes;
    }
    __syncthreads();
    // ---- one reservation for the whole tile (one atomic per pair on the same address would bound the kernel)
    if (w == 0) {
        const uint64_t mine = lane < ENC_TILE ? (size_all[lane] + 3) & ~3ull : 0;
        uint64_t incl = mine;
#pragma unroll
        for (int d = 1; d < (int)ENC_TILE; d <<= 1) {
            const uint64_t v = __shfl_up(incl, d, 64);
            if (lane >= (uint32_t)d) incl += v;
        }
        const uint64_t all = __shfl(incl, ENC_TILE - 1, 64);
        if (lane < ENC_TILE) offs_all[lane] = incl - mine;
        if (lane == 0)
            tile_base = all ? atomicAdd(reinterpret_cast<unsigned long long*>(total), (unsigned long long)all) : 0;
    }
    __syncthreads();

    // ---- walk 2: the bytes, into LDS and from there to memory as whole dwords (straight to memory if a stream is
    // longer than the LDS buffer)
    for (uint32_t k = 0; k < ENC_TP; k++) {
        const uint64_t p = p0 + k;
        if (p >= n_pairs) break;
        const uint64_t bytes = size_all[w * ENC_TP + k];
        const uint64_t at = tile_base + offs_all[w * ENC_TP + k];
        const bool fits = bytes <= stream_cap && at <= stream_cap - bytes;
        if (lane == 0) {
            off[p] = fits ? at : ~0ull;
            len[p] = bytes > 0xffffffffull ? 0xffffffffu : (uint32_t)bytes;
            if (!fits) atomicAdd(reinterpret_cast<unsigned long long*>(total + 1), 1ull);
        }
        if (!fits || bytes == 0) continue;
        const Seg sg = segment(p);
        const uint2 st = state_all[w][k][lane];
        const bool staged = bytes <= ENC_LDS_BYTES;
        auto walk2 = [&](uint8_t* o) {
            uint32_t pm = st.x;
            // an edit run of n after pm pending matches: c = pm >> 6 bytes 0x3F, the edit byte with pm & 63, n - 1 more
            // edit bytes.  The edit byte is stored by every lane that has one; the rare rest takes a side path.
            walk_segment(sg.my, sg.mine, [&](uint32_t run) {
                const uint32_t n = run & 0xffu, op = run >> 8;
                const bool e = op != (uint32_t)'=' && n != 0u;
                const uint32_t nn = n;
                const uint32_t c = pm >> 6;
                // 'X' 0x58, 'I' 0x49, 'D' 0x44 -> 1, 2, 3: two bits of a constant at position op & 31 (24, 9, 4)
                const uint32_t code = ((0x01000430u >> (op & 31u)) & 3u) << 6;
                if (e) o[c] = (uint8_t)(code | (pm & 63u));
                if (e && (c | (nn - 1u))) {
                    for (uint32_t i = 0; i < c; i++) o[i] = 0x3F;
                    for (uint32_t i = 1; i < nn; i++) o[c + i] = (uint8_t)code;
                }
                o += e ? c + nn : 0u;
                pm = e ? 0u : pm + n;
            });
        };
        if (staged) walk2(lds + st.y);
        else walk2(stream + at + st.y);
        if (staged) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t n_dw = (uint32_t)((bytes + 3) >> 2);
            const uint32_t tail_mask = (bytes & 3u) ? (0xffffffffu >> (32u - 8u * (uint32_t)(bytes & 3u))) : 0xffffffffu;
            uint32_t* const d32 = reinterpret_cast<uint32_t*>(stream + at);             // at is a multiple of 4
            const uint32_t* const l32 = reinterpret_cast<const uint32_t*>(lds);
            for (uint32_t i = lane; i < n_dw; i += 64) d32[i] = l32[i] & (i + 1 == n_dw ? tail_mask : 0xffffffffu);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

hipError_t launch_encode_edits(uint64_t n_pairs, const scrg_pair_desc* d_pairs, const uint16_t* d_runs,
                               const uint32_t* d_n_runs, uint8_t* d_stream, uint64_t stream_cap, uint64_t* d_off,
                               uint32_t* d_len, uint64_t* d_total, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(d_total, 0, 2 * sizeof(uint64_t), s);
    if (e != hipSuccess || n_pairs == 0) return e;
    const uint64_t blocks = (n_pairs + ENC_TILE - 1) / ENC_TILE;            // one workgroup per tile of 32 pairs
    hipLaunchKernelGGL(encode_edits_kernel, dim3((unsigned)blocks), dim3(256), 0, s, n_pairs, d_pairs, d_runs, d_n_runs,
                       d_stream, stream_cap, d_off, d_len, d_total);
    return hipGetLastError();
}

}  // namespace scrg
