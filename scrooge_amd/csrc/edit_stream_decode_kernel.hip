// edit_stream_decode_kernel.hip — edit streams -> scrg_run pairs, on the GPU (gfx950).
//
// The receiving side of the multi-GPU gather (the step's root gets every rank's CIGARs as edit streams) and
// of a D2H in stream form: what must come out is what the reference delivers, CIGARs with one run list per window
// (src/genasm_cpu.cpp:304-305, 400-403; host side of src/genasm_gpu.cu:955-968).
//
// One pair per LANE, 64 pairs per wavefront, the branch-free state machine of edit_stream.h (decode_lane_step: one stream
// byte per step — its matches, its edit or the window end; 30 VALU instructions and no branch: the lanes of a
// wavefront are at 64 different places of their streams, so every conditional would be taken by some lane every time).
// (Rounds 3-5 sent the edits without the window ends and replayed the window loop here: 90 instructions per step, a step
// per byte and per window, the stream bytes through a ring in LDS because a step might not consume its byte.)  Around it:
//   in : a lane reads its own stream in aligned 16-byte blocks straight into registers, all lanes at the same iterations
//        (an EPOCH = the 16 steps of one block; the block after it was asked for an epoch earlier); a step takes its byte
//        from a fixed position of the block.  Bytes of the first and the last block that belong to a neighbour's stream
//        are replaced by zeros, which the state machine passes over.  A 64-byte sector of the gathered buffer is asked for
//        four times within ~64 steps and is served by the L2 after the first.
//   out: runs are staged in a 64-run ring per lane in LDS, indexed by the run's position in the OUTPUT array modulo 64,
//        and leave as aligned 64-byte pieces (four 16-byte stores per lane); only the first and the last piece of a
//        pair, which it shares with its neighbours in the dense array, go out run by run.  Pieces are written when ONE
//        lane's ring is half full, by every lane that has a whole piece: fewer, fuller passes.
// 8 KB of LDS per wavefront (count-only: none).  Bound: VALU issue, next to 0.13 GB read + 0.43 GB written per 100 k
// 10 kb pairs.
#include <hipcub/hipcub.hpp>
#include <stdlib.h>

#include "edit_stream.h"

namespace scrg {

namespace {

constexpr uint32_t DEC_RING = 64;                     // runs per lane in LDS
constexpr uint32_t DEC_PIECE = 32;                    // runs per store pass: 64 bytes
constexpr uint32_t DEC_OUT_STRIDE = 2u * DEC_RING;            // a lane's row: 64 runs = 128 bytes = one pass over the 32 banks; its 16-byte chunks are swizzled by the lane (below); 4 wavefronts x 8 KB x 4 workgroups: 16 wavefronts per CU
constexpr uint32_t DEC_WAVE_LDS = 64u * DEC_OUT_STRIDE;
constexpr uint32_t DEC_FLUSH_AT = 32;                 // final runs in one lane's ring that start a store pass (looked at once per epoch: + <= 32 runs until the next look)
constexpr int DEC_EPOCH_STEPS = 16;                   // steps between two looks at the buffers = bytes of a block (<= 2 runs per step)
// The output ring is looked at ONCE per epoch.  After a look a lane holds fewer than DEC_FLUSH_AT final runs; until the next
// look it commits at most 2 runs per step (an '=' run and an edit run: the stream "=X=X=X..." does exactly that), i.e.
// 2 * DEC_EPOCH_STEPS more, plus the slot the free-running put() writes ahead — all of which must fit the ring, or a put()
// would overwrite a run that has not been stored yet.  Today that is 32 + 32 = 64 = DEC_RING exactly: no headroom, so the
// constants are tied together here (tests/test_gpu_scale.py::test_decode_two_runs_per_step_fills_the_ring drives the worst case).
static_assert(DEC_FLUSH_AT + 2u * (uint32_t)DEC_EPOCH_STEPS <= DEC_RING,
              "decoder: runs pending after a look + runs committed until the next look must fit the output ring");
static_assert(DEC_PIECE <= DEC_FLUSH_AT && DEC_RING % DEC_PIECE == 0u, "decoder: a store pass takes whole pieces of the ring");

struct DecodeArgs {
    uint64_t n_pairs;
    const uint8_t* stream;
    uint64_t stream_bytes;
    const uint64_t* off;
    const uint32_t* len;
    const uint64_t* read_len;
    uint64_t read_len_stride;
    const uint64_t* dense_off;
    uint16_t* dense;
    uint64_t dense_cap;         // runs `dense` has room for: a pair whose segment does not lie inside is reported, nothing of it written
    uint32_t* n_runs;
    uint32_t* bad;
    const uint32_t* order;      // optional: thread t takes pair order[t] (pairs sorted by stream length, longest first)
    uint32_t together;          // whole pieces are stored by the wavefront together (write_whole_pieces); 0: every lane its own
    uint32_t gate;              // 1: the one-pair-per-wavefront kernel looks at the lengths first and takes short streams lane by lane (dec_sample_long); 2: it does so anyway
};

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t dec_ffbl(uint32_t v)      // count trailing zeros; 0xffffffff for v == 0
{
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// Which decoding the streams are for, decided where their lengths are known (the host sees only the size of the buffer, and a
// capacity-sized buffer of short streams is not a buffer of long streams): the mean length of 256 evenly spaced pairs against
// DEC_LONG_STREAM bytes.  Every wavefront of a launch computes the same answer from the same 256 dwords.
constexpr uint32_t DEC_LONG_STREAM = 64;
constexpr uint32_t DEC_PLAIN_BELOW = 16;          // below this mean: every run stored on its own (decode_pairs_plain), no staging
__device__ __forceinline__ uint64_t dec_sample_mean256(const DecodeArgs& a)       // 256 x the mean length of the sample
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t sum = 0, fine = 0;                                  // (in units of 256 bytes and of one: no overflow)
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t v = a.len[((uint64_t)(lane + 64u * j) * a.n_pairs) >> 8];       // (n_pairs < 2^56)
        sum += v >> 8;
        fine += v & 255u;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sum += (uint32_t)__shfl_xor((int)sum, d);
        fine += (uint32_t)__shfl_xor((int)fine, d);
    }
    return (uint64_t)sum * 256u + fine;
}
__device__ __forceinline__ bool dec_sample_long(const DecodeArgs& a) { return dec_sample_mean256(a) >= 256ull * DEC_LONG_STREAM; }

// Short streams — in a buffer sized for long ones (the quad kernel's launch, dec_sample_long says "short"), or so short that
// staging their few runs costs more than it saves (the lane kernel's launch, mean below DEC_PLAIN_BELOW bytes): one pair per
// lane, the shared state machine (edit_stream.h: decode_lane_step) byte by byte, every run stored where it belongs as soon as it
// is made — no staging, no sorting: at 8-14 bytes per pair (150 bp reads) a wavefront per pair would spend its time on the
// per-pair overhead (2 M pairs: 0.83 ms, this loop: scripts/r06_gate_probe.sh).  Same checks and verdicts as decode_edits_kernel.
template <bool STORE>
__device__ __forceinline__ void decode_pairs_plain(const DecodeArgs& a, uint64_t first_pair, uint64_t stride)
{
#pragma unroll 1
    for (uint64_t p = first_pair; p < a.n_pairs; p += stride) {
        uint64_t off = a.off[p], g0 = 0;
        uint32_t len = a.len[p], cap = 0;
        const uint64_t rl64 = a.read_len[p * a.read_len_stride];
        bool bad_input = off == ~0ull || off > a.stream_bytes || len > a.stream_bytes - off || len > 0x3fffffffu || rl64 > 0x7fffffffull;
        const uint32_t rl = bad_input ? 0u : (uint32_t)rl64;
        if (bad_input) { off = 0; len = 0; }
        if (STORE) {
            g0 = a.dense_off[p];
            cap = a.n_runs[p];
            if (g0 > a.dense_cap || cap > a.dense_cap - g0) {
                bad_input = true;
                g0 = 0;
                cap = 0;
            }
        }
        uint16_t* const dst0 = STORE ? a.dense + g0 : nullptr;
        DecodeLane s;
        decode_lane_init(s, 0u);
        auto put = [&](uint32_t at, uint32_t word) {
            if (STORE && (at >> 1) < cap) dst0[at >> 1] = (uint16_t)word;         // (slot n, the free one, is rewritten by run n or lies past the segment)
        };
        const uint64_t end = off + len;
        const uint32_t last_byte = len ? (uint32_t)a.stream[end - 1u] : 0u;
        uint32_t steps = 0;
#pragma unroll 1
        for (uint64_t at = off & ~3ull; at < end; at += 4u) {   // aligned dwords (whole 16-byte blocks of the buffer may be read); bytes that are not mine become zeros
            uint32_t w = *reinterpret_cast<const uint32_t*>(a.stream + at);
            const uint32_t lo = at < off ? (uint32_t)(off - at) : 0u, hi = end - at < 4u ? (uint32_t)(end - at) : 4u;
            w &= (0xffffffffu << (8u * lo)) & (0xffffffffu >> (32u - 8u * hi));
#pragma unroll
            for (int k = 0; k < 4; k++) decode_lane_step(s, (w >> (8 * k)) & 0xffu, put);
            if ((++steps & 0xfffffu) == 0u) decode_lane_guard(s);
        }
        decode_lane_guard(s);
        const bool clean = decode_lane_clean(s, last_byte, rl) && !bad_input;
        const uint32_t n_end = decode_lane_runs(s);
        if (STORE) {
            if (!clean || n_end != cap) atomicAdd(a.bad, 1u);
        } else {
            a.n_runs[p] = clean ? n_end : 0xffffffffu;
            if (!clean) atomicAdd(a.bad, 1u);
        }
    }
}

}  // namespace

// STORE = false: count only (n_runs[p] is written).  STORE = true: n_runs[p] is the size of pair p's segment of `dense`
// (nothing is written past it) and a different count is an error.
template <bool STORE>
__global__ __launch_bounds__(256, 4) void decode_edits_kernel(DecodeArgs a)
{
    __shared__ __attribute__((aligned(128))) uint8_t lds_all[STORE ? 4 * DEC_WAVE_LDS : 128];
    const uint32_t lane = threadIdx.x & 63u;
    if (a.gate == 2u || (a.gate == 1u && dec_sample_mean256(a) < 256u * DEC_PLAIN_BELOW)) {       // (uniform over the launch)
        decode_pairs_plain<STORE>(a, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
        return;
    }
    // On the root of an N > 1 job this kernel shares the SIMDs with the aligner's wavefronts, and its own run time is set by
    // its longest lanes: it goes first.  (The align kernel rotates its priorities 0..3; 3 here is at least a tie.)
    __builtin_amdgcn_s_setprio(3);
    // LDS, as 32-bit addresses: my row of the wavefront's block (128-byte aligned), and the row with my swizzle in its low bits —
    // chunk c (16 bytes) of a row sits at chunk c ^ (lane & 7), so that lanes that write the same ring position hit eight
    // different bank groups; the address of byte offset q of the ring is (q & 126) ^ row_sw, one v_bitop3.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds_all;      // (an LDS address)
    const uint32_t wave_row0 = lds0 + (STORE ? (threadIdx.x >> 6) * DEC_WAVE_LDS : 0u);
    const uint32_t row_sw = (wave_row0 + lane * DEC_OUT_STRIDE) | ((lane & 7u) << 4);
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    typedef __attribute__((address_space(3))) u32x4_t lds_u32x4;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = tid < a.n_pairs;
    // Longest streams first: the pairs of a wavefront then need about the same number of steps (the wavefront runs until
    // its last lane is done — a batch of 10 kb reads has a few pairs in ten thousand whose alignment went astray and
    // whose stream is five times the usual length), and the long ones do not start last.
    const uint64_t p = valid ? (a.order ? (uint64_t)a.order[tid] : tid) : 0;

    uint64_t off = 0, g0 = 0;
    uint32_t len = 0, rl = 0, cap = 0;
    bool bad_input = false;
    if (valid) {
        off = a.off[p];
        len = a.len[p];
        const uint64_t rl64 = a.read_len[p * a.read_len_stride];
        // a stream that is not inside the buffer (offsets and lengths may come off a wire) or a pair marked "did not
        // fit" by the encoder is reported, never read
        bad_input = off == ~0ull || off > a.stream_bytes || len > a.stream_bytes - off || len > 0x3fffffffu || rl64 > 0x7fffffffull;
        rl = bad_input ? 0u : (uint32_t)rl64;
        if (bad_input) { off = 0; len = 0; }
        if (STORE) {
            g0 = a.dense_off[p];
            cap = a.n_runs[p];
            // run counts and offsets may come off a wire too: a segment that is not inside the dense array is never written
            if (g0 > a.dense_cap || cap > a.dense_cap - g0) {
                bad_input = true;
                g0 = 0;
                cap = 0;
            }
        }
    }

    // ---- input: my stream, in aligned 16-byte blocks; first / last: my bytes within the blocks, relative to the first block ----
    const uint64_t limit16 = (a.stream_bytes + 15u) & ~15ull;   // whole 16-byte blocks of the buffer may be read
    const uint64_t stream_end = off + len;
    const uint64_t blk0 = off & ~15ull;
    const uint32_t first = (uint32_t)off & 15u, last = first + len;
    auto load_block = [&](uint64_t at) -> u32x4_t {
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (at < stream_end && at < limit16) v = *reinterpret_cast<const u32x4_t*>(a.stream + at);
        return v;
    };
    // the last byte of the stream must be a window end (decode_lane_clean)
    const uint32_t last_byte = len ? (uint32_t)a.stream[stream_end - 1u] : 0u;
    u32x4_t cur = load_block(blk0), nxt = load_block(blk0 + 16u);
    uint32_t at16 = 0;                                           // where `cur` starts, relative to the first block

    // ---- output: run k of the pair is element g0 + k of the dense array; ring slot = that index modulo 64 ----
    const uint32_t slot0 = (uint32_t)g0 & (DEC_RING - 1u);
    int32_t kf = -(int32_t)((uint32_t)g0 & (DEC_PIECE - 1u));    // runs below kf are in memory (or not mine); g0 + kf is a multiple of 32
    uint16_t* const dst0 = STORE ? a.dense + g0 : nullptr;
    DecodeLane s;
    decode_lane_init(s, 2u * slot0);                             // (q: byte offset into the ring, taken modulo 128 where it is used)
    auto put = [&](uint32_t at, uint32_t word) {
        if (STORE)
            *reinterpret_cast<lds_u16*>((uintptr_t)__builtin_amdgcn_bitop3_b32(at, 2u * DEC_RING - 2u, row_sw, 0x6A)) = (uint16_t)word;       // (at & 126) ^ row_sw
    };
    auto n_now = [&]() -> uint32_t { return decode_lane_runs(s); };
    // chunk c (0..7) of lane l's row, as that lane's swizzle placed it
    auto chunk_of = [&](uint32_t l, uint32_t c) -> uint32_t { return wave_row0 + l * DEC_OUT_STRIDE + ((c ^ (l & 7u)) << 4); };
    // the 32 runs from kf on: an aligned 64-byte piece of the output; `upto`: runs below this index exist.  Whole pieces
    // inside my segment leave as four 16-byte stores; a pair's first and last piece, which it shares with its neighbours
    // in the dense array, run by run (a loop: twice per pair)
    auto write_piece = [&](uint32_t upto) {
        const uint32_t c0 = (((slot0 + (uint32_t)kf) & (DEC_RING - 1u)) << 1) >> 4;              // 0 or 4
        const uint32_t lim = upto < cap ? upto : cap;
        const bool whole = kf >= 0 && (uint32_t)kf + DEC_PIECE <= lim;
        if (whole) {
            u32x4_t w[4];
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = *reinterpret_cast<const lds_u32x4*>((uintptr_t)chunk_of(lane, c0 + (uint32_t)k));
            u32x4_t* const d = reinterpret_cast<u32x4_t*>(dst0 + kf);
#pragma unroll
            for (int k = 0; k < 4; k++) d[k] = w[k];
        }
        if (__any(!whole)) {
            if (!whole) {
#pragma unroll 1
                for (uint32_t k = 0; k < DEC_PIECE; k++) {
                    const int32_t idx = kf + (int32_t)k;
                    if (idx >= 0 && (uint32_t)idx < lim)
                        dst0[idx] = *reinterpret_cast<const lds_u16*>((uintptr_t)(chunk_of(lane, c0 + (k >> 3)) + 2u * (k & 7u)));
                }
            }
        }
        kf += (int32_t)DEC_PIECE;
    };
    // One pass over the lanes' whole pieces, written by the wavefront TOGETHER: four lanes take the four 16-byte quarters of one
    // lane's 64-byte piece (from that lane's ring in LDS; its address and ring position by ds_bpermute), sixteen pieces per store
    // instruction — sixteen fully written 64-byte segments instead of 64 scattered 16-byte ones.  (Every lane storing its own
    // piece, the stores were what eight slots in flight waited for: 2.9 ms with them, 2.1 without.)
    auto write_whole_pieces = [&](bool mine) {            // mine: my piece at kf is whole and inside my segment
        const uint32_t my_flag = mine ? ((((slot0 + (uint32_t)kf) & (DEC_RING - 1u)) << 1) >> 4) | 8u : 0u;      // first chunk of the piece (0 or 4) | valid
        const uint64_t my_dst = (uint64_t)(uintptr_t)(dst0 + kf);
        const uint32_t q4 = lane & 3u;
        // all twelve cross-lane reads first, then the four LDS reads, then the four stores: two waits for LDS instead of eight
        uint32_t f[4], lo[4], hi[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int src = 16 * r + (int)(lane >> 2);
            f[r] = (uint32_t)__shfl((int)my_flag, src, 64);
            lo[r] = (uint32_t)__shfl((int)(uint32_t)my_dst, src, 64);
            hi[r] = (uint32_t)__shfl((int)(uint32_t)(my_dst >> 32), src, 64);
        }
        u32x4_t v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t src = 16u * (uint32_t)r + (lane >> 2);
            v[r] = *reinterpret_cast<const lds_u32x4*>((uintptr_t)chunk_of(src, (f[r] & 4u) + q4));       // (a lane without a piece: chunk q4 of its row, not stored)
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            // (non-temporal: nothing reads the dense array back in this kernel; 2.41 ms where plain stores gave 2.41-2.50)
            if (f[r] & 8u)
                __builtin_nontemporal_store(v[r], reinterpret_cast<u32x4_t*>((((uint64_t)hi[r] << 32) | lo[r]) + 16u * q4));
        }
    };
    // final runs: all but run n - 1, which may still grow while the stream has bytes left.  A store pass starts when one lane
    // has DEC_FLUSH_AT of them waiting and takes every lane's whole pieces along.
    auto flush_pieces = [&]() {
        const uint32_t n = n_now();
        const int32_t fin = (int32_t)(n - (at16 < last ? 1u : 0u));
        if (!__any(fin - kf >= (int32_t)DEC_FLUSH_AT)) return;
        for (;;) {
            const bool need = kf + (int32_t)DEC_PIECE <= fin;
            if (!__any(need)) break;
            if (!a.together) {                            // (uniform) a launch that does not fill the GPU: fewer instructions count for more
                if (need) write_piece(n);
                continue;
            }
            const uint32_t lim = n < cap ? n : cap;
            const bool whole = need && kf >= 0 && (uint32_t)kf + DEC_PIECE <= lim;
            write_whole_pieces(whole);
            if (whole) kf += (int32_t)DEC_PIECE;
            if (__any(need && !whole)) {                  // a pair's first piece (shared with its neighbour) or one past its segment: run by run
                if (need && !whole) write_piece(n);
            }
        }
    };

#ifdef SCRG_DEC_PROBE           // (probe build only, scripts/decode_timing.py --probe: shader cycles per part of the loop, summed over wavefronts)
    uint64_t pc_steps = 0, pc_flush = 0, pc_input = 0, pc_iter = 0;
    const uint64_t pc_begin = __builtin_readcyclecounter();
    const uint64_t pr_begin = __builtin_amdgcn_s_memrealtime();
#define SCRG_DEC_T(var) const uint64_t var = __builtin_readcyclecounter()
#define SCRG_DEC_ACC(acc, a_, b_) acc += (b_) - (a_)
#else
#define SCRG_DEC_T(var)
#define SCRG_DEC_ACC(acc, a_, b_)
#endif
    while (__any(at16 < last)) {
        SCRG_DEC_T(t0);
        // bytes of this block in front of my stream (the first block) or behind it (the last one, and every block after it)
        // are a neighbour's, or zeros already: they become zeros — a window end after no matches, following a window end
        const bool edge = at16 < first || (at16 < last && at16 + 16u > last);      // (blocks behind the last one were never loaded: zeros)
        if (__any(edge)) {
            if (edge) {
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const int32_t lo = (int32_t)first - (int32_t)(at16 + 4u * (uint32_t)d);       // bytes of this dword below lo are not mine,
                    const int32_t hi = (int32_t)last - (int32_t)(at16 + 4u * (uint32_t)d);        // nor those from hi on
                    const uint32_t m_lo = lo <= 0 ? 0xffffffffu : (lo >= 4 ? 0u : 0xffffffffu << (8 * lo));
                    const uint32_t m_hi = hi >= 4 ? 0xffffffffu : (hi <= 0 ? 0u : 0xffffffffu >> (32 - 8 * hi));
                    cur[d] &= m_lo & m_hi;
                }
            }
        }
        SCRG_DEC_T(t1);
#pragma unroll
        for (int it = 0; it < DEC_EPOCH_STEPS; it++) {
            decode_lane_step(s, (cur[it >> 2] >> (8 * (it & 3))) & 0xffu, put);
            // (left alone, the scheduler interleaves all sixteen steps and keeps their run words and ring addresses alive
            // side by side: 150 VGPRs, three wavefronts per SIMD instead of four)
            if ((it & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        decode_lane_guard(s);
        SCRG_DEC_T(t2);
        // The block after the next is asked for now and looked at an epoch from now.  Stores and loads share one counter
        // (vmcnt) and the wait in front of the next epoch cannot tell the stores of a data-dependent pass from the load it is
        // after: the store passes come right behind the load, an epoch (~2 us) before anybody waits.
        cur = nxt;
        at16 += 16u;
        nxt = load_block(blk0 + at16 + 16u);
        if (STORE) flush_pieces();
        SCRG_DEC_T(t3);
        SCRG_DEC_ACC(pc_input, t0, t1);
        SCRG_DEC_ACC(pc_steps, t1, t2);
        SCRG_DEC_ACC(pc_flush, t2, t3);
#ifdef SCRG_DEC_PROBE
        pc_iter++;
#endif
    }
#ifdef SCRG_DEC_PROBE
    if (lane == 0) {
        unsigned long long* const probe = reinterpret_cast<unsigned long long*>(a.bad + 2);
        const uint64_t pr_end = __builtin_amdgcn_s_memrealtime();
        atomicAdd(probe + 0, (unsigned long long)pc_iter);
        atomicAdd(probe + 1, (unsigned long long)pc_steps);
        atomicAdd(probe + 2, (unsigned long long)pc_flush);
        atomicAdd(probe + 3, (unsigned long long)pc_input);
        atomicAdd(probe + 4, (unsigned long long)(__builtin_readcyclecounter() - pc_begin));
        atomicAdd(probe + 5, (unsigned long long)(pr_end - pr_begin));                // 100 MHz ticks
        atomicMax(probe + 6, (unsigned long long)pr_begin);                           // latest start
        atomicMax(probe + 7, (unsigned long long)((1ull << 62) - pr_begin));          // earliest start
        atomicMax(probe + 8, (unsigned long long)pr_end);                             // latest end
        atomicMax(probe + 9, (unsigned long long)((1ull << 62) - pr_end));            // earliest end
    }
#endif
    const bool clean = decode_lane_clean(s, last_byte, rl) && !bad_input;
    const uint32_t n_end = n_now();
    if (STORE) {
        // what is left in the rings: whole pieces of pairs that ended since the last pass, and every pair's last, partial one
        while (__any(kf < (int32_t)n_end)) {
            if (kf < (int32_t)n_end) write_piece(n_end);
        }
        if (valid && (!clean || n_end != cap)) atomicAdd(a.bad, 1u);
    } else if (valid) {
        a.n_runs[p] = clean ? n_end : 0xffffffffu;
        if (!clean) atomicAdd(a.bad, 1u);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same decoding with one pair per WAVEFRONT, 64 stream bytes per iteration, one byte per lane: what a byte adds to the
// run list depends on the byte before it only (edit_stream.h), so the 64 bytes of a chunk are decoded side by side —
//   * the byte before comes over DPP (wave_shr:1; lane 0 gets the last byte of the chunk before);
//   * "this byte starts a run of matches" (Q), "... starts a run of edits" (H = edit and not joined to the byte before, G)
//     are wavefront masks (v_cmp into an SGPR pair); a run's index in the pair is the number of Q and H bits below the
//     lane (v_mbcnt) plus the runs of the chunks before (a scalar);
//   * an edit run's length is the number of G bits directly above its first byte (a 64-bit shift of ~G by the lane and a
//     count of trailing zeros); the run that reaches the chunk's last byte stays OPEN — its first lane does not store it but
//     keeps its length, index and letter, adds the next chunks' leading G bits and stores when a chunk's first byte does
//     not join (as three scalars carried by the wavefront, with the scalar unit finding the lane and reading it out, the
//     same cost 4 % more: the scalar unit is what this kernel waits for);
//   * bytes 0x3F (63 matches and nothing else: only W-O > 63 has them) take a side path that counts the 0x3F lanes
//     directly below each lane.
// Loads are the 64 contiguous bytes of the chunk, stores the chunk's ~106 runs as two 2-byte stores per lane to one
// contiguous range: no LDS, no sorting by length, nothing for a wavefront to wait for but its own next chunk (asked for a
// chunk ahead).  Streams shorter than a few chunks leave most lanes idle: launch_decode_edits takes the lane-per-pair
// kernel above for those.
template <bool STORE>
__global__ __launch_bounds__(256) void decode_edits_wave_kernel(DecodeArgs a, uint32_t n_waves)
{
    __builtin_amdgcn_s_setprio(3);      // (a helper between align launches: it goes first, see compact_runs_kernel)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
    // bits below my lane, as two dwords (for the side path)
    const uint32_t below_lo = lane < 32u ? (1u << lane) - 1u : 0xffffffffu, below_hi = lane < 32u ? 0u : (1u << (lane - 32u)) - 1u;
    uint32_t n_bad = 0;                                          // (uniform)
    for (uint64_t p = wave0; p < a.n_pairs; p += n_waves) {
        // ---- the pair: everything here is the same in all lanes (scalar loads)
        auto uni32 = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
        auto uni64 = [&](uint64_t v) -> uint64_t { return ((uint64_t)uni32((uint32_t)(v >> 32)) << 32) | uni32((uint32_t)v); };
        uint64_t off = uni64(a.off[p]);
        uint32_t len = uni32(a.len[p]);
        const uint64_t rl64 = uni64(a.read_len[p * a.read_len_stride]);
        // a stream that is not inside the buffer (offsets and lengths may come off a wire) or a pair marked "did not
        // fit" by the encoder is reported, never read
        bool bad = off == ~0ull || off > a.stream_bytes || len > a.stream_bytes - off || len > 0x3fffffffu || rl64 > 0x7fffffffull;
        const uint32_t rl = bad ? 0u : (uint32_t)rl64;
        if (bad) { off = 0; len = 0; }
        uint64_t g0 = 0;
        uint32_t cap = 0;
        if (STORE) {
            g0 = uni64(a.dense_off[p]);
            cap = uni32(a.n_runs[p]);
            // run counts and offsets may come off a wire too: a segment that is not inside the dense array is never written
            if (g0 > a.dense_cap || cap > a.dense_cap - g0) { bad = true; g0 = 0; cap = 0; }
        }
        const uint8_t* const src = a.stream + off;
        uint16_t* const dst = STORE ? a.dense + g0 : nullptr;
        uint32_t base = 0;                   // runs of the chunks before
        uint32_t carry_b = 0;                // the last byte of the chunk before (0: a window end)
        uint32_t carry_more = 0;             // matches of the 0x3F bytes the chunk before ended with
        // the edit run that reached the last lane of a chunk stays with the lane that started it until a later chunk's first
        // byte does not join it: its length so far (0: none), its letter word, its index in the pair (per lane; one lane at a time)
        uint32_t pend_len = 0, pend_opw = 0, pend_idx = 0;
        uint32_t placed = 0, over = 0;       // (per lane) read characters placed; bits 8..: a match run longer than 255
        // (every lane loads, a lane behind the stream its last byte, and the byte is replaced by 0 — a window end after no matches,
        // which adds nothing — where it is used: a load under a condition is waited for on the spot)
        uint32_t b_next = len ? (uint32_t)src[min(lane, len - 1u)] : 0u;
        for (uint32_t c0 = 0; c0 < len; c0 += 64u) {
            const uint32_t b = b_next & es_neg_mask(c0 + lane - len);               // (a mask by arithmetic: v_cndmask on VCC issues at a seventh of the rate, edit_stream.h)
            const uint32_t k_next = c0 + 64u + lane;
            b_next = (uint32_t)src[min(k_next, len - 1u)];                              // (v_min_u32: no compare + conditional move)
            const uint32_t pb = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_b, (int)b, 0x138, 0xf, 0xf, false);       // wave_shr:1
            const uint32_t e = b >> 6, ln = b & 63u;
            const bool is_edit = b > 63u;
            const bool joins = is_edit && b == (pb & 0xC0u);             // the same edit as the byte before, and no match in between
            const uint64_t M = __ballot(b == EDIT_MORE);
            const uint64_t E = __ballot(is_edit);
            const uint64_t G = __ballot(joins);
            uint32_t t = ln;
            if (((uint32_t)M | (uint32_t)(M >> 32) | carry_more) != 0u) {
                // the 0x3F lanes directly below me: from the highest lane below that is not one (none: all of them, and the chunk before's)
                const uint32_t z_lo = ~(uint32_t)M & below_lo, z_hi = ~(uint32_t)(M >> 32) & below_hi;
                const bool none = (z_lo | z_hi) == 0u;
                const uint32_t top = z_hi ? 63u - (uint32_t)__builtin_clz(z_hi) : 31u - (uint32_t)__builtin_clz(z_lo | 1u);
                const uint32_t cnt = none ? lane : lane - 1u - top;
                t = ln + EDIT_MORE_MATCHES * cnt + (none ? carry_more : 0u);
                if (b == EDIT_MORE || c0 + lane >= len) t = 0;                        // (nor do the bytes behind the stream close a stretch)
                const uint64_t nM = ~M;
                const uint32_t trailing = nM ? (uint32_t)__builtin_clzll(nM) : 64u;  // 0x3F lanes the chunk ends with
                carry_more = EDIT_MORE_MATCHES * trailing + (trailing == 64u ? carry_more : 0u);
                over |= t;
            }
            const uint64_t Q = __ballot(t != 0u);
            const uint64_t H = E & ~G;
            // ---- the run that stayed open: joined by this chunk's first lanes (lead of them), written by its lane when it ends
            const uint64_t nG = ~G;
            if (__any(pend_len != 0u)) {
                const uint32_t lead = nG ? (uint32_t)__builtin_ctzll(nG) : 64u;       // lanes 0 .. lead-1 join the run before them
                if (pend_len != 0u) {
                    pend_len += lead;
                    if (lead < 64u) {
                        over |= pend_len;                                             // (more than 255: reported)
                        if (STORE && pend_idx < cap) dst[pend_idx] = (uint16_t)(pend_opw | (pend_len & 0xffu));
                        pend_len = 0;
                    }
                }
            }
            // ---- run indices: the Q and H bits below my lane, on top of the runs of the chunks before
            const uint32_t eq_idx = __builtin_amdgcn_mbcnt_hi((uint32_t)(H >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)H,
                                    __builtin_amdgcn_mbcnt_hi((uint32_t)(Q >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Q, base))));
            const uint32_t head_idx = eq_idx + (t != 0u ? 1u : 0u);
            // ---- the edit runs that start here: their lengths
            const uint64_t y = (nG >> lane) >> 1;                                    // bit j: lane + 1 + j does not join
            const uint32_t y_lo = (uint32_t)y, y_hi = (uint32_t)(y >> 32);
            const uint32_t f_lo = dec_ffbl(y_lo), f_hi = min(dec_ffbl(y_hi), 32u) + 32u;          // (v_ffbl_b32: 0xffffffff for 0)
            const uint32_t above = min(min(f_lo, f_hi), 63u - lane);                 // joined lanes directly above me
            const uint32_t opw = (((0x44495800u >> ((e << 3) & 31u)) & 0xffu) << 8);
            // the run that reaches the chunk's last lane stays open with the lane that started it (the next chunk may join it)
            const bool is_head = is_edit && !joins;
            const bool stays_open = is_head && lane + above == 63u;
            if (stays_open) {
                pend_len = above + 1u;
                pend_opw = opw;
                pend_idx = head_idx;
            }
            if (STORE) {
                // (nothing past the pair's segment: one compare per store — a separate unchecked form for chunks that lie
                // inside the segment saves the compares and costs more scalar instructions, which are what this kernel is short of)
                uint8_t* const d8 = reinterpret_cast<uint8_t*>(dst);
                if (t != 0u && eq_idx < cap) *reinterpret_cast<uint16_t*>(d8 + (eq_idx << 1)) = (uint16_t)(((uint32_t)'=' << 8) | t);
                if (is_head && !stays_open && head_idx < cap) *reinterpret_cast<uint16_t*>(d8 + (head_idx << 1)) = (uint16_t)(opw | (above + 1u));
            }
            placed += ln + ((6u >> e) & 1u);
            base += (uint32_t)__popcll(Q) + (uint32_t)__popcll(H);
            carry_b = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
        }
        // ---- the end of the pair
        if (pend_len != 0u) {              // (a stream that ends in an edit: reported below, its run written all the same)
            if (STORE && pend_idx < cap) dst[pend_idx] = (uint16_t)(pend_opw | (pend_len & 0xffu));
        }
        // read characters placed, over all lanes and chunks
        uint32_t tot = placed, ov = over;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            tot += (uint32_t)__shfl_xor((int)tot, d, 64);
            ov |= (uint32_t)__shfl_xor((int)ov, d, 64);
        }
        const uint32_t last_byte = len ? (uint32_t)src[len - 1u] : 0u;
        const bool clean = !bad && carry_more == 0u && (last_byte >> 6) == 0u && last_byte != EDIT_MORE && tot == rl && (ov >> 8) == 0u;
        if (STORE) {
            if (!clean || base != cap) n_bad++;
        } else {
            if (lane == 0u) a.n_runs[p] = clean ? base : 0xffffffffu;
            if (!clean) n_bad++;
        }
    }
    if (n_bad && lane == 0u) atomicAdd(a.bad, n_bad);
}

// ---------------------------------------------------------------------------------------------------------------------
// decode_edits_quad_kernel (round 6) — one pair per wavefront, FOUR stream bytes per lane: a trip takes 256 bytes.
//
// decode_edits_wave_kernel above is bound by the scalar unit (34 scalar instructions per 64-byte chunk — ballots, popcounts, the
// open run, the loop — one scalar unit per CU: profiles/r05_decode_timing.json) and stores its ~106 runs per chunk as two
// scattered 2-byte stores per lane.  Here
//   * a lane's four bytes are classified side by side in its dword (SWAR: "has matches", "is an edit", "is the same edit as the
//     byte before and has no match in front" are bit 7 of each byte), the byte before a lane's first comes over DPP;
//   * the runs a lane starts (0..8) are counted by two shift-adds, their indices in the pair come from ONE wavefront scan
//     (six DPP adds) on top of a scalar — no ballot, no popcount;
//   * every run is WRITTEN ONCE into a ring of 16-bit slots in LDS (1024 per wavefront, slot = the run's index in the dense
//     output array modulo 1024): a stretch of matches by its byte, an edit run by its LAST byte — the one the byte after does
//     not join — whose position in the run is the run's length: a segmented count over the lane's four bytes (two shift-adds)
//     plus, where the run began in the lane before, that lane's count (DPP; a lane whose four bytes ALL join takes the other
//     path, so the hand-over never chains).  The last byte of a trip writes as far as it knows; if the next trip's first bytes
//     join that run, their last one writes the slot again.  A byte that writes nothing writes to a per-lane slot nobody reads
//     (an address select by v_bitop3, no predication, no branch);
//   * the ring leaves in aligned 16-byte UNITS of eight runs, one unit per lane and trip (one ds_read_b128,
//     ONE 16-byte store instruction per trip, 1 KB per wavefront, fully coalesced — a fixed number of memory instructions per
//     trip, so the wait for the next trip's dword does not wait for this trip's stores); only the last run stays behind (the
//     next trip may still write it), and the units a pair shares with its neighbours in the dense array (its first and its
//     last) go out run by run at the end of the pair.
// (First version of this kernel, measured and not kept: every byte ADDED its contribution to its run's slot with ds_add_u32 — no
// run lengths to compute at all — and the slots were zeroed behind the units: the LDS atomics alone cost 0.074 ms of 0.25 ms per
// 100 k pairs, the zeroing 0.024: gpurun_out/r06_dec_probe.txt.)
// What the side-by-side form cannot do takes the per-byte form for that trip (four sub-chunks of 64 bytes, one byte per lane,
// ballots and v_mbcnt as in the wave kernel; there a joining byte ADDS 1 to its run's slot): a 0x3F byte in the trip or pending
// from the one before (only W-O > 63 writes them), a lane whose four bytes all join the run before them (five equal edits in a
// row with no match between them, dword-aligned: ~1e-4 of the trips at 10 % error) — which is also where a run longer than 255
// is noticed, by counting the bytes in a row that join (`chain`).  tests/tools/quad_decoder_model.py is this arithmetic lane by
// lane in numpy; tests/test_quad_decoder_model.py holds it to the format's definition on the CPU.
constexpr uint32_t QD_RING = 1024;                      // run slots (16 bits: a scrg_run) per wavefront: <= 7 waiting for their unit + the last run + 512 a trip
constexpr uint32_t QD_RING_BYTES = 2u * QD_RING;
constexpr uint32_t QD_WAVE_LDS = QD_RING_BYTES + 256u;  // + a dword per lane for the bytes that write nothing
// (9 KB per workgroup of four wavefronts: on the root of an N > 1 job this kernel runs beside the align kernel, whose four workgroups
// per CU hold 147 of the CU's 160 KB — a decoder workgroup of 17 KB, as with 32-bit slots, only started where an align workgroup
// had ended, and the two kernels took turns instead of sharing the CUs)

template <int SH>
__device__ __forceinline__ uint32_t qd_lshl_add_t(uint32_t a, uint32_t c)
{
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(SH), "v"(c));
    return r;
}
#define qd_lshl_add(a, sh, c) qd_lshl_add_t<sh>(a, c)
template <int SH>
__device__ __forceinline__ uint32_t qd_lshl_or_t(uint32_t a, uint32_t c)
{
    uint32_t r;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(SH), "v"(c));
    return r;
}
#define qd_lshl_or(a, sh, c) qd_lshl_or_t<sh>(a, c)
// truth tables: bit index = a * 4 + b * 2 + c (genasm_device.h: bitop3_table)
__device__ __forceinline__ uint32_t qd_or_and(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xA8); }      // (a | b) & c
__device__ __forceinline__ uint32_t qd_xor_and(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x78); }     // a ^ (b & c)

// inclusive prefix sum over the 64 lanes (values small enough not to overflow)
__device__ __forceinline__ uint32_t qd_wave_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);       // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);       // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);       // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);       // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);       // row_bcast:15 -> rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);       // row_bcast:31 -> rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t qd_total(uint32_t v)         // the sum over the 64 lanes, in every lane (uniform)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)qd_wave_scan(v), 63);
}

template <bool STORE>
__global__ __launch_bounds__(256) void decode_edits_quad_kernel(DecodeArgs a, uint32_t n_waves)
{
    __builtin_amdgcn_s_setprio(3);      // (a helper between align launches: it goes first, see compact_runs_kernel)
    __shared__ __attribute__((aligned(4 * QD_RING_BYTES))) uint32_t ring_all[STORE ? 4 * (QD_WAVE_LDS / 4u) : 32];      // (a slot's address is slot offset | ring base: the four rings first)
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    typedef __attribute__((address_space(3))) u32x4_t lds_u32x4;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t wave0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + wave));
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)ring_all;
    const uint32_t ring_b = lds0 + (STORE ? wave * QD_RING_BYTES : 0u);
    const uint32_t dump_b = lds0 + (STORE ? 4u * QD_RING_BYTES + wave * 256u + lane * 4u : 0u);
    auto uni32 = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto uni64 = [&](uint64_t v) -> uint64_t { return ((uint64_t)uni32((uint32_t)(v >> 32)) << 32) | uni32((uint32_t)v); };
    // LDS address of run slot `slot2 / 2` (slot2: 2 x the run's index in the dense array; any value, taken modulo the ring)
    auto slot_addr = [&](uint32_t slot2) -> uint32_t { return __builtin_amdgcn_bitop3_b32(slot2, QD_RING_BYTES - 2u, ring_b, 0xEA); };      // (slot2 & mask) | ring_b
    // `value` to that slot where bit `BIT` of `flags` is set, to the lane's own dump slot where it is not
    auto ring_put = [&](uint32_t slot2, uint32_t value, uint32_t flags, auto bit_tag) {
        constexpr int BIT = decltype(bit_tag)::value;
        const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)flags, BIT, 1);                       // 0 / ~0
        *(lds_u16*)(uintptr_t)__builtin_amdgcn_bitop3_b32(slot_addr(slot2), dump_b, m, 0xE4) = (uint16_t)value;  // (a & m) | (b & ~m)
    };
    uint32_t n_bad = 0;                                          // (uniform)
    // A pair's five numbers (every lane loads the same addresses; made scalars where they are used) are asked for while the pair
    // before it is decoded: a wavefront's pairs are a chain of dependent loads otherwise — numbers, first dwords, trips — and a
    // 10 kb pair is only five trips long.
    struct PairMeta { uint64_t off, rl, g0; uint32_t len, cap; };
    auto load_meta = [&](uint64_t p) -> PairMeta {
        PairMeta m = {0, 0, 0, 0, 0};
        if (p < a.n_pairs) {                                     // (uniform)
            m.off = a.off[p];
            m.len = a.len[p];
            m.rl = a.read_len[p * a.read_len_stride];
            if (STORE) {
                m.g0 = a.dense_off[p];
                m.cap = a.n_runs[p];
            }
        }
        return m;
    };
    PairMeta next_meta = load_meta(wave0);                       // (asked for before the sample of the lengths is waited for)
    if (a.gate == 2u || (a.gate == 1u && !dec_sample_long(a))) {        // (uniform over the launch)
        decode_pairs_plain<STORE>(a, (uint64_t)wave0 * 64u + lane, (uint64_t)n_waves * 64u);
        return;
    }
    for (uint64_t p = wave0; p < a.n_pairs; p += n_waves) {
        // ---- the pair: everything here is the same in all lanes
        const PairMeta meta = next_meta;
        next_meta = load_meta(p + n_waves);
        uint64_t off = uni64(meta.off);
        uint32_t len = uni32(meta.len);
        const uint64_t rl64 = uni64(meta.rl);
        // a stream that is not inside the buffer (offsets and lengths may come off a wire) or a pair marked "did not
        // fit" by the encoder is reported, never read
        bool bad = off == ~0ull || off > a.stream_bytes || len > a.stream_bytes - off || len > 0x3fffffffu || rl64 > 0x7fffffffull;
        const uint32_t rl = bad ? 0u : (uint32_t)rl64;
        if (bad) { off = 0; len = 0; }
        uint64_t g0 = 0;
        uint32_t cap = 0;
        if (STORE) {
            g0 = uni64(meta.g0);
            cap = uni32(meta.cap);
            // run counts and offsets may come off a wire too: a segment that is not inside the dense array is never written
            if (g0 > a.dense_cap || cap > a.dense_cap - g0) { bad = true; g0 = 0; cap = 0; }
        }
        // ---- input: dwords of the stream from its 4-byte aligned start; a stream that starts off a dword boundary takes two
        // loads per lane and one v_alignbit.  Every lane loads — a lane behind the stream the last dword that holds a stream
        // byte (inside the buffer: it is readable up to the next multiple of 16) — and the bytes behind the stream become zeros
        // where they are used: a window end after no matches, which writes nothing.
        const uint32_t delta = (uint32_t)off & 3u;
        const uint8_t* const src4 = a.stream + (off - delta);
        const uint32_t last_dw = len ? (delta + len - 1u) & ~3u : 0u;
        uint32_t base = 0;                   // runs so far
        // what a trip hands to the next one stays in VECTOR registers (lane 0 holds what the trip's last lane had: one DPP
        // wave_ror:1, and it is the `old` operand of the next trip's wave_shr:1): a scalar round trip (v_readlane, s_mul, v_mov)
        // costs the scalar unit, which all of a CU's wavefronts share
        uint32_t cx_v = 0;                   // (lane 0) the dword in front of the trip's first (its top byte: the byte before lane 0's first)
        uint32_t crp_v = 0;                  // (lane 0) x 0x01010101: the length so far of the edit run that reaches the end of the trip before (0: none does)
        uint32_t jn_prev = 0;                // Jn of the side-by-side trip before (its last lane's top bytes continue the chain); 0 after a per-byte trip
        uint32_t carry_more = 0;             // matches of the 0x3F bytes the trip before ended with
        uint32_t chain = 0;                  // bytes in a row, up to the end of the trip before, that join the edit run before them (+ jn_prev's)
        uint64_t force_other = 0;            // (0 / ~0) 0x3F matches pending, or a chain that is getting long: the next trip takes the per-byte path
        uint32_t over_s = 0;                 // (uniform) an edit run longer than 255
        uint32_t placed = 0, over = 0;       // (per lane) read characters placed; bits 8..: a run of matches longer than 255
        uint32_t cnt_acc = 0;                // (per lane, count only) runs started
        uint32_t x_last = 0;                 // the last trip's dwords (the stream's last byte is looked at when the pair ends)
        // Units of eight runs (16 bytes of the dense array).  Positions are counted from the start of the unit that holds the pair's
        // first run: run r of the pair is at position h + r, h = g0 % 8; unit u covers positions 8u .. 8u + 7; units below `uf` have
        // left the ring.  (32-bit arithmetic; 64 bits only where an address is formed.)
        const uint32_t g0l = (uint32_t)g0;                                  // (slot arithmetic is modulo the ring: the low bits do)
        const uint32_t h = g0l & 7u;
        const uint32_t unit0 = (uint32_t)(g0 >> 3);                         // (modulo 2^32: the ring only looks at its low bits)
        uint16_t* const dense_u = STORE ? a.dense + (g0 - h) : nullptr;     // position 0
        uint32_t uf = 0;
        uint32_t head_w = 0;                                                // (lanes 0..7) the slots of the pair's first unit, if the pair starts inside one: stored when the pair ends
        bool head_kept = false, head_pending = h != 0u;                     // (uniform)
        const uint32_t unit_first = h != 0u ? 1u : 0u, unit_stop = (h + cap) >> 3;
        auto read_unit = [&](uint32_t u) -> u32x4_t {
            return *(const lds_u32x4*)(uintptr_t)(ring_b + (((unit0 + u) & (QD_RING / 8u - 1u)) << 4));
        };
        // The trip loop asks for the next trip's dwords at its top and stores this trip's units at its bottom.  Loads and stores
        // share one counter (vmcnt) and complete in order; the compiler, which cannot know whether the (predicated) store of a
        // trip was issued, would wait for the counter to reach 0 in front of the next trip — i.e. for the store it has just
        // issued.  So both are issued where the compiler does not count them (inline assembly: exactly one or two loads, then
        // exactly ONE store instruction per trip — issued under an EXEC mask, possibly with no lane, never skipped), and the wait
        // at the trip's bottom is written out: vmcnt(1) — the loads have landed, the store may still be on its way.  (Memory
        // operations the compiler does not know about can only make ITS waits longer: completion is in order.)
        auto load_x_async = [&](uint32_t c0) -> uint32_t {                                             // (complete after qd_wait_loads)
            const uint32_t at = min(c0 + 4u * lane, last_dw);
            uint32_t w0;
            asm volatile("global_load_dword %0, %1, %2" : "=v"(w0) : "v"(at), "s"(src4));
            if (delta == 0u) return w0;                                                                // (uniform)
            const uint32_t at1 = min(at + 4u, last_dw);
            uint32_t w1;
            asm volatile("global_load_dword %0, %1, %2" : "=v"(w1) : "v"(at1), "s"(src4));
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0), "+v"(w1));                                  // (streams off a dword boundary: the plain way)
            return __builtin_amdgcn_alignbit(w1, w0, 8u * delta);
        };
        // (the first trip's dwords the same way: a load the compiler counts would make it wait — for everything — at the loop's top)
        uint32_t x = 0;
        if (len) {                                                                                     // (uniform)
            x = load_x_async(0u);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(x));
        }
        for (uint32_t c0 = 0; c0 < len; c0 += 256u) {
            if (c0 + 256u > len) {                                                                      // (uniform) the last trip: zeros behind the stream
                const int32_t rem = (int32_t)len - (int32_t)(c0 + 4u * lane);
                const uint32_t keep = rem >= 4 ? 0xffffffffu : (rem <= 0 ? 0u : (0xffffffffu >> (32 - 8 * rem)));
                x &= keep;
            }
            uint32_t x_next = load_x_async(c0 + 256u);                                                  // (behind the stream: its last dword again — never used)
            x_last = x;
            // ---- the four bytes side by side.  Bit 7 of byte k of ...
            const uint32_t px = (uint32_t)__builtin_amdgcn_update_dpp((int)cx_v, (int)x, 0x138, 0xf, 0xf, false);       // wave_shr:1; lane 0: the trip before
            const uint32_t pv = __builtin_amdgcn_alignbit(x, px, 24);                               // byte k: the byte before byte k
            const uint32_t T = x & 0x3F3F3F3Fu;                                                     // the matches in front
            const uint32_t Em = (T + 0x7F7F7F7Fu) & 0x80808080u;                                    // ... Em: there are matches (a run of '=')
            const uint32_t OPB = x & 0xC0C0C0C0u;
            const uint32_t Ed = qd_or_and(OPB, OPB << 1, 0x80808080u);                              // ... Ed: an edit
            const uint32_t D = qd_xor_and(x, pv, 0xC0C0C0C0u);                                      // 0 where the byte is the edit before it, bare
            const uint32_t nz = ((D & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | D;                              // bit 7: the byte of D is not 0
            const uint32_t Hd = nz & Ed;                                                            // ... Hd: an edit that starts a run
            const uint32_t Jn = es_andn(Ed, nz);                                                    // ... Jn: an edit that joins the run before it
            const uint32_t More = es_a_nb_c(T + 0x41414141u, Ed, 0x80808080u);                      // ... More: the byte is 0x3F
            const uint64_t trigger = __ballot(More != 0u) | __ballot(Jn == 0x80808080u) | force_other;
            if (trigger != 0ull) {
                // ---- one byte per lane, four times (see the head of the kernel)
                uint32_t carry_b = (uint32_t)__builtin_amdgcn_readfirstlane((int)cx_v) >> 24;
                // (after a side-by-side trip the chain is what joins at that trip's very end: the top bytes of its last lane's Jn)
                chain += (uint32_t)__builtin_clz(~((uint32_t)__builtin_amdgcn_readlane((int)jn_prev, 63) | 0x7F7F7F7Fu) | 1u) >> 3;
                jn_prev = 0;
                uint32_t b = 0;
#pragma unroll 1
                for (uint32_t sub = 0; sub < 4u; sub++) {
                    const uint32_t xs = (uint32_t)__shfl((int)x, (int)(sub * 16u + (lane >> 2)), 64);
                    b = (xs >> (8u * (lane & 3u))) & 0xffu;
                    const uint32_t pb = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_b, (int)b, 0x138, 0xf, 0xf, false);
                    const uint32_t e = b >> 6, ln = b & 63u;
                    const bool is_edit = b > 63u;
                    const bool joins = is_edit && b == (pb & 0xC0u);
                    const uint64_t M = __ballot(b == EDIT_MORE);
                    const uint64_t G = __ballot(joins);
                    uint32_t t = ln;
                    if (M != 0ull || carry_more != 0u) {
                        // the 0x3F lanes directly below me: from the highest lane below that is not one (none: all of them, and the trip before's)
                        const uint64_t below = (1ull << lane) - 1ull;
                        const uint64_t z = ~M & below;
                        const bool none = z == 0ull;
                        const uint32_t top = none ? 0u : 63u - (uint32_t)__builtin_clzll(z);
                        const uint32_t cnt = none ? lane : lane - 1u - top;
                        t = ln + EDIT_MORE_MATCHES * cnt + (none ? carry_more : 0u);
                        if (b == EDIT_MORE || c0 + 64u * sub + lane >= len) t = 0;                    // (nor do the bytes behind the stream close a stretch)
                        const uint64_t nM = ~M;
                        const uint32_t trailing = nM ? (uint32_t)__builtin_clzll(nM) : 64u;          // 0x3F lanes the sub-chunk ends with
                        carry_more = EDIT_MORE_MATCHES * trailing + (trailing == 64u ? carry_more : 0u);
                        over |= t;
                    }
                    const uint64_t Q = __ballot(t != 0u);
                    const uint64_t H = __ballot(is_edit && !joins);
                    // the bytes in a row that join: 255 of them make a run of 256
                    const uint64_t nG = ~G;
                    if (nG == 0ull) {
                        chain += 64u;
                    } else {
                        if (chain + (uint32_t)__builtin_ctzll(nG) >= 255u) over_s = 1u;
                        chain = (uint32_t)__builtin_clzll(nG);
                    }
                    if (chain >= 255u) over_s = 1u;
                    if (STORE) {
                        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(H >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)H,
                                                __builtin_amdgcn_mbcnt_hi((uint32_t)(Q >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Q, g0l + base))));
                        const uint32_t q1 = t != 0u ? 1u : 0u, h1 = (is_edit && !joins) ? 1u : 0u;
                        const uint32_t opw = ((0x44495800u >> ((e << 3) & 31u)) & 0xffu) << 8;
                        // matches and heads are written, then the joining bytes add 1 each (LDS operations of a wavefront run in order)
                        if (q1) *(lds_u16*)(uintptr_t)slot_addr(before << 1) = (uint16_t)(((uint32_t)'=' << 8) | t);
                        if (h1) *(lds_u16*)(uintptr_t)slot_addr((before + q1) << 1) = (uint16_t)(opw | 1u);
                        asm volatile("" ::: "memory");
                        if (joins) {                               // (the slot's dword, 1 in the slot's half: a count never carries — 256 is reported)
                            const uint32_t at = slot_addr((before - 1u) << 1);
                            (void)__hip_atomic_fetch_add((lds_u32*)(uintptr_t)(at & ~3u), 1u << ((at & 2u) << 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                    placed += ln + ((6u >> e) & 1u);
                    base += (uint32_t)__popcll(Q) + (uint32_t)__popcll(H);
                    carry_b = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
                }
                force_other = (carry_more != 0u || chain >= 248u) ? ~0ull : 0ull;
                // what the next side-by-side trip's first lane is handed: the length so far of the edit run that reaches this trip's end
                crp_v = carry_b > 63u ? ((chain + 1u) & 0xffu) * 0x01010101u : 0u;
            } else {
                // ---- runs started per byte (0..2), their inclusive prefix over the lane's bytes (<= 8 in the top byte), the lanes' prefix
                const uint32_t e7 = Em >> 7;
                const uint32_t R = e7 + (Hd >> 7);
                const uint32_t P1 = qd_lshl_add(R, 8, R);                                                               // (as written: the optimiser makes
                const uint32_t P = qd_lshl_add(P1, 16, P1);                                                             //  a quarter-rate v_mul_lo_u32 of the two)
                const uint32_t c = P >> 24;
                if (STORE) {
                    const uint32_t incl = qd_wave_scan(c);
                    // 2 x (index in the dense array of the first run this lane starts); 2 x the prefix within the lane, byte by byte
                    const uint32_t i2 = (incl - c + (g0l + base)) << 1;
                    const uint32_t P2 = P << 1;
                    // ---- an edit byte's position in its run: a segmented count over the lane's bytes (byte k: E_k, + the count of byte
                    // k - 1 if byte k joins, in two doubling steps), plus what the lane before hands over where bytes 0..k all join
                    const uint32_t E1 = Ed >> 7;
                    const uint32_t Jm = Jn | (Jn - (Jn >> 7));                                                          // 0xFF in the bytes that join
                    const uint32_t v1 = qd_lshl_add(E1 & (Jm >> 8), 8, E1);
                    const uint32_t f1 = Jm & (Jm << 8);
                    const uint32_t v2 = qd_lshl_add(v1 & (f1 >> 16), 16, v1);
                    const uint32_t g1 = Jm & qd_lshl_or(Jm, 8, 0xFFu);
                    const uint32_t F = g1 & qd_lshl_or(g1, 16, 0xFFFFu);                                                // bytes 0..k all join
                    const uint32_t B = __builtin_amdgcn_perm(v2, v2, 0x03030303u);                                      // the count of the lane's last byte, in all four
                    const uint32_t prB = (uint32_t)__builtin_amdgcn_update_dpp((int)crp_v, (int)B, 0x138, 0xf, 0xf, false);          // wave_shr:1; lane 0: the trip before
                    const uint32_t RP = v2 + (F & prB);
                    // an edit byte is the last of its run unless the byte after it joins (the last lane's last byte: as far as it knows)
                    const uint32_t Jnx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Jn, 0x130, 0xf, 0xf, false);       // wave_shl:1
                    const uint32_t Tl = es_andn(Ed, __builtin_amdgcn_alignbit(Jnx, Jn, 8));
                    // what the bytes write: matches '=' << 8 | t, the last byte of an edit run letter << 8 | its position
                    const uint32_t EQL = qd_or_and(Em - e7, Em, 0x3D3D3D3Du);                                            // '=' in the bytes that have matches
                    const uint32_t LET = __builtin_amdgcn_perm(0u, 0x44495800u, (x >> 6) & 0x03030303u);                 // "\0XID"[code]
                    const uint32_t p0 = P2 & 0xffu, p1 = (P2 >> 8) & 0xffu, p2 = (P2 >> 16) & 0xffu, p3 = P2 >> 24;
                    ring_put(i2, __builtin_amdgcn_perm(EQL, T, 0x0C0C0400u), Em, std::integral_constant<int, 7>{});
                    ring_put(i2 + p0 - 2u, __builtin_amdgcn_perm(LET, RP, 0x0C0C0400u), Tl, std::integral_constant<int, 7>{});
                    ring_put(i2 + p0, __builtin_amdgcn_perm(EQL, T, 0x0C0C0501u), Em, std::integral_constant<int, 15>{});
                    ring_put(i2 + p1 - 2u, __builtin_amdgcn_perm(LET, RP, 0x0C0C0501u), Tl, std::integral_constant<int, 15>{});
                    ring_put(i2 + p1, __builtin_amdgcn_perm(EQL, T, 0x0C0C0602u), Em, std::integral_constant<int, 23>{});
                    ring_put(i2 + p2 - 2u, __builtin_amdgcn_perm(LET, RP, 0x0C0C0602u), Tl, std::integral_constant<int, 23>{});
                    ring_put(i2 + p2, __builtin_amdgcn_perm(EQL, T, 0x0C0C0703u), Em, std::integral_constant<int, 31>{});
                    ring_put(i2 + p3 - 2u, __builtin_amdgcn_perm(LET, RP, 0x0C0C0703u), Tl, std::integral_constant<int, 31>{});
                    base += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    crp_v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)B, 0x13C, 0xf, 0xf, false);                              // wave_ror:1: lane 0 <- the last lane
                } else {
                    cnt_acc += c;
                }
                // read characters placed: the matches, and one per X or I (edit codes 1 and 2: bits 7 and 6 differ)
                placed = __builtin_amdgcn_sad_u8(T, 0u, placed);
                placed += (uint32_t)__builtin_popcount(((x >> 1) ^ x) & 0x40404040u);
                // the bytes at the trip's end that join the run before them (never all four of lane 63: that is the other path) are
                // counted when the other path is next entered; a chain that came in ended in this trip
                jn_prev = Jn;
                chain = 0;
            }
            cx_v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x13C, 0xf, 0xf, false);                                        // wave_ror:1: lane 0 <- the last lane
            if (STORE) {
                // Every unit below the last run — which the next trip may still write — is final: at most 64 of them (7 + 512 runs), one
                // per lane, ONE store instruction.  A unit that sticks out of the segment is not stored here: the pair's first unit is kept
                // by lanes 0..7 (a slot each) for the end of the pair; a unit past a segment that is too small (the pair is reported) is dropped.
                asm volatile("" ::: "memory");
                // (signed: no run yet and h = 0 make u_lim -1; units [u_first, u_stop) are stored: from the first unit that lies wholly in
                // the segment — unit 1 when the pair starts inside unit 0 — to the last unit wholly below the segment's end)
                const int32_t u_lim = (int32_t)(h + base - 1u) >> 3;
                const uint32_t u = uf + lane;
                const u32x4_t o = read_unit(u);
                const int32_t u_first = max((int32_t)uf, (int32_t)unit_first), u_stop = min(u_lim, (int32_t)unit_stop);
                const uint64_t inside = __ballot(u - (uint32_t)u_first < (uint32_t)max(u_stop - u_first, 0));
                uint16_t* const dst = dense_u + 8ull * u;
                uint64_t exec_keep;
                asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\tglobal_store_dwordx4 %2, %3, off nt\n\ts_mov_b64 exec, %0"
                             : "=&s"(exec_keep) : "s"(inside), "v"(dst), "v"(o) : "memory", "scc");       // (s_and_b64 writes SCC)
                if (head_pending) {                                                                     // (uniform) the pair starts inside unit 0 ...
                    if (u_lim > 0) {                                                                    // ... which leaves the ring now
                        head_w = *(const lds_u16*)(uintptr_t)slot_addr((8u * unit0 + (lane & 7u)) << 1);
                        head_kept = true;
                        head_pending = false;
                    }
                }
                uf = (uint32_t)max((int32_t)uf, u_lim);
                asm volatile("s_waitcnt vmcnt(1)" : "+v"(x_next));
            } else {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(x_next));
            }
            x = x_next;
        }
        // ---- the end of the pair
        if (!STORE) base += qd_total(cnt_acc);
        if (STORE) {
            asm volatile("" ::: "memory");
            // what has not left the ring goes out run by run, one run per lane: the pair's first unit where it was kept (lanes 0..7),
            // and the at most two units from `uf` on — the last run's and the one before it (lanes 0..15)
            const uint32_t lim = h + min(base, cap);
            if (head_kept) {                                                                            // (uniform)
                if (lane < 8u && lane >= h && lane < lim) dense_u[lane] = (uint16_t)head_w;
            }
            const uint32_t at = 8u * uf + lane;
            if (lane < 16u && at >= h && at < lim)
                dense_u[at] = *(const lds_u16*)(uintptr_t)slot_addr((8u * unit0 + at) << 1);
        }
        // read characters placed, over all lanes (64 bits: a lane's share stays below 2^30, their sum may not)
        const uint64_t tot = __any((placed >> 24) != 0u) ? (uint64_t)qd_total(placed & 0xffffu) + ((uint64_t)qd_total(placed >> 16) << 16)
                                                         : (uint64_t)qd_total(placed);            // (64 lanes below 2^24: no overflow)
        const bool ov = __any((over >> 8) != 0u) || over_s != 0u;
        // the stream's last byte must be a window end: it sits in the last trip's dwords
        const uint32_t last_byte = len ? ((uint32_t)__builtin_amdgcn_readlane((int)x_last, (int)(((len - 1u) & 255u) >> 2)) >> (8u * ((len - 1u) & 3u))) & 0xffu : 0u;
        const bool clean = !bad && carry_more == 0u && (last_byte >> 6) == 0u && last_byte != EDIT_MORE && tot == (uint64_t)rl && !ov;
        if (STORE) {
            if (!clean || base != cap) n_bad++;
        } else {
            if (lane == 0u) a.n_runs[p] = clean ? base : 0xffffffffu;
            if (!clean) n_bad++;
        }
    }
    if (n_bad && lane == 0u) atomicAdd(a.bad, n_bad);
}

__global__ void iota_kernel(uint32_t* v, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}

// Bits of the stream length the order is made from: 16-byte granularity, lengths up to 1 MB told apart.
constexpr int DEC_SORT_BEGIN_BIT = 4, DEC_SORT_END_BIT = 20;

// Which decoding a launch takes, in two steps.  The HOST picks the launch from the size of the buffer over the number of pairs —
// all it can see: one pair per wavefront (decode_edits_quad_kernel) at 64 bytes per pair and more, one pair per lane below.  The
// DEVICE then looks at the lengths themselves (dec_sample_mean256): the quad launch takes streams that turn out short lane by lane
// (a capacity-sized buffer of 150 bp reads' streams), and the lane launch drops its staging below 16 bytes per pair.  Measured, ms
// per launch, scripts/r06_gate_sweep.sh (profiles/r06_decoder_choice.json), plain / lane / quad: 2 M x 150 bp at 8 bytes per pair
// 0.043 / 0.12 / 0.83; 1 M x 300 bp ONT (41) 0.36 / 0.16 / 0.42; 600 k x 500 bp (67) 0.36 / 0.25 / 0.25; 300 k x 1 kb (132)
// 0.41 / 0.21 / 0.13; 75 k x 4 kb (521) 0.74 / 0.29 / 0.065; 100 k x 10 kb (1 299) 2.1 / 0.81 / 0.15.
// SCRG_DEC_KERNEL=lane|wave|quad|plain overrides (the tests run them on the same inputs; `wave` is round 5's one-byte-per-lane
// kernel, kept as the independent formulation).
static int decode_kernel_forced()                                          // -1: no override (read per call: the tests switch it)
{
    const char* e = getenv("SCRG_DEC_KERNEL");
    return !e ? -1 : e[0] == 'l' ? 0 : e[0] == 'w' ? 1 : e[0] == 'q' ? 2 : e[0] == 'p' ? 3 : -1;          // (p: the quad kernel's launch, every stream lane by lane)
}
int decode_kernel_choice(uint64_t n_pairs, uint64_t stream_bytes)          // 0: lane, 1: wave, 2: quad
{
    if (decode_kernel_forced() >= 0) return decode_kernel_forced();
    return n_pairs != 0 && stream_bytes / n_pairs >= DEC_LONG_STREAM ? 2 : 0;
}
bool decode_by_wavefront(uint64_t n_pairs, uint64_t stream_bytes) { return decode_kernel_choice(n_pairs, stream_bytes) != 0; }

size_t decode_sort_temp_bytes(uint64_t n_pairs)
{
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                                       (uint32_t*)nullptr, (int)n_pairs, DEC_SORT_BEGIN_BIT, DEC_SORT_END_BIT, (hipStream_t)0);
    return bytes;
}

// sort_ws: null (pairs are taken in index order) or a work area of 3 * n_pairs uint32 followed by decode_sort_temp_bytes()
// bytes (256-byte aligned): the pairs are then taken longest stream first.
hipError_t launch_decode_edits(uint64_t n_pairs, const uint8_t* d_stream, uint64_t stream_bytes,
                               const uint64_t* d_off, const uint32_t* d_len, const uint64_t* d_read_len,
                               uint64_t read_len_stride, const uint64_t* d_dense_off, uint16_t* d_dense, uint64_t dense_cap,
                               uint32_t* d_n_runs, uint32_t* d_bad, void* sort_ws, size_t sort_temp_bytes, hipStream_t s)
{
    if (n_pairs == 0) return hipSuccess;
    // Stored together, the pieces cost a third of the write time when launches fill the GPU (8 slots of 100 k pairs: 2.83 -> 2.47 ms,
    // 4 slots 1.64 -> 1.43) and a few more instructions per pass, which is what a launch of <= 2 wavefronts per SIMD feels
    // (1 slot: 1.06 -> 1.15 ms): by the size of the launch (200 000 pairs = three wavefronts on every SIMD of an MI355X).
    const uint32_t together = n_pairs > 200000 ? 1u : 0u;
    const uint32_t* order = nullptr;
    DecodeArgs a{n_pairs, d_stream, stream_bytes, d_off, d_len, d_read_len, read_len_stride, d_dense_off, d_dense, dense_cap, d_n_runs, d_bad, order, together, 0u};
    // Which kernel (decode_by_wavefront; scripts/decode_timing.py, 10 kb reads: one slot of 100 k pairs 0.30 ms by wavefront
    // against 0.82 ms by lane — 1 563 lane-per-pair wavefronts leave the GPU half empty —, eight slots 2.26 against 2.40 ms)
    const int choice = decode_kernel_choice(n_pairs, stream_bytes);
    if (choice != 0) {
        const uint64_t want = n_pairs < 8192u ? (n_pairs + 3u) & ~3ull : 8192u;          // 8 wavefronts on every SIMD of an MI355X
        const uint32_t n_waves = (uint32_t)want;
        const dim3 grid(n_waves / 4u), block(256);
        if (choice >= 2) {
            // The buffer is large enough for long streams; whether the streams ARE long only the device knows (a capacity-sized
            // buffer of 150 bp reads' streams — 8-14 bytes a pair — is no case for a wavefront per pair): the kernel looks at a
            // sample of the lengths first (a.gate) and takes short streams one pair per lane.
            a.gate = decode_kernel_forced() < 0 ? 1u : decode_kernel_forced() == 3 ? 2u : 0u;
            if (d_dense) hipLaunchKernelGGL(decode_edits_quad_kernel<true>, grid, block, 0, s, a, n_waves);
            else hipLaunchKernelGGL(decode_edits_quad_kernel<false>, grid, block, 0, s, a, n_waves);
        } else {
            if (d_dense) hipLaunchKernelGGL(decode_edits_wave_kernel<true>, grid, block, 0, s, a, n_waves);
            else hipLaunchKernelGGL(decode_edits_wave_kernel<false>, grid, block, 0, s, a, n_waves);
        }
        return hipGetLastError();
    }
    // Longest first pays for streams long enough to differ by many steps — which this kernel only sees when it is asked for by
    // name (long streams go to the other kernel).  On the short streams it is chosen for, the sort cost more than the decoding
    // (profiles/r06_decoder_choice.json: 1 M x 300 bp 0.30 ms sorted, 0.16 not; 2 M x 150 bp 0.21 / 0.12).
    if (sort_ws && n_pairs < 0x7fffffffull && (stream_bytes / n_pairs >= DEC_LONG_STREAM || getenv("SCRG_DEC_SORT"))) {
        uint32_t* const idx = static_cast<uint32_t*>(sort_ws);
        uint32_t* const keys_out = idx + n_pairs;
        uint32_t* const idx_out = keys_out + n_pairs;
        void* const temp = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(idx_out + n_pairs) + 255u) & ~(uintptr_t)255u);
        hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, s, idx, (uint32_t)n_pairs);
        size_t tb = sort_temp_bytes;
        hipError_t e = hipcub::DeviceRadixSort::SortPairsDescending(temp, tb, d_len, keys_out, idx, idx_out, (int)n_pairs,
                                                                    DEC_SORT_BEGIN_BIT, DEC_SORT_END_BIT, s);
        if (e != hipSuccess) return e;
        order = idx_out;
    }
    a.order = order;
    a.gate = decode_kernel_forced() < 0 ? 1u : 0u;
    const dim3 grid((unsigned)((n_pairs + 255) / 256)), block(256);
    if (d_dense) hipLaunchKernelGGL(decode_edits_kernel<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(decode_edits_kernel<false>, grid, block, 0, s, a);
    return hipGetLastError();
}

}  // namespace scrg
