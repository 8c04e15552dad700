// edit_stream_decode_kernel.hip — edit streams -> scrg_run pairs with the window breaks restored, on the GPU (gfx950).
//
// The receiving side of the multi-GPU gather (rank 0 gets every rank's CIGARs as edit streams) and of a
// D2H in stream form: what must come out is what the reference delivers, CIGARs with one run list per window
// (src/genasm_cpu.cpp:304-305, 400-403; host side of src/genasm_gpu.cu:955-968).
//
// One pair per LANE, 64 pairs per wavefront, the branch-free state machine of edit_stream.h (decode_lane_step: one
// stream byte, its matches cut at the window limits, the edit, the window end — O(edits + windows) steps per pair, no
// per-base work, about 90 VALU instructions per step and no branch: the lanes of a wavefront are at 64 different places
// of their streams, so every conditional would be taken by some lane every time).  Around it:
//   in : a lane reads its own stream in aligned 16-byte blocks, all lanes at the same iterations (a block is touched a
//        whole epoch of 16 steps after its load was issued), into a ring of two blocks per lane in LDS; a step looks at one
//        byte of it, asked for (ds_read_u8) a whole step earlier — as soon as the step before knows whether it consumes its
//        own byte.  (Round 3 and the first half of round 4 kept three blocks in registers and fed a 64-bit shift register one
//        dword at a time through a select tree: 49 of the 383 VALU instructions per four steps.)  A 64-byte sector of the
//        gathered buffer is asked for four times within ~100 steps and is served by the L2 after the first.
//   out: runs are staged in a 64-run ring per lane in LDS, indexed by the run's position in the OUTPUT array modulo 64,
//        and leave as aligned 64-byte pieces (four 16-byte stores per lane); only the first and the last piece of a
//        pair, which it shares with its neighbours in the dense array, go out run by run.  Pieces are written when ONE
//        lane's ring is three quarters full, by every lane that has a whole piece: fewer, fuller passes.
// 10 KB of LDS per wavefront (count-only: 3 KB).  Bound: VALU issue, next to 0.1 GB read + 0.43 GB written per 100 k
// 10 kb pairs.
#include <hipcub/hipcub.hpp>

#include "edit_stream.h"

namespace scrg {

namespace {

constexpr uint32_t DEC_RING = 64;                     // runs per lane in LDS
constexpr uint32_t DEC_PIECE = 32;                    // runs per store pass: 64 bytes
constexpr uint32_t DEC_IN_RING = 32;                   // bytes of the lane's stream in LDS: two aligned 16-byte blocks
constexpr uint32_t DEC_OUT_STRIDE = 2u * DEC_RING + DEC_IN_RING;      // a lane's row: [64 runs | 32 stream bytes]; 16-byte aligned; 4 wavefronts x 10 KB x 4 workgroups = the CU's 160 KB (they fit: 16 wavefronts per CU)
constexpr uint32_t DEC_COUNT_STRIDE = DEC_IN_RING + 16u;              // count only: the stream bytes alone
constexpr uint32_t DEC_WAVE_LDS = 64u * DEC_OUT_STRIDE;
constexpr uint32_t DEC_FLUSH_AT = 32;                 // final runs in one lane's ring that start a store pass (looked at once per epoch: + <= 32 runs until the next look)
constexpr int DEC_STEPS_PER_CHECK = 4;                // steps between two looks at the buffers (<= 2 runs and 1 byte per step)
constexpr uint32_t DEC_EPOCH = 4;                     // iterations between two block moves: 16 steps, at most 16 bytes
// The output ring is looked at ONCE per epoch.  After a look a lane holds fewer than DEC_FLUSH_AT final runs; until the next
// look it commits at most 2 runs per step (an '=' run and an edit run: the stream "=X=X=X..." does exactly that), i.e.
// 2 * DEC_EPOCH * DEC_STEPS_PER_CHECK more, plus the slot the free-running put() writes ahead — all of which must fit the ring, or
// a put() would overwrite a run that has not been stored yet.  Today that is 32 + 32 = 64 = DEC_RING exactly: no headroom, so the
// constants are tied together here (tests/test_gpu_scale.py::test_decode_two_runs_per_step_fills_the_ring drives the worst case).
static_assert(DEC_FLUSH_AT + 2u * DEC_EPOCH * (uint32_t)DEC_STEPS_PER_CHECK <= DEC_RING,
              "decoder: runs pending after a look + runs committed until the next look must fit the output ring");
static_assert(DEC_PIECE <= DEC_FLUSH_AT && DEC_RING % DEC_PIECE == 0u, "decoder: a store pass takes whole pieces of the ring");

struct DecodeArgs {
    uint64_t n_pairs;
    uint32_t W, O;
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
};

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

}  // namespace

// STORE = false: count only (n_runs[p] is written).  STORE = true: n_runs[p] is the size of pair p's segment of `dense`
// (nothing is written past it) and a different count is an error.
template <bool STORE>
__global__ __launch_bounds__(256) void decode_edits_kernel(DecodeArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[STORE ? 4 * DEC_WAVE_LDS : 4 * 64 * DEC_COUNT_STRIDE];
    const uint32_t lane = threadIdx.x & 63u;
    // On the root of an N > 1 job this kernel shares the SIMDs with the aligner's wavefronts, and its own run time is set by
    // its longest lanes: it goes first.  (The align kernel rotates its priorities 0..3; 3 here is at least a tie.)
    __builtin_amdgcn_s_setprio(3);
    uint8_t* const out_me = lds_all + (STORE ? (threadIdx.x >> 6) * DEC_WAVE_LDS + lane * DEC_OUT_STRIDE : 0u);
    uint8_t* const in_me = STORE ? out_me + 2u * DEC_RING : lds_all + threadIdx.x * DEC_COUNT_STRIDE;
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
    DecodeLane s;
    decode_lane_init(s, a.W, a.O, 0u, len, rl);
    if (!valid) s.aliveM = 0;

    // ---- input: my stream, in aligned 16-byte blocks, through a ring of two blocks in LDS ----
    const uint64_t limit16 = (a.stream_bytes + 15u) & ~15ull;   // whole 16-byte blocks of the buffer may be read
    const uint64_t stream_end = off + len;
    uint64_t blk = off & ~15ull;                                 // the older block of the ring
    auto load_block = [&](uint64_t at) -> u32x4_t {
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (at < stream_end && at < limit16) v = *reinterpret_cast<const u32x4_t*>(a.stream + at);
        return v;
    };
    // The ring holds the block the lane is in and the next one; the one after that is being LOADED into registers.  Blocks
    // move up (the loaded one into the ring, the next load issued) only every DEC_EPOCH iterations, for all lanes at once: a
    // block is first touched a whole epoch after its load was issued, so the wait in front of the move costs nothing — issued
    // lane by lane as blocks run out, some lane's load would be seconds old at every look and the wavefront would wait for
    // memory each time.  rp = my position relative to the first block; base = the older ring block's.  At a move rp < base + 32
    // (a lane uses at most 16 bytes per epoch and was inside the older block after the move before), after it rp < base + 16.
    uint32_t rp = (uint32_t)off & 15u, base = 0;
    u32x4_t nx2;
    {
        const u32x4_t b0 = load_block(blk), b1 = load_block(blk + 16u);
        nx2 = load_block(blk + 32u);
        *reinterpret_cast<u32x4_t*>(in_me) = b0;
        *reinterpret_cast<u32x4_t*>(in_me + 16u) = b1;
    }
    auto advance_blocks = [&]() {
        const bool rot = rp - base >= 16u;
        if (__any(rot)) {
            if (rot) {
                *reinterpret_cast<u32x4_t*>(in_me + (base & 16u)) = nx2;       // over the block that has been used up
                base += 16u;
                blk += 16u;
                nx2 = load_block(blk + 32u);
            }
        }
    };
    auto byte_at = [&](uint32_t r) -> uint32_t { return in_me[r & (DEC_IN_RING - 1u)]; };
    uint32_t bn = byte_at(rp);                                   // the byte the next step looks at

    // ---- output: run k of the pair is element g0 + k of the dense array; ring slot = that index modulo 64 ----
    const uint32_t slot0 = (uint32_t)g0 & (DEC_RING - 1u);
    int32_t kf = -(int32_t)((uint32_t)g0 & (DEC_PIECE - 1u));    // runs below kf are in memory (or not mine); g0 + kf is a multiple of 32
    uint16_t* const dst0 = STORE ? a.dense + g0 : nullptr;
    auto put = [&](uint32_t k, uint32_t word) {
        if (STORE) *reinterpret_cast<uint16_t*>(out_me + (((slot0 + k) & (DEC_RING - 1u)) << 1)) = (uint16_t)word;
    };
    // the 32 runs from kf on: an aligned 64-byte piece of the output; `upto`: runs below this index exist
    auto write_piece = [&](uint32_t upto) {
        const u32x4_t* const src = reinterpret_cast<const u32x4_t*>(out_me + (((slot0 + (uint32_t)kf) & (DEC_RING - 1u)) << 1));
        u32x4_t w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = src[k];
        const uint32_t lim = upto < cap ? upto : cap;
        const bool whole = kf >= 0 && (uint32_t)kf + DEC_PIECE <= lim;
        if (whole) {
            u32x4_t* const d = reinterpret_cast<u32x4_t*>(dst0 + kf);
#pragma unroll
            for (int k = 0; k < 4; k++) d[k] = w[k];
        }
        if (__any(!whole)) {
            if (!whole) {
#pragma unroll
                for (int k = 0; k < (int)DEC_PIECE; k++) {
                    const int32_t idx = kf + k;
                    const uint32_t dw = w[k >> 3][(k >> 1) & 3];
                    if (idx >= 0 && (uint32_t)idx < lim) dst0[idx] = (uint16_t)(dw >> (16 * (k & 1)));
                }
            }
        }
        kf += (int32_t)DEC_PIECE;
    };
    // final runs: all but run n - 1, which may still grow while the pair is alive.  A store pass starts when one lane
    // has DEC_FLUSH_AT of them waiting and takes every lane's whole pieces along.
    // One pass over the lanes' whole pieces, written by the wavefront TOGETHER: four lanes take the four 16-byte quarters of one
    // lane's 64-byte piece (from that lane's ring in LDS; its address and ring position by ds_bpermute), sixteen pieces per store
    // instruction — sixteen fully written 64-byte segments instead of 64 scattered 16-byte ones.  (Every lane storing its own
    // piece, the stores were what eight slots in flight waited for: 2.9 ms with them, 2.1 without.)
    uint8_t* const wave_out = lds_all + (STORE ? (threadIdx.x >> 6) * DEC_WAVE_LDS : 0u);
    auto write_whole_pieces = [&](bool mine) {            // mine: my piece at kf is whole and inside my segment
        const uint32_t my_flag = mine ? (((slot0 + (uint32_t)kf) & (DEC_RING - 1u)) << 1) | 1u : 0u;      // ring offset (0 or 64) | valid
        const uint64_t my_dst = (uint64_t)(uintptr_t)(dst0 + kf);
        const uint32_t q = lane & 3u;
        const uint64_t have = __ballot(mine);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (((have >> (16 * r)) & 0xffffull) == 0) continue;                 // none of these sixteen lanes has a piece (uniform)
            const int src = 16 * r + (int)(lane >> 2);
            const uint32_t f = (uint32_t)__shfl((int)my_flag, src, 64);
            const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)my_dst, src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(my_dst >> 32), src, 64);
            if (f & 1u) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(wave_out + (uint32_t)src * DEC_OUT_STRIDE + (f & ~1u) + 16u * q);
                // (non-temporal: nothing reads the dense array back in this kernel; 2.41 ms where plain stores gave 2.41-2.50)
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>((((uint64_t)hi << 32) | lo) + 16u * q));
            }
        }
    };
    auto flush_pieces = [&]() {
        const int32_t fin = (int32_t)(s.n - (s.aliveM & 1u));
        if (!__any(fin - kf >= (int32_t)DEC_FLUSH_AT)) return;
        for (;;) {
            const bool need = kf + (int32_t)DEC_PIECE <= fin;
            if (!__any(need)) break;
            if (!a.together) {                            // (uniform) a launch that does not fill the GPU: fewer instructions count for more
                if (need) write_piece(s.n);
                continue;
            }
            const uint32_t lim = s.n < cap ? s.n : cap;
            const bool whole = need && kf >= 0 && (uint32_t)kf + DEC_PIECE <= lim;
            write_whole_pieces(whole);
            if (whole) kf += (int32_t)DEC_PIECE;
            if (__any(need && !whole)) {                  // a pair's first piece (shared with its neighbour) or one past its segment: run by run
                if (need && !whole) write_piece(s.n);
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
    for (uint32_t iter = 0;; iter++) {
        SCRG_DEC_T(t0);
#pragma unroll
        for (int it = 0; it < DEC_STEPS_PER_CHECK; it++) {
            // the blocks move before the LAST step of an epoch asks for its successor's byte: that byte may be the first of
            // the block that arrives with the move (15 steps since the move before: rp < base + 31, the step's own byte is in
            // the ring and already on its way)
            if (it == DEC_STEPS_PER_CHECK - 1 && (iter & (DEC_EPOCH - 1u)) == DEC_EPOCH - 1u) advance_blocks();
            uint32_t next = 0;
            (void)decode_lane_step(s, bn, put, [&](uint32_t takeM) {
                rp -= takeM;
                next = byte_at(rp);
                __builtin_amdgcn_sched_barrier(0);       // (left alone, the scheduler sinks the read to its use, a step later: every step then waits for LDS)
            });
            bn = next;
        }
        SCRG_DEC_T(t1);
        // Stores and the waits for stream blocks share one counter (vmcnt), and the wait in front of a block move cannot tell
        // the stores of a data-dependent pass from the loads it is after: it waits for everything.  So the store passes run in
        // the FIRST iteration of an epoch and the block move in the LAST (above): a store has three iterations (~2 us) to be
        // acknowledged before anybody waits (passes in any iteration: the wait of the next move met stores a few hundred
        // nanoseconds old — 8 slots 2.99 ms with stores against 1.99 ms without; now see DESIGN.md §3.7).
        if (STORE && (iter & (DEC_EPOCH - 1u)) == 0u) flush_pieces();
        SCRG_DEC_T(t2);
        SCRG_DEC_T(t3);
        SCRG_DEC_ACC(pc_steps, t0, t1);
        SCRG_DEC_ACC(pc_flush, t1, t2);
        SCRG_DEC_ACC(pc_input, t2, t3);
#ifdef SCRG_DEC_PROBE
        pc_iter++;
#endif
        if (!__any(s.aliveM != 0u)) break;
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
    const bool clean = decode_lane_clean(s) && !bad_input;
    if (STORE) {
        // what is left in the rings: whole pieces of pairs that ended since the last pass, and every pair's last, partial one
        while (__any(kf < (int32_t)s.n)) {
            if (kf < (int32_t)s.n) write_piece(s.n);
        }
        if (valid && (!clean || s.n != cap)) atomicAdd(a.bad, 1u);
    } else if (valid) {
        a.n_runs[p] = clean ? s.n : 0xffffffffu;
        if (!clean) atomicAdd(a.bad, 1u);
    }
}

__global__ void iota_kernel(uint32_t* v, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}

// Bits of the stream length the order is made from: 16-byte granularity, lengths up to 1 MB told apart.
constexpr int DEC_SORT_BEGIN_BIT = 4, DEC_SORT_END_BIT = 20;

size_t decode_sort_temp_bytes(uint64_t n_pairs)
{
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                                       (uint32_t*)nullptr, (int)n_pairs, DEC_SORT_BEGIN_BIT, DEC_SORT_END_BIT, (hipStream_t)0);
    return bytes;
}

// sort_ws: null (pairs are taken in index order) or a work area of 3 * n_pairs uint32 followed by decode_sort_temp_bytes()
// bytes (256-byte aligned): the pairs are then taken longest stream first.
hipError_t launch_decode_edits(uint64_t n_pairs, uint32_t W, uint32_t O, const uint8_t* d_stream, uint64_t stream_bytes,
                               const uint64_t* d_off, const uint32_t* d_len, const uint64_t* d_read_len,
                               uint64_t read_len_stride, const uint64_t* d_dense_off, uint16_t* d_dense, uint64_t dense_cap,
                               uint32_t* d_n_runs, uint32_t* d_bad, void* sort_ws, size_t sort_temp_bytes, hipStream_t s)
{
    if (n_pairs == 0) return hipSuccess;
    const uint32_t* order = nullptr;
    if (sort_ws && n_pairs < 0x7fffffffull) {
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
    // Stored together, the pieces cost a third of the write time when launches fill the GPU (8 slots of 100 k pairs: 2.83 -> 2.47 ms,
    // 4 slots 1.64 -> 1.43) and a few more instructions per pass, which is what a launch of <= 2 wavefronts per SIMD feels
    // (1 slot: 1.06 -> 1.15 ms): by the size of the launch (200 000 pairs = three wavefronts on every SIMD of an MI355X).
    const uint32_t together = n_pairs > 200000 ? 1u : 0u;
    DecodeArgs a{n_pairs, W, O, d_stream, stream_bytes, d_off, d_len, d_read_len, read_len_stride, d_dense_off, d_dense, dense_cap, d_n_runs, d_bad, order, together};
    const dim3 grid((unsigned)((n_pairs + 255) / 256)), block(256);
    if (d_dense) hipLaunchKernelGGL(decode_edits_kernel<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(decode_edits_kernel<false>, grid, block, 0, s, a);
    return hipGetLastError();
}

}  // namespace scrg
