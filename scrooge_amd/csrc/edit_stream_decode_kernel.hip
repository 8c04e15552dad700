// edit_stream_decode_kernel.hip — edit streams -> scrg_run pairs with the window breaks restored, on the GPU (gfx950).
//
// The receiving side of the multi-GPU gather (rank 0 gets every rank's CIGARs as edit streams) and of a
// D2H in stream form: what must come out is what the reference delivers, CIGARs with one run list per window
// (src/genasm_cpu.cpp:304-305, 400-403; host side of src/genasm_gpu.cu:955-968).
//
// One pair per LANE, 64 pairs per wavefront, the state machine of edit_stream.h (decode_lane_step: one stream byte,
// its matches cut at the window limits, the edit, the window end — O(edits + windows) steps per pair, no per-base work).
// What makes it fast is the memory side:
//   in : a lane's stream is consumed byte by byte, but it is FETCHED by the wavefront: 64-byte chunks, four lanes
//        x 16 bytes per pair and chunk, sixteen pairs per load instruction, into a 128-byte ring per lane in LDS
//        (every 64-byte block of the gathered buffer is read exactly once, as one request);
//   out: runs are staged in a 32-run ring per lane in LDS, indexed by the run's position in the OUTPUT array modulo 32,
//        and leave as aligned 32-byte pieces (two 16-byte stores); only the first and the last piece of a pair, which
//        it shares with its neighbours in the dense array, go out run by run.
// 13.25 KB of LDS per wavefront, 12 wavefronts per CU.  Bound: VALU issue (about 60 instructions per step, ~1300 steps
// for a 10 kb read at 10 % error) next to 0.1 GB read + 0.43 GB written per 100 k pairs.
#include "edit_stream.h"

namespace scrg {

namespace {

constexpr uint32_t DEC_IN_RING = 128;                 // bytes of stream per lane in LDS
constexpr uint32_t DEC_IN_STRIDE = DEC_IN_RING + 16;  // 16-byte aligned (ds_write_b128), lanes spread over the banks
constexpr uint32_t DEC_CHUNK = 64;                    // bytes fetched per pair and refill
constexpr uint32_t DEC_OUT_STRIDE = 68;               // 32 runs + one dword
constexpr uint32_t DEC_WAVE_LDS = 64u * (DEC_IN_STRIDE + DEC_OUT_STRIDE);
constexpr int DEC_STEPS_PER_CHECK = 4;                // steps between two looks at the rings (<= 2 runs and 1 byte per step)

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
    uint32_t* n_runs;
    uint32_t* bad;
};

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

}  // namespace

// STORE = false: count only (n_runs[p] is written).  STORE = true: n_runs[p] is the size of pair p's segment of `dense`
// (nothing is written past it) and a different count is an error.
template <bool STORE>
__global__ __launch_bounds__(256) void decode_edits_kernel(DecodeArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[4 * DEC_WAVE_LDS];
    const uint32_t lane = threadIdx.x & 63u;
    uint8_t* const wave_lds = lds_all + (threadIdx.x >> 6) * DEC_WAVE_LDS;
    uint8_t* const in_me = wave_lds + lane * DEC_IN_STRIDE;
    uint8_t* const out_me = wave_lds + 64u * DEC_IN_STRIDE + lane * DEC_OUT_STRIDE;
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = p < a.n_pairs;

    uint64_t off = 0, g0 = 0;
    uint32_t len = 0, rl = 0, cap = 0;
    bool bad_input = false;
    if (valid) {
        off = a.off[p];
        len = a.len[p];
        const uint64_t rl64 = a.read_len[p * a.read_len_stride];
        // a stream that is not inside the buffer (offsets and lengths may come off a wire) or a pair marked "did not
        // fit" by the encoder is reported, never read
        bad_input = off == ~0ull || off > a.stream_bytes || len > a.stream_bytes - off || rl64 > 0x7fffffffull;
        rl = bad_input ? 0u : (uint32_t)rl64;
        if (bad_input) { off = 0; len = 0; }
        if (STORE) {
            g0 = a.dense_off[p];
            cap = a.n_runs[p];
        }
    }
    const uint64_t base = off & ~(uint64_t)(DEC_CHUNK - 1u);    // the lane's stream positions are relative to this block
    const uint64_t limit16 = (a.stream_bytes + 15u) & ~15ull;   // whole 16-byte blocks of the buffer may be read
    uint32_t loaded = 0;                                        // stream bytes in my ring: [loaded - 128, loaded), a multiple of 64

    DecodeLane s;
    decode_lane_init(s, a.W, a.O, (uint32_t)(off - base), (uint32_t)(off - base) + len, rl);
    if (!valid) s.alive = 0;

    // ---- input: the wavefront fetches the next 64-byte chunk of every lane that has room for it ----
    auto refill = [&]() {
        const bool want = s.alive && loaded < s.end && (int32_t)(loaded - s.pos) <= (int32_t)(DEC_IN_RING - DEC_CHUNK);
        if (!__any(want)) return;
#pragma unroll
        for (uint32_t r = 0; r < 4; r++) {
            const int q = (int)(16u * r + (lane >> 2));          // the pair (lane) this lane fetches for, and which quarter
            const uint32_t sub = (lane & 3u) * 16u;
            const uint64_t qbase = (uint64_t)__shfl((long long)base, q, 64);
            const uint32_t qloaded = (uint32_t)__shfl((int)loaded, q, 64);
            const bool qwant = __shfl((int)want, q, 64) != 0;
            const uint64_t at = qbase + qloaded + sub;
            if (qwant && at < limit16) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(a.stream + at);
                *reinterpret_cast<u32x4_t*>(wave_lds + (uint32_t)q * DEC_IN_STRIDE + ((qloaded + sub) & (DEC_IN_RING - 1u))) = v;
            }
        }
        if (want) loaded += DEC_CHUNK;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    refill();
    refill();
    // two bytes of lookahead in registers: the byte at pos and the one after it
    uint32_t bn0 = in_me[s.pos & (DEC_IN_RING - 1u)], bn1 = in_me[(s.pos + 1u) & (DEC_IN_RING - 1u)];

    // ---- output: run k of the pair is element g0 + k of the dense array; ring slot = that index modulo 32 ----
    const uint32_t slot0 = (uint32_t)g0 & 31u;
    int32_t kf = -(int32_t)((uint32_t)g0 & 15u);                // runs below kf are in memory (or not mine); g0 + kf is a multiple of 16
    uint16_t* const dst0 = STORE ? a.dense + g0 : nullptr;
    auto put = [&](uint32_t k, uint32_t word) {
        if (STORE) *reinterpret_cast<uint16_t*>(out_me + (((slot0 + k) & 31u) << 1)) = (uint16_t)word;
    };
    // the 16 runs from kf on: an aligned 32-byte piece of the output; `upto`: runs below this index exist
    auto write_piece = [&](uint32_t upto) {
        const uint32_t* const src = reinterpret_cast<const uint32_t*>(out_me + (((slot0 + (uint32_t)kf) & 31u) << 1));
        uint32_t w[8];
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = src[k];
        const uint32_t lim = upto < cap ? upto : cap;
        const bool whole = kf >= 0 && (uint32_t)kf + 16u <= lim;
        if (whole) {
            uint4* const d = reinterpret_cast<uint4*>(dst0 + kf);
            d[0] = make_uint4(w[0], w[1], w[2], w[3]);
            d[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
        if (__any(!whole)) {
            if (!whole) {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int32_t idx = kf + k;
                    if (idx >= 0 && (uint32_t)idx < lim) dst0[idx] = (uint16_t)(w[k >> 1] >> (16 * (k & 1)));
                }
            }
        }
        kf += 16;
    };
    // every piece whose runs are final (run n - 1 may still grow while the pair is alive)
    auto flush_pieces = [&]() {
        for (;;) {
            const bool need = kf + 16 <= (int32_t)(s.n - (s.alive ? 1u : 0u));
            if (!__any(need)) break;
            if (need) write_piece(s.n);
        }
    };

    for (;;) {
#pragma unroll
        for (int it = 0; it < DEC_STEPS_PER_CHECK; it++) {
            decode_lane_step(s, [&]() { return bn0; }, [&]() { bn0 = bn1; }, put);
            bn1 = in_me[(s.pos + 1u) & (DEC_IN_RING - 1u)];
        }
        if (STORE) flush_pieces();
        refill();
        if (!__any(s.alive != 0)) break;
    }
    const bool clean = decode_lane_clean(s) && !bad_input;
    if (STORE) {
        // the last, partial piece
        while (__any(kf < (int32_t)s.n)) {
            if (kf < (int32_t)s.n) write_piece(s.n);
        }
        if (valid && (!clean || s.n != cap)) atomicAdd(a.bad, 1u);
    } else if (valid) {
        a.n_runs[p] = clean ? s.n : 0xffffffffu;
        if (!clean) atomicAdd(a.bad, 1u);
    }
}

hipError_t launch_decode_edits(uint64_t n_pairs, uint32_t W, uint32_t O, const uint8_t* d_stream, uint64_t stream_bytes,
                               const uint64_t* d_off, const uint32_t* d_len, const uint64_t* d_read_len,
                               uint64_t read_len_stride, const uint64_t* d_dense_off, uint16_t* d_dense, uint32_t* d_n_runs,
                               uint32_t* d_bad, hipStream_t s)
{
    if (n_pairs == 0) return hipSuccess;
    DecodeArgs a{n_pairs, W, O, d_stream, stream_bytes, d_off, d_len, d_read_len, read_len_stride, d_dense_off, d_dense, d_n_runs, d_bad};
    const dim3 grid((unsigned)((n_pairs + 255) / 256)), block(256);
    if (d_dense) hipLaunchKernelGGL(decode_edits_kernel<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(decode_edits_kernel<false>, grid, block, 0, s, a);
    return hipGetLastError();
}

}  // namespace scrg
