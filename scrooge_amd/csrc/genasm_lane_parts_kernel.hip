// genasm_lane_parts_kernel.hip — the lane-per-pair aligner for 64 <= W-O <= 127 (W <= 256; and W > 128 with W-O <= 63): the formulation of
// genasm_lane_kernel.hip (every lane aligns its own pair; the window's table holds the differences of the edit-distance
// matrix behind the GenASM bitvectors, src/genasm_cpu.cpp:210-409 — see the header of that file for why this gives the
// reference's edit distance and CIGAR bit for bit) with multi-word vectors (NW = ceil(W/64) words of 64 pattern rows,
// word 0 the most significant; src/bitvector.hpp:45-48, 124-139) for the reference's large-window sweep points
// (scripts/profile.py:180-185: W = 160 ... 256 with O = W/2 + 1), where a window's traceback may consume up to 127
// characters: a table row is two 64-bit words, W-O columns of two such rows are 8 (W-O) dwords = 4 KB per lane — thirty
// times the registers of a wavefront.  genasm_lane_mw_kernel.hip keeps that table in HBM (0.5 MB written and read back per
// wavefront and window: 3.4 TB/s with the VALU idle half of the time).
//
// Here the table exists only in PARTS of 16 columns, all held in the same 128 registers.  The recurrence runs from the
// last text column down, the walk from column 0 up, so:
//   1. ONE sweep over all W columns (a run-time loop over chunks of 16 columns, the 16 unrolled) keeps the table of
//      columns 0..15 and, on its way, leaves a CHECKPOINT — the difference vectors Pv, Mv in front of a chunk, 4 NW
//      dwords per lane — for each of the chunks 1 .. ceil((W-O)/16) - 1 in a slab of HBM (word-interleaved over the
//      lanes: 512 contiguous bytes per store; 7 x 64 bytes per lane and window at W = 256 instead of 2 x 4 KB);
//   2. the walk consumes part 0; then, part by part, the chunk's 16 columns are swept AGAIN from their checkpoint,
//      this time keeping the table, and the walk goes on from where it stood.
// W + (W-O) - 16 swept columns per window instead of W, no table traffic, no data-dependent slow path.  Chunks that lie
// past the end of every lane's text are skipped (a column past the end leaves the vectors as they are).
//
// Each part ends with its own second pass (masks -> runs or edit-stream bytes: the 32-bit code of genasm_lane_kernel,
// 16 columns at a time); a run that crosses from one part into the next is ONE run of the window (the reference merges
// within a window, src/genasm_cpu.cpp:372-404): see genasm_lane_wide_kernel.hip, whose two halves work the same way.
// tests/proto/lane_proto.c (lane_align_codes_mw) restates the multi-word arithmetic; tests/test_gpu_parity.py holds this
// kernel against the CPU checker, the table-in-HBM kernel and the reference-built fixtures at W/O = 192/97, 200/50,
// 256/129, 128/20.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "genasm_kernels.h"
#include "genasm_device.h"

namespace scrg {

namespace {

constexpr int PT_COLS = 16;                      // columns per part / chunk
constexpr uint32_t PT_RING_BYTES = 68;           // 32 runs + one dword: lanes land on distinct LDS banks
constexpr uint32_t PT_SCRATCH_BYTES = 20;        // insertion-run length of each column of a part, one byte each (+ bank skew)
constexpr int PT_EQ_AHEAD = 2;                   // Eq words are read from LDS this many columns ahead of their use

constexpr int PT_XH = bitop3_table([](int sum, int pv, int eq) { return (sum ^ pv) | eq; });
constexpr int PT_PH = bitop3_table([](int mv, int xh, int pv) { return mv | ~(xh | pv); });
constexpr int PT_PVN = bitop3_table([](int mhs, int xv, int phs) { return mhs | ~(xv | phs); });
constexpr int PT_NOR3 = bitop3_table([](int a, int b, int c) { return ~(a | b | c); });
constexpr int PT_ANDN = bitop3_table([](int a, int b, int) { return a & ~b; });
constexpr int PT_BFI = bitop3_table([](int a, int b, int c) { return (a & c) | (b & ~c); });
constexpr int PT_ANDOR = bitop3_table([](int a, int b, int c) { return (a & b) | c; });
constexpr int PT_V0 = bitop3_table([](int pvn, int ph, int xh) { return pvn | ~(ph | xh); });

typedef uint32_t pt_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) pt_u32x2 pt_lds_u32x2;
__device__ __forceinline__ uint2 pt_lds_read64(uint32_t addr)
{
    const pt_u32x2 v = *reinterpret_cast<const pt_lds_u32x2*>((uintptr_t)addr);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void pt_lds_write64(uint32_t addr, uint2 v)
{
    pt_u32x2 w;
    w.x = v.x;
    w.y = v.y;
    *reinterpret_cast<pt_lds_u32x2*>((uintptr_t)addr) = w;
}
__device__ __forceinline__ uint32_t pt_ffbh(uint32_t v)      // count leading zeros; 0xffffffff for v == 0
{
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ uint64_t pt_shl64(uint64_t v, uint32_t s)      // one v_lshlrev_b64 (count modulo 64)
{
    uint64_t r;
    asm("v_lshlrev_b64 %0, %1, %2" : "=v"(r) : "v"(s), "v"(v));
    return r;
}
__device__ __forceinline__ uint64_t pt_shr64(uint64_t v, uint32_t s)
{
    uint64_t r;
    asm("v_lshrrev_b64 %0, %1, %2" : "=v"(r) : "v"(s), "v"(v));
    return r;
}
__device__ __forceinline__ uint32_t pt_clz64(uint64_t v)                  // 64 for v == 0 (through 0xffffffff + 32 -> min)
{
    return min(pt_ffbh((uint32_t)(v >> 32)), pt_ffbh((uint32_t)v) + 32u);
}
// the 64 bits of the 128-bit row {w0 (rows 0..63, row r at bit 63-r), w1 (rows 64..127)} from row s on (s < 128), zeros after the row's end
__device__ __forceinline__ uint64_t pt_from_row(uint64_t w0, uint64_t w1, uint32_t s)
{
    const bool far = s >= 64u;
    const uint64_t hi = far ? w1 : w0, lo = far ? 0ull : w1;
    const uint32_t b = s & 63u;
    return pt_shl64(hi, b) | pt_shr64(lo >> 1, 63u - b);
}

// The difference vectors between two text columns: NW 64-bit words, word 0 the most significant (bit 63-k of word w
// belongs to pattern character 64 w + k), each as two dwords (.x low, .y high).
template <int NW> struct PtState {
    uint2 pv[NW], mv[NW];
};

// a + b over 2 NW dwords, least significant first (word NW-1 low dword ... word 0 high dword): one carry chain
template <int NW> __device__ __forceinline__ void pt_add_chain(const uint32_t (&a)[2 * NW], const uint32_t (&b)[2 * NW], uint32_t (&s)[2 * NW])
{
    if constexpr (NW == 2) {
        asm("v_add_co_u32 %0, vcc, %4, %8\n\t"
            "v_addc_co_u32 %1, vcc, %5, %9, vcc\n\t"
            "v_addc_co_u32 %2, vcc, %6, %10, vcc\n\t"
            "v_addc_co_u32 %3, vcc, %7, %11, vcc"
            : "=&v"(s[0]), "=&v"(s[1]), "=&v"(s[2]), "=&v"(s[3])
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3])
            : "vcc");
    } else if constexpr (NW == 3) {
        asm("v_add_co_u32 %0, vcc, %6, %12\n\t"
            "v_addc_co_u32 %1, vcc, %7, %13, vcc\n\t"
            "v_addc_co_u32 %2, vcc, %8, %14, vcc\n\t"
            "v_addc_co_u32 %3, vcc, %9, %15, vcc\n\t"
            "v_addc_co_u32 %4, vcc, %10, %16, vcc\n\t"
            "v_addc_co_u32 %5, vcc, %11, %17, vcc"
            : "=&v"(s[0]), "=&v"(s[1]), "=&v"(s[2]), "=&v"(s[3]), "=&v"(s[4]), "=&v"(s[5])
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5])
            : "vcc");
    } else {
        asm("v_add_co_u32 %0, vcc, %8, %16\n\t"
            "v_addc_co_u32 %1, vcc, %9, %17, vcc\n\t"
            "v_addc_co_u32 %2, vcc, %10, %18, vcc\n\t"
            "v_addc_co_u32 %3, vcc, %11, %19, vcc\n\t"
            "v_addc_co_u32 %4, vcc, %12, %20, vcc\n\t"
            "v_addc_co_u32 %5, vcc, %13, %21, vcc\n\t"
            "v_addc_co_u32 %6, vcc, %14, %22, vcc\n\t"
            "v_addc_co_u32 %7, vcc, %15, %23, vcc"
            : "=&v"(s[0]), "=&v"(s[1]), "=&v"(s[2]), "=&v"(s[3]), "=&v"(s[4]), "=&v"(s[5]), "=&v"(s[6]), "=&v"(s[7])
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
              "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
            : "vcc");
    }
}

// One chunk: its 16 columns, 15 .. 0 (descending).  xe / xo hold the chunk's text characters, two bits each, next to each
// other (xe: bits c, c + 1 = lo, hi bit of every EVEN column c; xo: bits c - 1, c of every ODD column c — see
// genasm_lane_kernel.hip), so that the LDS address of a column's Eq words is one shift and one v_bitop3.
// STORE: the columns go to tab[c] = {~(V1 | stop), V0 | stop}, rows 0..127 (words 0 and 1 of the vectors).
// SHORT: some lane's text ends inside the chunk: columns c >= nrel read the Eq words "no character matches", which leave
// the vectors as they are (the boundary column D[n][j] = m-j, genasm_cpu.cpp:239-245).
template <int NW, bool STORE, bool SHORT>
__device__ __forceinline__ void pt_sweep16(PtState<NW>& st, const uint32_t xe, const uint32_t xo, const int32_t nrel, const uint2 (&stop)[2],
                                           uint64_t (&tab)[PT_COLS][2][2], const uint32_t eq_b, const uint32_t nomatch_b)
{
    constexpr int SLOT_SHIFT = NW == 2 ? 4 : 5;                      // a base's NW words: 16 or 32 bytes (NW = 3: padded)
    auto eq_addr = [&](int c) -> uint32_t {
        const uint32_t x = (c & 1) ? xo : xe;
        const int f = (c & 1) ? c - 1 : c;                                      // the field's low bit; it goes to bit SLOT_SHIFT
        const uint32_t u = f >= SLOT_SHIFT ? x >> (f - SLOT_SHIFT) : x << (SLOT_SHIFT - f);
        const uint32_t a = bitop3<PT_ANDOR>(u, 3u << SLOT_SHIFT, eq_b);
        return (!SHORT || c < nrel) ? a : nomatch_b;
    };
    uint2 eqw[PT_EQ_AHEAD][NW];
#pragma unroll
    for (int k = 0; k < PT_EQ_AHEAD; k++) {
        const uint32_t ad = eq_addr(PT_COLS - 1 - k);
#pragma unroll
        for (int q = 0; q < NW; q++) eqw[k][q] = pt_lds_read64(ad + 8u * q);
    }
#pragma unroll
    for (int c = PT_COLS - 1; c >= 0; c--) {
        uint2 eq[NW];
#pragma unroll
        for (int q = 0; q < NW; q++) eq[q] = eqw[(PT_COLS - 1 - c) % PT_EQ_AHEAD][q];
        if (c - PT_EQ_AHEAD >= 0) {
            const uint32_t ad = eq_addr(c - PT_EQ_AHEAD);
#pragma unroll
            for (int q = 0; q < NW; q++) eqw[(PT_COLS - 1 - c) % PT_EQ_AHEAD][q] = pt_lds_read64(ad + 8u * q);
        }
        uint2 xv[NW], xh[NW], ph[NW], mh[NW];
        {   // the add (Eq & Pv) + Pv: carries run from the last word to word 0
            uint32_t aa[2 * NW], bb[2 * NW], ss[2 * NW];
#pragma unroll
            for (int q = 0; q < NW; q++) {                       // dword 2 k, 2 k + 1 of the chain = word NW-1-k
                aa[2 * q] = eq[NW - 1 - q].x & st.pv[NW - 1 - q].x;
                aa[2 * q + 1] = eq[NW - 1 - q].y & st.pv[NW - 1 - q].y;
                bb[2 * q] = st.pv[NW - 1 - q].x;
                bb[2 * q + 1] = st.pv[NW - 1 - q].y;
            }
            pt_add_chain<NW>(aa, bb, ss);
#pragma unroll
            for (int q = 0; q < NW; q++) {
                xh[NW - 1 - q].x = bitop3<PT_XH>(ss[2 * q], st.pv[NW - 1 - q].x, eq[NW - 1 - q].x);
                xh[NW - 1 - q].y = bitop3<PT_XH>(ss[2 * q + 1], st.pv[NW - 1 - q].y, eq[NW - 1 - q].y);
            }
        }
#pragma unroll
        for (int q = 0; q < NW; q++) {
            xv[q].x = eq[q].x | st.mv[q].x;
            xv[q].y = eq[q].y | st.mv[q].y;
            ph[q].x = bitop3<PT_PH>(st.mv[q].x, xh[q].x, st.pv[q].x);
            ph[q].y = bitop3<PT_PH>(st.mv[q].y, xh[q].y, st.pv[q].y);
            mh[q].x = st.pv[q].x & xh[q].x;
            mh[q].y = st.pv[q].y & xh[q].y;
        }
        // << 1 over all the words: row 0 of the matrix is all zeros, 0 comes in at the bottom
        uint2 phs[NW], mhs[NW];
        {
            const uint64_t p = shl1(((uint64_t)ph[NW - 1].y << 32) | ph[NW - 1].x), m = shl1(((uint64_t)mh[NW - 1].y << 32) | mh[NW - 1].x);
            phs[NW - 1] = make_uint2((uint32_t)p, (uint32_t)(p >> 32));
            mhs[NW - 1] = make_uint2((uint32_t)m, (uint32_t)(m >> 32));
        }
#pragma unroll
        for (int q = NW - 2; q >= 0; q--) {
            phs[q].x = __builtin_amdgcn_alignbit(ph[q].x, ph[q + 1].y, 31);
            phs[q].y = __builtin_amdgcn_alignbit(ph[q].y, ph[q].x, 31);
            mhs[q].x = __builtin_amdgcn_alignbit(mh[q].x, mh[q + 1].y, 31);
            mhs[q].y = __builtin_amdgcn_alignbit(mh[q].y, mh[q].x, 31);
        }
#pragma unroll
        for (int q = 0; q < NW; q++) {
            st.pv[q].x = bitop3<PT_PVN>(mhs[q].x, xv[q].x, phs[q].x);
            st.pv[q].y = bitop3<PT_PVN>(mhs[q].y, xv[q].y, phs[q].y);
            st.mv[q].x = phs[q].x & xv[q].x;
            st.mv[q].y = phs[q].y & xv[q].y;
        }
        if (STORE) {
#pragma unroll
            for (int r = 0; r < 2; r++) {
                tab[c][0][r] = ((uint64_t)bitop3<PT_NOR3>(st.pv[r].y, ph[r].y, stop[r].y) << 32) | bitop3<PT_NOR3>(st.pv[r].x, ph[r].x, stop[r].x);
                tab[c][1][r] = ((uint64_t)(bitop3<PT_V0>(st.pv[r].y, ph[r].y, xh[r].y) | stop[r].y) << 32) | (bitop3<PT_V0>(st.pv[r].x, ph[r].x, xh[r].x) | stop[r].x);
            }
        }
    }
}

}  // namespace

// Workgroups are four independent wavefronts (as genasm_lane_kernel); two workgroups per CU: a part's 128 table registers
// leave room for two wavefronts per SIMD.
template <int NW, bool EDITS>
__global__ __launch_bounds__(256, 2) void genasm_lane_parts_kernel(AlignArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    char* const lds_b = reinterpret_cast<char*>(lds);
    uint8_t* const lds8 = reinterpret_cast<uint8_t*>(lds);
    constexpr uint32_t SLOT = NW == 2 ? 16u : 32u;                 // bytes of one base's Eq words (NW = 3: padded to 32)
    constexpr uint32_t EQ_BYTES = 4u * SLOT, NOMATCH_BYTES = SLOT, TEXT_BYTES = 16u * NW;     // per lane
    constexpr uint32_t WAVE_LDS = 64u * (PT_RING_BYTES + PT_SCRATCH_BYTES + EQ_BYTES + NOMATCH_BYTES + TEXT_BYTES);
    constexpr uint32_t CP_DWORDS = 4u * NW;                        // a checkpoint: Pv and Mv, 2 NW dwords each

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_b = (threadIdx.x >> 6) * WAVE_LDS;
    const uint32_t ring_b = wave_b + lane * PT_RING_BYTES;
    const uint32_t scr_b = wave_b + 64u * PT_RING_BYTES + lane * PT_SCRATCH_BYTES;
    // (LDS ADDRESSES; the Eq tables start at a multiple of 4 SLOT: nothing static precedes the dynamic LDS, and the ring
    // and scratch areas of a wavefront are 64 x 88 bytes = a multiple of 128)
    const uint32_t eq_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_b + wave_b + 64u * (PT_RING_BYTES + PT_SCRATCH_BYTES);
    const uint32_t eq_b = eq_base + lane * EQ_BYTES;
    const uint32_t nomatch_b = eq_base + 64u * EQ_BYTES + lane * NOMATCH_BYTES;
    const uint32_t text_b = eq_base + 64u * (EQ_BYTES + NOMATCH_BYTES) + lane * TEXT_BYTES;       // xe / xo of dword d at text_b + 8 d
    const uint32_t swz = NW == 2 ? (lane >> 2) & 3u : (lane >> 1) & 3u;     // lanes that share LDS banks use different slots for the same base
    const uint32_t W = (uint32_t)a.W;
    const uint32_t TBL = (uint32_t)a.tb_limit;                     // W - O: 64..127, or 1..63 with W > 128
    const uint32_t P = (TBL + (uint32_t)PT_COLS - 1u) / (uint32_t)PT_COLS;     // parts, 1..8
    const int32_t ktop = (int32_t)((W + (uint32_t)PT_COLS - 1u) / (uint32_t)PT_COLS) - 1;      // the first chunk of the sweep
    // my wavefront's checkpoints: dword d of checkpoint k at ((k * CP_DWORDS + d) * 64 + lane)
    uint32_t* const cps = a.spill + ((uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * 8u * CP_DWORDS * 64u + lane;

    // ---- per-lane pair state (as in genasm_lane_kernel) ----
    bool has_pair = false;
    uint32_t pair = 0;
    uint64_t text_off = 0, read_off = 0, cigar_off = 0;
    bool rev = false;                  // my pair's read is aligned as its reverse complement (genasm_device.h: revcomp_pattern_word)
    uint32_t text_len = 0, read_len = 0, cigar_cap = 0;
    uint32_t ref_idx = 0, read_idx = 0, edits = 0;
    int32_t nr = -1;                   // index of the last committed run; n_runs = nr + 1
    uint32_t flushed = 0;              // runs below this index are in HBM (a multiple of 16); EDITS: bytes, a multiple of 32
    uint32_t pos = 0;                  // EDITS: bytes of the pair's stream so far
    uint32_t mbase = 0;                // EDITS: matches pending at column c of the current part = mbase + c
    bool queue_empty = false;          // wave-uniform
    uint32_t st_rounds = 0;

    auto write_piece = [&]() {
        const uint32_t rd = EDITS ? (ring_b >> 2) + ((flushed & 32u) >> 2) : (ring_b >> 2) + ((flushed & 16u) >> 1);
        uint32_t w[8];
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = lds[rd + k];
        const bool room = EDITS ? flushed + 32u <= 2u * (uint64_t)cigar_cap : flushed + 16u <= cigar_cap;
        if (room) {
            uint4* const dst = EDITS ? reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(a.runs + cigar_off) + flushed)
                                     : reinterpret_cast<uint4*>(a.runs + cigar_off + flushed);
            dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
            dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
        flushed += EDITS ? 32u : 16u;
    };
    // write out every piece that consists of finished runs only (the run at index nr may still grow)
    auto flush_pieces = [&]() {
        for (;;) {
            const bool need = has_pair && (EDITS ? pos - flushed >= 32u : nr - (int32_t)flushed >= 16);
            if (!__any(need)) break;
            if (need) write_piece();
        }
    };

    // `count` characters starting at character k of a sequence -> planes, one dword per 32 columns (only the words that
    // hold one of those characters are read: nothing past the end of the sequence)
    auto load_planes = [&](uint64_t off, uint32_t k, uint32_t count, uint32_t stride, uint32_t (&lo)[2 * NW], uint32_t (&hi)[2 * NW]) {
        const uint32_t inner = ((uint32_t)off & 31u) + k;
        const uint64_t w0 = (off >> 5) + (uint64_t)(inner >> 5) * stride;
        const uint32_t s = inner & 31u;
        uint64_t v[2 * NW + 1];
#pragma unroll
        for (int q = 0; q <= 2 * NW; q++) v[q] = 32u * (uint32_t)q < s + count ? a.seq[w0 + (uint64_t)q * stride] : 0ull;
#pragma unroll
        for (int q = 0; q < 2 * NW; q++) {
            lo[q] = __builtin_amdgcn_alignbit((uint32_t)v[q + 1], (uint32_t)v[q], s);
            hi[q] = __builtin_amdgcn_alignbit((uint32_t)(v[q + 1] >> 32), (uint32_t)(v[q] >> 32), s);
        }
    };
    auto save_checkpoint = [&](uint32_t k, const PtState<NW>& st) {
        uint32_t* const dst = cps + (uint64_t)k * CP_DWORDS * 64u;
#pragma unroll
        for (int q = 0; q < NW; q++) {
            dst[(4 * q + 0) * 64] = st.pv[q].x;
            dst[(4 * q + 1) * 64] = st.pv[q].y;
            dst[(4 * q + 2) * 64] = st.mv[q].x;
            dst[(4 * q + 3) * 64] = st.mv[q].y;
        }
    };
    auto load_checkpoint = [&](uint32_t k, PtState<NW>& st) {
        const uint32_t* const src = cps + (uint64_t)k * CP_DWORDS * 64u;
#pragma unroll
        for (int q = 0; q < NW; q++) {
            st.pv[q] = make_uint2(src[(4 * q + 0) * 64], src[(4 * q + 1) * 64]);
            st.mv[q] = make_uint2(src[(4 * q + 2) * 64], src[(4 * q + 3) * 64]);
        }
    };

    const uint32_t wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
    uint32_t rot = wave_slot;          // priority rotation, one step per round (not keyed on the clock: see genasm_lane_kernel)
    for (;;) {
        // (priority rotation: see genasm_lane_kernel)
        if (!SCRG_SW(a, 1)) {
            const uint32_t pr = rot++ & 3u;
            if (pr == 0) __builtin_amdgcn_s_setprio(0);
            else if (pr == 1) __builtin_amdgcn_s_setprio(1);
            else if (pr == 2) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(3);
        }
        // ---------------- retire finished pairs, fetch new ones (genasm_cpu.cpp:440-460) ----------------
        for (;;) {
            const bool fin = has_pair && read_idx >= read_len;
            if (__any(fin)) {
                if (EDITS && fin) {
                    while (pos - flushed >= 32u) write_piece();
                    const uint32_t rem = pos - flushed;
                    const uint32_t rd = (ring_b >> 2) + ((flushed & 32u) >> 2);
                    uint32_t* const dst = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(a.runs + cigar_off) + flushed);
                    for (uint32_t k = 0; 4u * k < rem; k++) {
                        const uint32_t left = rem - 4u * k;
                        const uint32_t keep = left >= 4u ? 0xffffffffu : (0xffffffffu >> (32u - 8u * left));
                        if (flushed + 4u * k < 2u * (uint64_t)cigar_cap) dst[k] = lds[rd + k] & keep;
                    }
                    a.ed[pair] = (int64_t)edits;
                    a.n_runs[pair] = pos;
                    a.status[pair] = pos > 2u * (uint64_t)cigar_cap ? 1u : 0u;
                    if (a.run_count) a.run_count[pair] = (uint32_t)(nr + 1);
                } else if (fin) {
                    const uint32_t n_runs = (uint32_t)(nr + 1);
                    while (n_runs - flushed >= 16u) write_piece();
                    const uint32_t rem = n_runs - flushed;
                    const uint32_t rd = (ring_b >> 2) + ((flushed & 16u) >> 1);
                    uint32_t* const dst = reinterpret_cast<uint32_t*>(a.runs + cigar_off + flushed);
                    for (uint32_t k = 0; 2u * k < rem; k++)
                        if (flushed + 2u * k < cigar_cap) dst[k] = lds[rd + k];
                    a.ed[pair] = (int64_t)edits;
                    a.n_runs[pair] = n_runs;
                    a.status[pair] = n_runs > cigar_cap ? 1u : 0u;
                }
                has_pair = has_pair && !fin;
            }
            const bool want = !has_pair && !queue_empty;
            if (!__any(want)) break;
            const uint64_t askers = __ballot(want);
            const int first = __ffsll((unsigned long long)askers) - 1;
            uint32_t base = 0;
            if ((int)lane == first) base = atomicAdd(a.counter, (uint32_t)__popcll(askers));
            base = (uint32_t)__shfl((int)base, first);
            const uint32_t idx = base + (uint32_t)__popcll(askers & ((1ull << lane) - 1ull));
            const bool got = want && idx < a.n_pairs;
            if (__any(want && idx >= a.n_pairs)) queue_empty = true;
            if (got) {
                const scrg_pair_desc pd = a.pairs[idx];
                pair = idx;
                text_off = pd.text_off;
                read_off = a.stranded ? pd.read_off & ~SCRG_READ_REVCOMP : pd.read_off;
                rev = a.stranded && (pd.read_off & SCRG_READ_REVCOMP) != 0;
                text_len = pd.text_len > 0xffffffffull ? 0xffffffffu : (uint32_t)pd.text_len;
                read_len = (uint32_t)pd.read_len;
                cigar_off = pd.cigar_off;
                cigar_cap = pd.cigar_cap > 0xffffffffull ? 0xffffffffu : (uint32_t)pd.cigar_cap;
                ref_idx = read_idx = edits = flushed = pos = mbase = 0;
                nr = -1;
                has_pair = true;
            }
        }
        if (!__any(has_pair)) break;

        // ---------------- window setup (genasm_cpu.cpp:417-420) ----------------
        const uint32_t n = (has_pair && ref_idx < text_len) ? min(W, text_len - ref_idx) : 0u;
        const uint32_t m = has_pair ? min(W, read_len - read_idx) : 1u;      // >= 1 for live pairs
        const uint32_t jlim = has_pair ? min(m, TBL) : 0u;                   // the walk ends when j gets here (:301, :310)
        uint2 stop[2];                                                        // the stop row (bit 63 - jlim % 64 of word jlim / 64)
        {
            const uint64_t sb = 0x8000000000000000ull >> (jlim & 63u);
            const uint64_t s0 = jlim < 64u ? sb : 0ull, s1 = jlim < 64u ? 0ull : sb;
            stop[0] = make_uint2((uint32_t)s0, (uint32_t)(s0 >> 32));
            stop[1] = make_uint2((uint32_t)s1, (uint32_t)(s1 >> 32));
        }
        PtState<NW> st;                  // the boundary column: D[n][j] = m-j, every vertical step is +1
        {
            uint32_t plo[2 * NW], phi[2 * NW], tl[2 * NW], th[2 * NW];
#pragma unroll
            for (int q = 0; q < 2 * NW; q++) { plo[q] = phi[q] = tl[q] = th[q] = 0; }
            if (has_pair) {
                load_planes(text_off, ref_idx, n, a.text_stride, tl, th);
                load_planes(read_off, read_idx, m, a.read_stride, plo, phi);
            }
            // the reversed pattern, LEFT-aligned over the NW words: bit 63-k of word w <-> pattern[64 w + k]; below the
            // pattern Eq = 1, Pv = Mv = 0 (no carry starts there, 0 comes in at its lowest bit)
            const uint32_t x = eq_b | (swz * SLOT);
#pragma unroll
            for (int q = 0; q < NW; q++) {
                // word q: characters 64 q .. 64 q + 63 = plane dwords 2q (-> high dword, reversed) and 2q + 1 (-> low dword)
                uint32_t rl1 = __builtin_bitreverse32(plo[2 * q]), rl0 = __builtin_bitreverse32(plo[2 * q + 1]);
                uint32_t rh1 = __builtin_bitreverse32(phi[2 * q]), rh0 = __builtin_bitreverse32(phi[2 * q + 1]);
                if (a.stranded && __any(has_pair && rev)) {       // (uniform) minus-strand pairs: the word comes reversed from the read's forward copy
                    const Planes rv = revcomp_pattern_word(a.seq, read_off, read_len, has_pair ? read_idx : read_len, (uint32_t)q, a.read_stride);
                    if (has_pair && rev) {
                        rl1 = (uint32_t)(rv.lo >> 32); rl0 = (uint32_t)rv.lo;
                        rh1 = (uint32_t)(rv.hi >> 32); rh0 = (uint32_t)rv.hi;
                    }
                }
                const uint32_t lo_chars = 64u * (uint32_t)q;
                const uint64_t valid = m >= lo_chars + 64u ? ~0ull : (m <= lo_chars ? 0ull : ~0ull << (64u - (m - lo_chars)));
                const uint32_t iv0 = ~(uint32_t)valid, iv1 = ~(uint32_t)(valid >> 32);
                pt_lds_write64((x ^ (0u * SLOT)) + 8u * q, make_uint2(~(rl0 | rh0) | iv0, ~(rl1 | rh1) | iv1));
                pt_lds_write64((x ^ (1u * SLOT)) + 8u * q, make_uint2((rl0 & ~rh0) | iv0, (rl1 & ~rh1) | iv1));
                pt_lds_write64((x ^ (2u * SLOT)) + 8u * q, make_uint2((~rl0 & rh0) | iv0, (~rl1 & rh1) | iv1));
                pt_lds_write64((x ^ (3u * SLOT)) + 8u * q, make_uint2((rl0 & rh0) | iv0, (rl1 & rh1) | iv1));
                pt_lds_write64(nomatch_b + 8u * q, make_uint2(iv0, iv1));
                st.pv[q] = make_uint2((uint32_t)valid, (uint32_t)(valid >> 32));
                st.mv[q] = make_uint2(0u, 0u);
            }
            // the text, slot swizzle folded in, its two planes interleaved (genasm_lane_kernel.hip), to LDS: a chunk reads its 16 columns from there
            const uint32_t swl = 0u - (swz & 1u), swh = 0u - (swz >> 1);
#pragma unroll
            for (int q = 0; q < 2 * NW; q++) {
                const uint32_t l = tl[q] ^ swl, h = th[q] ^ swh;
                pt_lds_write64(text_b + 8u * q, make_uint2(bitop3<PT_BFI>(l, h << 1, 0x55555555u), bitop3<PT_BFI>(h, l >> 1, 0xaaaaaaaau)));
            }
        }

        uint64_t tab[PT_COLS][2][2];
        // one chunk of the sweep: the variant by what the wavefront's lanes need (uniform)
        auto sweep_chunk = [&](int32_t k, auto store_tag) {
            constexpr bool STORE = decltype(store_tag)::value;
            const int32_t nrel = (int32_t)n - (int32_t)PT_COLS * k;               // columns c < nrel of the chunk are text
            if (!__any(has_pair && nrel > 0)) {                                     // past the end of every lane's text: the vectors stay
                if (STORE) {                                                        // (its table: "insertion in every row", what the sweep would give)
                    PtState<NW> keep = st;
                    pt_sweep16<NW, true, true>(keep, 0u, 0u, nrel, stop, tab, eq_b, nomatch_b);
                }
                return;
            }
            const uint2 xx = pt_lds_read64(text_b + 8u * (uint32_t)(k >> 1));
            const uint32_t sh = (uint32_t)(k & 1) * 16u;
            const uint32_t xe = xx.x >> sh, xo = xx.y >> sh;
            if (__any(has_pair && nrel < PT_COLS)) pt_sweep16<NW, STORE, true>(st, xe, xo, nrel, stop, tab, eq_b, nomatch_b);
            else pt_sweep16<NW, STORE, false>(st, xe, xo, nrel, stop, tab, eq_b, nomatch_b);
        };
        // ---------------- the sweep over all the columns: checkpoints, and the table of part 0 ----------------
#pragma unroll 1
        for (int32_t k = ktop; k >= 1; k--) {
            if ((uint32_t)k < P) save_checkpoint((uint32_t)k, st);                  // the vectors in front of chunk k
            sweep_chunk(k, std::false_type{});
        }
        sweep_chunk(0, std::true_type{});
        // the next part's checkpoint is asked for as soon as the vectors are free, so that the walk hides the round trip — where
        // the registers allow: with four-word vectors the 16 dwords in flight across the walk would spill (20-26 registers)
        constexpr bool PREFETCH = NW <= 3;
        if (PREFETCH && P > 1u) load_checkpoint(1u, st);

        // ---------------- the parts: (table,) walk, runs ----------------
        uint32_t j = 0;                                    // pattern row of the walk
        uint32_t last_dx = 0;                              // previous part: D and X bits of its last column (bit 1, bit 0) if the lane was alive to the end, else 4
        bool alive = has_pair;                             // still walking after the previous part
#pragma unroll 1
        for (uint32_t part = 0; part < P; part++) {
            const uint32_t ncols = min((uint32_t)PT_COLS, TBL - (uint32_t)PT_COLS * part);
            if (part != 0u) {
                if (!__any(alive)) break;
                if (!PREFETCH) load_checkpoint(part, st);
                sweep_chunk((int32_t)part, std::true_type{});           // (st: the checkpoint in front of this chunk)
                if (PREFETCH && part + 1u < P) load_checkpoint(part + 1u, st);
            }
            // pass 1 (see genasm_lane_kernel): the walk through this part's columns.  A table row is 128 bits (rows 0..63 in word 0,
            // 64..127 in word 1), and reading it at an arbitrary row costs a funnel of two 64-bit shifts and two selects per word:
            // ~45 instructions per column.  But the walk of part k is near row 16 k in EVERY lane, so most parts stay inside one
            // word for the whole wavefront: then a row access is ONE 64-bit shift (~16 instructions per column).
            //   MODE 1: every lane that holds a pair starts the part at a row <= 40 — walk on word 0 alone.  Rows past 63 then read
            //           as "insertion", so a lane that really leaves the word ends the part at a row > 63: the part is walked
            //           again on the full rows (rare: more than 7 insertions in 16 columns).
            //   MODE 2: every lane that holds a pair is at a row >= 64 (it can only go up) — walk on word 1 alone.
            //   MODE 0: the full 128-bit rows.
            const uint32_t j0 = j;
            uint32_t nDm = 0, Xm = 0, nIm = 0;
            const uint64_t stop0 = ((uint64_t)stop[0].y << 32) | stop[0].x, stop1 = ((uint64_t)stop[1].y << 32) | stop[1].x;
            auto walk_part = [&](auto mode_tag) {
                constexpr int MODE = decltype(mode_tag)::value;
                nDm = Xm = nIm = 0;
                j = j0;
                if constexpr (MODE == 0) {
#pragma unroll
                    for (int s = 0; s < PT_COLS; s++) {
                        if ((uint32_t)s >= ncols) continue;                 // (uniform)
                        // not (insertion), or the stop row, from row j on: the run of insertions is its leading zeros (the stop bit ends it)
                        const uint64_t x0 = tab[s][0][0] | ~tab[s][1][0] | stop0, x1 = tab[s][0][1] | ~tab[s][1][1] | stop1;
                        const uint64_t top = pt_from_row(x0, x1, j);
                        const uint64_t nxt = j < 64u ? pt_shl64(x1, j) : 0ull;      // the 64 rows after those (only if the run is that long)
                        const uint32_t ni = (top != 0ull) ? pt_clz64(top) : 64u + pt_clz64(nxt);
                        lds8[scr_b + s] = (uint8_t)ni;
                        nIm = __builtin_amdgcn_alignbit(nIm, (uint32_t)(top >> 32), 31);
                        j += ni;
                        const uint32_t nt1 = (uint32_t)(pt_from_row(tab[s][0][0], tab[s][0][1], j) >> 32);     // sign: not a deletion
                        const uint32_t t0 = (uint32_t)(pt_from_row(tab[s][1][0], tab[s][1][1], j) >> 32);      // sign: substitution
                        nDm = __builtin_amdgcn_alignbit(nDm, nt1, 31);
                        Xm = __builtin_amdgcn_alignbit(Xm, t0, 31);
                        j -= neg_mask(nt1);                                 // j += sign bit of nt1: a deletion (or the stop row) keeps j
                    }
                } else {
                    constexpr int WD = MODE - 1;                            // the word the whole wavefront stays in
                    const uint64_t stopw = WD == 0 ? stop0 : stop1;
                    uint32_t jr = j - 64u * (uint32_t)WD;                   // row inside that word
#pragma unroll
                    for (int s = 0; s < PT_COLS; s++) {
                        if ((uint32_t)s >= ncols) continue;                 // (uniform)
                        const uint64_t x = tab[s][0][WD] | ~tab[s][1][WD] | stopw;
                        const uint64_t top = pt_shl64(x, jr);               // (rows past the word: zeros = "insertion": see MODE 1 above)
                        // leading zeros, 64 for 0: the run of insertions
                        const uint32_t ni = min(pt_ffbh((uint32_t)(top >> 32)), min(pt_ffbh((uint32_t)top), 32u) + 32u);
                        lds8[scr_b + s] = (uint8_t)ni;
                        nIm = __builtin_amdgcn_alignbit(nIm, (uint32_t)(top >> 32), 31);
                        jr += ni;
                        const uint32_t nt1 = (uint32_t)(pt_shl64(tab[s][0][WD], jr) >> 32);      // sign: not a deletion
                        const uint32_t t0 = (uint32_t)(pt_shl64(tab[s][1][WD], jr) >> 32);       // sign: substitution
                        nDm = __builtin_amdgcn_alignbit(nDm, nt1, 31);
                        Xm = __builtin_amdgcn_alignbit(Xm, t0, 31);
                        jr -= neg_mask(nt1);
                    }
                    j = jr + 64u * (uint32_t)WD;
                }
            };
            if (TBL <= 63u) {                                   // (W > 128 with a small W-O: no walk ever reaches row 64 — jlim <= W-O)
                walk_part(std::integral_constant<int, 1>{});
            } else if (!__any(has_pair && j0 < 64u)) {
                walk_part(std::integral_constant<int, 2>{});
            } else if (!__any(has_pair && j0 > 40u)) {
                walk_part(std::integral_constant<int, 1>{});
                if (__any(has_pair && j > 63u)) walk_part(std::integral_constant<int, 0>{});       // some lane left word 0: once more, on the full rows
            } else {
                walk_part(std::integral_constant<int, 0>{});
            }
            // column s of the part -> bit 31-s; the lane was alive in the ti columns before the first "deletion and
            // substitution" (the stop row)
            const uint32_t nsh = 32u - ncols;
            const uint32_t Draw = ~(nDm << nsh), Xraw = Xm << nsh;
            const uint32_t ti = has_pair ? min(pt_ffbh(Draw & Xraw), ncols) : 0u;       // (a lane without a pair: the one-word walks read it garbage)
            const uint32_t A = ~(uint32_t)pt_shr64(0xffffffffull, ti);      // the top ti bits (ti = 0..16)
            const uint32_t D = Draw & A, X = Xraw & A;
            const uint32_t Im = ~nIm << nsh;
            uint32_t B = ((D ^ (D >> 1)) | (X ^ (X >> 1)) | Im | 0x80000000u) & A;    // a D / X / = run starts here
            edits += (j - j0) - ti + 2u * (uint32_t)__builtin_popcount(D) + (uint32_t)__builtin_popcount(X);
            ref_idx += ti;
            // a part whose first step continues the previous part's last run: no run starts at its column 0
            uint32_t cont = 0;
            if (part != 0u) {
                const uint32_t first_dx = ((D >> 31) << 1) | (X >> 31);
                cont = (ti != 0u && (Im >> 31) == 0u && first_dx == last_dx) ? 0x80000000u : 0u;
                B &= ~cont;
            }
            last_dx = ti == (uint32_t)PT_COLS ? ((((D >> 16) & 1u) << 1) | ((X >> 16) & 1u)) : 4u;      // (column 15 <-> bit 16)
            alive = has_pair && ti == (uint32_t)PT_COLS;

            if constexpr (EDITS) {
                // pass 2, edit stream (genasm_lane_kernel<true>): the columns that hold an edit.  (A lane that has no event left
                // has c = 0xffffffff and takes its mask bits with a field width of 0.)
                uint32_t E = D | X | Im;
                nr += (int32_t)(__builtin_popcount(B) + __builtin_popcount(Im));
                uint32_t c = pt_ffbh(E);
                uint32_t ni = lds8[scr_b + (c & 15u)];
                const uint32_t DX = D | X;
                auto put = [&](uint32_t at, uint32_t b) { lds8[ring_b + (at & 63u)] = (uint8_t)b; };
                auto event = [&]() {
                    const uint32_t sh = 31u - c;
                    const uint32_t bit = 0x80000000u >> (c & 31u);
                    const uint32_t lv = ~c >> 31;
                    uint32_t iB = __builtin_amdgcn_ubfe(Im, sh, lv), dx = __builtin_amdgcn_ubfe(DX, sh, lv);
                    const uint32_t xB = __builtin_amdgcn_ubfe(X, sh, lv);
                    const uint32_t t = mbase + c;                             // matches pending: the window's own, <= W-O - 1 <= 126
                    E = bitop3<PT_ANDN>(E, bit, bit);
                    const uint32_t nx = pt_ffbh(E);
                    const uint32_t step = 0xC0u - 0x80u * xB;                  // 'D' 3 << 6, 'X' 1 << 6
                    const uint32_t live = iB | dx;                             // (0 only for a lane that is done)
                    uint32_t k63 = ((t >= 63u ? 1u : 0u) + (t >= 126u ? 1u : 0u)) * live;      // bytes 0x3F (63 matches each) owed before the edit byte
                    const uint32_t r = (t - 63u * k63) & 63u;
                    const bool side = max(ni * iB, 2u * k63) > 3u;             // more than 3 insertions or 125 matches pending
                    if (__any(side)) {
                        if (side) {
                            auto emit = [&](uint32_t b) {
                                put(pos, b);
                                pos++;
                                if (pos - flushed >= 32u) write_piece();
                            };
                            for (uint32_t q = k63; q; q--) emit(0x3Fu);
                            if (iB) {
                                emit(0x80u | r);
                                for (uint32_t q = 1; q < ni; q++) emit(0x80u);
                                mbase = 0u - c;
                            }
                            if (dx) {
                                emit(step | (iB ? 0u : r));
                                mbase = ~c;
                            }
                            iB = dx = k63 = 0;
                        }
                    }
                    // in line: one byte 0x3F (63..125 matches pending), up to three insertions, the step
                    put(pos, 0x3Fu);
                    pos += k63;
                    put(pos, 0x80u | r);
                    put(pos + 1u, 0x80u);
                    put(pos + 2u, 0x80u);
                    pos += iB ? ni : 0u;
                    put(pos, step | (iB ? 0u : r));
                    pos += dx;
                    mbase = dx ? ~c : (iB ? 0u - c : mbase);
                    ni = lds8[scr_b + (nx & 15u)];
                    c = nx;
                };
                uint32_t trips = 0;
                while (__any(E != 0u)) {
                    event();
                    event();
                    if (++trips == 2u) {                       // <= 4 x 5 new bytes between checks + 4 speculative ones: the 64-byte ring cannot wrap
                        trips = 0;
                        flush_pieces();
                    }
                }
                flush_pieces();
                mbase += ti;
            } else {
                // pass 2, runs (genasm_lane_kernel<false>)
                uint32_t E = B | Im;
                uint32_t c = pt_ffbh(E);
                if (cont) {        // the steps up to the first event belong to the run committed last
                    uint16_t* const prev = reinterpret_cast<uint16_t*>(lds_b + ring_b + ((2u * (uint32_t)nr) & 62u));
                    *prev = (uint16_t)(*prev + min(c, ti));
                }
                uint32_t ni = lds8[scr_b + (c & 15u)];
                uint32_t nr2 = 2u * (uint32_t)nr;          // byte offset of the last committed run
                // (a lane that has no event left has c = 0xffffffff: its mask bits are taken with a field width of 0, nothing is committed)
                auto event = [&]() {
                    const uint32_t sh = 31u - c;
                    const uint32_t bit = 0x80000000u >> (c & 31u);
                    const uint32_t live = ~c >> 31;
                    *reinterpret_cast<uint16_t*>(lds_b + ring_b + ((nr2 + 2u) & 62u)) = (uint16_t)(((uint32_t)'I' << 8) | ni);
                    nr2 += 2u * __builtin_amdgcn_ubfe(Im, sh, live);
                    E = bitop3<PT_ANDN>(E, bit, bit);
                    const uint32_t nx = pt_ffbh(E);
                    ni = lds8[scr_b + (nx & 15u)];
                    const uint32_t len = min(nx, ti) - c;                       // up to the next event or the end of the walk
                    const uint32_t w = (((uint32_t)'=' << 8) + len) + __builtin_amdgcn_ubfe(D, sh, live) * (7u << 8) + __builtin_amdgcn_ubfe(X, sh, live) * (27u << 8);
                    *reinterpret_cast<uint16_t*>(lds_b + ring_b + ((nr2 + 2u) & 62u)) = (uint16_t)w;
                    nr2 += 2u * __builtin_amdgcn_ubfe(B, sh, live);
                    c = nx;
                };
                uint32_t trips = 0;
                while (__any(E != 0u)) {
                    event();
                    event();
                    if (++trips == 3u) {                       // <= 12 new runs between checks + 1 speculative slot: the 32-run ring cannot wrap
                        trips = 0;
                        nr = (int32_t)nr2 >> 1;
                        flush_pieces();
                    }
                }
                nr = (int32_t)nr2 >> 1;
                flush_pieces();
            }
        }
        read_idx += j;
        if constexpr (EDITS) {
            // the window ends (edit_stream.h): the matches since its last edit (<= W-O <= 127: up to two bytes 0x3F of 63 each), then the mark
            const uint32_t k63 = has_pair ? (mbase >= 63u ? 1u : 0u) + (mbase >= 126u ? 1u : 0u) : 0u;
            lds8[ring_b + (pos & 63u)] = (uint8_t)0x3Fu;
            lds8[ring_b + ((pos + 1u) & 63u)] = (uint8_t)0x3Fu;
            pos += k63;
            lds8[ring_b + (pos & 63u)] = (uint8_t)(mbase - 63u * k63);
            pos += has_pair ? 1u : 0u;
            mbase = 0;
            flush_pieces();
        }
        st_rounds++;
    }
    if (SCRG_TIMING(a) && lane == 0) atomicAdd((unsigned long long*)&a.stats[0], (unsigned long long)st_rounds);
}

hipError_t launch_align_lane_parts(const AlignArgs& a, int grid, size_t lds_bytes, hipStream_t s, bool edits)
{
    // grid counts wavefronts, lds_bytes is per wavefront
    const dim3 g((grid + 3) / 4), b(256);
    const int nw = (a.W + 63) / 64;
    if (nw == 2) {
        if (edits) hipLaunchKernelGGL((genasm_lane_parts_kernel<2, true>), g, b, 4 * lds_bytes, s, a);
        else hipLaunchKernelGGL((genasm_lane_parts_kernel<2, false>), g, b, 4 * lds_bytes, s, a);
    } else if (nw == 3) {
        if (edits) hipLaunchKernelGGL((genasm_lane_parts_kernel<3, true>), g, b, 4 * lds_bytes, s, a);
        else hipLaunchKernelGGL((genasm_lane_parts_kernel<3, false>), g, b, 4 * lds_bytes, s, a);
    } else if (nw == 4) {
        if (edits) hipLaunchKernelGGL((genasm_lane_parts_kernel<4, true>), g, b, 4 * lds_bytes, s, a);
        else hipLaunchKernelGGL((genasm_lane_parts_kernel<4, false>), g, b, 4 * lds_bytes, s, a);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace scrg
