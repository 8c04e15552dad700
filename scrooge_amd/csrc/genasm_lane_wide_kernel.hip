// genasm_lane_wide_kernel.hip — the lane-per-pair aligner for 32 <= W-O <= 63 and W <= 128: the formulation of
// genasm_lane_kernel.hip (every lane aligns its own pair; the window's table holds the differences of the
// edit-distance matrix behind the GenASM bitvectors, src/genasm_cpu.cpp:210-409 — see the header of that file for why
// this gives the reference's edit distance and CIGAR bit for bit), for windows whose traceback may consume up to 63
// characters: a table row is one 64-bit word (rows 0 .. W-O <= 63; for W > 64 that is word 0 of the two-word vectors,
// src/bitvector.hpp:45-48), and W-O columns of two such rows are 4 (W-O) dwords — twice the registers of a wavefront.
//
// The table is therefore built in TWO HALVES of 32 columns, both held in the same 128 registers: the sweep over the
// text columns runs from the last column down (the recurrence only goes that way) and keeps columns 0..31, the walk
// consumes them, then the sweep is REPEATED from the top down to column 32, keeping columns 32..W-O-1, and the walk
// goes on from where it stood.  Recomputing 32 columns of difference vectors (21 instructions each for one word)
// replaces 2 x 16 bytes per lane and column of table traffic through HBM (genasm_lane_mw_kernel.hip, which this kernel
// replaces for these W/O: 3 TB/s at the reference's W=64/O=2 sweep point) and there is no data-dependent slow path.
// For W > 64 the columns 127..64 are common to both sweeps: their result (Pv, Mv: 8 dwords) is kept and both halves
// start from it.
//
// Each half ends with its own second pass (masks -> runs or edit-stream bytes, the 32-bit code of genasm_lane_kernel).
// A run that crosses from column 31 to column 32 is ONE run of the window (the reference merges within a window,
// src/genasm_cpu.cpp:372-404, and starts a new run at every window): the second half does not force a run start at its
// first column when the step there continues the first half's last run, and adds its length to that run, which is
// still in the staging ring (the last committed run never leaves before the next one is committed).
//
// tests/proto/lane_proto.c (lane_align_codes_mw, RW = 1) restates the arithmetic; tests/test_gpu_parity.py holds the
// kernel against the CPU checker and the reference-built fixtures at W/O = 64/2, 64/16, 64/32, 128/65, 96/49, ...

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "genasm_kernels.h"
#include "genasm_device.h"

namespace scrg {

namespace {

#ifndef WD_BLOCKS_PER_CU
#define WD_BLOCKS_PER_CU 2
#endif
constexpr int WD_HALF = 32;                      // columns per half
constexpr uint32_t WD_RING_BYTES = 68;           // 32 runs + one dword: lanes land on distinct LDS banks
constexpr uint32_t WD_SCRATCH_BYTES = 36;        // insertion-run length of each column of a half, one byte each (+ bank skew)
constexpr int WD_EQ_AHEAD = 8;                   // Eq words are read from LDS this many columns ahead of their use

constexpr int WT_XH = bitop3_table([](int sum, int pv, int eq) { return (sum ^ pv) | eq; });
constexpr int WT_PH = bitop3_table([](int mv, int xh, int pv) { return mv | ~(xh | pv); });
constexpr int WT_PVN = bitop3_table([](int mhs, int xv, int phs) { return mhs | ~(xv | phs); });
constexpr int WT_NOR3 = bitop3_table([](int a, int b, int c) { return ~(a | b | c); });
constexpr int WT_NIV = bitop3_table([](int nv1, int v0, int stop) { return nv1 | ~v0 | stop; });
constexpr int WT_ANDN = bitop3_table([](int a, int b, int) { return a & ~b; });
constexpr int WT_BFI = bitop3_table([](int a, int b, int c) { return (a & c) | (b & ~c); });
constexpr int WT_ANDOR = bitop3_table([](int a, int b, int c) { return (a & b) | c; });
constexpr int WT_V0 = bitop3_table([](int pvn, int ph, int xh) { return pvn | ~(ph | xh); });

typedef uint32_t wd_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) wd_u32x2 wd_lds_u32x2;
__device__ __forceinline__ uint2 wd_lds_read64(uint32_t addr)
{
    const wd_u32x2 v = *reinterpret_cast<const wd_lds_u32x2*>((uintptr_t)addr);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void wd_lds_write64(uint32_t addr, uint2 v)
{
    wd_u32x2 w;
    w.x = v.x;
    w.y = v.y;
    *reinterpret_cast<wd_lds_u32x2*>((uintptr_t)addr) = w;
}
__device__ __forceinline__ uint32_t wd_ffbh(uint32_t v)      // count leading zeros; 0xffffffff for v == 0
{
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ uint64_t wd_shl64(uint64_t v, uint32_t s)      // one v_lshlrev_b64 (count modulo 64)
{
    uint64_t r;
    asm("v_lshlrev_b64 %0, %1, %2" : "=v"(r) : "v"(s), "v"(v));
    return r;
}
__device__ __forceinline__ uint64_t wd_shr64(uint64_t v, uint32_t s)
{
    uint64_t r;
    asm("v_lshrrev_b64 %0, %1, %2" : "=v"(r) : "v"(s), "v"(v));
    return r;
}

// The difference vectors between two text columns: NW 64-bit words, word 0 the most significant (bit 63-k of word w
// belongs to pattern character 64 w + k), each as two dwords (.x low, .y high).
template <int NW> struct WdState {
    uint2 pv[NW], mv[NW];
};

// Per-lane constants of a window's sweeps.
template <int NW> struct WdWindow {
    uint32_t tl[2 * NW], th[2 * NW];     // text planes (swizzled for the Eq slots): dword d holds columns 32 d .. 32 d + 31
    uint32_t n;                          // text columns of the window
    uint2 stop;                          // the stop row (bit 63 - jlim), as two dwords
};

// Columns HI .. LO (descending) of the window's table; the columns >= STORE (at most 32 of them: STORE .. STORE + 31)
// go to tab[column - STORE] as {~(V1 | stop), V0 | stop}; STORE < 0: nothing is kept.
// SHORT_N as in genasm_lane_kernel: columns >= n read the Eq word "no character matches".
// SHORT_N: 0 every lane has its text column; 1 columns >= n read "no character matches", selected by compare + v_cndmask; 2 the same
// selected by arithmetic (subtract, smear the sign, v_bitop3: 5 counted cycles per column instead of 15 — a v_cndmask on VCC issues at
// a seventh of the rate, profiles/r03_valu_issue_rates.txt — but the compiler keeps the masks: 248 registers for one-word vectors;
// the edit-stream variant and two-word vectors, which have none to spare, would spill and take form 1)
template <int NW, int SHORT_N, int HI, int LO, int STORE>
__device__ __forceinline__ void wd_sweep(WdState<NW>& st, const WdWindow<NW>& w, uint64_t (&tab)[WD_HALF][2],
                                         const uint32_t eq_b, const uint32_t nomatch_b)
{
    // (the planes pass through an opaque copy: the address arithmetic of the 64 columns must not be shared between the
    // sweeps of a window — kept alive across the walk it would take 64 registers)
    uint32_t wtl[2 * NW], wth[2 * NW];
#pragma unroll
    for (int q = 0; q < 2 * NW; q++) {
        wtl[q] = w.tl[q];
        wth[q] = w.th[q];
        asm volatile("" : "+v"(wtl[q]), "+v"(wth[q]));
    }
    // the two planes interleaved (genasm_lane_kernel.hip): xe: bits b, b + 1 = lo, hi bit of every EVEN column b of the dword; xo:
    // bits b - 1, b of every ODD column b — a column's address is then one shift and one v_bitop3 (2 instructions instead of 4)
    uint32_t xe[2 * NW], xo[2 * NW];
#pragma unroll
    for (int q = 0; q < 2 * NW; q++) {
        if (32 * q > HI || 32 * q + 31 < LO) continue;
        xe[q] = bitop3<WT_BFI>(wtl[q], wth[q] << 1, 0x55555555u);
        xo[q] = bitop3<WT_BFI>(wth[q], wtl[q] >> 1, 0xaaaaaaaau);
    }
    auto eq_addr = [&](int i) -> uint32_t {
        constexpr int SH = NW == 1 ? 3 : 4;                                  // a base's NW words: 8 or 16 bytes
        const int b = i & 31;
        const uint32_t x = (b & 1) ? xo[i >> 5] : xe[i >> 5];
        const int f = (b & 1) ? b - 1 : b;                                  // the field's low bit; it goes to bit SH
        const uint32_t u = f >= SH ? x >> (f - SH) : x << (SH - f);
        const uint32_t a = bitop3<WT_ANDOR>(u, 3u << SH, eq_b);
        if (SHORT_N == 0) return a;
        if (SHORT_N == 2) return bitop3<WT_BFI>(a, nomatch_b, neg_mask((uint32_t)i - w.n));
        return (uint32_t)i < w.n ? a : nomatch_b;
    };
    uint2 eqw[WD_EQ_AHEAD][NW];
#pragma unroll
    for (int k = 0; k < WD_EQ_AHEAD; k++) {
        if (HI - k < LO) continue;
        const uint32_t ad = eq_addr(HI - k);
#pragma unroll
        for (int q = 0; q < NW; q++) eqw[k][q] = wd_lds_read64(ad + 8u * q);
    }
#pragma unroll
    for (int i = HI; i >= LO; i--) {
        uint2 eq[NW];
#pragma unroll
        for (int q = 0; q < NW; q++) eq[q] = eqw[(HI - i) % WD_EQ_AHEAD][q];
        if (i - WD_EQ_AHEAD >= LO) {
            const uint32_t ad = eq_addr(i - WD_EQ_AHEAD);
#pragma unroll
            for (int q = 0; q < NW; q++) eqw[(HI - i) % WD_EQ_AHEAD][q] = wd_lds_read64(ad + 8u * q);
        }
        uint2 xv[NW], xh[NW], ph[NW], mh[NW];
        // the add: carries run from the last word to word 0
        if constexpr (NW == 1) {
            const uint32_t t0 = eq[0].x & st.pv[0].x, t1 = eq[0].y & st.pv[0].y;
            const uint64_t sum = add64(((uint64_t)t1 << 32) | t0, ((uint64_t)st.pv[0].y << 32) | st.pv[0].x);
            xh[0].x = bitop3<WT_XH>((uint32_t)sum, st.pv[0].x, eq[0].x);
            xh[0].y = bitop3<WT_XH>((uint32_t)(sum >> 32), st.pv[0].y, eq[0].y);
        } else {
            const uint32_t a0 = eq[1].x & st.pv[1].x, a1 = eq[1].y & st.pv[1].y, a2 = eq[0].x & st.pv[0].x, a3 = eq[0].y & st.pv[0].y;
            uint32_t s0, s1, s2, s3;
            asm("v_add_co_u32 %0, vcc, %4, %8\n\t"
                "v_addc_co_u32 %1, vcc, %5, %9, vcc\n\t"
                "v_addc_co_u32 %2, vcc, %6, %10, vcc\n\t"
                "v_addc_co_u32 %3, vcc, %7, %11, vcc"
                : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3)
                : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(st.pv[1].x), "v"(st.pv[1].y), "v"(st.pv[0].x), "v"(st.pv[0].y)
                : "vcc");
            xh[1].x = bitop3<WT_XH>(s0, st.pv[1].x, eq[1].x);
            xh[1].y = bitop3<WT_XH>(s1, st.pv[1].y, eq[1].y);
            xh[0].x = bitop3<WT_XH>(s2, st.pv[0].x, eq[0].x);
            xh[0].y = bitop3<WT_XH>(s3, st.pv[0].y, eq[0].y);
        }
#pragma unroll
        for (int q = 0; q < NW; q++) {
            xv[q].x = eq[q].x | st.mv[q].x;
            xv[q].y = eq[q].y | st.mv[q].y;
            ph[q].x = bitop3<WT_PH>(st.mv[q].x, xh[q].x, st.pv[q].x);
            ph[q].y = bitop3<WT_PH>(st.mv[q].y, xh[q].y, st.pv[q].y);
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
            st.pv[q].x = bitop3<WT_PVN>(mhs[q].x, xv[q].x, phs[q].x);
            st.pv[q].y = bitop3<WT_PVN>(mhs[q].y, xv[q].y, phs[q].y);
            st.mv[q].x = phs[q].x & xv[q].x;
            st.mv[q].y = phs[q].y & xv[q].y;
        }
        if (STORE >= 0 && i >= STORE && i < STORE + WD_HALF) {
            tab[i - STORE][0] = ((uint64_t)bitop3<WT_NOR3>(st.pv[0].y, ph[0].y, w.stop.y) << 32) | bitop3<WT_NOR3>(st.pv[0].x, ph[0].x, w.stop.x);
            tab[i - STORE][1] = ((uint64_t)(bitop3<WT_V0>(st.pv[0].y, ph[0].y, xh[0].y) | w.stop.y) << 32) | (bitop3<WT_V0>(st.pv[0].x, ph[0].x, xh[0].x) | w.stop.x);
        }
    }
}

}  // namespace

// Workgroups are four independent wavefronts (as genasm_lane_kernel); two workgroups per CU: the table's 128 registers
// leave room for two wavefronts per SIMD.
template <int NW, bool EDITS>
__global__ __launch_bounds__(256, WD_BLOCKS_PER_CU) void genasm_lane_wide_kernel(AlignArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    char* const lds_b = reinterpret_cast<char*>(lds);
    uint8_t* const lds8 = reinterpret_cast<uint8_t*>(lds);
    constexpr uint32_t EQ_BYTES = 32u * NW, NOMATCH_BYTES = 8u * NW;
    constexpr uint32_t WAVE_LDS = 64u * (WD_RING_BYTES + WD_SCRATCH_BYTES + EQ_BYTES + NOMATCH_BYTES);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_b = (threadIdx.x >> 6) * WAVE_LDS;
    const uint32_t ring_b = wave_b + lane * WD_RING_BYTES;
    const uint32_t scr_b = wave_b + 64u * WD_RING_BYTES + lane * WD_SCRATCH_BYTES;
    // (LDS ADDRESSES, multiples of 8 NW: nothing static precedes the dynamic LDS)
    const uint32_t eq_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_b + wave_b + 64u * (WD_RING_BYTES + WD_SCRATCH_BYTES);
    const uint32_t eq_b = eq_base + lane * EQ_BYTES;
    const uint32_t nomatch_b = eq_base + 64u * EQ_BYTES + lane * NOMATCH_BYTES;
    const uint32_t swz = NW == 1 ? (lane >> 3) & 3u : (lane >> 2) & 3u;     // lanes that share LDS banks use different slots for the same base
    const uint32_t W = (uint32_t)a.W;
    const uint32_t TBL = (uint32_t)a.tb_limit;         // W - O, 32..63
    const uint32_t HB = TBL - (uint32_t)WD_HALF;       // columns of the second half, 0..31

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
    uint32_t mbase = 0;                // EDITS: matches pending at column c of the current half = mbase + c
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
        const uint64_t stop64 = 0x8000000000000000ull >> jlim;
        WdWindow<NW> win;
        win.n = n;
        win.stop = make_uint2((uint32_t)stop64, (uint32_t)(stop64 >> 32));
        WdState<NW> st0;                 // the vectors in front of column 63: the boundary column (W <= 64) or the result of columns 127..64
        {
            uint32_t plo[2 * NW], phi[2 * NW];
#pragma unroll
            for (int q = 0; q < 2 * NW; q++) { plo[q] = phi[q] = win.tl[q] = win.th[q] = 0; }
            if (has_pair) {
                load_planes(text_off, ref_idx, n, a.text_stride, win.tl, win.th);
                load_planes(read_off, read_idx, m, a.read_stride, plo, phi);
            }
            // the reversed pattern, LEFT-aligned over the NW words: bit 63-k of word w <-> pattern[64 w + k]; below the
            // pattern Eq = 1, Pv = Mv = 0 (no carry starts there, 0 comes in at its lowest bit)
            const uint32_t x = eq_b | (swz << (NW == 1 ? 3 : 4));
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
                constexpr uint32_t SLOT = 8u * NW;
                wd_lds_write64((x ^ (0u * SLOT)) + 8u * q, make_uint2(~(rl0 | rh0) | iv0, ~(rl1 | rh1) | iv1));
                wd_lds_write64((x ^ (1u * SLOT)) + 8u * q, make_uint2((rl0 & ~rh0) | iv0, (rl1 & ~rh1) | iv1));
                wd_lds_write64((x ^ (2u * SLOT)) + 8u * q, make_uint2((~rl0 & rh0) | iv0, (~rl1 & rh1) | iv1));
                wd_lds_write64((x ^ (3u * SLOT)) + 8u * q, make_uint2((rl0 & rh0) | iv0, (rl1 & rh1) | iv1));
                wd_lds_write64(nomatch_b + 8u * q, make_uint2(iv0, iv1));
                st0.pv[q] = make_uint2((uint32_t)valid, (uint32_t)(valid >> 32));       // D[n][j] = m-j: every vertical step is +1
                st0.mv[q] = make_uint2(0u, 0u);
            }
            // the slot swizzle folded into the text planes
            const uint32_t swl = 0u - (swz & 1u), swh = 0u - (swz >> 1);
#pragma unroll
            for (int q = 0; q < 2 * NW; q++) { win.tl[q] ^= swl; win.th[q] ^= swh; }
        }
        // short_n: some lane's text ends inside the columns 0..63; short_pro (W > 64): ... inside the columns 64 .. the
        // first column of the prologue sweep
        const bool short_n = __any(has_pair && n < 64u);
        const bool short_pro = NW == 2 && __any(has_pair && n != ((W + 15u) & ~15u));
        uint64_t tab[WD_HALF][2];
        if constexpr (NW == 2) {
            // columns 64 .. W-1, common to both halves.  (A column past the end of the text leaves the boundary vectors as
            // they are — Eq = "no match" gives Xh = ~valid, Ph = Mh = 0 — so the sweep starts at the first column a
            // window of W characters can have, rounded up to 16.)
            if (W <= 80u) {
                if (short_pro) wd_sweep<NW, ((EDITS || NW == 2) ? 1 : 2), 79, 64, -1>(st0, win, tab, eq_b, nomatch_b);
                else wd_sweep<NW, 0, 79, 64, -1>(st0, win, tab, eq_b, nomatch_b);
            } else if (W <= 96u) {
                if (short_pro) wd_sweep<NW, ((EDITS || NW == 2) ? 1 : 2), 95, 64, -1>(st0, win, tab, eq_b, nomatch_b);
                else wd_sweep<NW, 0, 95, 64, -1>(st0, win, tab, eq_b, nomatch_b);
            } else if (W <= 112u) {
                if (short_pro) wd_sweep<NW, ((EDITS || NW == 2) ? 1 : 2), 111, 64, -1>(st0, win, tab, eq_b, nomatch_b);
                else wd_sweep<NW, 0, 111, 64, -1>(st0, win, tab, eq_b, nomatch_b);
            } else {
                if (short_pro) wd_sweep<NW, ((EDITS || NW == 2) ? 1 : 2), 127, 64, -1>(st0, win, tab, eq_b, nomatch_b);
                else wd_sweep<NW, 0, 127, 64, -1>(st0, win, tab, eq_b, nomatch_b);
            }
        }

        // ---------------- the two halves: table, walk, runs ----------------
        uint32_t j = 0;                                    // pattern row of the walk
        uint32_t last_dx = 0;                              // first half: D and X bits of its last column (bit 1, bit 0) if the lane was alive to the end, else 4
        bool alive = has_pair;                             // still walking after the first half
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            const uint32_t ncols = half == 0 ? (uint32_t)WD_HALF : HB;
            if (half == 1 && (HB == 0u || !__any(alive))) break;
            {
                WdState<NW> st = st0;
                if (half == 0) {
                    if (short_n) wd_sweep<NW, ((EDITS || NW == 2) ? 1 : 2), 63, 0, 0>(st, win, tab, eq_b, nomatch_b);
                    else wd_sweep<NW, 0, 63, 0, 0>(st, win, tab, eq_b, nomatch_b);
                } else {
                    if (short_n) wd_sweep<NW, ((EDITS || NW == 2) ? 1 : 2), 63, WD_HALF, WD_HALF>(st, win, tab, eq_b, nomatch_b);
                    else wd_sweep<NW, 0, 63, WD_HALF, WD_HALF>(st, win, tab, eq_b, nomatch_b);
                }
            }
            // pass 1 (see genasm_lane_kernel): the walk through this half's columns, on 64-bit rows
            const uint32_t j0 = j;
            uint32_t nDm = 0, Xm = 0, nIm = 0;
#pragma unroll
            for (int s = 0; s < WD_HALF; s++) {
                if ((uint32_t)s >= ncols) continue;                 // (uniform)
                const uint32_t xl = bitop3<WT_NIV>((uint32_t)tab[s][0], (uint32_t)tab[s][1], win.stop.x);
                const uint32_t xu = bitop3<WT_NIV>((uint32_t)(tab[s][0] >> 32), (uint32_t)(tab[s][1] >> 32), win.stop.y);
                const uint64_t x = wd_shl64(((uint64_t)xu << 32) | xl, j);      // not (insertion), or the stop row, from row j on
                const uint32_t ni = min(wd_ffbh((uint32_t)(x >> 32)), wd_ffbh((uint32_t)x) + 32u);     // (the stop bit makes x non-zero)
                lds8[scr_b + s] = (uint8_t)ni;
                nIm = __builtin_amdgcn_alignbit(nIm, (uint32_t)(x >> 32), 31);
                j += ni;
                const uint32_t nt1 = (uint32_t)(wd_shl64(tab[s][0], j) >> 32);     // sign: not a deletion
                const uint32_t t0 = (uint32_t)(wd_shl64(tab[s][1], j) >> 32);      // sign: substitution
                nDm = __builtin_amdgcn_alignbit(nDm, nt1, 31);
                Xm = __builtin_amdgcn_alignbit(Xm, t0, 31);
                j -= neg_mask(nt1);                                 // j += sign bit of nt1: a deletion (or the stop row) keeps j
            }
            // column s of the half -> bit 31-s; the lane was alive in the ti columns before the first "deletion and
            // substitution" (the stop row)
            const uint32_t nsh = 32u - ncols;
            const uint32_t Draw = ~(nDm << nsh), Xraw = Xm << nsh;
            const uint32_t ti = min(wd_ffbh(Draw & Xraw), ncols);
            const uint32_t A = ~(uint32_t)wd_shr64(0xffffffffull, ti);      // the top ti bits (ti = 0..32)
            const uint32_t D = Draw & A, X = Xraw & A;
            const uint32_t Im = ~nIm << nsh;
            uint32_t B = ((D ^ (D >> 1)) | (X ^ (X >> 1)) | Im | 0x80000000u) & A;    // a D / X / = run starts here
            edits += (j - j0) - ti + 2u * (uint32_t)__builtin_popcount(D) + (uint32_t)__builtin_popcount(X);
            ref_idx += ti;
            // a second half whose first step continues the first half's last run: no run starts at its column 0
            uint32_t cont = 0;
            if (half == 1) {
                const uint32_t first_dx = ((D >> 31) << 1) | (X >> 31);
                cont = (ti != 0u && (Im >> 31) == 0u && first_dx == last_dx) ? 0x80000000u : 0u;
                B &= ~cont;
            } else {
                last_dx = ti == (uint32_t)WD_HALF ? (((D & 1u) << 1) | (X & 1u)) : 4u;
                alive = has_pair && ti == (uint32_t)WD_HALF;
            }

            if constexpr (EDITS) {
                // pass 2, edit stream (genasm_lane_kernel<true>): the columns that hold an edit
                // Only columns with an edit are visited: an insertion run (before the column's step), then a deletion or
                // substitution.  mbase + c = matches pending when column c is reached (the window's own: < W-O <= 63, its
                // END byte follows the second half below); an insertion at c leaves none at c (mbase = -c), a
                // deletion/substitution none at c + 1.  Every byte goes to the slot after the last committed one; only
                // committing moves on.  Three insertions are handled in line, longer runs on a side path.  (A lane that has no
                // event left has c = 0xffffffff and takes its mask bits with a field width of 0: a half has a column 31.)
                uint32_t E = D | X | Im;
                nr += (int32_t)(__builtin_popcount(B) + __builtin_popcount(Im));
                uint32_t c = wd_ffbh(E);
                uint32_t ni = lds8[scr_b + (c & 31u)];
                const uint32_t DX = D | X;
                auto put = [&](uint32_t at, uint32_t b) { lds8[ring_b + (at & 63u)] = (uint8_t)b; };
                auto event = [&]() {
                    const uint32_t sh = 31u - c;
                    const uint32_t bit = 0x80000000u >> (c & 31u);
                    const uint32_t lv = ~c >> 31;
                    uint32_t iB = __builtin_amdgcn_ubfe(Im, sh, lv), dx = __builtin_amdgcn_ubfe(DX, sh, lv);
                    const uint32_t xB = __builtin_amdgcn_ubfe(X, sh, lv);
                    const uint32_t t = (mbase + c) & 63u;                       // (< 63 wherever a byte is committed)
                    E = bitop3<WT_ANDN>(E, bit, bit);
                    const uint32_t nx = wd_ffbh(E);
                    const uint32_t step = 0xC0u - 0x80u * xB;                  // 'D' 3 << 6, 'X' 1 << 6
                    const bool side = ni * iB > 3u;                            // more than 3 insertions
                    if (__any(side)) {
                        if (side) {
                            auto emit = [&](uint32_t b) {
                                put(pos, b);
                                pos++;
                                if (pos - flushed >= 32u) write_piece();
                            };
                            emit(0x80u | t);
                            for (uint32_t q = 1; q < ni; q++) emit(0x80u);
                            mbase = 0u - c;
                            if (dx) {
                                emit(step);
                                mbase = ~c;
                            }
                            iB = dx = 0;
                        }
                    }
                    // in line: up to three insertions, the step
                    put(pos, 0x80u | t);
                    put(pos + 1u, 0x80u);
                    put(pos + 2u, 0x80u);
                    pos += iB ? ni : 0u;
                    put(pos, step | (iB ? 0u : t));
                    pos += dx;
                    mbase = dx ? ~c : (iB ? 0u - c : mbase);
                    ni = lds8[scr_b + (nx & 31u)];
                    c = nx;
                };
                uint32_t trips = 0;
                while (__any(E != 0u)) {
                    event();
                    event();
                    if (++trips == 2u) {                       // <= 4 x 4 new bytes between checks + 3 speculative ones (+ 2 of a window end): the 64-byte ring cannot wrap
                        trips = 0;
                        flush_pieces();
                    }
                }
                flush_pieces();
                mbase += ti;
            } else {
                // pass 2, runs (genasm_lane_kernel<false>)
                uint32_t E = B | Im;
                uint32_t c = wd_ffbh(E);
                if (cont) {        // the steps up to the first event belong to the run committed last
                    uint16_t* const prev = reinterpret_cast<uint16_t*>(lds_b + ring_b + ((2u * (uint32_t)nr) & 62u));
                    *prev = (uint16_t)(*prev + min(c, ti));
                }
                uint32_t ni = lds8[scr_b + (c & 31u)];
                uint32_t nr2 = 2u * (uint32_t)nr;          // byte offset of the last committed run
                // (a lane that has no event left has c = 0xffffffff; a half has a column 31, unlike genasm_lane_kernel's
                // windows, so its mask bits are taken with a field width of 0: nothing is committed)
                auto event = [&]() {
                    const uint32_t sh = 31u - c;
                    const uint32_t bit = 0x80000000u >> (c & 31u);
                    const uint32_t live = ~c >> 31;
                    *reinterpret_cast<uint16_t*>(lds_b + ring_b + ((nr2 + 2u) & 62u)) = (uint16_t)(((uint32_t)'I' << 8) | ni);
                    nr2 += 2u * __builtin_amdgcn_ubfe(Im, sh, live);
                    E = bitop3<WT_ANDN>(E, bit, bit);
                    const uint32_t nx = wd_ffbh(E);
                    ni = lds8[scr_b + (nx & 31u)];
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
            // the window ends (edit_stream.h): the matches since its last edit — 63 of them, a whole window of W-O = 63
            // without an edit, are a byte 0x3F first — and the mark
            const uint32_t more = mbase >= 63u ? 1u : 0u;
            lds8[ring_b + (pos & 63u)] = (uint8_t)0x3Fu;
            pos += has_pair ? more : 0u;
            lds8[ring_b + (pos & 63u)] = (uint8_t)(mbase - 63u * more);
            pos += has_pair ? 1u : 0u;
            mbase = 0;
            flush_pieces();
        }
        st_rounds++;
    }
    if (SCRG_TIMING(a) && lane == 0) atomicAdd((unsigned long long*)&a.stats[0], (unsigned long long)st_rounds);
}

hipError_t launch_align_lane_wide(const AlignArgs& a, int grid, size_t lds_bytes, hipStream_t s, bool edits)
{
    // grid counts wavefronts, lds_bytes is per wavefront
    const dim3 g((grid + 3) / 4), b(256);
    if (a.W <= 64) {
        if (edits) hipLaunchKernelGGL((genasm_lane_wide_kernel<1, true>), g, b, 4 * lds_bytes, s, a);
        else hipLaunchKernelGGL((genasm_lane_wide_kernel<1, false>), g, b, 4 * lds_bytes, s, a);
    } else {
        if (edits) hipLaunchKernelGGL((genasm_lane_wide_kernel<2, true>), g, b, 4 * lds_bytes, s, a);
        else hipLaunchKernelGGL((genasm_lane_wide_kernel<2, false>), g, b, 4 * lds_bytes, s, a);
    }
    return hipGetLastError();
}

}  // namespace scrg
