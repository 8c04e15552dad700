// genasm_lane_kernel.hip — the lane-per-pair aligner for gfx950 (W <= 64, W-O <= 31): every lane of a
// wavefront aligns its own read pair, 64 pairs per wavefront, the window's traceback table in VGPRs.
// (W-O > 31 and W > 64: genasm_lane_mw_kernel.hip, the same arithmetic with multi-word rows and the table in HBM.)
//
// What is computed is the reference's windowed GenASM (src/genasm_cpu.cpp:411-438: window loop, :210-288
// distance calculation, :290-409 traceback); how the table is held is different.  Bit j of R[i][d] is clear
// exactly when D[i][j] <= d, D[i][j] being the least number of edits that align pattern[j..m) to a prefix
// of text[i..n): :225-252 is the Wu-Manber recurrence of that matrix, the boundary column R[n][d] = ones<<d
// (:239-245) is D[n][j] = m-j, the zero shifted in by `<< 1` is D[i][m] = 0.  The traceback keeps
// d == D[i][j] and asks, in the order insertion, deletion, substitution (:319-370),
//     D[i][j+1] == d-1 ?    D[i+1][j] == d-1 ?    D[i+1][j+1] == d-1 ?
// i.e. it only ever looks at the vertical, horizontal and diagonal DIFFERENCES of D.  Those differences
// are what the Myers/Hyyro bit-vector recurrence carries (Pv/Mv vertical, Ph/Mh horizontal, Xh|Mv the
// zero diagonal steps), so one text column — all 64 pattern rows and every distance d at once — costs 21
// VALU instructions, there is no loop over d, and the cost of a window does not depend on its distance
// (early termination, :278-283, has nothing left to skip).  Per text column i < W-O two dwords are kept
// for the traceback (SENE + DENT, :63-78, :200-208, :258-267, in this form):
//     V1 = Pv' | Ph            insertion or deletion             (kept negated)
//     V0 = Pv' | ~(Ph | Xh)    insertion or substitution         (both = insertion, neither = match)
// left-aligned: bit 31-j belongs to pattern character j; both carry a stop bit at the row where the lane's
// walk ends.  The traceback is column-synchronous — in column i the run of insertions is one
// count-leading-zeros over V1 & V0, then one D / X / = step moves every lane to column i+1 — so the table
// is indexed by compile-time constants and lives in 62 VGPRs (31 register pairs).
// tests/proto/lane_proto.c restates this arithmetic in C; tests/test_lane_proto.py checks that restatement
// against the reference algorithm on the CPU.
//
// LDS (9.2 KB per wavefront): the CIGAR staging ring (32 runs per lane, written out in aligned 32-byte
// pieces), 31 bytes of insertion-run lengths per lane, and the window's Eq words for the four bases (+ the
// "no character matches" word), which the table columns look up by text character.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "genasm_kernels.h"
#include "genasm_device.h"

namespace scrg {

constexpr int LANE_TB_COLS = 31;                 // W-O <= 31 columns can be consumed per window
constexpr uint32_t LANE_RING_BYTES = 68;         // 32 runs + one dword: lanes land on distinct LDS banks
constexpr uint32_t LANE_SCRATCH_BYTES = 36;      // insertion-run length of each traceback column, one byte each (+ bank skew)
constexpr uint32_t LANE_EQ_BYTES = 32;           // Eq of the window's pattern for each of the four bases, 8 bytes each
constexpr uint32_t LANE_NOMATCH_BYTES = 8;       // the Eq word of "no character matches" (columns past the end of the text)
constexpr uint32_t LANE_WAVE_LDS_BYTES = 64u * (LANE_RING_BYTES + LANE_SCRATCH_BYTES + LANE_EQ_BYTES + LANE_NOMATCH_BYTES);
// The Eq words of a wavefront are stored SLOT-MAJOR: base c of lane l at eq region + c * 512 + l * 8, the region a multiple of
// 2048 bytes.  The LDS bank of a read then depends on the lane only (512 = two rows of 64 banks), so the 64 reads of one
// ds_read_b64 never collide whatever mix of bases the lanes look up, and a column's address is still one shift and one
// v_bitop3: (x >> s) & 0x600 | (region + l * 8), the mask in an SGPR.  (Until round 4 a lane's four words were 32 contiguous
// bytes and lanes eight apart shared banks: 3.6 conflict cycles per LDS instruction, SQ_LDS_BANK_CONFLICT.)
constexpr uint32_t LANE_EQ_SLOT_STRIDE = 512;                          // 64 lanes x 8 bytes
constexpr uint32_t LANE_EQ_REGION_BYTES = 4u * LANE_EQ_SLOT_STRIDE;    // 2048 per wavefront
constexpr uint32_t LANE_EQ_FIELD_MASK = 3u * LANE_EQ_SLOT_STRIDE;      // 0x600: the base's two bits as an address field
constexpr uint32_t LANE_REST_BYTES = LANE_WAVE_LDS_BYTES - LANE_EQ_REGION_BYTES;      // ring + scratch + "no match" words of a wavefront
constexpr int LANE_EQ_AHEAD = 8;                 // Eq words are read from LDS this many columns ahead of their use

// truth tables (inputs a, b, c in that order)
// (two-input operations are left to plain and/or/xor: 4-byte encodings, a v_bitop3_b32 takes 8)
constexpr int TT_XH  = bitop3_table([](int sum, int pv, int eq) { return (sum ^ pv) | eq; });
constexpr int TT_PH  = bitop3_table([](int mv, int xh, int pv) { return mv | ~(xh | pv); });
constexpr int TT_PVN = bitop3_table([](int mhs, int xv, int phs) { return mhs | ~(xv | phs); });
constexpr int TT_NOR3 = bitop3_table([](int a, int b, int c) { return ~(a | b | c); });
constexpr int TT_NIV  = bitop3_table([](int nv1, int v0, int stop) { return nv1 | ~v0 | stop; });     // not (insertion), or the stop row
constexpr int TT_ANDN = bitop3_table([](int a, int b, int) { return a & ~b; });
constexpr int TT_BFI = bitop3_table([](int a, int b, int c) { return (a & c) | (b & ~c); });       // bits of a where c is set, else b
constexpr int TT_ANDOR = bitop3_table([](int a, int b, int c) { return (a & b) | c; });
constexpr int TT_V0  = bitop3_table([](int pvn, int ph, int xh) { return pvn | ~(ph | xh); });

// LDS accesses by 32-bit LDS address (no generic-pointer arithmetic in front of the ds instruction)
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2_t lds_u32x2_t;
__device__ __forceinline__ uint2 lds_read64(uint32_t addr)
{
    const u32x2_t v = *reinterpret_cast<const lds_u32x2_t*>((uintptr_t)addr);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void lds_write64(uint32_t addr, uint2 v)
{
    u32x2_t w;
    w.x = v.x;
    w.y = v.y;
    *reinterpret_cast<lds_u32x2_t*>((uintptr_t)addr) = w;
}

__device__ __forceinline__ uint32_t ffbh_u32(uint32_t v)      // count leading zeros; 0xffffffff for v == 0
{
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// q + 2 * bit as ONE instruction, and as written (the optimiser otherwise sums the bits of an iteration first and rebuilds every
// slot offset from the offset at the start of the iteration: one more instruction per slot)
__device__ __forceinline__ uint32_t add_twice(uint32_t q, uint32_t bit)
{
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(r) : "v"(bit), "v"(q));
    return r;
}

__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c)      // a * b + c for a, b < 2^24, as written (v_mad_u32_u24)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}

__device__ __forceinline__ uint32_t add3(uint32_t a, uint32_t b, uint32_t c)       // a + b + c (c wave-uniform), as written (v_add3_u32)
{
    uint32_t r;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}

__device__ __forceinline__ uint32_t ffbl_u32(uint32_t v)      // count trailing zeros; 0xffffffff for v == 0
{
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// ---- reads taken as their REVERSE COMPLEMENT (scrg_params.stranded, bit 63 of scrg_pair_desc.read_off: SURVEY.md §8 f4 — the
// reference drops reverse-strand candidates, src/tests.cu:346-355) — from the ONE packed copy of the read.  Character k of the
// window at read_idx of the reverse complement r' is the complement of read[len-1-read_idx-k]; complement = both planes inverted
// (A0 C1 G2 T3).  The table wants the pattern reversed and left-aligned (bit 63-k <-> pattern[k]): for a forward read that is the
// bit reversal of the 64 bases from read_idx; for r' it is simply the INVERTED 64 bases that END at len - read_idx — bit 63-k of
// that window is read[len-read_idx-1-k] — so a reverse-strand window costs no bit reversal at all: a window that starts 64
// before the place where the forward window would end, shifted up when fewer than 64 bases are left (the bits below the
// pattern are masked by the table anyway).
// lane_read_offset: the first base the window's words are loaded from (forward: read_idx; reverse: len - read_idx - 64, not below 0)
__device__ __forceinline__ uint32_t lane_read_offset(uint32_t fwd_offset, uint32_t read_idx, uint32_t read_len, uint32_t revm)
{
    return bitop3<TT_BFI>(sub_sat_u32(read_len - read_idx, 64u), fwd_offset, revm);      // (a finished read: 0 either way)
}
// lane_pattern_rev: the window's planes as loaded -> reversed and left-aligned, per lane by its strand (revm: 0 / ~0)
__device__ __forceinline__ void lane_pattern_rev(const Planes pw, uint32_t left, uint32_t revm, bool any_rev, uint64_t& rlo, uint64_t& rhi)
{
    rlo = brev64(pw.lo);
    rhi = brev64(pw.hi);
    if (any_rev) {                                     // (uniform: a wavefront of forward pairs pays one scalar branch)
        const uint32_t sh = sub_sat_u32(64u, left);  // `left` = read characters from read_idx on: fewer than 64 -> the window was loaded from base 0
        const uint64_t flo = ~(pw.lo << sh), fhi = ~(pw.hi << sh);
        rlo = ((uint64_t)bitop3<TT_BFI>((uint32_t)(flo >> 32), (uint32_t)(rlo >> 32), revm) << 32) | bitop3<TT_BFI>((uint32_t)flo, (uint32_t)rlo, revm);
        rhi = ((uint64_t)bitop3<TT_BFI>((uint32_t)(fhi >> 32), (uint32_t)(rhi >> 32), revm) << 32) | bitop3<TT_BFI>((uint32_t)fhi, (uint32_t)rhi, revm);
    }
}

// SHORT_N = false: every lane of the wave has a full text window (n = 64); true: any n <= 64 per lane (the text ends
// inside the window): columns >= n read the Eq word "no character matches", which leaves the boundary column
// D[n][j] = m-j (genasm_cpu.cpp:239-245) as it is and gives table words that say "insertion" in every row — what the
// reference's traceback finds there (:312) — so those columns need no predication, only a select of the address.
// Any pattern length m <= 64 is served by the same code:
// the reversed pattern is LEFT-aligned (bit 63-k <-> pattern[k], i.e. the reference's layout genasm_cpu.cpp:178-198
// shifted left by 64-m), the 64-m bits below it are kept neutral — Eq = 1 there (written into the table once per
// window), Pv = Mv = 0 — so no carry starts below the pattern, "0 comes in" at its lowest bit, and the rows the
// traceback reads are always the upper dword.
//
// Eq of a column is LOOKED UP: the four possible words (pattern == A / C / G / T) are written to LDS once per window
// (slot-major, LANE_EQ_SLOT_STRIDE above), and a column reads the one its text character selects — address =
// lane's column of the region | 512 * character from one shift and one v_bitop3_b32, the ds_read_b64 itself does not occupy
// the VALU.  (Computing Eq per column costs 2 v_bfe_i32 + 2 v_xor + 2 v_bitop3; the reads are issued LANE_EQ_AHEAD columns early.)
// rlo / rhi: the two planes of the window's pattern REVERSED and left-aligned (bit 63-k <-> pattern character k): lane_pattern_rev.
template <bool SHORT_N>
__device__ __forceinline__ void lane_window_table(const Planes tw, const uint64_t rlo, const uint64_t rhi, const uint32_t n, const uint32_t m,
                                                  const uint32_t stop, uint64_t (&tab)[LANE_TB_COLS],
                                                  const uint32_t eq_b, const uint32_t nomatch_b)
{
    // tab[i] = ~(V1 | stop) in the upper dword, V0 in the lower one (a register pair: the traceback shifts both with one
    // 64-bit shift): stop has the one bit of the row at which this lane's walk ends (jlim), so a finished lane reads
    // "deletion" there — it stays put without a test; how many columns it was alive in follows from the masks (below)
    const uint64_t valid = ~0ull << (64u - m);                    // (m >= 1)
    const uint32_t rl0 = (uint32_t)rlo, rl1 = (uint32_t)(rlo >> 32), rh0 = (uint32_t)rhi, rh1 = (uint32_t)(rhi >> 32);
    const uint32_t iv0 = ~(uint32_t)valid, iv1 = ~(uint32_t)(valid >> 32);
    {   // base c = 2*hi + lo: Eq_c = (lo plane == c&1) & (hi plane == c>>1), and 1 below the pattern; slot c of my column of the region
        lds_write64(eq_b, make_uint2(~(rl0 | rh0) | iv0, ~(rl1 | rh1) | iv1));
        lds_write64(eq_b + LANE_EQ_SLOT_STRIDE, make_uint2((rl0 & ~rh0) | iv0, (rl1 & ~rh1) | iv1));
        lds_write64(eq_b + 2u * LANE_EQ_SLOT_STRIDE, make_uint2((~rl0 & rh0) | iv0, (~rl1 & rh1) | iv1));
        lds_write64(eq_b + 3u * LANE_EQ_SLOT_STRIDE, make_uint2((rl0 & rh0) | iv0, (rl1 & rh1) | iv1));
        if (SHORT_N) lds_write64(nomatch_b, make_uint2(iv0, iv1));
    }
    const uint32_t tl0 = (uint32_t)tw.lo, tl1 = (uint32_t)(tw.lo >> 32), th0 = (uint32_t)tw.hi, th1 = (uint32_t)(tw.hi >> 32);
    // LDS address of column i's Eq word: eq_b | 512 * (2 * hi bit + lo bit).  The two planes are interleaved once per window so
    // that a column's two bits sit next to each other — xe: bit b = lo bit, bit b + 1 = hi bit of every EVEN column b of the
    // dword; xo: bit b - 1 = lo bit, bit b = hi bit of every ODD column b — and the address is one shift (the field to bits
    // 9..10) and one v_bitop3: 2 instructions per column instead of 4, for 8 per window.
    const uint32_t xe0 = bitop3<TT_BFI>(tl0, th0 << 1, 0x55555555u), xo0 = bitop3<TT_BFI>(th0, tl0 >> 1, 0xaaaaaaaau);
    const uint32_t xe1 = bitop3<TT_BFI>(tl1, th1 << 1, 0x55555555u), xo1 = bitop3<TT_BFI>(th1, tl1 >> 1, 0xaaaaaaaau);
    auto eq_addr = [&](int i) -> uint32_t {
        const int b = i & 31;
        const uint32_t x = (b & 1) ? (i < 32 ? xo0 : xo1) : (i < 32 ? xe0 : xe1);
        const int f = (b & 1) ? b - 1 : b;                                  // the field's low bit; it goes to bit 9
        const uint32_t u = f >= 9 ? x >> (f - 9) : x << (9 - f);
        const uint32_t a = bitop3<TT_ANDOR>(u, LANE_EQ_FIELD_MASK, eq_b);
        if (!SHORT_N) return a;
        // column i >= n: the "no character matches" word.  A select by arithmetic (i - n < 0: the column exists): compare +
        // v_cndmask on VCC costs 15 counted cycles per column against 5 for subtract, smear the sign, v_bitop3
        // (profiles/r03_valu_issue_rates.txt) — a third of the table's time in the rounds in which some lane's text ends
        return bitop3<TT_BFI>(a, nomatch_b, neg_mask((uint32_t)i - n));
    };
    uint2 eqw[LANE_EQ_AHEAD];
#pragma unroll
    for (int k = 0; k < LANE_EQ_AHEAD; k++) eqw[k] = lds_read64(eq_addr(63 - k));
    uint32_t pv0 = (uint32_t)valid, pv1 = (uint32_t)(valid >> 32), mv0 = 0u, mv1 = 0u;   // D[n][j] = m-j: every vertical step is +1
#pragma unroll
    for (int i = 63; i >= 0; i--) {
        const uint2 eq_now = eqw[(63 - i) % LANE_EQ_AHEAD];
        if (i - LANE_EQ_AHEAD >= 0) eqw[(63 - i) % LANE_EQ_AHEAD] = lds_read64(eq_addr(i - LANE_EQ_AHEAD));
        {
            const uint32_t eq0 = eq_now.x, eq1 = eq_now.y;
            const uint32_t xv0 = eq0 | mv0, xv1 = eq1 | mv1;
            const uint32_t t0 = eq0 & pv0, t1 = eq1 & pv1;
            const uint64_t sum = add64(((uint64_t)t1 << 32) | t0, ((uint64_t)pv1 << 32) | pv0);
            const uint32_t xh0 = bitop3<TT_XH>((uint32_t)sum, pv0, eq0), xh1 = bitop3<TT_XH>((uint32_t)(sum >> 32), pv1, eq1);
            const uint32_t ph0 = bitop3<TT_PH>(mv0, xh0, pv0), ph1 = bitop3<TT_PH>(mv1, xh1, pv1);
            const uint32_t mh0 = pv0 & xh0, mh1 = pv1 & xh1;
            const uint64_t phs = shl1(((uint64_t)ph1 << 32) | ph0);       // row 0 of the matrix is all zeros: 0 comes in
            const uint64_t mhs = shl1(((uint64_t)mh1 << 32) | mh0);
            pv0 = bitop3<TT_PVN>((uint32_t)mhs, xv0, (uint32_t)phs);
            pv1 = bitop3<TT_PVN>((uint32_t)(mhs >> 32), xv1, (uint32_t)(phs >> 32));
            mv0 = (uint32_t)phs & xv0;
            mv1 = (uint32_t)(phs >> 32) & xv1;
            if (i < LANE_TB_COLS)
                tab[i] = ((uint64_t)bitop3<TT_NOR3>(pv1, ph1, stop) << 32) | bitop3<TT_V0>(pv1, ph1, xh1);
        }
    }
}

// Workgroups are four independent wavefronts (nothing is shared, there is no barrier): the four land on the four
// SIMDs of one CU, so a partly filled GPU has the same number of wavefronts on every SIMD of a CU.  (Single-wave
// workgroups were placed unevenly — 3072 of them ran no faster than 4096.)
//
// EDITS = true (scrg_align_device_edits): the same alignment, delivered as an EDIT STREAM (edit_stream.h: one byte
// per edit carrying the number of matches before it) instead of runs.  Only the second traceback pass and the
// stores differ: it visits the columns that hold an edit (~3 per window at 10 % error instead of ~6.5 run
// boundaries), the matches pending since the last edit are one register carried from window to window, and the
// staging ring holds 64 bytes per lane that leave in the same aligned 32-byte pieces.
template <bool EDITS>
__global__ __launch_bounds__(256, 4) void genasm_lane_kernel(AlignArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    char* const lds_b = reinterpret_cast<char*>(lds);

    const uint32_t lane = threadIdx.x & 63u;
    // LDS of a workgroup: the wavefronts' Eq regions first (2048 bytes each, slot-major: the dynamic LDS starts at address 0
    // because nothing static precedes it, so every region is a multiple of 2048), then each wavefront's ring, insertion-run
    // lengths and "no match" words
    const uint32_t wave = threadIdx.x >> 6, wpg = blockDim.x >> 6;
    const uint32_t wave_b = wpg * LANE_EQ_REGION_BYTES + wave * LANE_REST_BYTES;
    const uint32_t ring_b = wave_b + lane * LANE_RING_BYTES;
    const uint32_t scr_b = wave_b + 64u * LANE_RING_BYTES + lane * LANE_SCRATCH_BYTES;
    uint8_t* const lds8 = reinterpret_cast<uint8_t*>(lds);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_b;      // (an LDS ADDRESS: 0)
    const uint32_t eq_b = lds0 + wave * LANE_EQ_REGION_BYTES + lane * 8u;
    const uint32_t nomatch_b = lds0 + wave_b + 64u * (LANE_RING_BYTES + LANE_SCRATCH_BYTES) + lane * LANE_NOMATCH_BYTES;
    const uint32_t W = (uint32_t)a.W;
    const uint32_t TBL = (uint32_t)a.tb_limit;         // W - O, 1..31

    // ---- per-lane pair state ----
    bool has_pair = false;
    uint32_t pair = 0;
    const uint64_t* text_w = a.seq;    // the first word of my text / read, and (tr_in: bits 4..0 / 12..8) where in that word they begin
    const uint64_t* read_w = a.seq;
    uint32_t tr_in = 0;
    uint64_t cigar_off = 0;
    uint32_t text_len = 0, read_len = 0, cigar_cap = 0;
    uint32_t ref_idx = 0, read_idx = 0, edits = 0;
    uint32_t revm = 0;                 // ~0: my pair's read is aligned as its reverse complement (a.stranded and bit 63 of read_off)
    int32_t nr = -1;                   // index of the run in progress (or of the last finished one); n_runs = nr + 1
    uint32_t flushed = 0;              // runs below this index are in HBM (a multiple of 16); EDITS: bytes, a multiple of 32
    uint32_t pos = 0;                  // EDITS: bytes of the pair's stream so far
    uint32_t mbase = 0;                // EDITS: matches pending at column c of the current window = mbase + c
    WindowWords twords = {0, 0, 0, 0}, pwords = {0, 0, 0, 0};     // the words of my next text / read window (loaded ahead)
    bool queue_empty = false;          // wave-uniform
    const bool timing = SCRG_TIMING(a);            // (compile-time false in the shipped build: genasm_kernels.h)
    uint64_t cy_fetch = 0, cy_setup = 0, cy_dc = 0, cy_tb = 0, cy_p1 = 0;
    uint32_t st_rounds = 0, st_gen = 0;
    const uint64_t rt0 = timing ? __builtin_amdgcn_s_memrealtime() : 0;      // 100 MHz wall clock: wavefront start

    // one 16-run piece of my ring -> my slice (two 16-byte stores); pieces past the slice's capacity are dropped
    // (EDITS: the slice holds bytes — it starts at byte 2 * cigar_off and is 2 * cigar_cap bytes long — and a piece is
    // 32 bytes of the stream)
    auto write_piece = [&]() {
        const uint32_t rd = EDITS ? (ring_b >> 2) + ((flushed & 32u) >> 2) : (ring_b >> 2) + ((flushed & 16u) >> 1);
        uint32_t w[8];
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = lds[rd + k];
        const bool room = EDITS ? flushed + 32u <= 2u * (uint64_t)cigar_cap : flushed + 16u <= cigar_cap;
        if (room && !SCRG_ABL(a, 16)) {          // (16: ablation, profiling only: no stores)
            uint4* const dst = EDITS ? reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(a.runs + cigar_off) + flushed)
                                     : reinterpret_cast<uint4*>(a.runs + cigar_off + flushed);
            dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
            dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
        flushed += EDITS ? 32u : 16u;
    };
    // Write out a whole piece of committed output where a lane has one: one piece per lane and look.
    //
    // WHEN matters more than how.  Loads and stores share one counter (vmcnt) and complete in order, so a store that is
    // issued after the next window's words have been asked for (they are asked for right after traceback pass 1) makes the wait
    // for those words at the start of the next round a wait for the store as well — several hundred cycles per round for a
    // wavefront that has its SIMD to itself (ablation: 0.1 of 2.6 ms).  So the regular look is BEFORE the table (7 k cycles
    // without a memory instruction: the stores are long done when the loads are issued), and the second traceback pass only
    // looks when a ring is really about to run full (ring_guard, below): in well under a tenth of the rounds.
    // Bound: after the regular look a lane holds at most 15 unwritten runs (EDITS: 31 bytes).
    auto flush_pieces = [&]() {
        const bool need = has_pair && (EDITS ? pos - flushed >= 32u : nr - (int32_t)flushed >= 15);
        if (__any(need)) {
            if (need) write_piece();
        }
    };
    // Inside the second pass, before a trip that may commit `add` more runs (EDITS: bytes; plus the slots written ahead): only a
    // lane whose ring could not take them makes the wavefront look — then every lane that has a whole piece writes it.
    auto ring_guard = [&](uint32_t fill, uint32_t limit) {      // fill: runs (EDITS: bytes) committed and not yet written out
        if (__any(has_pair && fill >= limit)) {
            if (has_pair && fill >= (EDITS ? 32u : 16u)) write_piece();
        }
    };

    // hardware wave slot on my SIMD (HW_ID bits 3:0)
    const uint32_t wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
    uint32_t rot = wave_slot;          // priority rotation: one step per round
    for (;;) {
        // The SIMD's arbiter issues oldest-wave-first: left alone, the first wavefront on a SIMD runs at the speed
        // of a lone wave and the last one finishes 2.7x later, with the SIMD half idle at the end of a launch.
        // Rotating the priorities (one step per round, starting from the wave slot: a different wavefront is on top from
        // round to round) lets the wavefronts of a SIMD progress, and finish, together.  (Until round 4 the rotation was keyed
        // on the clock: s_memtime and the wait for it — which is a wait for every LDS operation in flight as well — cost a
        // wavefront that has its SIMD to itself ~1 000 cycles per round: one launch of 100 k pairs 2.45 -> 2.30 ms without it.)
        if (!SCRG_SW(a, 1)) {
            const uint32_t pr = rot++ & 3u;
            if (pr == 0) __builtin_amdgcn_s_setprio(0);
            else if (pr == 1) __builtin_amdgcn_s_setprio(1);
            else if (pr == 2) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(3);
        }
        const uint64_t tm0 = timing ? __builtin_readcyclecounter() : 0;
        // ---------------- retire finished pairs, fetch new ones (genasm_cpu.cpp:440-460) ----------------
        for (;;) {
            const bool fin = has_pair && read_idx >= read_len;
            if (__any(fin)) {
                if (EDITS && fin) {
                    // (the matches after the last edit are implied by the read length)
                    while (pos - flushed >= 32u) write_piece();
                    const uint32_t rem = pos - flushed;                      // < 32: whole dwords of the last, partial piece,
                    const uint32_t rd = (ring_b >> 2) + ((flushed & 32u) >> 2);      // bytes past the end zeroed
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
                    // the tail: whole dwords of the last, partial piece
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
            // one atomic per wavefront for all the lanes that want a pair
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
                text_w = a.seq + (pd.text_off >> 5);
                const uint64_t r_off = a.stranded ? pd.read_off & ~SCRG_READ_REVCOMP : pd.read_off;
                revm = (a.stranded && (pd.read_off & SCRG_READ_REVCOMP)) ? 0xffffffffu : 0u;
                read_w = a.seq + (r_off >> 5);
                tr_in = ((uint32_t)pd.text_off & 31u) | (((uint32_t)r_off & 31u) << 8);
                text_len = pd.text_len > 0xffffffffull ? 0xffffffffu : (uint32_t)pd.text_len;
                read_len = (uint32_t)pd.read_len;
                cigar_off = pd.cigar_off;
                cigar_cap = pd.cigar_cap > 0xffffffffull ? 0xffffffffu : (uint32_t)pd.cigar_cap;
                ref_idx = read_idx = edits = flushed = pos = mbase = 0;
                nr = -1;
                has_pair = true;
                twords = load_window_words_at(text_w, tr_in & 31u, 0u, a.text_stride);
                pwords = load_window_words_at(read_w, tr_in >> 8, lane_read_offset(0u, 0u, read_len, revm), a.read_stride);
            }
        }
        if (!__any(has_pair)) break;
        const bool wave_rev = a.stranded && __any(has_pair && revm != 0u);       // (uniform) some lane aligns a reverse complement

        const uint64_t tm1 = timing ? __builtin_readcyclecounter() : 0;
        // ---------------- window setup (genasm_cpu.cpp:417-420) ----------------
        const uint32_t n = has_pair ? min(W, sub_sat_u32(text_len, ref_idx)) : 0u;      // text characters left, at most a window's
        const uint32_t m = has_pair ? min(W, read_len - read_idx) : 1u;      // >= 1 for live pairs
        // (the words were loaded when the pair was fetched, or before the previous round's second traceback pass.  EVERY lane
        // takes them, not only those that hold a pair: a lane without one computes a table of garbage that its walk never reads —
        // jlim = 0 freezes it in row 0 — and the wait for the loads is then unconditional, i.e. over when the stores below are
        // issued: a wait inside `if (has_pair)` leaves the compiler unsure whether the loads have landed, and it would wait
        // again — now for the stores too — in front of the first table instruction that reuses one of their registers)
        const Planes tw = window_planes(twords), pw = window_planes(pwords);
        // (the planes are in registers — the loads have landed — before any store below is issued: the scheduler is not to sink
        // the funnel shifts, and with them the wait, behind the stores)
        asm volatile("" :: "v"(tw.lo), "v"(tw.hi), "v"(pw.lo), "v"(pw.hi) : "memory");
        flush_pieces();                             // (the stores of this round: issued here, done long before pass 1 ends)
        const uint64_t tm2 = timing ? __builtin_readcyclecounter() : 0;

        // ---------------- the window's table: all distances at once (genasm_cpu.cpp:210-288) ----------------
        uint64_t tab[LANE_TB_COLS];
        const uint32_t jlim = has_pair ? min(m, TBL) : 0u;          // the walk ends when j gets here (:301, :310)
        const uint32_t stop = 0x80000000u >> jlim;
        const bool short_n = __any(has_pair && n != 64u);
        uint64_t rlo, rhi;
        lane_pattern_rev(pw, read_len - read_idx, revm, wave_rev, rlo, rhi);
        if (SCRG_ABL(a, 2)) {                       // ablation (profiling only): no table computation
#pragma unroll
            for (int i = 0; i < LANE_TB_COLS; i++) tab[i] = ((uint64_t)~stop << 32) | (((uint32_t)tw.lo * (uint32_t)(i + 1)) | stop);
        } else if (short_n) {
            lane_window_table<true>(tw, rlo, rhi, n, m, stop, tab, eq_b, nomatch_b);
            st_gen++;
        } else {
            lane_window_table<false>(tw, rlo, rhi, n, m, stop, tab, eq_b, nomatch_b);
        }
        const uint64_t tm3 = timing ? __builtin_readcyclecounter() : 0;

        // ---------------- traceback, column-synchronous (genasm_cpu.cpp:290-409) ----------------
        // Lanes stop by themselves when j reaches jlim = min(m, W-O) (:301, :310); i < W-O (:309) is the
        // loop bound; i < n (:312) needs no test because columns >= n hold "insertion" in every row.  The
        // last-character rule (:336-343, insertion whenever there is budget) is what the matrix says anyway:
        // D[i][m] = 0, so an insertion is possible exactly when D[i][m-1] > 0.
        //
        // Pass 1 walks the columns without a branch and only records the path: per column one bit each for
        // "no insertion run starts here" (a run's length goes to a byte of LDS), "the step out of this column
        // is not a deletion" and "... is a substitution", pushed into three masks from the right, and the
        // number of columns the lane was alive in.  The stop bit (row jlim) ends an insertion run and freezes
        // a finished lane (it reads "deletion": j stays), so the loop has no min() and no test.  Pass 2 turns
        // the masks into runs: one iteration per column at which a run starts (count-leading-zeros over the
        // boundary mask), so its length is the number of runs of the lane with the most runs, not the number
        // of columns.
        {
            uint32_t j = 0, ti, nDm = 0, Xm = 0, nIm = 0;
            auto walk = [&](auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;       // W-O = 31: no per-column test of the column limit
#pragma unroll
                for (int i = 0; i < LANE_TB_COLS; i++) {
                    if (!FULL && (uint32_t)i >= TBL) continue;         // (uniform)
                    // the insertions in a row from (i, j): leading zeros of "not insertion, or stop" << j
                    const uint32_t x = bitop3<TT_NIV>((uint32_t)(tab[i] >> 32), (uint32_t)tab[i], stop) << j;
                    const uint32_t ni = ffbh_u32(x);
                    lds8[scr_b + i] = (uint8_t)ni;
                    nIm = __builtin_amdgcn_alignbit(nIm, x, 31);               // (nIm << 1) | (ni == 0)
                    j += ni;
                    // sign bits of both dwords after ONE 64-bit shift of the pair (what spills from v0 into the low bits of
                    // the upper dword is never looked at): not a deletion, substitution
                    const uint64_t both = tab[i] << j;
                    const uint32_t nt1 = (uint32_t)(both >> 32), t0 = (uint32_t)both;
                    nDm = __builtin_amdgcn_alignbit(nDm, nt1, 31);
                    Xm = __builtin_amdgcn_alignbit(Xm, t0, 31);
                    j -= neg_mask(nt1);                                        // j += sign bit of nt1: a deletion (or the stop row) keeps j (v_ashrrev, v_sub: full rate)
                }
            };
            if (SCRG_ABL(a, 8)) { j = jlim; nDm = ~0u; }                    // ablation (profiling only): no walk
            else if (TBL == (uint32_t)LANE_TB_COLS) walk(std::true_type{});
            else walk(std::false_type{});
            if (timing) cy_p1 += __builtin_readcyclecounter() - tm3;
            // column i -> bit 31-i; only the ti columns the lane was alive in count (insertion runs are exact as recorded).
            // A lane whose walk has not reached its last row (j < jlim) was alive in every column.  One that has (:307-310)
            // stopped in the column after its last step that was not a deletion — a diagonal step took it to row jlim — or in
            // the column of its last insertion run, if that run took it there (a dead lane reads "deletion, no insertion"
            // from then on); with no such column at all it never moved: ti = 0.
            const uint32_t TBc = min(TBL, (uint32_t)LANE_TB_COLS);
            const uint32_t nsh = 32u - TBc;
            const uint32_t notD = nDm << nsh, Xraw = Xm << nsh;
            const uint32_t Draw = ~notD;
            const uint32_t Im = ~nIm << nsh;
            const uint32_t ti_stopped = (31u - ffbl_u32((notD >> 1) | Im)) & 31u;      // bit 31-c: the lane stopped in column c or later; none: 0
            ti = bitop3<TT_BFI>(TBc, ti_stopped, neg_mask(j - jlim));          // j < jlim ? TBc : ti_stopped, without a v_cndmask on VCC
            const uint32_t A = ~(0xffffffffu >> ti);
            const uint32_t D = Draw & A, X = Xraw & A;
            const uint32_t B = ((D ^ (D >> 1)) | (X ^ (X >> 1)) | Im | 0x80000000u) & A;    // a D / X / = run starts here
            const uint32_t nD = (uint32_t)__builtin_popcount(D), nX = (uint32_t)__builtin_popcount(X);
            edits += j - ti + 2u * nD + nX;             // insertions (j - (ti - nD)) + deletions + substitutions
            ref_idx += ti;
            read_idx += j;
            // the next window's words, asked for now: the second pass below hides the latency.  (Every lane loads: a
            // pair that is finished reads its padding, a lane without a pair its last pair's, and the registers are
            // not alive across the table that way.)
            twords = load_window_words_at(text_w, tr_in & 31u, ref_idx, a.text_stride);
            // (never past a finished read: index 0 then.  read_idx <= read_len always, so "finished" is "equal": x | -x has its sign
            // bit set for every x != 0 — exact for any 32-bit length, and no v_cndmask on VCC)
            const uint32_t left_x = read_idx ^ read_len;
            const uint32_t fwd_at = read_idx & neg_mask(left_x | (0u - left_x));
            pwords = load_window_words_at(read_w, tr_in >> 8, wave_rev ? lane_read_offset(fwd_at, read_idx, read_len, revm) : fwd_at, a.read_stride);

            // Pass 2.  The next column with an event: an insertion run, then (if B) the run of steps that starts
            // there.  Both words go to the slot after the last committed run; only committing moves on.  (A lane
            // that is done computes garbage from column "31", which no mask ever has.)  The length byte of the
            // next insertion run is read one iteration ahead.
            if constexpr (EDITS) {
                // Pass 2, edit stream (edit_stream.h, version 2).  Only columns with an edit are visited: an insertion run
                // (before the column's step), then a deletion or substitution.  mbase + c = matches pending when column c
                // is reached (< W-O <= 31: the window's own matches only, every window closes with its END byte below);
                // an insertion at c leaves none at c (mbase = -c), a deletion/substitution none at c + 1.  Every byte
                // goes to the slot after the last committed one; only committing moves on.  Three insertions are handled
                // in line, longer runs on a side path (well under one per cent of the iterations at 10 % error).
                uint32_t E = SCRG_ABL(a, 4) ? 0u : (D | X | Im);
                // (the runs this window has in the other output format: one per insertion run, one per D / X / = run start)
                nr += (int32_t)(__builtin_popcount(B) + __builtin_popcount(Im));
                uint32_t c = ffbh_u32(E);
                uint32_t ni = lds8[scr_b + c];
                auto put = [&](uint32_t at, uint32_t b) { lds8[ring_b + (at & 63u)] = (uint8_t)b; };
                auto event = [&]() {
                    const uint32_t sh = 31u - c;
                    const uint32_t bit = 0x80000000u >> (c & 31u);
                    uint32_t iB = __builtin_amdgcn_ubfe(Im, sh, 1), dx = __builtin_amdgcn_ubfe(D | X, sh, 1);
                    const uint32_t xB = __builtin_amdgcn_ubfe(X, sh, 1);
                    const uint32_t t = mbase + c;
                    E = bitop3<TT_ANDN>(E, bit, bit);
                    const uint32_t nx = ffbh_u32(E);
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
                    ni = lds8[scr_b + nx];
                    c = nx;
                };
                // a trip commits at most 2 x 4 bytes and writes 3 more ahead: 31 left by the regular look + two trips fit the
                // 64-byte ring; from the third trip on a lane with 50 or more bytes pending makes the wavefront look
                uint32_t trips = 0;
                while (__any(E != 0u)) {
                    if (trips >= 2u) ring_guard(pos - flushed, 50u);
                    event();
                    event();
                    trips++;
                }
                // the window ends: the matches since its last edit, and the mark (a lane without a pair writes a byte nobody commits)
                put(pos, mbase + ti);
                pos += has_pair ? 1u : 0u;
                mbase = 0;
            } else {
            // Two events per trip (c0 / c1: the columns of the next two events; E holds c1 and what comes after it).  The length
            // bytes of both events' insertion runs are asked for at the TOP of the trip and used at its very end — the two ring
            // writes of an event come after all its arithmetic — so the loop carries no LDS read from trip to trip (the wait in
            // front of a carried read would also wait for the writes issued just before the branch) and a wavefront that has its
            // SIMD to itself waits for LDS less than once per event.
            uint32_t E = SCRG_ABL(a, 4) ? 0u : (B | Im);                  // (ablation, profiling only: no runs)
            uint32_t c0 = ffbh_u32(E);
            E = bitop3<TT_ANDN>(E, 0x80000000u >> (c0 & 31u), 0u);
            uint32_t c1 = ffbh_u32(E);
            uint32_t q = 2u * (uint32_t)nr + 2u;       // byte offset of the next free slot (the run after the last committed one)
            // one event: column c, its length byte ni; nx = the column of the event after it.  The insertion run goes to the slot
            // after the last committed run and is committed by its mask bit; the D / X / = run that starts here goes to the slot
            // after that (the same slot if there is no insertion run: it is written second) and is committed by B.
            auto event = [&](uint32_t& c, const uint32_t ni, const uint32_t nx) {
                const uint32_t sh = 31u - c;
                const uint32_t aI = ring_b + (q & 62u);
                q = add_twice(q, __builtin_amdgcn_ubfe(Im, sh, 1));
                // the run's length = min(nx, ti) - c (up to the next event or the end of the walk) = min(nx, ti) + sh - 31; its letter
                // '=' 0x3D, 'X' 0x58 = '=' + 27, 'D' 0x44 = '=' + 7: two multiply-adds on top of sh and one three-operand add
                const uint32_t w = add3(mad24(__builtin_amdgcn_ubfe(X, sh, 1), 27u << 8, mad24(__builtin_amdgcn_ubfe(D, sh, 1), 7u << 8, sh)),
                                        min(nx, ti), ((uint32_t)'=' << 8) - 31u);
                const uint32_t aS = ring_b + (q & 62u);
                q = add_twice(q, __builtin_amdgcn_ubfe(B, sh, 1));
                E = bitop3<TT_ANDN>(E, 0x80000000u >> (nx & 31u), 0u);      // the event after nx takes this one's place
                c = ffbh_u32(E);
                *reinterpret_cast<uint16_t*>(lds_b + aI) = (uint16_t)(((uint32_t)'I' << 8) | ni);
                *reinterpret_cast<uint16_t*>(lds_b + aS) = (uint16_t)w;
            };
            // a trip commits at most 4 runs and writes one slot ahead: the 15 left by the regular look + four trips fit the 32-run
            // ring; from the fifth trip on a lane with 27 or more runs pending makes the wavefront look
            uint32_t trips = 0;
            while (__any(c0 < 32u)) {                      // (a lane without events left has c0 = 0xffffffff: it commits nothing)
                const uint32_t n0 = lds8[scr_b + c0], n1 = lds8[scr_b + c1];
                if (trips >= 4u) ring_guard((q >> 1) - flushed, 27u);
                event(c0, n0, c1);
                event(c1, n1, c0);
                trips++;
            }
            nr = ((int32_t)q >> 1) - 1;
            }
        }
        st_rounds++;
        if (timing) {
            const uint64_t tm4 = __builtin_readcyclecounter();
            cy_fetch += tm1 - tm0;
            cy_setup += tm2 - tm1;
            cy_dc += tm3 - tm2;
            cy_tb += tm4 - tm3;
        }
    }
    if (timing && lane == 0) {
        atomicAdd((unsigned long long*)&a.stats[0], (unsigned long long)st_rounds);
        atomicAdd((unsigned long long*)&a.stats[1], (unsigned long long)st_gen);
        atomicAdd((unsigned long long*)&a.stats[2], (unsigned long long)cy_p1);
        atomicAdd((unsigned long long*)&a.stats[3], (unsigned long long)cy_fetch);
        atomicAdd((unsigned long long*)&a.stats[4], (unsigned long long)cy_setup);
        atomicAdd((unsigned long long*)&a.stats[5], (unsigned long long)cy_dc);
        atomicAdd((unsigned long long*)&a.stats[6], (unsigned long long)cy_tb);
        // wavefront life times on the 100 MHz wall clock: sum, latest start, earliest start (as 2^62 - t), latest end
        const uint64_t rt1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd((unsigned long long*)&a.stats[7], (unsigned long long)(rt1 - rt0));
        atomicMax((unsigned long long*)&a.stats[8], (unsigned long long)rt0);
        atomicMax((unsigned long long*)&a.stats[9], (unsigned long long)((1ull << 62) - rt0));
        atomicMax((unsigned long long*)&a.stats[10], (unsigned long long)rt1);
        atomicMax((unsigned long long*)&a.stats[11], (unsigned long long)((1ull << 62) - rt1));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// genasm_lane_split_kernel — the same alignment (runs output) with a window's work split over TWO wavefronts.
//
// A pair is a chain of ~330 windows, each of which needs the previous one's advance, so a wavefront that has a SIMD to itself
// is bound by that chain — its instruction count times the issue interval of ONE wavefront (4.9 cycles, however empty the SIMD
// is) plus the latencies nobody hides — not by issue slots: 25 000 ... 65 000 pairs (at most one wavefront per SIMD) take 1.93 ms
// whatever their number.  The chain itself is only fetch -> window setup -> table -> traceback pass 1 (which yields the window's
// advance): the SECOND pass (masks -> runs, the CIGAR ring, the stores, the retiring of a pair) depends on nothing the next
// window needs.  So a workgroup is four PRODUCER wavefronts (everything up to pass 1) and four CONSUMER wavefronts (pass 2 and
// all output), producer i handing consumer i one record per lane and window through LDS, double buffered, one s_barrier per
// window round: the consumer works on window r - 1 while the producer computes window r, on issue slots the launch would have
// left empty.  Measured (scripts/split_time.py, 10 kb pairs): 25 k and 50 k pairs 1.93 -> 1.50 ms (+29 %); 100 k pairs (539 of the
// 1024 SIMDs hold two producers and two consumers) 2.51 ms against 2.52 ms — two wavefronts of this kernel on a SIMD already take
// 1.3 x the time of one, and those SIMDs end the launch either way.  So the library takes this kernel for launches of at most one
// wavefront per SIMD (scrg_api.cpp: the chunks of the host entry points, small batches) and genasm_lane_kernel otherwise — also
// because it has about 3 % more instructions in total, which is what counts once launches fill the GPU or overlap.
// Same results bit for bit (tests/test_gpu_parity.py::test_one_and_two_wavefronts_per_window_agree).
//
// The record of a lane and round (structure of arrays, one dword per field and lane): D, X, Im (the masks of pass 1, raw),
// ti | flags (the lane holds a pair / this is its first window / its last), the pair's index, its edits so far; plus the
// 31 insertion-run lengths of the window.
constexpr uint32_t SPLIT_FIELDS = 6;
constexpr uint32_t SPLIT_BUF_BYTES = SPLIT_FIELDS * 256u + 64u * LANE_SCRATCH_BYTES;                // 3840
constexpr uint32_t SPLIT_PAIR_LDS_BYTES = 64u * (LANE_RING_BYTES + LANE_EQ_BYTES + LANE_NOMATCH_BYTES) + 2u * SPLIT_BUF_BYTES;     // 14592 per producer / consumer pair
constexpr uint32_t SPLIT_WG_LDS_BYTES = 4u * SPLIT_PAIR_LDS_BYTES + 64u;                              // + the producers' "done" flags, two rounds x four
constexpr uint32_t SPLIT_VALID = 1u << 8, SPLIT_FIRST = 1u << 9, SPLIT_LAST = 1u << 10;

__global__ __launch_bounds__(512, 2) void genasm_lane_split_kernel(AlignArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    char* const lds_b = reinterpret_cast<char*>(lds);
    uint8_t* const lds8 = reinterpret_cast<uint8_t*>(lds);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6, duo = wave & 3u;
    const bool consumer = wave >= 4u;
    // LDS of a workgroup: the four producers' Eq regions first (slot-major, 2048 bytes each, see LANE_EQ_SLOT_STRIDE), then
    // each producer / consumer pair's ring, "no match" words and record buffers
    const uint32_t duo_b = 4u * LANE_EQ_REGION_BYTES + duo * (SPLIT_PAIR_LDS_BYTES - LANE_EQ_REGION_BYTES);
    const uint32_t ring_b = duo_b + lane * LANE_RING_BYTES;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_b;      // (an LDS ADDRESS: 0)
    const uint32_t eq_b = lds0 + duo * LANE_EQ_REGION_BYTES + lane * 8u;
    const uint32_t nomatch_b = lds0 + duo_b + 64u * LANE_RING_BYTES + lane * LANE_NOMATCH_BYTES;
    const uint32_t bufs_b = duo_b + 64u * (LANE_RING_BYTES + LANE_NOMATCH_BYTES);
    const uint32_t flags_w = (4u * SPLIT_PAIR_LDS_BYTES) >> 2;                 // dword index of done[2][4]
    const uint32_t W = (uint32_t)a.W;
    const uint32_t TBL = (uint32_t)a.tb_limit;         // W - O, 1..31
    const uint32_t TBc = min(TBL, (uint32_t)LANE_TB_COLS);
    const uint32_t nsh = 32u - TBc;
    auto rec_w = [&](uint32_t buf, uint32_t field) -> uint32_t { return ((bufs_b + buf * SPLIT_BUF_BYTES) >> 2) + field * 64u + lane; };
    auto len_b = [&](uint32_t buf) -> uint32_t { return bufs_b + buf * SPLIT_BUF_BYTES + SPLIT_FIELDS * 256u + lane * LANE_SCRATCH_BYTES; };
    auto all_done = [&](uint32_t r) -> bool {
        const uint32_t f = flags_w + (r & 1u) * 4u;
        return (lds[f] & lds[f + 1] & lds[f + 2] & lds[f + 3]) != 0u;
    };

    if (!consumer) {
        // ================= producer: queue, window setup, table, traceback pass 1 =================
        bool has_pair = false, active = true;
        uint32_t pair = 0;
        uint64_t text_off = 0, read_off = 0;
        uint32_t text_len = 0, read_len = 0;
        uint32_t ref_idx = 0, read_idx = 0, edits = 0;
        uint32_t revm = 0;                 // (see genasm_lane_kernel)
        WindowWords twords = {0, 0, 0, 0}, pwords = {0, 0, 0, 0};
        bool queue_empty = false;          // wave-uniform
        uint32_t st_rounds = 0;
        const uint32_t wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
        for (uint32_t r = 0;; r++) {
            const uint32_t buf = r & 1u;
            uint32_t first = 0;
            if (active) {
                if (!SCRG_SW(a, 1)) {
                    const uint32_t pr = (r + wave_slot) & 3u;          // (one step per round: see genasm_lane_kernel)
                    if (pr == 0) __builtin_amdgcn_s_setprio(0);
                    else if (pr == 1) __builtin_amdgcn_s_setprio(1);
                    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
                    else __builtin_amdgcn_s_setprio(3);
                }
                // a finished pair has been handed over with its last window: the lane is free (genasm_cpu.cpp:440-460)
                has_pair = has_pair && read_idx < read_len;
                for (;;) {
                    const bool want = !has_pair && !queue_empty;
                    if (!__any(want)) break;
                    const uint64_t askers = __ballot(want);
                    const int firstl = __ffsll((unsigned long long)askers) - 1;
                    uint32_t base = 0;
                    if ((int)lane == firstl) base = atomicAdd(a.counter, (uint32_t)__popcll(askers));
                    base = (uint32_t)__shfl((int)base, firstl);
                    const uint32_t idx = base + (uint32_t)__popcll(askers & ((1ull << lane) - 1ull));
                    const bool got = want && idx < a.n_pairs;
                    if (__any(want && idx >= a.n_pairs)) queue_empty = true;
                    if (got) {
                        const scrg_pair_desc pd = a.pairs[idx];
                        pair = idx;
                        text_off = pd.text_off;
                        read_off = a.stranded ? pd.read_off & ~SCRG_READ_REVCOMP : pd.read_off;
                        revm = (a.stranded && (pd.read_off & SCRG_READ_REVCOMP)) ? 0xffffffffu : 0u;
                        text_len = pd.text_len > 0xffffffffull ? 0xffffffffu : (uint32_t)pd.text_len;
                        read_len = (uint32_t)pd.read_len;
                        ref_idx = read_idx = edits = 0;
                        has_pair = true;
                        first = SPLIT_FIRST;
                        twords = load_window_words(a.seq, text_off, 0u, a.text_stride);
                        pwords = load_window_words(a.seq, read_off, lane_read_offset(0u, 0u, read_len, revm), a.read_stride);
                    }
                    // (an empty read is a pair of no windows: it is handed over as first and last at once, below)
                }
                if (!__any(has_pair)) active = false;
            }
            if (active) {
                // ---------------- window setup (genasm_cpu.cpp:417-420) ----------------
                const bool live = has_pair && read_idx < read_len;           // (false only for an empty read)
                const uint32_t n = (live && ref_idx < text_len) ? min(W, text_len - ref_idx) : 0u;
                const uint32_t m = live ? min(W, read_len - read_idx) : 1u;
                Planes tw = {0, 0}, pw = {0, 0};
                if (live) {
                    tw = window_planes(twords);
                    pw = window_planes(pwords);
                }
                uint64_t tab[LANE_TB_COLS];
                const uint32_t jlim = live ? min(m, TBL) : 0u;
                const uint32_t stop = 0x80000000u >> jlim;
                const bool short_n = __any(live && n != 64u);
                const bool wave_rev = a.stranded && __any(live && revm != 0u);
                uint64_t rlo, rhi;
                lane_pattern_rev(pw, read_len - read_idx, revm, wave_rev, rlo, rhi);
                if (short_n) lane_window_table<true>(tw, rlo, rhi, n, m, stop, tab, eq_b, nomatch_b);
                else lane_window_table<false>(tw, rlo, rhi, n, m, stop, tab, eq_b, nomatch_b);
                // ---------------- traceback pass 1 (genasm_cpu.cpp:290-409; see genasm_lane_kernel) ----------------
                uint32_t j = 0, nDm = 0, Xm = 0, nIm = 0;
                const uint32_t lb = len_b(buf);
                auto walk = [&](auto full_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                    for (int i = 0; i < LANE_TB_COLS; i++) {
                        if (!FULL && (uint32_t)i >= TBL) continue;         // (uniform)
                        const uint32_t x = bitop3<TT_NIV>((uint32_t)(tab[i] >> 32), (uint32_t)tab[i], stop) << j;
                        const uint32_t ni = ffbh_u32(x);
                        lds8[lb + i] = (uint8_t)ni;
                        nIm = __builtin_amdgcn_alignbit(nIm, x, 31);
                        j += ni;
                        const uint64_t both = tab[i] << j;
                        const uint32_t nt1 = (uint32_t)(both >> 32), t0 = (uint32_t)both;
                        nDm = __builtin_amdgcn_alignbit(nDm, nt1, 31);
                        Xm = __builtin_amdgcn_alignbit(Xm, t0, 31);
                        j -= neg_mask(nt1);
                    }
                };
                if (TBL == (uint32_t)LANE_TB_COLS) walk(std::true_type{});
                else walk(std::false_type{});
                const uint32_t notD = nDm << nsh, Xraw = Xm << nsh;
                const uint32_t Im = ~nIm << nsh;
                const uint32_t ti_stopped = (31u - ffbl_u32((notD >> 1) | Im)) & 31u;
                const uint32_t ti = bitop3<TT_BFI>(TBc, ti_stopped, neg_mask(j - jlim));       // j < jlim ? TBc : ti_stopped
                const uint32_t A = ~(0xffffffffu >> ti);
                const uint32_t D = ~notD & A, X = Xraw & A;
                edits += j - ti + 2u * (uint32_t)__builtin_popcount(D) + (uint32_t)__builtin_popcount(X);
                ref_idx += ti;
                read_idx += j;
                // the next window's words, asked for now (a pair that is finished reads its padding)
                twords = load_window_words(a.seq, text_off, ref_idx, a.text_stride);
                const uint32_t left_x = read_idx ^ read_len;                     // (see genasm_lane_kernel)
                pwords = load_window_words(a.seq, read_off, lane_read_offset(read_idx & neg_mask(left_x | (0u - left_x)), read_idx, read_len, revm), a.read_stride);
                const uint32_t last = (has_pair && read_idx >= read_len) ? SPLIT_LAST : 0u;
                lds[rec_w(buf, 0)] = D;
                lds[rec_w(buf, 1)] = X;
                lds[rec_w(buf, 2)] = Im;
                lds[rec_w(buf, 3)] = ti | (has_pair ? SPLIT_VALID : 0u) | first | last;
                lds[rec_w(buf, 4)] = pair;
                lds[rec_w(buf, 5)] = edits;
                st_rounds++;
            } else {
                lds[rec_w(buf, 3)] = 0u;                                   // nothing for the consumer
            }
            if (lane == 0) lds[flags_w + buf * 4u + duo] = active ? 0u : 1u;
            __syncthreads();
            if (all_done(r)) break;
        }
        if (SCRG_TIMING(a) && lane == 0) atomicAdd((unsigned long long*)&a.stats[0], (unsigned long long)st_rounds);
        return;
    }

    // ================= consumer: traceback pass 2, the CIGAR ring, the stores, retiring pairs =================
    bool open = false;                 // my lane has a pair in progress
    uint32_t pair = 0;
    uint64_t cigar_off = 0;
    uint32_t cigar_cap = 0;
    int32_t nr = -1;                   // index of the last committed run; n_runs = nr + 1
    uint32_t flushed = 0;              // runs below this index are in HBM (a multiple of 16)
    auto write_piece = [&]() {
        const uint32_t rd = (ring_b >> 2) + ((flushed & 16u) >> 1);
        uint32_t w[8];
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = lds[rd + k];
        if (flushed + 16u <= cigar_cap) {
            uint4* const dst = reinterpret_cast<uint4*>(a.runs + cigar_off + flushed);
            dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
            dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
        flushed += 16u;
    };
    auto flush_pieces = [&]() {                      // (one piece per lane and look: see genasm_lane_kernel)
        const bool need = open && nr - (int32_t)flushed >= 16;
        if (__any(need)) {
            if (need) write_piece();
        }
    };
    auto consume = [&](uint32_t buf) {
        const uint32_t meta = lds[rec_w(buf, 3)];
        if (!__any((meta & SPLIT_VALID) != 0u)) return;
        const bool valid = (meta & SPLIT_VALID) != 0u;
        const uint32_t ti = meta & 31u;
        const uint32_t D = valid ? lds[rec_w(buf, 0)] : 0u, X = valid ? lds[rec_w(buf, 1)] : 0u, Im = valid ? lds[rec_w(buf, 2)] : 0u;
        if (valid && (meta & SPLIT_FIRST)) {
            pair = lds[rec_w(buf, 4)];
            const scrg_pair_desc pd = a.pairs[pair];
            cigar_off = pd.cigar_off;
            cigar_cap = pd.cigar_cap > 0xffffffffull ? 0xffffffffu : (uint32_t)pd.cigar_cap;
            nr = -1;
            flushed = 0;
            open = true;
        }
        const uint32_t A = ~(0xffffffffu >> ti);
        const uint32_t B = ((D ^ (D >> 1)) | (X ^ (X >> 1)) | Im | 0x80000000u) & A;    // a D / X / = run starts here
        const uint32_t scr_b = len_b(buf);
        uint32_t E = valid ? (B | Im) : 0u;
        uint32_t c = ffbh_u32(E);
        uint32_t ni = lds8[scr_b + c];
        uint32_t q = 2u * (uint32_t)nr + 2u;       // byte offset of the next free slot (see genasm_lane_kernel)
        auto event = [&]() {
            const uint32_t sh = 31u - c;
            const uint32_t bit = 0x80000000u >> (c & 31u);
            *reinterpret_cast<uint16_t*>(lds_b + ring_b + (q & 62u)) = (uint16_t)(((uint32_t)'I' << 8) | ni);
            q = add_twice(q, __builtin_amdgcn_ubfe(Im, sh, 1));
            E = bitop3<TT_ANDN>(E, bit, bit);
            const uint32_t nx = ffbh_u32(E);
            ni = lds8[scr_b + nx];
            const uint32_t w = add3(mad24(__builtin_amdgcn_ubfe(X, sh, 1), 27u << 8, mad24(__builtin_amdgcn_ubfe(D, sh, 1), 7u << 8, sh)),
                                    min(nx, ti), ((uint32_t)'=' << 8) - 31u);
            *reinterpret_cast<uint16_t*>(lds_b + ring_b + (q & 62u)) = (uint16_t)w;
            q = add_twice(q, __builtin_amdgcn_ubfe(B, sh, 1));
            c = nx;
        };
        uint32_t trips = 0;
        while (__any(E != 0u)) {
            event();
            event();
            if (++trips == 3u) {
                trips = 0;
                nr = ((int32_t)q >> 1) - 1;
                flush_pieces();
            }
        }
        nr = ((int32_t)q >> 1) - 1;
        flush_pieces();
        // retire the pairs whose last window this was (genasm_cpu.cpp:440-460)
        const bool fin = valid && (meta & SPLIT_LAST) != 0u;
        if (__any(fin)) {
            if (fin) {
                const uint32_t n_runs = (uint32_t)(nr + 1);
                while (n_runs - flushed >= 16u) write_piece();
                const uint32_t rem = n_runs - flushed;
                const uint32_t rd = (ring_b >> 2) + ((flushed & 16u) >> 1);
                uint32_t* const dst = reinterpret_cast<uint32_t*>(a.runs + cigar_off + flushed);
                for (uint32_t k = 0; 2u * k < rem; k++)
                    if (flushed + 2u * k < cigar_cap) dst[k] = lds[rd + k];
                a.ed[pair] = (int64_t)lds[rec_w(buf, 5)];
                a.n_runs[pair] = n_runs;
                a.status[pair] = n_runs > cigar_cap ? 1u : 0u;
                open = false;
            }
        }
    };
    for (uint32_t r = 0;; r++) {
        if (r != 0u) consume((r - 1u) & 1u);
        __syncthreads();
        if (all_done(r)) {
            consume(r & 1u);
            break;
        }
    }
}

hipError_t launch_align_lane_split(const AlignArgs& a, int grid, hipStream_t s)
{
    // grid counts PRODUCER wavefronts (64 pairs in flight each); a workgroup is four of them and their four consumers
    const dim3 g((grid + 3) / 4), b(512);
    hipLaunchKernelGGL(genasm_lane_split_kernel, g, b, SPLIT_WG_LDS_BYTES, s, a);
    return hipGetLastError();
}

hipError_t launch_align_lane(const AlignArgs& a, int grid, size_t lds_bytes, hipStream_t s, bool edits)
{
    // grid counts wavefronts, lds_bytes is per wavefront
    const unsigned wpg = SCRG_SW(a, 64) ? 1u : (SCRG_SW(a, 128) ? 2u : 4u);      // (profiling builds: wavefronts per workgroup)
    const dim3 g((grid + wpg - 1) / wpg), b(64 * wpg);
    if (edits) hipLaunchKernelGGL(genasm_lane_kernel<true>, g, b, wpg * lds_bytes, s, a);
    else hipLaunchKernelGGL(genasm_lane_kernel<false>, g, b, wpg * lds_bytes, s, a);
    return hipGetLastError();
}

}  // namespace scrg
