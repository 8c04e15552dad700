// genasm_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels for the
// GenASM/Scrooge hot path.  No MFMA: this is 64-bit logic/shift work on the
// VALU with the traceback table in LDS.
//
// What is computed (reference semantics, src/genasm_cpu.cpp):
//   window loop            :411-438
//   GenASM-DC (distance)   :210-288   R[i][d] = mat & sub & ins & del
//   GenASM-TB (traceback)  :290-409   priority I > D > X > =, per-window run flush
//   pattern masks          :178-198
// and Scrooge's three optimisations:
//   SENE  only the centre entry R[i][d] is stored (:63-78)
//   DENT  only W-O+1 columns and the top min(W-O+1, m) bits are stored (:200-208, :258-267)
//   ET    the sweep stops at the first row that reaches the goal bit (:278-283)
//
// Mapping onto a wavefront: a wave holds 64/G independent pairs
// ("slots"), G lanes each.  Lane t of a slot owns text columns
// [t*CPL, (t+1)*CPL), CPL = 64/G, keeps row d-1 of those columns and their
// match masks in VGPRs, and sweeps rows skewed by one step per lane (lane t is
// at row step-(G-1-t)): the only cross-lane traffic is one 64-bit DPP shift per
// row.  G = 64 is the "one pair per wavefront, one lane per bitvector word"
// mapping; G = 8 packs 8 pairs per wave and is the default.  Rows of R
// needed by the traceback go to LDS (rows >= lds_rows spill to an HBM scratch).
// The traceback itself is lane-parallel: lane l of a slot tests the cell l steps
// down the current diagonal, a ballot finds the first non-match.
//
// The G = 8 instantiation has a second, faster layout for the common window (W = 64,
// full text window, distance <= 15): the same table indexed by DIAGONAL, where a row
// is a carry chain solved by one 64-bit addition per diagonal, no skew is needed and
// the traceback finds each edit with one count-leading-zeros ("diagonal-major window"
// below).  Rounds of the two kinds interleave freely inside a pair.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "genasm_kernels.h"
#include "genasm_device.h"

namespace scrg {

// ----------------------------------------------------------------------------
// the aligner (device helpers: genasm_device.h)
// ----------------------------------------------------------------------------

// WIDE stores every column's whole 64-bit entry instead of the DENT dword of columns 0..31: needed
// when the traceback may consume more than 31 characters per window (W-O > 31, e.g. the reference's
// O sweeps, scripts/profile.py:88-100).  It costs 4x the LDS per row and uses the generic traceback.
template <int G, bool WIDE>
__global__ __launch_bounds__(64, (G >= 8 ? 3 : 2)) void genasm_align_kernel(AlignArgs a)
{
    constexpr int CPL = 64 / G;          // text columns per lane
    constexpr int SLOTS = 64 / G;        // pairs per wavefront
    constexpr uint32_t ROWDW = WIDE ? 128u : 32u;   // dwords per stored row of R
    constexpr uint32_t OBUF_DWORDS = 16; // CIGAR runs leave the CU in aligned 32-byte pieces (16 runs) out of a 32-run ring
    constexpr uint32_t GMASK = (G == 32) ? 0xffffffffu : ((G == 64) ? 0xffffffffu : ((1u << (G & 31)) - 1u));

    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];

    const int lane = threadIdx.x;
    const int t = lane % G;              // lane within slot
    const int gbase = lane - t;          // first lane of my slot
    const int slot = lane / G;
    const bool leader = (t == 0);

    const int W = a.W;
    const int TBL = a.tb_limit;          // W - O
    const int RB = a.lds_rows;
    // LDS: [SLOTS x 16 dwords of CIGAR staging: two 32-byte pieces][SLOTS x 1 scratch dword]
    //      [SLOTS x (RB rows x 32 dwords + 1)][8 dwords pad]
    const uint32_t obuf = (uint32_t)slot * OBUF_DWORDS;            // 32 runs staged per slot, written out 16 at a time
    const uint32_t scratch_dw = SLOTS * OBUF_DWORDS + (uint32_t)slot;   // target of masked-off staging writes
    const uint32_t slot_stride = slot_stride_dwords(W, TBL, G, RB);  // RB rows + 1 word: conflict-free slot banks (>= 397 for the diagonal path)
    const uint32_t lds_slot = SLOTS * (OBUF_DWORDS + 1u) + slot * slot_stride;   // R[d][i] at lds[lds_slot + d*32 + i], d < RB
    uint16_t* const lds16 = reinterpret_cast<uint16_t*>(lds);
    uint32_t* const Rs = a.spill + ((size_t)blockIdx.x * SLOTS + slot) * (size_t)(SPILL_ROWS * ROWDW);
    const uint32_t spill_slot_b = (blockIdx.x * SLOTS + slot) * (uint32_t)(SPILL_ROWS * ROWDW * 4);   // byte offset of my slot's spill rows

    // mask with bit (first lane of slot s) set for every slot
    constexpr uint64_t leaders = leader_mask(G);

    // ---- per-slot state (replicated in the slot's G lanes) ----
    bool has_pair = false;
    uint32_t pair = 0;
    uint64_t text_off = 0, read_off = 0, cigar_off = 0;
    uint32_t text_len = 0, read_len = 0, cigar_cap = 0;
    uint32_t ref_idx = 0, read_idx = 0, n_runs = 0, edits = 0;
    bool overflow = false;
    bool queue_empty = false;            // wave-uniform
    uint32_t st_rounds = 0, st_steps = 0, st_macro = 0;   // profiling counters (a.stats != nullptr)
    uint32_t st_diag = 0, st_diag_fail = 0;
    uint64_t cy_ddc = 0, cy_dtb = 0;
    uint32_t diag_skip = 0, diag_fails = 0;     // rounds to stay off the diagonal-major path after repeated failures
    bool force_column = false;                  // the next round must be column-major (some slot sat the last one out)
    uint64_t cy_fetch = 0, cy_setup = 0, cy_dc = 0, cy_tb = 0, cy_tbloop = 0;
    const bool timing = SCRG_TIMING(a);            // (compile-time false in the shipped build: genasm_kernels.h)

    for (;;) {
        const uint64_t tm0 = timing ? __builtin_readcyclecounter() : 0;
        // ---------------- retire finished pairs, fetch new ones ----------------
        for (;;) {
            const bool fin = has_pair && read_idx >= read_len;
            if (fin) {
                // write out the runs still staged in LDS (the slice is a multiple of 16 runs long, so
                // rounding the tail up to whole dwords stays inside it)
                const uint32_t done = n_runs < cigar_cap ? n_runs : cigar_cap;
                const uint32_t rem = done & 15u;
                const uint32_t piece = ((done >> 4) & 1u) * 8u;          // which half of the ring holds the tail
                uint32_t* const dst = reinterpret_cast<uint32_t*>(a.runs + cigar_off + (done - rem));
                for (uint32_t k = (uint32_t)t; 2u * k < rem; k += (uint32_t)G) dst[k] = lds[obuf + piece + k];
                if (leader) {
                    a.ed[pair] = (int64_t)edits;
                    a.n_runs[pair] = n_runs;
                    a.status[pair] = overflow ? 1u : 0u;
                }
            }
            has_pair = has_pair && !fin;
            const bool want = !has_pair && !queue_empty;
            if (!__any(want)) break;

            // one atomic per wavefront for all the slots that want a pair (a queue of millions of short
            // reads is otherwise bound by same-address atomics at L2)
            uint32_t idx = 0xffffffffu;
            {
                const uint64_t askers = __ballot(want && leader);
                const int first = __ffsll((unsigned long long)askers) - 1;
                uint32_t base = 0;
                if (lane == first) base = atomicAdd(a.counter, (uint32_t)__popcll(askers));
                base = (uint32_t)__shfl((int)base, first);
                if (want && leader) idx = base + (uint32_t)__popcll(askers & ((1ull << lane) - 1ull));
            }
            idx = (uint32_t)__shfl((int)idx, gbase);
            const bool got = want && idx < a.n_pairs;
            if (__any(want && idx >= a.n_pairs)) queue_empty = true;
            if (got) {
                const scrg_pair_desc pd = a.pairs[idx];
                pair = idx;
                text_off = pd.text_off;
                read_off = pd.read_off;
                text_len = pd.text_len > 0xffffffffull ? 0xffffffffu : (uint32_t)pd.text_len;
                read_len = (uint32_t)pd.read_len;
                cigar_off = pd.cigar_off;
                cigar_cap = pd.cigar_cap > 0xffffffffull ? 0xffffffffu : (uint32_t)pd.cigar_cap;
                ref_idx = read_idx = n_runs = edits = 0;
                overflow = false;
                has_pair = true;
            }
        }
        if (!__any(has_pair)) break;

        const uint64_t tm1 = timing ? __builtin_readcyclecounter() : 0;
        // ---------------- window setup (genasm_cpu.cpp:417-420) ----------------
        const uint32_t n = (has_pair && ref_idx < text_len) ? min((uint32_t)W, text_len - ref_idx) : 0u;
        const uint32_t m = has_pair ? min((uint32_t)W, read_len - read_idx) : 1u;   // >= 1 for live pairs

        // ---------------- diagonal-major window (full windows with a small distance) ----------------
        // The common case — W = 64, a full window (n = m = 64) and a window distance <= 15 — is computed
        // in a different layout: one 64-bit word per DIAGONAL delta = j - i (pattern index minus text
        // index), bit p = 63 - i, in positive logic (a = ~R: 1 = "alignable").  Then
        //   a(i,j,d) = y(i,j,d) | (match(i,j) & a(i+1,j+1,d)),   y = a(i+1,j+1,d-1) | a(i,j+1,d-1) | a(i+1,j,d-1)
        // (genasm_cpu.cpp:246-252 restated): the in-row dependency runs along the word, from bit p-1
        // to bit p, i.e. it is a carry chain with generate y and propagate match, solved for all 64
        // positions by ONE 64-bit addition: C = ((y|mt) + y) ^ (y|mt) ^ y, a = y | (mt & C).  Rows
        // therefore need no skew across lanes (the column layout below needs G-1 extra steps per window)
        // and y only needs the previous row of the two neighbouring diagonals: y_x = S_x | A_{x+1} | S_{x-1}
        // with S = A << 1.  Only the band |delta| <= d can influence the goal cell (0,0,d) and the cells
        // the traceback visits, so 32 diagonals delta = -16..15 (4 per lane) are exact for d <= 15; the
        // text/pattern ends enter as carry-ins (boundary column i = 64: a = [64-j <= d], genasm_cpu.cpp:239-245)
        // and as forced cells below each diagonal's first valid bit (pattern end j = 64: a = 1).
        // The traceback walks a diagonal with one count-leading-zeros per edit.  Windows this path does
        // not cover (a short text window, larger distances) take the column-major path below; results are
        // identical (tests/proto/diag_proto.c restates this arithmetic on the CPU for tests/test_diag_proto.py).
        if constexpr (G == 8 && !WIDE) {
            // LDS rows of this layout: high dwords (positions i <= 31) of the 32 diagonals; from row 8 on the
            // traceback can only be within |delta| <= 7, so those rows keep the 16 diagonals of lanes 2..5
            // and 16 rows take 8*32 + 8*16 + 4 (parking) = 396 dwords
            constexpr int cmp_row = (int)DIAG_WIDE_ROWS, max_rows = 15;      // (slot_stride_dwords() reserves the 397 dwords: 16 rows)
            // slots with a short text window (the text ends inside it) sit a diagonal round out; so do slots whose
            // window turns out to need more than max_rows rows.  Either kind makes the NEXT round a
            // column-major one (which serves every slot), so a diagonal round pays off when more than half
            // of the live slots can use it.
            const bool capable = has_pair && n == 64u && n_runs + 64u <= cigar_cap;   // (a window adds < 64 runs)
            const uint32_t n_live = (uint32_t)__popcll(__ballot(has_pair));
            const uint32_t n_cap = (uint32_t)__popcll(__ballot(capable));
            bool try_diag = W == 64 && !SCRG_SEL(a.debug, SCRG_SWITCH_NO_DIAG) && !force_column && (n_cap == n_live || n_cap >= 5u * G);
            force_column = false;
            if (try_diag && diag_skip) {
                diag_skip--;
                try_diag = false;
            }
            if (try_diag) {
                // (an opaque copy of the lane index: per-lane constants of this path are recomputed each round
                // instead of living in — and being spilled from — registers across the column-major path)
                int tq = t;
                asm volatile("" : "+v"(tq));
                // ---- setup: match words of my four diagonals x = 4t+k (delta = x-16) ----
                uint64_t mt[4];
                int32_t cnt[4];            // carry-in of diagonal delta <= 0 at row d is [d >= -delta]: sign of -delta-1-d
                uint64_t A0[4], S0[4], A1[4], S1[4];
                {
                    Planes tw = {0, 0}, pw = {0, 0};
                    if (has_pair) {
                        tw = load_window(a.seq, text_off + ref_idx);
                        pw = load_window(a.seq, read_off + read_idx);
                    }
                    const uint64_t trl = brev64(tw.lo), trh = brev64(tw.hi);     // bit p = text char 63-p
                    const uint64_t prl = brev64(pw.lo), prh = brev64(pw.hi);     // bit p = pattern char 63-p
                    // pattern planes moved onto diagonal delta: P << delta (>> for delta < 0) = (P << 15) >> (15 - delta)
                    const uint32_t l0 = (uint32_t)prl << 15, l1 = __builtin_amdgcn_alignbit((uint32_t)(prl >> 32), (uint32_t)prl, 17),
                                   l2 = (uint32_t)(prl >> 32) >> 17;
                    const uint32_t h0 = (uint32_t)prh << 15, h1 = __builtin_amdgcn_alignbit((uint32_t)(prh >> 32), (uint32_t)prh, 17),
                                   h2 = (uint32_t)(prh >> 32) >> 17;
                    // A pattern window of m < 64 characters (the last windows of a read) moves the pattern end, and
                    // with it every boundary role, by 64-m diagonals: with e = delta + 64 - m, diagonals e >= 1 start
                    // at the pattern end (forced cell at bit e-1, valid bits from e), e = 0 has carry-in 1, e < 0
                    // carry-in [d >= -e].  m = 64 is the e = delta special case with the masks from one funnel shift.
                    const bool short_pattern = __any(capable && m != 64u);
                    auto match_word = [&](int k, uint32_t v_lo, uint32_t v_hi) {
                        const uint32_t sh = 31u - (uint32_t)(4 * tq + k);
                        const uint32_t pl_lo = __builtin_amdgcn_alignbit(l1, l0, sh), pl_hi = __builtin_amdgcn_alignbit(l2, l1, sh);
                        const uint32_t ph_lo = __builtin_amdgcn_alignbit(h1, h0, sh), ph_hi = __builtin_amdgcn_alignbit(h2, h1, sh);
                        const uint32_t m_lo = v_lo & ~(((uint32_t)trl ^ pl_lo) | ((uint32_t)trh ^ ph_lo));
                        const uint32_t m_hi = v_hi & ~(((uint32_t)(trl >> 32) ^ pl_hi) | ((uint32_t)(trh >> 32) ^ ph_hi));
                        mt[k] = ((uint64_t)m_hi << 32) | m_lo;
                        A0[k] = 0;
                    };
                    if (!short_pattern) {
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t x = (uint32_t)(4 * tq + k);
                            const uint32_t sh = 31u - x;
                            // valid positions 0 <= j < 64: both ends from one funnel shift of 64 ones
                            match_word(k, __builtin_amdgcn_alignbit(0xffffffffu, 0xffff8000u, sh),
                                       __builtin_amdgcn_alignbit(0x00007fffu, 0xffffffffu, sh));
                            cnt[k] = x <= 16u ? (int32_t)(15u - x) : 0x40000000;
                            S0[k] = x >= 17u ? (1ull << (x - 17u)) : 0ull;     // "row -1": only the forced pattern-end cells
                        }
                    } else {
                        asm volatile("" ::: "memory");       // (a real branch: only the last windows of a read come here)
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t x = (uint32_t)(4 * tq + k);
                            const int32_t e = (int32_t)x - 16 + (int32_t)(64u - m);
                            const uint64_t lowm = e <= 0 ? ~0ull : (e >= 64 ? 0ull : (~0ull << (e & 63)));     // j < m
                            match_word(k, (uint32_t)lowm,
                                       (uint32_t)(lowm >> 32) & __builtin_amdgcn_alignbit(0x00007fffu, 0xffffffffu, 31u - x));   // j >= 0
                            cnt[k] = e <= 0 ? -e - 1 : 0x40000000;
                            S0[k] = (e >= 1 && e <= 64) ? (1ull << ((e - 1) & 63)) : 0ull;
                        }
                    }
                }
                const uint64_t tmd0 = timing ? __builtin_readcyclecounter() : 0;
                // ---- rows ----
                uint32_t not_first = (tq != 0) ? ~0u : 0u, not_last = (tq != G - 1) ? ~0u : 0u;
                asm volatile("" : "+v"(not_first), "+v"(not_last));
                uint32_t ddw = 0;
                int32_t hit_cmp = (capable && tq == 4) ? 0 : INT32_MIN;      // lane 4, k = 0 holds delta = 0: goal = bit 63
                asm volatile("" : "+v"(hit_cmp));                            // (a register, not a select recomputed per row)
                bool found = false;
                int waiting = __popcll(__ballot(capable) & leaders);        // slots that have not reached the goal yet
                uint32_t waddr = lds_slot + 4u * (uint32_t)tq, wstride = 32u;
                int d = 0;
                constexpr int TT_A = bitop3_table([](int sum, int y, int mm) { return y | (mm & (sum ^ (y | mm) ^ y)); });
                auto row = [&](const uint64_t (&Ap)[4], const uint64_t (&Sp)[4], uint64_t (&Ac)[4], uint64_t (&Sc)[4]) {
                    // A of diagonal x+1 for my k = 3 (from the next lane), S of diagonal x-1 for my k = 0 (from the previous one)
                    uint32_t un_lo, un_hi, dp_lo, dp_hi;
                    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %4 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_mov_b32_dpp %1, %5 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_mov_b32_dpp %2, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_mov_b32_dpp %3, %7 wave_shr:1 row_mask:0xf bank_mask:0xf"
                                 : "=&v"(un_lo), "=&v"(un_hi), "=&v"(dp_lo), "=&v"(dp_hi)
                                 : "v"((uint32_t)Ap[0]), "v"((uint32_t)(Ap[0] >> 32)), "v"((uint32_t)Sp[3]), "v"((uint32_t)(Sp[3] >> 32)));
                    const uint64_t up_n = ((uint64_t)(un_hi & not_last) << 32) | (un_lo & not_last);
                    const uint64_t dn_p = ((uint64_t)(dp_hi & not_first) << 32) | (dp_lo & not_first);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint64_t up = k < 3 ? Ap[k < 3 ? k + 1 : 3] : up_n;
                        const uint64_t dn = k > 0 ? Sp[k > 0 ? k - 1 : 0] : dn_p;
                        const uint32_t u = (uint32_t)cnt[k] >> 31;   // boundary cell a(64, j+1, d) of this diagonal: the chain's carry-in
                        cnt[k]--;
                        asm volatile("" : "+v"(cnt[k]));              // keep it a VGPR counter (an SGPR-operand add issues at half rate)
                        // generate: y, plus the carry-in where the first cell matches
                        const uint32_t y_lo = bitop3<0xFE>((uint32_t)Sp[k], (uint32_t)up, (uint32_t)dn);
                        const uint32_t y_hi = bitop3<0xFE>((uint32_t)(Sp[k] >> 32), (uint32_t)(up >> 32), (uint32_t)(dn >> 32));
                        const uint32_t g_lo = bitop3<0xF8>(y_lo, (uint32_t)mt[k], u);                  // y | (mt & u)
                        const uint64_t g = ((uint64_t)y_hi << 32) | g_lo;
                        const uint64_t sum = add64(g | mt[k], g);
                        const uint32_t a_lo = bitop3<TT_A>((uint32_t)sum, g_lo, (uint32_t)mt[k]);
                        const uint32_t a_hi = bitop3<TT_A>((uint32_t)(sum >> 32), y_hi, (uint32_t)(mt[k] >> 32));
                        Ac[k] = ((uint64_t)a_hi << 32) | a_lo;
                        Sc[k] = shl1_add64(Ac[k], (uint64_t)u);     // (a << 1) | boundary cell: the next row's sub/del source
                    }
                    // the traceback reads positions i <= 31: the high dwords (every row 0..15 has its place)
#pragma unroll
                    for (int k = 0; k < 4; k++) lds[waddr + k] = (uint32_t)(Ac[k] >> 32);
                    waddr += wstride;
                    const uint64_t hits = __ballot((int32_t)(Ac[0] >> 32) < hit_cmp);
                    if (hits) {
                        const uint64_t lead = (hits >> 4) & leaders;
                        const uint64_t newly = ((uint64_t)((uint32_t)lead * 0xffu)) | ((uint64_t)((uint32_t)(lead >> 32) * 0xffu) << 32);
                        if ((newly >> lane) & 1ull) {
                            hit_cmp = INT32_MIN;
                            ddw = (uint32_t)d;
                            found = true;
                        }
                        asm volatile("" : "+v"(hit_cmp));
                        waiting -= __popcll(lead);
                    }
                    d++;
                };
                auto rows_until = [&](int end) {                      // an even number of rows unless every slot is done
                    for (;;) {
                        row(A0, S0, A1, S1);
                        if (waiting == 0 || d == end) break;
                        row(A1, S1, A0, S0);
                        if (waiting == 0 || d == end) break;
                    }
                };
                rows_until(cmp_row);
                if (waiting != 0) {                                   // compact rows: lanes 2..5 keep storing, the rest park
                    const bool mid = (tq >= 2 && tq <= 5);
                    waddr = lds_slot + (mid ? DIAG_LATE_BASE + 16u * DIAG_WIDE_ROWS + 4u * (uint32_t)tq : DIAG_PARK_DWORD);
                    wstride = mid ? 16u : 0u;
                    rows_until(max_rows + 1);
                }
                const bool all_done = waiting == 0;
                const uint64_t tmd1 = timing ? __builtin_readcyclecounter() : 0;
                st_steps += (uint32_t)d;
                {
                    // ---- traceback along diagonals (genasm_cpu.cpp:290-409).  The text limit :312 cannot trigger
                    // (n = 64 > W-O), and the last-character rule :336-343 (insertion whenever there is budget)
                    // is what the forced pattern-end cell produces by itself: it sits on diagonal x+1 at the
                    // position of the last character, and insertions have the highest priority.
                    // One iteration = the '=' run up to the next edit (count-leading-zeros over the three
                    // neighbouring diagonals of row d-1) plus that edit.  Runs go to the staging ring as soon
                    // as they start — every lane of the slot writes the same halfword — and a repeated edit
                    // rewrites the last run, the only one that can still grow.  Room for a whole window's runs
                    // was checked up front (capable), so nothing here tests the capacity. ----
                    uint32_t actmask = found ? ~0u : 0u;
                    uint32_t ti = 0, tj = 0, dd = ddw;
                    const uint32_t jlim = m < (uint32_t)TBL ? m : (uint32_t)TBL;     // j < m && j < W-O (:307-310)
                    uint32_t xoff = lds_slot + 15u;                  // lds_slot + (x - 1), x = j - i + 16
                    uint32_t nr2 = 2u * n_runs;                       // byte position of the next run
                    uint32_t flushed = n_runs & ~15u;                 // runs below this are in HBM (the column path's invariant)
                    const uint32_t obuf_b = 4u * obuf, dummy_b = 4u * scratch_dw;
                    char* const lds_b = reinterpret_cast<char*>(lds);
                    // 16-run pieces leave for HBM once a later run exists (or the window is over)
                    auto flush_check = [&](bool final) {
                        for (;;) {
                            const uint32_t nr = nr2 >> 1;
                            const uint32_t upto = final ? nr : (nr ? nr - 1u : 0u);
                            const bool need = found && (upto & ~15u) > flushed;
                            if (!__any(need)) break;
                            if (need) {
                                if (!SCRG_ABL(a, 4)) {
                                    const uint32_t piece = ((flushed >> 4) & 1u) * 8u;
                                    uint32_t* const dst = reinterpret_cast<uint32_t*>(a.runs + cigar_off + flushed);
                                    for (uint32_t k = (uint32_t)t; k < 8u; k += (uint32_t)G) dst[k] = lds[obuf + piece + k];
                                }
                                flushed += 16u;
                            }
                        }
                    };
                    int iter = 0;
                    uint32_t cur = 0;                                 // last run written if it is an edit run of this window: count | op << 8
                    uint32_t lim = (uint32_t)TBL < jlim ? (uint32_t)TBL : jlim;     // cells left before i or j reaches its limit (:307-310)
                    do {
                        st_macro++;
                        // row d-1 of diagonals x-1, x, x+1; garbage when d == 0 (masked below)
                        const uint32_t r = dd ? dd - 1u : 0u;
                        const uint32_t rmin = r < (uint32_t)cmp_row ? r : (uint32_t)cmp_row;
                        const uint32_t base = xoff + 16u * (r + rmin);              // 32 r for r < 8, DIAG_LATE_BASE + 16 r after
                        const uint32_t w_del = lds[base];              // a(i+1, j)   on diagonal x-1, one position down
                        const uint32_t w_sub = lds[base + 1];          // a(i+1, j+1) on diagonal x,   one position down
                        const uint32_t w_ins = lds[base + 2];          // a(i, j+1)   on diagonal x+1
                        const uint32_t roomm = neg_mask(0u - dd) & actmask;          // d > 0 (:313)
                        const uint32_t E = (((w_del | w_sub) << 1) | w_ins) & (0xffffffffu >> ti) & roomm;
                        const uint32_t run = (uint32_t)__clz((int)E) - ti;         // clz = 32 when no edit is available
                        const uint32_t edit = neg_mask(run - lim) & actmask;       // an edit ends the run inside the window
                        const uint32_t n_eq = (run < lim ? run : lim) & actmask;
                        // the '=' run
                        const uint32_t eqm = nz_mask(n_eq);
                        *reinterpret_cast<uint16_t*>(lds_b + bitop3<0xCA>(eqm, obuf_b + (nr2 & 62u), dummy_b)) =
                            (uint16_t)(n_eq | ((uint32_t)'=' << 8));
                        nr2 += eqm & 2u;
                        cur &= ~eqm;
                        // the edit that ends it, at position i + n_eq: priority I, D, X (:346-370)
                        const uint32_t ie = ti + n_eq;
                        const uint32_t sh = 31u - ie;
                        const uint32_t is_i = (uint32_t)__builtin_amdgcn_sbfe((int)w_ins, sh, 1u);
                        const uint32_t is_d = (uint32_t)__builtin_amdgcn_sbfe((int)(w_del << 1), sh, 1u) & ~is_i;
                        const uint32_t op8 = bitop3<0xCA>(is_i, (uint32_t)'I' << 8, bitop3<0xCA>(is_d, (uint32_t)'D' << 8, (uint32_t)'X' << 8));
                        const uint32_t merge = edit & neg_mask((cur ^ op8) - 256u);   // same op as the last run (cur != 0)
                        cur = bitop3<0xCA>(merge, cur, op8) + 1u;
                        *reinterpret_cast<uint16_t*>(lds_b + bitop3<0xCA>(edit, obuf_b + ((nr2 - (merge & 2u)) & 62u), dummy_b)) = (uint16_t)cur;
                        nr2 += edit & ~merge & 2u;
                        cur &= edit;
                        const uint32_t e1 = edit & 1u;
                        ti = ie + (e1 & ~is_i);
                        tj += n_eq + (e1 & ~is_d);
                        xoff += (e1 & is_i) - (e1 & is_d);
                        dd -= e1;
                        const uint32_t li = (uint32_t)TBL - ti, lj = jlim - tj;
                        lim = li < lj ? li : lj;
                        actmask = edit & nz_mask(lim);
                        if ((++iter & 7) == 0) flush_check(false);    // at most 16 runs between checks: the 32-run ring cannot wrap
                    } while (__any(actmask != 0u));
                    flush_check(true);          // runs never merge across windows (:400-403): everything staged is final
                    n_runs = nr2 >> 1;
                    if (found) {
                        edits += ddw - dd;
                        ref_idx += ti;
                        read_idx += tj;
                    }
                    st_rounds++;
                    st_diag++;
                    // slots that sat out or ran out of rows: serve them column-major next; after repeated
                    // row failures stay off this path for a while (high-error reads fail most rounds)
                    if (__any(has_pair && !found)) force_column = true;
                    if (all_done) {
                        diag_fails = diag_fails ? diag_fails - 1u : 0u;
                    } else {
                        st_diag_fail++;
                        diag_fails = diag_fails < 12u ? diag_fails + 1u : 12u;
                        if (diag_fails >= 3u) diag_skip = 1u << (diag_fails - 2u);      // 2, 4, ... 1024 rounds
                    }
                    if (timing) {
                        const uint64_t tmd2 = __builtin_readcyclecounter();
                        cy_fetch += tm1 - tm0;
                        cy_setup += tmd0 - tm1;
                        cy_ddc += tmd1 - tmd0;
                        cy_dtb += tmd2 - tmd1;
                    }
                    continue;
                }
            }
        }

        // Bit layout inside a window ("left-aligned"): pattern character j lives at bit 63-j,
        // i.e. every bitvector is the reference's (genasm_cpu.cpp:178-198, bit b <-> pattern[m-1-b])
        // shifted left by s = 64-m with zeros below.  The recurrence is shift-invariant as long as
        // the low s bits stay zero, the goal bit is always bit 63, and the DENT word (top
        // min(m,32) bits, genasm_cpu.cpp:200-208) is always the high dword.
        const uint64_t V = ~0ull << (64u - m);          // valid bits (m >= 1)
        uint64_t M[CPL];      // match mask of my columns: bit 63-j == 0 <=> pattern[j] == text[col]
        {
            Planes tw = {0, 0}, pw = {0, 0};
            if (has_pair) {
                tw = load_window(a.seq, text_off + ref_idx);
                pw = load_window(a.seq, read_off + read_idx);
            }
            const uint64_t plo = brev64(pw.lo);        // char k of the window -> bit 63-k
            const uint64_t phi = brev64(pw.hi);
            const uint32_t tlo = (uint32_t)(tw.lo >> (t * CPL));
            const uint32_t thi = (uint32_t)(tw.hi >> (t * CPL));
#pragma unroll
            for (int k = 0; k < CPL; k++) {
                const uint64_t sl = (uint64_t)(int64_t)(-(int32_t)((tlo >> k) & 1u));
                const uint64_t sh = (uint64_t)(int64_t)(-(int32_t)((thi >> k) & 1u));
                M[k] = ((plo ^ sl) | (phi ^ sh)) & V;
            }
            // columns at or past n behave as the all-insertions boundary column
            // (genasm_cpu.cpp:239-245): with mask V the recurrence reproduces V<<d there by itself
            if (__any(n < 64u)) {
#pragma unroll
                for (int k = 0; k < CPL; k++)
                    if ((uint32_t)(t * CPL + k) >= n) M[k] = V;
            }
        }

        // ---------------- GenASM-DC, skewed row sweep (genasm_cpu.cpp:210-288) ----------------
        // Lane t runs row d = step-(G-1-t).  Per cell (column k, row d), with the previous row's
        // entries kept both plain (prev) and pre-shifted (prevs = prev<<1):
        //   ins & sub & del = prevs[k] & prevs[k+1] & prev[k+1]      (genasm_cpu.cpp:248-250)
        //   c   = ((R[k+1][d] << 1) | M[k]) & ins & sub & del          (:247, :251)
        // = two 3-input logic ops per dword (v_bitop3_b32, full rate) + one 64-bit shift.
        // Row 0 (:232-238) falls out by starting from all-ones stand-ins for "row -1".
        // Two register sets (A/B) alternate as "previous row" and "current row" so that no
        // register copies are needed between steps.
        uint64_t pA[CPL], psA[CPL], pB[CPL], psB[CPL];
#pragma unroll
        for (int k = 0; k < CPL; k++) pA[k] = psA[k] = pB[k] = psB[k] = ~0ull;
        uint64_t rnA = ~0ull, rnsA = ~0ull, rnB = ~0ull, rnsB = ~0ull;   // right neighbour column, per set
        uint64_t bnd = V;                          // virtual column 64 at the current row: V << d
        int d = -(G - 1 - t);                      // my row at step 0
        uint32_t dw = 0;                           // window edit distance once found
        uint64_t done_mask = __ballot(!has_pair);  // wave-uniform: lanes of finished slots
        bool all_done = (done_mask & leaders) == leaders;
        const uint32_t col0 = (uint32_t)(t * CPL);
        constexpr int ST = WIDE ? G : (32 + CPL - 1) / CPL;   // lanes 0..ST-1 of a slot own the stored columns (DENT: 0..31, :258-259)
        // Per-lane thresholds keep the per-step control flow to one compare each:
        //   rows d < st_limit of a live slot's storer lanes go to LDS,
        //   a live slot's leader reports a hit when the high dword of column 0 is > hit_thr (= bit 63 clear).
        int32_t st_limit = (has_pair && t < ST) ? RB : INT32_MIN;
        int32_t hit_thr = (has_pair && leader) ? -1 : INT32_MAX;
        uint32_t saddr = lds_slot + col0 * (WIDE ? 2u : 1u);   // LDS word index of my columns in row d (once d >= 0)
        int step = 0;
        uint32_t lastmask = (t == G - 1) ? ~0u : 0u;
        asm volatile("" : "+v"(lastmask));          // keep it a VGPR mask (v_bitop3 select, full rate) rather than v_cndmask
        const uint64_t tm2 = timing ? __builtin_readcyclecounter() : 0;

        // one skewed step: reads the row in (pi, psi, rni, rnsi), writes the next one to (po, pso, rno, rnso)
        auto dc_step = [&](const uint64_t (&pi)[CPL], const uint64_t (&psi)[CPL], const uint64_t rni,
                           const uint64_t rnsi, uint64_t (&po)[CPL], uint64_t (&pso)[CPL], uint64_t& rno,
                           uint64_t& rnso) {
            // right neighbour's first column at my row: it finished that row one step ago
            uint64_t rn = dpp_from_next64(pi[0]);
            const uint64_t bnds = shl1(bnd);
            {   // last lane of the slot: column 64 (always >= n) instead of the neighbour slot's data
                const uint32_t lo = bitop3<0xCA>(lastmask, (uint32_t)bnd, (uint32_t)rn);
                const uint32_t hi = bitop3<0xCA>(lastmask, (uint32_t)(bnd >> 32), (uint32_t)(rn >> 32));
                rn = ((uint64_t)hi << 32) | lo;
            }
            const uint64_t rns = shl1(rn);
            bnd = bnds;

            if (d >= 0) {
                uint32_t rs_lo = (uint32_t)rns, rs_hi = (uint32_t)(rns >> 32);        // (R[k+1][d]) << 1
                uint32_t tr_lo = (uint32_t)rni, tr_hi = (uint32_t)(rni >> 32);        // R[k+1][d-1]
                uint32_t ts_lo = (uint32_t)rnsi, ts_hi = (uint32_t)(rnsi >> 32);      // R[k+1][d-1] << 1
#pragma unroll
                for (int k = CPL - 1; k >= 0; k--) {
                    const uint32_t p_lo = (uint32_t)pi[k], p_hi = (uint32_t)(pi[k] >> 32);
                    const uint32_t q_lo = (uint32_t)psi[k], q_hi = (uint32_t)(psi[k] >> 32);
                    // ins & sub & del, then (match) & that: 2 x v_bitop3_b32 per dword
                    const uint32_t x_lo = __builtin_amdgcn_bitop3_b32(q_lo, ts_lo, tr_lo, 0x80);
                    const uint32_t x_hi = __builtin_amdgcn_bitop3_b32(q_hi, ts_hi, tr_hi, 0x80);
                    const uint32_t c_lo = __builtin_amdgcn_bitop3_b32(rs_lo, (uint32_t)M[k], x_lo, 0xA8);
                    const uint32_t c_hi = __builtin_amdgcn_bitop3_b32(rs_hi, (uint32_t)(M[k] >> 32), x_hi, 0xA8);
                    const uint64_t c = ((uint64_t)c_hi << 32) | c_lo;
                    const uint64_t cs = shl1(c);
                    tr_lo = p_lo; tr_hi = p_hi;
                    ts_lo = q_lo; ts_hi = q_hi;
                    po[k] = c;
                    pso[k] = cs;
                    rs_lo = (uint32_t)cs; rs_hi = (uint32_t)(cs >> 32);
                }
                rno = rn;
                rnso = rns;
                // SENE + DENT store of the row just computed (genasm_cpu.cpp:258-267): plain
                // ds_write of the high dwords (never a flat access)
                if (d < st_limit) {
                    if (WIDE) {
#pragma unroll
                        for (int k = 0; k < CPL; k++) {
                            lds[saddr + 2 * k] = (uint32_t)po[k];
                            lds[saddr + 2 * k + 1] = (uint32_t)(po[k] >> 32);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < CPL; k++) lds[saddr + k] = (uint32_t)(po[k] >> 32);
                    }
                }
                saddr += ROWDW;
            }
            // rows >= RB of a live slot go to the HBM spill area (rare; L1-bypassing agent-scope stores)
            if (step >= RB + (G - ST)) {
                int d_here = d;
                asm volatile("" : "+v"(d_here));      // keeps the per-lane tests inside this rarely taken scalar branch
                if (d_here >= RB && st_limit > 0) {
                    // plain (L2-resident) stores; the traceback reads them back with L1-bypassing loads
                    // after an s_waitcnt vmcnt(0)
                    // uniform base + 32-bit per-lane byte offset: no 64-bit address or packing registers
                    const uint32_t off = spill_slot_b + ((uint32_t)(d_here < SPILL_ROWS ? d_here : SPILL_ROWS - 1) * ROWDW + col0 * (WIDE ? 2u : 1u)) * 4u;
                    char* const sb = reinterpret_cast<char*>(a.spill);
                    if (WIDE) {
#pragma unroll
                        for (int k = 0; k < CPL; k++) *reinterpret_cast<uint64_t*>(sb + off + 8 * k) = po[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < CPL; k++) *reinterpret_cast<uint32_t*>(sb + off + 4 * k) = (uint32_t)(po[k] >> 32);
                    }
                }
            }

            // early termination: column 0 reaches the goal bit (genasm_cpu.cpp:278-283)
            const uint64_t hits = __ballot((int32_t)(po[0] >> 32) > hit_thr);
            if (hits) {
                // expand each hit leader bit to its slot's G lanes
                const uint64_t newly = (G == 64) ? ~0ull
                    : (((uint64_t)((uint32_t)hits * GMASK)) | ((uint64_t)((uint32_t)(hits >> 32) * GMASK) << 32));
                if ((newly >> lane) & 1ull) {
                    st_limit = INT32_MIN;
                    hit_thr = INT32_MAX;
                    dw = (uint32_t)(step - (G - 1));
                }
                done_mask |= newly;
                all_done = (done_mask & leaders) == leaders;
            }
            d++;
            step++;
        };
        if (!SCRG_ABL(a, 2)) {
            while (!all_done) {
                dc_step(pA, psA, rnA, rnsA, pB, psB, rnB, rnsB);
                if (all_done) break;
                dc_step(pB, psB, rnB, rnsB, pA, psA, rnA, rnsA);
            }
        }
        const uint64_t tm3 = timing ? __builtin_readcyclecounter() : 0;
        st_rounds++;
        st_steps += (uint32_t)step;
        // some row of a live slot was spilled if the sweep ran past row RB on a storer lane
        // the traceback reads rows 0..dw-1: it needs the HBM spill rows only if some live slot's window
        // distance exceeds the LDS rows (the sweep itself overshoots RB far more often than that)
        const bool spilled = __any(has_pair && dw > (uint32_t)RB);

        // spilled rows were written by other lanes of this wave: make sure they reached L2
        // before the traceback reads them back (it reads them with L1-bypassing loads)
        if (spilled) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        // ---------------- GenASM-TB, lane-parallel diagonal scan (genasm_cpu.cpp:290-409) ----------------
        // The walk (i,j,d) is replicated in the slot's lanes.  In one macro-step lane l evaluates the
        // reference's per-step decision (:319-370) for the cell l steps down the current diagonal,
        // assuming every earlier step was a match; a DPP min-reduction over the slot finds the first
        // lane whose cell is not a plain match, so a whole '=' run plus the edit that ends it retire
        // per macro-step.  The common variant touches only LDS (no vmcnt waits in the loop); the
        // variant that may read spilled rows from HBM is a separate instantiation.
        auto traceback = [&](auto spill_tag) {
            constexpr bool SPILL = decltype(spill_tag)::value;
            uint32_t i = 0, j = 0, dd = dw;
            uint32_t cur_op = 0, cur_cnt = 0;
            bool act = has_pair;
            const uint32_t jlim = min(m, (uint32_t)TBL);       // j < m && j < W-O  (:307-310)

            auto emit = [&](bool en, uint32_t op, uint32_t cnt) {
                const bool same = (op == cur_op);
                if (en && !same && cur_cnt != 0) {             // run ended: stage {count, op}
                    if (n_runs < cigar_cap) {
                        if (leader) lds16[2u * obuf + (n_runs & 31u)] = (uint16_t)(cur_cnt | (cur_op << 8));
                        if ((n_runs & 15u) == 15u && !SCRG_ABL(a, 4)) {   // 16 runs complete: one 32-byte store per slot
                            const uint32_t piece = ((n_runs >> 4) & 1u) * 8u;
                            uint32_t* const dst = reinterpret_cast<uint32_t*>(a.runs + cigar_off + (n_runs - 15u));
                            for (uint32_t k = (uint32_t)t; k < 8u; k += (uint32_t)G) dst[k] = lds[obuf + piece + k];
                        }
                    } else {
                        overflow = true;
                    }
                    n_runs++;
                }
                cur_cnt = en ? (same ? cur_cnt + cnt : cnt) : cur_cnt;
                cur_op = en ? op : cur_op;
            };

            while (__any(act)) {
                st_macro++;
                const uint32_t il = i + t, jl = j + t;
                const bool pos_ok = (jl < jlim) && (il < (uint32_t)TBL);
                const bool room = dd > 0;                       // d_limit, :313
                const uint32_t r = room ? dd - 1 : 0u;
                // entries as 64-bit values with pattern char j at bit 63-j (a DENT dword is the high half)
                constexpr uint32_t ICMAX = WIDE ? 62u : 30u;
                const uint32_t ic = il < ICMAX ? il : ICMAX;
                uint64_t e0, e1;                                // R[i][d-1], R[i+1][d-1]
                if (SPILL && r >= (uint32_t)RB) {
                    uint32_t* rp = Rs + (size_t)r * ROWDW + ic * (WIDE ? 2u : 1u);
                    if (WIDE) {
                        const uint32_t a0 = __hip_atomic_load(rp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint32_t a1 = __hip_atomic_load(rp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint32_t a2 = __hip_atomic_load(rp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint32_t a3 = __hip_atomic_load(rp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        e0 = ((uint64_t)a1 << 32) | a0;
                        e1 = ((uint64_t)a3 << 32) | a2;
                    } else {
                        e0 = (uint64_t)__hip_atomic_load(rp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32;
                        e1 = (uint64_t)__hip_atomic_load(rp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32;
                    }
                } else {
                    const uint32_t base = lds_slot + (r < (uint32_t)RB ? r : 0u) * ROWDW + ic * (WIDE ? 2u : 1u);
                    if (WIDE) {
                        e0 = ((uint64_t)lds[base + 1] << 32) | lds[base];
                        e1 = ((uint64_t)lds[base + 3] << 32) | lds[base + 2];
                    } else {
                        e0 = (uint64_t)lds[base] << 32;
                        e1 = (uint64_t)lds[base + 1] << 32;
                    }
                }
                // TB_BIT(j+1) = bit 62-j, TB_BIT(j) = bit 63-j (genasm_cpu.cpp:57-60 in the left-aligned layout)
                const uint32_t sh = 62u - (jl < 62u ? jl : 62u);
                const bool last = (jl + 1u == m);                // last pattern character, :336-343
                const bool tl = il < n;                          // !i_limit, :312
                const bool ins = room && (last || ((e0 >> sh) & 1ull) == 0ull);
                const bool del = room && tl && !last && ((e1 >> (sh + 1u)) & 1ull) == 0ull;
                const bool sub = room && tl && (last || ((e1 >> sh) & 1ull) == 0ull);
                uint32_t ev = ins ? 1u : (del ? 2u : (sub ? 3u : 0u));     // priority I, D, X, = (:346-370)
                ev = pos_ok ? ev : 4u;                           // 4 = the window's walk ends here
                if (SCRG_ABL(a, 1)) ev = pos_ok ? 0u : 4u;
                const uint32_t key = (act && ev) ? (((uint32_t)t << 3) | ev) : 0xffffu;
                const uint32_t kmin = slot_min<G>(key);
                const bool none = kmin == 0xffffu;
                const uint32_t n_eq = none ? (uint32_t)G : (kmin >> 3);
                const uint32_t evf = none ? 0u : (kmin & 7u);

                emit(act && n_eq != 0u, '=', n_eq);
                emit(act && evf >= 1u && evf <= 3u, evf == 1u ? 'I' : (evf == 2u ? 'D' : 'X'), 1u);
                if (act) {
                    i += n_eq + ((evf == 2u || evf == 3u) ? 1u : 0u);
                    j += n_eq + ((evf == 1u || evf == 3u) ? 1u : 0u);
                    dd -= (evf >= 1u && evf <= 3u) ? 1u : 0u;
                    act = evf != 4u;
                }
            }
            emit(has_pair, 0u, 0u);                              // per-window flush (:400-403)
            if (has_pair) {
                edits += dw - dd;
                ref_idx += i;
                read_idx += j;
            }
        };
        // ---- fast traceback: the same macro-step, branch-free, in integer/mask arithmetic ----------
        // Valid when no row was spilled and, in every live slot, the window keeps more than W-O
        // pattern characters and at least W-O text characters (so neither the last-character rule
        // :336-343 nor the text limit :312 can trigger inside the W-O cells a window may consume) —
        // i.e. everywhere except the last window or two of a pair.  A wave is bound by the length of
        // its own instruction stream here, so the step has two branches (loop, rare 32-byte flush)
        // instead of one per decision, and conditions are 0/~0 masks: on gfx950 compares, selects,
        // min/max and left shifts issue at half the rate of and/or/xor/add/sub/right-shift/bitop3.
        auto traceback_fast = [&]() {
            constexpr uint32_t LOG2G = (G == 4) ? 2 : (G == 8) ? 3 : (G == 16) ? 4 : (G == 32) ? 5 : 6;
            const uint32_t K0 = ~0x40000000u, K1 = ~0x30000000u;
            uint32_t actmask = has_pair ? ~0u : 0u;
            uint32_t dd = dw;
            uint32_t irem = (uint32_t)TBL, jrem = (uint32_t)TBL;       // cells left before i or j reaches W-O
            uint32_t it4 = (uint32_t)t * 4u;                            // (i + t) * 4
            uint32_t rj = 0u - (uint32_t)t;                             // -(j + t): rotate amount, mod 32
            uint32_t rowaddr = (lds_slot + (dd - 1u) * 32u) * 4u;       // byte address of row d-1 (garbage if d == 0)
            uint32_t cur_op8 = 0, cur_cnt = 0, ovf = 0;                 // run in progress: op << 8, length
            uint32_t nr2 = 2u * n_runs;                                 // 2 * n_runs: byte position of the next staged run
            const uint32_t l8 = (uint32_t)t << 3;
            const uint32_t leadmask = leader ? ~0u : 0u;
            const uint32_t obuf_b = 4u * obuf, dummy_b = 4u * scratch_dw;
            const uint32_t cap2 = 2u * cigar_cap;
            char* const lds_b = reinterpret_cast<char*>(lds);

            // stage the run in progress if `op8` differs from it; only the slot leader's write lands in
            // the ring, every other lane (and every masked-off case) writes its slot's scratch cell
            auto emit = [&](uint32_t enmask, uint32_t op8, uint32_t cnt) {
                const uint32_t same = ~nz_mask(op8 ^ cur_op8);
                const uint32_t flush = bitop3<0x40>(enmask, nz_mask(cur_cnt), same);   // en & havecur & ~same
                const uint32_t room_ok = neg_mask(nr2 - cap2);                         // ~0 iff n_runs < cap
                const uint32_t wr = flush & room_ok & leadmask;
                const uint32_t addr = bitop3<0xCA>(wr, obuf_b + (nr2 & 62u), dummy_b);
                *reinterpret_cast<uint16_t*>(lds_b + addr) = (uint16_t)(cur_cnt | cur_op8);
                ovf |= flush & ~room_ok;
                nr2 += flush & 2u;
                cur_cnt = bitop3<0xD0>(cur_cnt, same, enmask) + (cnt & enmask);        // cur_cnt & (same | ~en)
                cur_op8 = bitop3<0xCA>(enmask, op8, cur_op8);
            };
            // write out a 16-run piece of the ring once the staged count has moved past it
            auto flush_pieces = [&](uint32_t before2) {
                const uint32_t crossed = (before2 ^ nr2) & 32u;                        // bit 5 of 2*n flips every 16 runs
                if (__any(crossed != 0u)) {
                    if (crossed && !SCRG_ABL(a, 4)) {
                        const uint32_t first = (nr2 >> 5) * 16u - 16u;                 // first run of the completed piece
                        if (first + 16u <= cigar_cap) {
                            const uint32_t piece = ((first >> 4) & 1u) * 8u;
                            uint32_t* const dst = reinterpret_cast<uint32_t*>(a.runs + cigar_off + first);
                            for (uint32_t k = (uint32_t)t; k < 8u; k += (uint32_t)G) dst[k] = lds[obuf + piece + k];
                        }
                    }
                }
            };

            const uint64_t tl0 = timing ? __builtin_readcyclecounter() : 0;
            while (__any(actmask != 0u)) {
                st_macro++;
                const uint32_t before2 = nr2;
                // R[i+l][d-1], R[i+l+1][d-1]
                const uint32_t* const rp = reinterpret_cast<const uint32_t*>(lds_b + rowaddr + it4);
                const uint32_t w0 = rp[0], w1 = rp[1];
                // TB_BIT(j+1) = bit 30-j of the stored dword; rotate it to bit 30 (ins, from w0) and
                // bits 28 (sub) / 29 (del) (from w1); a clear bit means the edit is available
                const uint32_t r0 = __builtin_amdgcn_alignbit(w0, w0, rj);
                const uint32_t r1 = __builtin_amdgcn_alignbit(w1, w1, rj + 2u);
                const uint32_t avail = bitop3<0x1F>(r0 | K0, r1, K1);             // ~(a & (b | c)): bit30 ins, bit29 del, bit28 sub
                const uint32_t code = (uint32_t)__builtin_clz(avail | 0x08000000u);   // 1 I, 2 D, 3 X, 4 none (priority :346-370)
                // no edit possible without budget (d == 0) or in a finished slot
                const uint32_t blocked = neg_mask(dd - 1u) | ~actmask;
                const uint32_t key = l8 | code | (0u - (code >> 2)) | blocked;   // all ones when nothing ends the run here
                uint32_t kmin = slot_min<G>(key);
                // the walk stops at the first cell past the W-O limits (:307-310); with more than G cells
                // left the macro-step simply ends after G matches (code 4)
                const uint32_t lim = irem < jrem ? irem : jrem;
                const uint32_t limc = lim < (uint32_t)G ? lim : (uint32_t)G;
                const uint32_t stopkey = ((limc << 3) | ((limc >> (LOG2G - 2u)) & 4u)) & actmask;
                kmin = kmin < stopkey ? kmin : stopkey;
                const uint32_t n_eq = kmin >> 3;
                const uint32_t c = kmin & 7u;                                 // 0 stop, 1 I, 2 D, 3 X, 4 continue
                const uint32_t is_edit = ((c & 3u) + 3u) >> 2;                // 1 for I/D/X
                const uint32_t inc_i = (c >> 1) & 1u, inc_j = c & 1u;

                if (!SCRG_ABL(a, 16)) {
                emit(nz_mask(n_eq), (uint32_t)'=' << 8, n_eq);
                // 'I' 0x49, 'D' 0x44, 'X' 0x58 in a table word indexed by the code
                emit(0u - is_edit, ((0x58444900u >> ((c & 3u) * 8u)) & 0xffu) << 8, 1u);
                flush_pieces(before2);
                }

                const uint32_t ai = n_eq + inc_i, aj = n_eq + inc_j;
                irem -= ai;
                jrem -= aj;
                it4 += ((kmin >> 1) & ~3u) + (c & 2u) + (c & 2u);             // 4 * ai
                rj -= aj;
                dd -= is_edit;
                rowaddr -= (0u - is_edit) & 128u;
                actmask &= 0u - ((c + 7u) >> 3);                              // code 0: this slot's walk is over
            }
            if (timing) cy_tbloop += __builtin_readcyclecounter() - tl0;
            {
                const uint32_t before2 = nr2;
                emit(has_pair ? ~0u : 0u, 0u, 0u);                            // per-window flush (:400-403)
                flush_pieces(before2);
            }
            overflow = overflow || (ovf != 0u);
            n_runs = nr2 >> 1;
            if (has_pair) {
                edits += dw - dd;
                ref_idx += (uint32_t)TBL - irem;
                read_idx += (uint32_t)TBL - jrem;
            }
        };
        const bool fast_ok = !WIDE && !spilled && !__any(has_pair && (m <= (uint32_t)TBL || n < (uint32_t)TBL)) && !SCRG_ABL(a, 8);
        if (fast_ok) traceback_fast();
        else if (spilled) traceback(std::true_type{});
        else traceback(std::false_type{});
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
        atomicAdd((unsigned long long*)&a.stats[1], (unsigned long long)st_steps);
        atomicAdd((unsigned long long*)&a.stats[2], (unsigned long long)st_macro);
        atomicAdd((unsigned long long*)&a.stats[3], (unsigned long long)cy_fetch);
        atomicAdd((unsigned long long*)&a.stats[4], (unsigned long long)cy_setup);
        atomicAdd((unsigned long long*)&a.stats[5], (unsigned long long)cy_dc);
        atomicAdd((unsigned long long*)&a.stats[6], (unsigned long long)cy_tb);
        atomicAdd((unsigned long long*)&a.stats[7], (unsigned long long)cy_tbloop);
        atomicAdd((unsigned long long*)&a.stats[8], (unsigned long long)st_diag);
        atomicAdd((unsigned long long*)&a.stats[9], (unsigned long long)st_diag_fail);
        atomicAdd((unsigned long long*)&a.stats[10], (unsigned long long)cy_ddc);
        atomicAdd((unsigned long long*)&a.stats[11], (unsigned long long)cy_dtb);
    }
}


// ----------------------------------------------------------------------------
// host-side launchers
// ----------------------------------------------------------------------------
template <int G, bool WIDE>
static hipError_t launch_align_tw(const AlignArgs& a, int grid, size_t lds_bytes, hipStream_t s)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&genasm_align_kernel<G, WIDE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((genasm_align_kernel<G, WIDE>), dim3(grid), dim3(64), lds_bytes, s, a);
    return hipGetLastError();
}
template <int G>
static hipError_t launch_align_t(const AlignArgs& a, int grid, size_t lds_bytes, hipStream_t s)
{
    // the traceback consumes up to W-O characters per window: more than 31 needs whole entries of all columns
    return a.tb_limit > 31 ? launch_align_tw<G, true>(a, grid, lds_bytes, s) : launch_align_tw<G, false>(a, grid, lds_bytes, s);
}

hipError_t launch_align(int lanes_per_pair, const AlignArgs& a, int grid, size_t lds_bytes, hipStream_t s)
{
    switch (lanes_per_pair) {
    case 4:  return launch_align_t<4>(a, grid, lds_bytes, s);
    case 8:  return launch_align_t<8>(a, grid, lds_bytes, s);
    case 16: return launch_align_t<16>(a, grid, lds_bytes, s);
    case 32: return launch_align_t<32>(a, grid, lds_bytes, s);
    case 64: return launch_align_t<64>(a, grid, lds_bytes, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace scrg
