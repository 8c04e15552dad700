// genasm_lane_mw_kernel.hip — the lane-per-pair aligner for W > 64 (64 < W <= 256): the formulation of
// genasm_lane_kernel.hip — every lane aligns its own pair, the window's table holds the DIFFERENCES of the
// edit-distance matrix behind the GenASM bitvectors (see the header of that file for why this gives the reference's
// edit distance and CIGAR bit for bit, src/genasm_cpu.cpp:210-409; multi-word entries: src/bitvector.hpp:45-48) — with
// multi-word vectors: a pattern vector has NW = ceil(W/64) words (word 0 the most significant: bit 63-k of word w
// belongs to pattern character 64 w + k), and as the traceback only looks at rows j <= W-O, a table row keeps the
// top RW = (W-O)/64 + 1 words.  One text column — all W pattern rows, every distance at once — is ~10 NW 64-bit
// operations, there is no loop over the distance, and a window's cost does not depend on its distance; the GenASM-row
// kernel this replaces as the default for W > 64 (genasm_kernel_multiword.hip) sweeps W+1 distance rows.
//
// The table (two RW-word rows for each of the W-O columns: 1 to 16 KB per lane) lives in HBM, one slab per wavefront,
// word-interleaved over the lanes so that a store or load of the wavefront is 512 contiguous bytes; the first
// traceback pass reads it back four to sixteen columns at a time.  LDS holds the CIGAR staging ring and the insertion-run
// lengths only.  Written with plain 64-bit operations (this is the knob-sweep configuration, scripts/profile.py:180-185,
// not the tuned one); tests/proto/lane_proto.c (lane_align_codes_mw) restates the arithmetic and is checked against
// the reference algorithm on the CPU (tests/test_lane_proto.py).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "genasm_kernels.h"
#include "genasm_device.h"
#include "row_ops.h"

namespace scrg {

namespace {

constexpr uint32_t MW_RING_BYTES = 68;             // 32 runs + one dword per lane (bank skew)

}  // namespace

// EDITS = true (scrg_align_device_edits): the alignment leaves as an edit stream (edit_stream.h: one byte per edit
// carrying the number of matches before it) instead of runs; see genasm_lane_kernel<true>.
template <int NW, int RW, bool EDITS>
__global__ __launch_bounds__(64) void genasm_lane_mw_kernel(AlignArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    char* const lds_b = reinterpret_cast<char*>(lds);
    uint8_t* const lds8 = reinterpret_cast<uint8_t*>(lds);

    const uint32_t lane = threadIdx.x;
    const uint32_t W = (uint32_t)a.W;
    const uint32_t TBL = (uint32_t)a.tb_limit;         // W - O: 64 (RW - 1) <= TBL < 64 RW
    const uint32_t ring_b = lane * MW_RING_BYTES;
    const uint32_t len_b = 64u * MW_RING_BYTES + lane * lane_mw_len_bytes(a.tb_limit);
    // my wavefront's slab of the table: word ((column * 2 + which) * RW + r) * 64 + lane
    uint64_t* const tab = reinterpret_cast<uint64_t*>(a.spill) + (uint64_t)blockIdx.x * (uint64_t)TBL * 2u * RW * 64u + lane;

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
    uint32_t mbase = 0;                // EDITS: matches pending at column c of the current window = mbase + c
    bool queue_empty = false;          // wave-uniform

    // (EDITS: the slice holds bytes — it starts at byte 2 * cigar_off and is 2 * cigar_cap bytes long — and a piece is
    // 32 bytes of the stream)
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
    auto flush_pieces = [&]() {
        for (;;) {
            const bool need = has_pair && nr + 1 - (int32_t)flushed >= 16;
            if (!__any(need)) break;
            if (need) write_piece();
        }
    };
    auto push_run = [&](uint32_t op, uint32_t count) {
        nr++;
        *reinterpret_cast<uint16_t*>(lds_b + ring_b + (((uint32_t)nr & 31u) << 1)) = (uint16_t)(count | (op << 8));
    };
    auto emit = [&](uint32_t b) {          // EDITS: one byte of the stream; whole pieces leave at once
        lds8[ring_b + (pos & 63u)] = (uint8_t)b;
        pos++;
        if (pos - flushed >= 32u) write_piece();
    };

    for (;;) {
        // ---------------- retire finished pairs, fetch new ones (genasm_cpu.cpp:440-460) ----------------
        for (;;) {
            const bool fin = has_pair && read_idx >= read_len;
            if (__any(fin)) {
                if (EDITS && fin) {
                    // (the matches after the last edit are implied by the read length; fewer than 32 bytes are staged)
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
        uint64_t tlo[NW], thi[NW], rlo[NW], rhi[NW], valid[NW];
#pragma unroll
        for (int w = 0; w < NW; w++) {
            Planes t = {0, 0}, p = {0, 0};
            if (has_pair && 64u * (uint32_t)w < n) t = load_window_strided(a.seq, text_off, ref_idx + 64u * (uint32_t)w, a.text_stride);
            if (has_pair && 64u * (uint32_t)w < m) p = load_window_strided(a.seq, read_off, read_idx + 64u * (uint32_t)w, a.read_stride);
            tlo[w] = t.lo;
            thi[w] = t.hi;
            rlo[w] = brev64(p.lo);                       // reversed: bit 63-k <-> pattern character 64 w + k
            rhi[w] = brev64(p.hi);
            if (a.stranded && __any(has_pair && rev)) {  // (uniform)
                const Planes rv = revcomp_pattern_word(a.seq, read_off, read_len, has_pair ? read_idx : read_len, (uint32_t)w, a.read_stride);
                if (has_pair && rev) { rlo[w] = rv.lo; rhi[w] = rv.hi; }
            }
            const uint32_t lo = 64u * (uint32_t)w;
            valid[w] = m >= lo + 64u ? ~0ull : (m <= lo ? 0ull : ~0ull << (64u - (m - lo)));
        }
        const uint32_t jlim = has_pair ? min(m, TBL) : 0u;          // the walk ends when j gets here (:301, :310)
        const Row<RW> stop = row_bit<RW>(jlim);

        // ---------------- the window's table (genasm_cpu.cpp:210-288 in difference form) ----------------
        // The bits below the pattern are neutral (Eq = 1, Pv = Mv = 0); columns past the end of the text take the Eq
        // word "no character matches", which leaves the boundary column D[n][j] = m-j as it is and yields "insertion
        // in every row".  Column i < W-O keeps ~(V1 | stop) and V0 | stop (V1 = Pv' | Ph, V0 = Pv' | ~(Ph | Xh)).
        {
            uint64_t pv[NW], mv[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) { pv[w] = valid[w]; mv[w] = 0; }
#pragma unroll
            for (int wi = NW - 1; wi >= 0; wi--) {
                for (int b = 63; b >= 0; b--) {
                    const uint32_t i = 64u * (uint32_t)wi + (uint32_t)b;
                    if (i >= W) continue;                             // (uniform)
                    const uint64_t sl = 0ull - ((tlo[wi] >> b) & 1ull), sh = 0ull - ((thi[wi] >> b) & 1ull);
                    const bool in_text = i < n;
                    uint64_t xv[NW], xh[NW], ph[NW], mh[NW];
                    uint64_t carry = 0;
#pragma unroll
                    for (int w = NW - 1; w >= 0; w--) {              // the add: carries run from the last word to word 0
                        const uint64_t eq = (in_text ? ~((rlo[w] ^ sl) | (rhi[w] ^ sh)) : 0ull) | ~valid[w];
                        xv[w] = eq | mv[w];
                        const uint64_t t = eq & pv[w];
                        const uint64_t s1 = t + pv[w];
                        const uint64_t s2 = s1 + carry;
                        carry = (uint64_t)(s1 < t) | (uint64_t)(s2 < s1);
                        xh[w] = (s2 ^ pv[w]) | eq;
                        ph[w] = mv[w] | ~(xh[w] | pv[w]);
                        mh[w] = pv[w] & xh[w];
                    }
#pragma unroll
                    for (int w = 0; w < NW; w++) {
                        const uint64_t ph1 = (ph[w] << 1) | (w + 1 < NW ? ph[w + 1] >> 63 : 0ull);    // row 0 of the matrix is all
                        const uint64_t mh1 = (mh[w] << 1) | (w + 1 < NW ? mh[w + 1] >> 63 : 0ull);    // zeros: 0 comes in
                        pv[w] = mh1 | ~(xv[w] | ph1);
                        mv[w] = ph1 & xv[w];
                    }
                    if (i < TBL) {
#pragma unroll
                        for (int r = 0; r < RW; r++) {
                            // (O = 0 at W = 64, 128, 256: the stop row W is the first bit of a word the vectors do not have —
                            // RW = NW + 1 — which holds nothing but that bit)
                            const uint64_t pvr = r < NW ? pv[r < NW ? r : 0] : 0ull, phr = r < NW ? ph[r < NW ? r : 0] : 0ull, xhr = r < NW ? xh[r < NW ? r : 0] : ~0ull;
                            tab[((uint64_t)(i * 2u) * RW + r) * 64u] = ~((pvr | phr) | stop.w[r]);
                            tab[((uint64_t)(i * 2u + 1u) * RW + r) * 64u] = (pvr | ~(phr | xhr)) | stop.w[r];
                        }
                    }
                }
            }
        }

        // ---------------- traceback (genasm_cpu.cpp:290-409): the two passes of genasm_lane_kernel on RW-word rows ----------------
        {
            uint32_t j = 0;
            Row<RW> nDm = row_zero<RW>(), Xm = row_zero<RW>(), nIm = row_zero<RW>();
            constexpr int CH = RW == 1 ? 16 : (RW == 2 ? 8 : 4);    // columns in flight: the loads do not depend on the walk
            for (uint32_t i0 = 0; i0 < TBL; i0 += CH) {
                Row<RW> nv1[CH], v0[CH];
#pragma unroll
                for (int q = 0; q < CH; q++) {
                    const uint32_t i = min(i0 + (uint32_t)q, TBL - 1u);
#pragma unroll
                    for (int r = 0; r < RW; r++) {
                        nv1[q].w[r] = tab[((uint64_t)(i * 2u) * RW + r) * 64u];
                        v0[q].w[r] = tab[((uint64_t)(i * 2u + 1u) * RW + r) * 64u];
                    }
                }
#pragma unroll
                for (int q = 0; q < CH; q++) {
                    const uint32_t i = i0 + (uint32_t)q;
                    if (i >= TBL) continue;                          // (uniform)
                    Row<RW> x;
#pragma unroll
                    for (int r = 0; r < RW; r++) x.w[r] = nv1[q].w[r] | ~v0[q].w[r] | stop.w[r];   // not (insertion), or the stop row
                    x = row_shl<RW>(x, j);
                    const uint32_t ni = row_clz<RW>(x);             // (the stop bit makes x non-zero)
                    lds8[len_b + i] = (uint8_t)ni;
                    nIm = row_shl1_in<RW>(nIm, x.w[0] >> 63);
                    j += ni;
                    const Row<RW> nt1 = row_shl<RW>(nv1[q], j), t0 = row_shl<RW>(v0[q], j);    // top bits: not a deletion, substitution
                    nDm = row_shl1_in<RW>(nDm, nt1.w[0] >> 63);
                    Xm = row_shl1_in<RW>(Xm, t0.w[0] >> 63);
                    j += (uint32_t)(nt1.w[0] >> 63);                 // a deletion (or the stop row) keeps j
                }
            }
            // column i -> bit i from the top; a finished lane reads "deletion and substitution" (the stop row)
            const uint32_t nsh = 64u * RW - TBL;
            Row<RW> Draw = row_shl<RW>(nDm, nsh), Im;
            const Row<RW> Xraw = row_shl<RW>(Xm, nsh);
#pragma unroll
            for (int r = 0; r < RW; r++) { Draw.w[r] = ~Draw.w[r]; Im.w[r] = ~nIm.w[r]; }
            Im = row_shl<RW>(Im, nsh);
            const Row<RW> lim = row_bit<RW>(TBL);
            Row<RW> dead;
#pragma unroll
            for (int r = 0; r < RW; r++) dead.w[r] = (Draw.w[r] & Xraw.w[r]) | lim.w[r];
            const uint32_t ti = row_clz<RW>(dead);
            const Row<RW> A = row_top<RW>(ti);
            Row<RW> D, X, B, E;
#pragma unroll
            for (int r = 0; r < RW; r++) { D.w[r] = Draw.w[r] & A.w[r]; X.w[r] = Xraw.w[r] & A.w[r]; }
            const Row<RW> D1 = row_shr1<RW>(D), X1 = row_shr1<RW>(X);
#pragma unroll
            for (int r = 0; r < RW; r++) {
                B.w[r] = ((D.w[r] ^ D1.w[r]) | (X.w[r] ^ X1.w[r]) | Im.w[r] | (r == 0 ? TOP : 0ull)) & A.w[r];
                E.w[r] = B.w[r] | Im.w[r];
            }
            edits += j - ti + 2u * row_pop<RW>(D) + row_pop<RW>(X);
            ref_idx += ti;
            read_idx += j;

            if constexpr (EDITS) {
                // the columns that hold an edit: an insertion run (before the column's step), then a deletion or a
                // substitution; mbase + c = matches pending when column c is reached
                Row<RW> Ev;
#pragma unroll
                for (int r = 0; r < RW; r++) Ev.w[r] = D.w[r] | X.w[r] | Im.w[r];
                nr += (int32_t)(row_pop<RW>(B) + row_pop<RW>(Im));       // the runs this window has in the other output format
                while (__any(row_any<RW>(Ev))) {
                    if (row_any<RW>(Ev)) {
                        const uint32_t c = row_clz<RW>(Ev);
                        const Row<RW> bit = row_bit<RW>(c);
#pragma unroll
                        for (int r = 0; r < RW; r++) Ev.w[r] &= ~bit.w[r];
                        uint32_t t = mbase + c;
                        if (row_test<RW>(Im, c)) {
                            const uint32_t ni = lds8[len_b + c];
                            for (; t >= 63u; t -= 63u) emit(0x3Fu);          // (edit_stream.h: 63 matches and nothing else)
                            emit(0x80u | t);
                            for (uint32_t q = 1; q < ni; q++) emit(0x80u);
                            t = 0;
                            mbase = 0u - c;
                        }
                        const bool isD = row_test<RW>(D, c), isX = row_test<RW>(X, c);
                        if (isD || isX) {
                            for (; t >= 63u; t -= 63u) emit(0x3Fu);
                            emit((isX ? 0x40u : 0xC0u) | t);
                            mbase = ~c;
                        }
                    }
                }
                // the window ends: the matches since its last edit, then the mark (every window of a pair, the last one too)
                if (__any(has_pair)) {
                    if (has_pair) {
                        uint32_t t = mbase + ti;
                        for (; t >= 63u; t -= 63u) emit(0x3Fu);
                        emit(t);
                    }
                }
                mbase = 0;
            } else
            while (__any(row_any<RW>(E))) {
                if (row_any<RW>(E)) {
                    const uint32_t c = row_clz<RW>(E);
                    if (row_test<RW>(Im, c)) push_run((uint32_t)'I', lds8[len_b + c]);
                    const Row<RW> bit = row_bit<RW>(c);
#pragma unroll
                    for (int r = 0; r < RW; r++) E.w[r] &= ~bit.w[r];
                    const uint32_t nx = min(row_clz<RW>(E), ti);     // the next event or the end of the walk
                    if (row_test<RW>(B, c))
                        push_run(row_test<RW>(D, c) ? (uint32_t)'D' : (row_test<RW>(X, c) ? (uint32_t)'X' : (uint32_t)'='), nx - c);
                }
                flush_pieces();                                         // at most two new runs per iteration
            }
        }
    }
}

template <int NW, int RW> static hipError_t launch_mw(const AlignArgs& a, int grid, size_t lds_bytes, hipStream_t s, bool edits)
{
    if (edits) hipLaunchKernelGGL((genasm_lane_mw_kernel<NW, RW, true>), dim3(grid), dim3(64), lds_bytes, s, a);
    else hipLaunchKernelGGL((genasm_lane_mw_kernel<NW, RW, false>), dim3(grid), dim3(64), lds_bytes, s, a);
    return hipGetLastError();
}

hipError_t launch_align_lane_mw(const AlignArgs& a, int grid, size_t lds_bytes, hipStream_t s, bool edits)
{
    const int nw = a.W <= 64 ? 1 : (a.W <= 128 ? 2 : 4), rw = a.tb_limit / 64 + 1;
    if (nw == 1 && rw == 1) return launch_mw<1, 1>(a, grid, lds_bytes, s, edits);
    if (nw == 1 && rw == 2) return launch_mw<1, 2>(a, grid, lds_bytes, s, edits);       // (W = 64, O = 0: the stop bit is row 64)
    if (nw == 2 && rw == 3) return launch_mw<2, 3>(a, grid, lds_bytes, s, edits);       // (W = 128, O = 0)
    if (nw == 4 && rw == 5) return launch_mw<4, 5>(a, grid, lds_bytes, s, edits);       // (W = 256, O = 0)
    if (nw == 2 && rw == 1) return launch_mw<2, 1>(a, grid, lds_bytes, s, edits);
    if (nw == 2 && rw == 2) return launch_mw<2, 2>(a, grid, lds_bytes, s, edits);
    if (nw == 4 && rw == 1) return launch_mw<4, 1>(a, grid, lds_bytes, s, edits);
    if (nw == 4 && rw == 2) return launch_mw<4, 2>(a, grid, lds_bytes, s, edits);
    if (nw == 4 && rw == 3) return launch_mw<4, 3>(a, grid, lds_bytes, s, edits);
    if (nw == 4 && rw == 4) return launch_mw<4, 4>(a, grid, lds_bytes, s, edits);
    return hipErrorInvalidValue;
}

}  // namespace scrg
