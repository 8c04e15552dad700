// genasm_kernel_multiword.hip — the aligner for 64 < W <= 256: NW = 2 or 4 64-bit words per bitvector
// (the reference builds these from arrays of 32-bit elements, src/bitvector.hpp:45-48, 124-139, and its
// sweeps go up to W = 256 with O = W/2+1, scripts/profile.py:69-82, 180-185).
//
// Same algorithm and the same wavefront mapping as genasm_kernels.hip (see the comments there): a
// slot of G lanes owns one pair, lane t the text columns [t*CPL, (t+1)*CPL) with CPL = 64*NW/G, rows
// are swept skewed by one step per lane, the traceback is the lane-parallel diagonal scan.  Differences:
//   * an entry is NW words, left-aligned: pattern character j at bit 63-(j%64) of word j/64;
//   * the traceback consumes at most TBL = W-O characters per window, so it only reads characters
//     0..TBL of columns 0..TBL: with SW = TBL/64+1 that is words 0..SW-1 of columns 0..64*SW-1, the
//     stored "DENT" part (genasm_cpu.cpp:200-208, 258-267) — 512 B per row for the reference's
//     O = W/2+1 up to W = 128, 2 KB up to W = 256;
//   * K = W <= 256 rows can exist: rows >= lds_rows go to a (W+1)-row HBM spill area per slot.
// Throughput is secondary here (one or two pairs per wavefront); parity is not: results are
// bit-identical to the reference built with -DCLI_W=128, 192, 256 ... (tests/golden/pairs_w*_o*.json).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "genasm_kernels.h"
#include "genasm_device.h"
#include "row_ops.h"

namespace scrg {

template <int NW> __device__ __forceinline__ BV<NW> bv_dpp_from_next(const BV<NW>& v)
{
    BV<NW> r;
#pragma unroll
    for (int i = 0; i < NW; i++) r.w[i] = dpp_from_next64(v.w[i]);
    return r;
}

template <int G, int NW>
__global__ __launch_bounds__(64, 1) void genasm_align_multiword_kernel(AlignArgs a)
{
    constexpr int CPL = 64 * NW / G;     // text columns per lane (2, 4 or 8)
    constexpr int SLOTS = 64 / G;        // pairs per wavefront
    constexpr uint32_t OBUF_DWORDS = 16;
    constexpr uint32_t GMASK = 0xffffffffu;
    typedef BV<NW> bv;

    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];

    const int lane = threadIdx.x;
    const int t = lane % G;
    const int gbase = lane - t;
    const int slot = lane / G;
    const bool leader = (t == 0);

    const int W = a.W;
    const int TBL = a.tb_limit;                      // W - O
    const int RB = a.lds_rows;
    const uint32_t SW = (uint32_t)TBL / 64u + 1u;    // stored words per entry (<= NW since TBL < W)
    const uint32_t SCOLS = 64u * SW;                 // stored columns
    const uint32_t ROWDW = SCOLS * SW * 2u;          // dwords per stored row
    const int ST = (int)(SCOLS / CPL);               // lanes 0..ST-1 of a slot own the stored columns
    const uint32_t spill_rows = (uint32_t)W + 1u;
    const uint32_t obuf = (uint32_t)slot * OBUF_DWORDS;
    const uint32_t slot_stride = (uint32_t)RB * ROWDW + 1u;
    const uint32_t lds_slot = SLOTS * (OBUF_DWORDS + 1u) + slot * slot_stride;
    uint16_t* const lds16 = reinterpret_cast<uint16_t*>(lds);
    uint32_t* const Rs = a.spill + ((size_t)blockIdx.x * SLOTS + slot) * ((size_t)spill_rows * ROWDW);
    constexpr uint64_t leaders = leader_mask(G);

    bool has_pair = false;
    uint32_t pair = 0;
    uint64_t text_off = 0, read_off = 0, cigar_off = 0;
    uint32_t text_len = 0, read_len = 0, cigar_cap = 0;
    uint32_t ref_idx = 0, read_idx = 0, n_runs = 0, edits = 0;
    bool overflow = false;
    bool queue_empty = false;

    for (;;) {
        // ---------------- retire finished pairs, fetch new ones ----------------
        for (;;) {
            const bool fin = has_pair && read_idx >= read_len;
            if (fin) {
                const uint32_t done = n_runs < cigar_cap ? n_runs : cigar_cap;
                const uint32_t rem = done & 15u;
                const uint32_t piece = ((done >> 4) & 1u) * 8u;
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

        // ---------------- window setup (genasm_cpu.cpp:417-420) ----------------
        const uint32_t n = (has_pair && ref_idx < text_len) ? min((uint32_t)W, text_len - ref_idx) : 0u;
        const uint32_t m = has_pair ? min((uint32_t)W, read_len - read_idx) : 1u;
        bv V;           // valid bits: characters 0..m-1
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t lo = 64u * w, hi = 64u * (w + 1);
            V.w[w] = m >= hi ? ~0ull : (m <= lo ? 0ull : (~0ull << (hi - m)));
        }

        const uint32_t col0 = (uint32_t)(t * CPL);
        bv M[CPL];      // match mask of my columns: the bit of char j is 0 <=> pattern[j] == text[col]
        {
            Planes tw = {0, 0};                       // my own CPL text characters (bit k = column col0+k)
            if (has_pair && col0 < n) tw = load_window(a.seq, text_off + ref_idx + col0);
            const uint32_t tlo = (uint32_t)tw.lo, thi = (uint32_t)tw.hi;
            uint64_t plo[NW], phi[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) {
                Planes p = {0, 0};
                if (has_pair && 64u * w < m) p = load_window(a.seq, read_off + read_idx + 64u * w);
                plo[w] = brev64(p.lo);               // character 64w+k -> bit 63-k
                phi[w] = brev64(p.hi);
            }
#pragma unroll
            for (int k = 0; k < CPL; k++) {
                const uint64_t sl = (uint64_t)(int64_t)(-(int32_t)((tlo >> k) & 1u));
                const uint64_t sh = (uint64_t)(int64_t)(-(int32_t)((thi >> k) & 1u));
                // columns at or past n behave as the all-insertions boundary column (genasm_cpu.cpp:239-245)
                const bool past = col0 + (uint32_t)k >= n;
#pragma unroll
                for (int w = 0; w < NW; w++)
                    M[k].w[w] = past ? V.w[w] : (((plo[w] ^ sl) | (phi[w] ^ sh)) & V.w[w]);
            }
        }

        // ---------------- GenASM-DC, skewed row sweep (genasm_cpu.cpp:210-288) ----------------
        bv pA[CPL], psA[CPL], pB[CPL], psB[CPL];
#pragma unroll
        for (int k = 0; k < CPL; k++) pA[k] = psA[k] = pB[k] = psB[k] = bv_fill<NW>(~0ull);
        bv rnA = bv_fill<NW>(~0ull), rnsA = rnA, rnB = rnA, rnsB = rnA;
        bv bnd = V;                                // virtual column 64*NW at the current row: V << d
        int d = -(G - 1 - t);
        uint32_t dw = 0;
        uint64_t done_mask = __ballot(!has_pair);
        bool all_done = (done_mask & leaders) == leaders;
        int32_t st_limit = (has_pair && t < ST) ? RB : INT32_MIN;
        int32_t hit_thr = (has_pair && leader) ? -1 : INT32_MAX;
        uint32_t saddr = lds_slot + col0 * SW * 2u;
        int step = 0;

        auto dc_step = [&](const bv (&pi)[CPL], const bv (&psi)[CPL], const bv& rni, const bv& rnsi,
                           bv (&po)[CPL], bv (&pso)[CPL], bv& rno, bv& rnso) {
            bv rn = bv_dpp_from_next<NW>(pi[0]);
            const bv bnds = bv_shl1<NW>(bnd);
            if (t == G - 1) rn = bnd;              // last lane: the boundary column (always >= n)
            const bv rns = bv_shl1<NW>(rn);
            bnd = bnds;
            if (d >= 0) {
                bv right_s = rns, tr = rni, trs = rnsi;
#pragma unroll
                for (int k = CPL - 1; k >= 0; k--) {
                    bv c;
#pragma unroll
                    for (int w = 0; w < NW; w++) {
                        // ins & sub & del = (top<<1) & (topright<<1) & topright     (genasm_cpu.cpp:248-250)
                        const uint64_t x = psi[k].w[w] & trs.w[w] & tr.w[w];
                        c.w[w] = (right_s.w[w] | M[k].w[w]) & x;              // :247, :251
                    }
                    const bv cs = bv_shl1<NW>(c);
                    tr = pi[k];
                    trs = psi[k];
                    po[k] = c;
                    pso[k] = cs;
                    right_s = cs;
                }
                rno = rn;
                rnso = rns;
                if (d < st_limit) {                 // SENE + DENT store: words 0..SW-1 of the stored columns
#pragma unroll
                    for (int k = 0; k < CPL; k++) {
#pragma unroll
                        for (int w = 0; w < NW; w++) {
                            if ((uint32_t)w < SW) {
                                lds[saddr + ((uint32_t)k * SW + w) * 2u] = (uint32_t)po[k].w[w];
                                lds[saddr + ((uint32_t)k * SW + w) * 2u + 1u] = (uint32_t)(po[k].w[w] >> 32);
                            }
                        }
                    }
                }
                saddr += ROWDW;
            }
            if (step >= RB + (G - ST)) {            // rows >= RB of a live slot go to the HBM spill area
                int d_here = d;
                asm volatile("" : "+v"(d_here));
                if (d_here >= RB && st_limit > 0) {
                    const uint32_t row = (uint32_t)d_here < spill_rows ? (uint32_t)d_here : spill_rows - 1u;
                    uint64_t* const rowp = reinterpret_cast<uint64_t*>(Rs + (size_t)row * ROWDW + col0 * SW * 2u);
#pragma unroll
                    for (int k = 0; k < CPL; k++) {
#pragma unroll
                        for (int w = 0; w < NW; w++)
                            if ((uint32_t)w < SW) rowp[(uint32_t)k * SW + w] = po[k].w[w];
                    }
                }
            }
            // early termination: column 0 reaches the goal bit = character 0 (genasm_cpu.cpp:278-283)
            const uint64_t hits = __ballot((int32_t)(po[0].w[0] >> 32) > hit_thr);
            if (hits) {
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
        while (!all_done) {
            dc_step(pA, psA, rnA, rnsA, pB, psB, rnB, rnsB);
            if (all_done) break;
            dc_step(pB, psB, rnB, rnsB, pA, psA, rnA, rnsA);
        }
        const bool spilled = __any(has_pair && dw > (uint32_t)RB);
        if (spilled) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        // ---------------- GenASM-TB, lane-parallel diagonal scan (genasm_cpu.cpp:290-409) ----------------
        {
            uint32_t i = 0, j = 0, dd = dw;
            uint32_t cur_op = 0, cur_cnt = 0;
            bool act = has_pair;
            const uint32_t jlim = min(m, (uint32_t)TBL);

            auto emit = [&](bool en, uint32_t op, uint32_t cnt) {
                const bool same = (op == cur_op);
                if (en && !same && cur_cnt != 0) {
                    if (n_runs < cigar_cap) {
                        if (leader) lds16[2u * obuf + (n_runs & 31u)] = (uint16_t)(cur_cnt | (cur_op << 8));
                        if ((n_runs & 15u) == 15u) {
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
                const uint32_t il = i + t, jl = j + t;
                const bool pos_ok = (jl < jlim) && (il < (uint32_t)TBL);
                const bool room = dd > 0;
                const uint32_t r = room ? dd - 1 : 0u;
                // the stored bit of character ch in column col of row r (speculative lanes are clamped
                // into the stored part; their result is discarded by pos_ok)
                const uint32_t cc = il < SCOLS - 2u ? il : SCOLS - 2u;
                const uint32_t cj = jl < SCOLS - 2u ? jl : SCOLS - 2u;
                const bool from_spill = spilled && r >= (uint32_t)RB;
                const uint32_t* const rowp = Rs + (size_t)r * ROWDW;
                const uint32_t lrow = lds_slot + (r < (uint32_t)RB ? r : 0u) * ROWDW;
                auto stored_bit = [&](uint32_t col, uint32_t ch) -> uint32_t {
                    const uint32_t idx = (col * SW + (ch >> 6)) * 2u + (((ch >> 5) & 1u) ^ 1u);
                    uint32_t v;
                    if (from_spill) v = __hip_atomic_load(rowp + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else v = lds[lrow + idx];
                    return (v >> (31u - (ch & 31u))) & 1u;
                };
                const uint32_t b_ins = stored_bit(cc, cj + 1u);          // R[i][d-1],   TB_BIT(j+1)
                const uint32_t b_del = stored_bit(cc + 1u, cj);          // R[i+1][d-1], TB_BIT(j)
                const uint32_t b_sub = stored_bit(cc + 1u, cj + 1u);     // R[i+1][d-1], TB_BIT(j+1)
                const bool last = (jl + 1u == m);
                const bool tl = il < n;
                const bool ins = room && (last || b_ins == 0u);
                const bool del = room && tl && !last && b_del == 0u;
                const bool sub = room && tl && (last || b_sub == 0u);
                uint32_t ev = ins ? 1u : (del ? 2u : (sub ? 3u : 0u));
                ev = pos_ok ? ev : 4u;
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
            emit(has_pair, 0u, 0u);
            if (has_pair) {
                edits += dw - dd;
                ref_idx += i;
                read_idx += j;
            }
        }
    }
}

template <int G, int NW>
static hipError_t launch_multiword_t(const AlignArgs& a, int grid, size_t lds_bytes, hipStream_t s)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&genasm_align_multiword_kernel<G, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((genasm_align_multiword_kernel<G, NW>), dim3(grid), dim3(64), lds_bytes, s, a);
    return hipGetLastError();
}

hipError_t launch_align_multiword(int lanes_per_pair, const AlignArgs& a, int grid, size_t lds_bytes, hipStream_t s)
{
    const bool four = a.W > 128;
    switch (lanes_per_pair) {
    case 32: return four ? launch_multiword_t<32, 4>(a, grid, lds_bytes, s) : launch_multiword_t<32, 2>(a, grid, lds_bytes, s);
    case 64: return four ? launch_multiword_t<64, 4>(a, grid, lds_bytes, s) : launch_multiword_t<64, 2>(a, grid, lds_bytes, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace scrg
