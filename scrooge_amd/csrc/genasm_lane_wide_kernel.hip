// genasm_lane_wide_kernel.hip — the lane-per-pair aligner for window overlaps with W-O > 31 (W <= 64): the same
// formulation as genasm_lane_kernel.hip — every lane aligns its own pair, the window's table holds the DIFFERENCES
// of the edit-distance matrix behind the GenASM bitvectors (see the header of that file for why this gives the
// reference's edit distance and CIGAR bit for bit, src/genasm_cpu.cpp:210-409) — but a window may now consume up to
// W-O <= 63 characters, so a table row has 64 bits and there are up to 63 columns: 2 x 8 bytes x 63 columns per
// lane do not fit the register file and live in LDS (35 to 66 KB per wavefront, four to two wavefronts per CU).  This is the
// configuration of the reference's overlap sweeps (scripts/profile.py:88-100), not the tuned one: the code is written
// with plain 64-bit arithmetic and loops over the columns at run time; what it shares with the tuned kernel is the
// arithmetic (tests/proto/lane_proto.c restates both), the CIGAR staging ring and the work queue.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "genasm_kernels.h"
#include "genasm_device.h"

namespace scrg {

constexpr uint32_t WIDE_RING_BYTES = 68;            // 32 runs + one dword per lane (bank skew)
constexpr uint32_t WIDE_LEN_BYTES = 68;             // insertion-run length of each of up to 63 columns, one byte each
// (the table: two 64-bit words per column, lane_wide_tab_bytes(W-O) per lane: 35 KB per wavefront at W-O = 32, 66 KB at 62)
static_assert(WIDE_RING_BYTES == 68 && WIDE_LEN_BYTES == 68, "lane_wide_lds_bytes() in genasm_kernels.h assumes these");

__device__ __forceinline__ uint32_t clz64(uint64_t v) { return v ? (uint32_t)__builtin_clzll(v) : 64u; }

__global__ __launch_bounds__(64) void genasm_lane_wide_kernel(AlignArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    char* const lds_b = reinterpret_cast<char*>(lds);
    uint8_t* const lds8 = reinterpret_cast<uint8_t*>(lds);

    const uint32_t lane = threadIdx.x;
    const uint32_t ring_b = lane * WIDE_RING_BYTES;
    const uint32_t len_b = 64u * WIDE_RING_BYTES + lane * WIDE_LEN_BYTES;
    const uint32_t tab_b = 64u * (WIDE_RING_BYTES + WIDE_LEN_BYTES) + lane * lane_wide_tab_bytes(a.tb_limit);
    const uint32_t W = (uint32_t)a.W;
    const uint32_t TBL = (uint32_t)a.tb_limit;         // W - O, 32..63 (any value 1..63 works)

    // ---- per-lane pair state (as in genasm_lane_kernel) ----
    bool has_pair = false;
    uint32_t pair = 0;
    uint64_t text_off = 0, read_off = 0, cigar_off = 0;
    uint32_t text_len = 0, read_len = 0, cigar_cap = 0;
    uint32_t ref_idx = 0, read_idx = 0, edits = 0;
    int32_t nr = -1;                   // index of the last committed run; n_runs = nr + 1
    uint32_t flushed = 0;              // runs below this index are in HBM (a multiple of 16)
    bool queue_empty = false;          // wave-uniform

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
    // every run up to index nr is final here (runs are committed whole): keep fewer than 16 of them staged
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

    for (;;) {
        // ---------------- retire finished pairs, fetch new ones (genasm_cpu.cpp:440-460) ----------------
        for (;;) {
            const bool fin = has_pair && read_idx >= read_len;
            if (__any(fin)) {
                if (fin) {
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
                read_off = pd.read_off;
                text_len = pd.text_len > 0xffffffffull ? 0xffffffffu : (uint32_t)pd.text_len;
                read_len = (uint32_t)pd.read_len;
                cigar_off = pd.cigar_off;
                cigar_cap = pd.cigar_cap > 0xffffffffull ? 0xffffffffu : (uint32_t)pd.cigar_cap;
                ref_idx = read_idx = edits = flushed = 0;
                nr = -1;
                has_pair = true;
            }
        }
        if (!__any(has_pair)) break;

        // ---------------- window setup (genasm_cpu.cpp:417-420) ----------------
        const uint32_t n = (has_pair && ref_idx < text_len) ? min(W, text_len - ref_idx) : 0u;
        const uint32_t m = has_pair ? min(W, read_len - read_idx) : 1u;      // >= 1 for live pairs
        Planes tw = {0, 0}, pw = {0, 0};
        if (has_pair) {
            tw = load_window_strided(a.seq, text_off, ref_idx, a.text_stride);
            pw = load_window_strided(a.seq, read_off, read_idx, a.read_stride);
        }
        const uint32_t jlim = has_pair ? min(m, TBL) : 0u;          // the walk ends when j gets here (:301, :310)
        const uint64_t stop = 0x8000000000000000ull >> jlim;          // (jlim <= 63)

        // ---------------- the window's table (genasm_cpu.cpp:210-288 in difference form) ----------------
        // Reversed pattern left-aligned (bit 63-k <-> pattern[k]); the 64-m bits below it are neutral (Eq = 1,
        // Pv = Mv = 0); columns past the end of the text take the Eq word "no character matches", which keeps the
        // boundary column D[n][j] = m-j and yields "insertion in every row".  Column i < W-O keeps
        // ~(V1 | stop) and V0 | stop, V1 = Pv' | Ph, V0 = Pv' | ~(Ph | Xh), bit 63-j <-> pattern character j.
        {
            const uint64_t rlo = brev64(pw.lo), rhi = brev64(pw.hi);
            const uint64_t valid = ~0ull << (64u - m);
            uint64_t pv = valid, mv = 0;
            for (int i = 63; i >= 0; i--) {
                const uint64_t sl = 0ull - ((tw.lo >> i) & 1ull), sh = 0ull - ((tw.hi >> i) & 1ull);
                const uint64_t eq = ((uint32_t)i < n ? ~((rlo ^ sl) | (rhi ^ sh)) : 0ull) | ~valid;
                const uint64_t xv = eq | mv;
                const uint64_t xh = (((eq & pv) + pv) ^ pv) | eq;
                const uint64_t ph = mv | ~(xh | pv);
                const uint64_t mh = pv & xh;
                const uint64_t phs = ph << 1, mhs = mh << 1;          // row 0 of the matrix is all zeros: 0 comes in
                const uint64_t pvn = mhs | ~(xv | phs);
                mv = phs & xv;
                pv = pvn;
                if ((uint32_t)i < TBL) {
                    uint64_t* const t = reinterpret_cast<uint64_t*>(lds_b + tab_b + 16u * (uint32_t)i);
                    t[0] = ~((pvn | ph) | stop);
                    t[1] = (pvn | ~(ph | xh)) | stop;
                }
            }
        }

        // ---------------- traceback (genasm_cpu.cpp:290-409), the two passes of genasm_lane_kernel on 64-bit rows ----------------
        {
            uint32_t j = 0;
            uint64_t nDm = 0, Xm = 0, nIm = 0;
            for (uint32_t i = 0; i < TBL; i++) {
                const uint64_t* const t = reinterpret_cast<const uint64_t*>(lds_b + tab_b + 16u * i);
                const uint64_t nv1 = t[0], v0 = t[1];
                const uint64_t x = (nv1 | ~v0 | stop) << j;            // not (insertion), or the stop row
                const uint32_t ni = clz64(x);                          // (the stop bit makes x non-zero)
                lds8[len_b + i] = (uint8_t)ni;
                nIm = (nIm << 1) | (x >> 63);
                j += ni;
                const uint64_t nt1 = nv1 << j, t0 = v0 << j;           // top bits: not a deletion, substitution
                nDm = (nDm << 1) | (nt1 >> 63);
                Xm = (Xm << 1) | (t0 >> 63);
                j += (uint32_t)(nt1 >> 63);                            // a deletion (or the stop row) keeps j
            }
            // column i -> bit 63-i; a finished lane reads "deletion and substitution" (the stop row)
            const uint32_t nsh = 64u - TBL;
            const uint64_t Draw = ~(nDm << nsh), Xraw = Xm << nsh;
            const uint32_t ti = clz64((Draw & Xraw) | (0x8000000000000000ull >> TBL));
            const uint64_t A = ti ? ~(~0ull >> ti) : 0ull;
            const uint64_t D = Draw & A, X = Xraw & A, Im = ~nIm << nsh;
            const uint64_t B = ((D ^ (D >> 1)) | (X ^ (X >> 1)) | Im | 0x8000000000000000ull) & A;
            edits += j - ti + 2u * (uint32_t)__popcll(D) + (uint32_t)__popcll(X);
            ref_idx += ti;
            read_idx += j;

            uint64_t E = B | Im;
            while (__any(E != 0ull)) {
                if (E) {
                    const uint32_t c = clz64(E);
                    const uint64_t bit = 0x8000000000000000ull >> c;
                    if (Im & bit) push_run((uint32_t)'I', lds8[len_b + c]);
                    E &= ~bit;
                    const uint32_t nx = min(clz64(E), ti);              // the next event or the end of the walk
                    if (B & bit) push_run((D & bit) ? (uint32_t)'D' : ((X & bit) ? (uint32_t)'X' : (uint32_t)'='), nx - c);
                }
                flush_pieces();                                         // at most two new runs per iteration
            }
        }
    }
}

hipError_t launch_align_lane_wide(const AlignArgs& a, int grid, size_t lds_bytes, hipStream_t s)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&genasm_lane_wide_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(genasm_lane_wide_kernel, dim3(grid), dim3(64), lds_bytes, s, a);
    return hipGetLastError();
}

}  // namespace scrg
