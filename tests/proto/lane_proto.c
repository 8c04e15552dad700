/*
 * lane_proto.c — CPU prototype (not shipped, not part of the oracle) of the lane-per-pair window of the
 * HIP kernel (genasm_lane_kernel.hip).  Used to validate its arithmetic against the oracle before and
 * after porting it.
 *
 * The GenASM table is a thresholded edit-distance matrix: bit j of R[i][d] is clear exactly when
 * D[i][j] <= d, where D[i][j] is the least number of edits that align pattern[j..m) to a prefix of
 * text[i..n) (genasm_cpu.cpp:225-252 is the Wu-Manber recurrence of that matrix; D[n][j] = m-j is the
 * boundary column :239-245, D[i][m] = 0 the bit shifted in by `<< 1`).  The traceback (:290-409) keeps
 * d == D[i][j] and asks three questions per cell, in the order I, D, X:
 *     D[i][j+1] == d-1 ?   D[i+1][j] == d-1 ?   D[i+1][j+1] == d-1 ?
 * i.e. it looks at the vertical, horizontal and diagonal DIFFERENCES of D.  Those differences are what
 * the Myers/Hyyro bit-vector recurrence carries (Pv/Mv vertical, Ph/Mh horizontal, D0 diagonal), so a
 * whole window column — all 64 pattern rows, every distance at once — is ~20 word operations, with no
 * loop over d and nothing that depends on the window distance.  Per text column i (processed n-1..0)
 * the prototype keeps two words for the traceback, in the order the kernel uses them:
 *     V1 = Pv' | Ph          "insertion or deletion"
 *     V0 = Pv' | ~(Ph | Xh)  "insertion or substitution"        (both set = insertion)
 * left-aligned so that bit 31-j belongs to pattern character j (the DENT word, :200-208).
 * The traceback is column-synchronous: in column i the run of insertions is one count-leading-zeros
 * over V1 & V0, then one D / X / = step moves to column i+1 (lane_tb below restates the kernel's two passes).
 * For W-O > 31 (not served by the lane kernel) lane_tb_wide walks the same table cell by cell.
 */
#include "../../oracle/genasm_oracle.c"

static uint64_t lp_brev64(uint64_t v)
{
    uint64_t r = 0;
    for (int k = 0; k < 64; k++) if ((v >> k) & 1) r |= 1ull << (63 - k);
    return r;
}

typedef struct lane_stats { uint64_t windows, columns, tb_columns; } lane_stats;

/* window table: V1[i], V0[i] for i < TBL (<= 63): 64-bit left-aligned (bit 63-j <-> pattern char j) */
static void lane_dc(const uint8_t *t, int n, const uint8_t *q, int m, int TBL, uint64_t *V1, uint64_t *V0, lane_stats *ls)
{
    /* planes as load_window() delivers them: bit k <-> character k; characters past the end are whatever
     * follows in memory (a fixed filler here) */
    uint64_t Tlo = 0, Thi = 0, Plo = 0, Phi = 0;
    for (int k = 0; k < 64; k++) {
        uint8_t tc = k < n ? t[k] : (uint8_t)((k * 5 + 1) & 3);
        uint8_t qc = k < m ? q[k] : (uint8_t)((k * 7 + 3) & 3);
        Tlo |= (uint64_t)(tc & 1) << k; Thi |= (uint64_t)(tc >> 1) << k;
        Plo |= (uint64_t)(qc & 1) << k; Phi |= (uint64_t)(qc >> 1) << k;
    }
    /* reversed pattern, LEFT-aligned: bit 63-k <-> pattern[k], i.e. the reference's layout (bit b <-> pattern[m-1-b],
     * genasm_cpu.cpp:178-198) shifted left by 64-m; the 64-m bits below the pattern are kept neutral: Eq = 1,
     * Pv = Mv = 0, so no carry starts there and 0 comes in at the pattern's lowest bit */
    const uint64_t Rlo = lp_brev64(Plo), Rhi = lp_brev64(Phi);
    const uint64_t valid = ~0ull << (64 - m);
    uint64_t Pv = valid, Mv = 0;
    /* columns >= n (the text ends inside the window) take the Eq word "no character matches": the boundary column
     * D[n][j] = m-j stays as it is and the table words come out as "insertion in every row" by themselves */
    for (int i = 63; i >= 0; i--) {
        const uint64_t sl = 0ull - ((Tlo >> i) & 1), sh = 0ull - ((Thi >> i) & 1);
        const uint64_t Eq = (i < n ? ~((Rlo ^ sl) | (Rhi ^ sh)) : 0ull) | ~valid;  /* (the kernel reads it from a table in LDS) */
        const uint64_t Xv = Eq | Mv;
        const uint64_t Xh = ((((Eq & Pv) + Pv) ^ Pv) | Eq);
        const uint64_t Ph = Mv | ~(Xh | Pv);
        const uint64_t Mh = Pv & Xh;
        const uint64_t Ph1 = Ph << 1, Mh1 = Mh << 1;              /* row 0 of the matrix is all zeros: shift in 0 */
        const uint64_t Pvn = Mh1 | ~(Xv | Ph1);
        const uint64_t Mvn = Ph1 & Xv;
        if (i < TBL) {
            V1[i] = Pvn | Ph;
            V0[i] = Pvn | ~(Ph | Xh);
        }
        Pv = Pvn; Mv = Mvn;
        ls->columns++;
    }
}

static unsigned lp_ffbh32(uint32_t v) { return v ? (unsigned)__builtin_clz(v) : 0xffffffffu; }   /* v_ffbh_u32 */
static uint32_t lp_alignbit(uint32_t hi, uint32_t lo, unsigned s) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> s); }

/* The kernel's traceback (W-O <= 31), restated step by step: pass 1 walks the columns and records the path in
 * three masks + one byte per column, pass 2 turns the masks into runs.  nv1[i] = ~(V1 | stop), v0[i] = V0 (high
 * dwords: bit 31-j <-> pattern character j, with the stop bit); stop has the bit of row jlim. */
static int lane_tb(const uint64_t *V1, const uint64_t *V0, int m, int TBL, size_t *tu, size_t *pu, run_sink *out, lane_stats *ls)
{
    const uint32_t jlim = (uint32_t)(m < TBL ? m : TBL);    /* j < m && j < W-O, genasm_cpu.cpp:307-310 */
    const uint32_t stop = 0x80000000u >> jlim;
    uint32_t j = 0, ti = 0, nDm = 0, Xm = 0, nIm = 0;
    uint8_t ilen[32];
    for (int i = 0; i < TBL; i++) {
        const uint32_t nv1 = ~((uint32_t)(V1[i] >> 32) | stop), v0 = (uint32_t)(V0[i] >> 32) | stop;
        const uint32_t x = (nv1 | ~v0 | stop) << j;             /* not (insertion), or the stop row */
        const uint32_t ni = lp_ffbh32(x);
        ilen[i] = (uint8_t)ni;
        nIm = lp_alignbit(nIm, x, 31);
        j += ni;
        const uint32_t nt1 = nv1 << j, t0 = v0 << j;
        nDm = lp_alignbit(nDm, nt1, 31);
        Xm = lp_alignbit(Xm, t0, 31);
        j += nt1 >> 31;                                         /* a deletion (or the stop row) keeps j */
        ls->tb_columns++;
    }
    const unsigned nsh = 32u - (unsigned)TBL;
    /* a finished lane reads "deletion and substitution" (the stop row): the first such column ends the walk */
    const uint32_t Draw = ~(nDm << nsh), Xraw = Xm << nsh;
    ti = lp_ffbh32((Draw & Xraw) | (0x80000000u >> TBL));
    const uint32_t A = ti ? ~(0xffffffffu >> ti) : 0u;
    const uint32_t D = Draw & A, X = Xraw & A, Im = ~nIm << nsh;
    const uint32_t B = ((D ^ (D >> 1)) | (X ^ (X >> 1)) | Im | 0x80000000u) & A;
    const int edits = (int)(j - ti + 2u * (unsigned)__builtin_popcount(D) + (unsigned)__builtin_popcount(X));
    uint32_t E = B | Im;
    while (E) {
        const unsigned c = lp_ffbh32(E);
        const uint32_t bit = 0x80000000u >> c;
        if (Im & bit) sink_push(out, 'I', ilen[c]);
        E &= ~bit;
        unsigned nx = lp_ffbh32(E);
        if (nx > ti) nx = ti;
        if (B & bit) sink_push(out, (D & bit) ? 'D' : ((X & bit) ? 'X' : '='), nx - c);
    }
    *tu = ti; *pu = j;
    return edits;
}

/* The kernel's edit-stream variant of pass 2 (genasm_lane_kernel<true>): the same pass 1, then one visit per column
 * that holds an edit.  mbase + c = matches pending when column c is reached (carried from window to window);
 * an insertion at c leaves none at c, a deletion/substitution none at c + 1. */
typedef struct { uint8_t *p; size_t cap, n; uint32_t mbase; } edit_sink;
static void es_put(edit_sink *o, uint32_t b) { if (o->n < o->cap) o->p[o->n] = (uint8_t)b; o->n++; }

static int lane_tb_edits(const uint64_t *V1, const uint64_t *V0, int m, int TBL, size_t *tu, size_t *pu, edit_sink *out, lane_stats *ls)
{
    const uint32_t jlim = (uint32_t)(m < TBL ? m : TBL);
    const uint32_t stop = 0x80000000u >> jlim;
    uint32_t j = 0, ti = 0, nDm = 0, Xm = 0, nIm = 0;
    uint8_t ilen[32];
    for (int i = 0; i < TBL; i++) {
        const uint32_t nv1 = ~((uint32_t)(V1[i] >> 32) | stop), v0 = (uint32_t)(V0[i] >> 32) | stop;
        const uint32_t x = (nv1 | ~v0 | stop) << j;
        const uint32_t ni = lp_ffbh32(x);
        ilen[i] = (uint8_t)ni;
        nIm = lp_alignbit(nIm, x, 31);
        j += ni;
        const uint32_t nt1 = nv1 << j, t0 = v0 << j;
        nDm = lp_alignbit(nDm, nt1, 31);
        Xm = lp_alignbit(Xm, t0, 31);
        j += nt1 >> 31;
        ls->tb_columns++;
    }
    const unsigned nsh = 32u - (unsigned)TBL;
    const uint32_t Draw = ~(nDm << nsh), Xraw = Xm << nsh;
    ti = lp_ffbh32((Draw & Xraw) | (0x80000000u >> TBL));
    const uint32_t A = ti ? ~(0xffffffffu >> ti) : 0u;
    const uint32_t D = Draw & A, X = Xraw & A, Im = ~nIm << nsh;
    const int edits = (int)(j - ti + 2u * (unsigned)__builtin_popcount(D) + (unsigned)__builtin_popcount(X));
    uint32_t E = D | X | Im;
    while (E) {
        const unsigned c = lp_ffbh32(E);
        const uint32_t bit = 0x80000000u >> c;
        uint32_t t = out->mbase + c;
        E &= ~bit;
        if (Im & bit) {
            for (uint32_t q = t >> 6; q; q--) es_put(out, 0x3F);
            es_put(out, 0x80u | (t & 63u));
            for (uint32_t q = 1; q < ilen[c]; q++) es_put(out, 0x80u);
            t = 0;
            out->mbase = 0u - c;
        }
        if ((D | X) & bit) {
            for (uint32_t q = t >> 6; q; q--) es_put(out, 0x3F);
            es_put(out, ((X & bit) ? 0x40u : 0xC0u) | (t & 63u));
            out->mbase = ~c;
        }
    }
    out->mbase += ti;
    *tu = ti; *pu = j;
    return edits;
}

int lane_align_edits(const uint8_t *text, size_t text_len, const uint8_t *read, size_t read_len, int W, int O,
                     uint8_t *stream, size_t cap, size_t *n_bytes, long long *edit_distance, lane_stats *ls)
{
    if (W < 2 || W > 64 || O < 1 || O >= W || W - O > 31) return GO_ERR_PARAMS;
    edit_sink out = { stream, cap, 0, 0 };
    size_t ti = 0, ri = 0; long long total = 0;
    const int TBL = W - O;
    uint64_t V1[64], V0[64];
    while (ri < read_len) {
        size_t n = text_len - ti < (size_t)W ? text_len - ti : (size_t)W;
        size_t m = read_len - ri < (size_t)W ? read_len - ri : (size_t)W;
        size_t tu, pu;
        lane_dc(text + ti, (int)n, read + ri, (int)m, TBL, V1, V0, ls);
        total += lane_tb_edits(V1, V0, (int)m, TBL, &tu, &pu, &out, ls);
        ls->windows++;
        ti += tu; ri += pu;
    }
    *n_bytes = out.n; *edit_distance = total;
    return out.n > cap ? GO_ERR_CAPACITY : GO_OK;
}

static int lp_clz64(uint64_t v) { return v ? __builtin_clzll(v) : 64; }

static int lane_tb_wide(const uint64_t *V1, const uint64_t *V0, int m, int TBL, size_t *tu, size_t *pu, run_sink *out, lane_stats *ls)
{
    const int jlim = m < TBL ? m : TBL;
    int i = 0, j = 0, edits = 0;
    char cur = 0; unsigned cur_len = 0;
#define LP_EVENT(op_, n_) do { if (cur == (op_)) cur_len += (n_); else { if (cur_len) sink_push(out, cur, cur_len); cur = (op_); cur_len = (n_); } } while (0)
    for (i = 0; i < TBL && j < jlim; i++) {
        ls->tb_columns++;
        const uint64_t Iv = V1[i] & V0[i];
        int r = lp_clz64(~(Iv << j));
        int ni = r < jlim - j ? r : jlim - j;
        if (ni) { LP_EVENT('I', (unsigned)ni); j += ni; edits += ni; }
        if (j >= jlim) break;
        const int c1 = (int)((V1[i] << j) >> 63), c0 = (int)((V0[i] << j) >> 63);
        if (c1) { LP_EVENT('D', 1u); edits++; }
        else if (c0) { LP_EVENT('X', 1u); j++; edits++; }
        else { LP_EVENT('=', 1u); j++; }
    }
    if (cur_len) sink_push(out, cur, cur_len);
    *tu = (size_t)i; *pu = (size_t)j;
    return edits;
}

int lane_align_codes(const uint8_t *text, size_t text_len, const uint8_t *read, size_t read_len, int W, int O,
                     go_run *runs, size_t cap, size_t *n_runs, long long *edit_distance, lane_stats *ls)
{
    if (W < 2 || W > 64 || O < 1 || O >= W) return GO_ERR_PARAMS;
    run_sink out = { runs, cap, 0, 0 };
    size_t ti = 0, ri = 0; long long total = 0;
    const int TBL = W - O;
    uint64_t V1[64], V0[64];
    while (ri < read_len) {
        size_t n = text_len - ti < (size_t)W ? text_len - ti : (size_t)W;
        size_t m = read_len - ri < (size_t)W ? read_len - ri : (size_t)W;
        size_t tu, pu;
        lane_dc(text + ti, (int)n, read + ri, (int)m, TBL, V1, V0, ls);
        total += (TBL <= 31 ? lane_tb : lane_tb_wide)(V1, V0, (int)m, TBL, &tu, &pu, &out, ls);
        ls->windows++;
        ti += tu; ri += pu;
    }
    *n_runs = out.n; *edit_distance = total;
    return out.overflow ? GO_ERR_CAPACITY : GO_OK;
}
