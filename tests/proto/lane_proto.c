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

/* columns a lane was alive in, from the masks of pass 1 (column i <-> bit 31-i): notD = the step out of the column is
 * not a deletion, Im = an insertion run starts in it (genasm_lane_kernel.hip: the table's V0 words carry no stop bit) */
static uint32_t lane_alive_columns(uint32_t notD, uint32_t Im, uint32_t j, uint32_t jlim, uint32_t TBL)
{
    const uint32_t P = (notD >> 1) | Im;                          /* bit 31-c: the lane stopped in column c or later */
    const uint32_t ffbl = P ? (uint32_t)__builtin_ctz(P) : 0xffffffffu;     /* v_ffbl_b32 */
    const uint32_t stopped = (31u - ffbl) & 31u;                  /* (no such column: it never moved, 0) */
    return j < jlim ? TBL : stopped;
}

/* The kernel's traceback (W-O <= 31), restated step by step: pass 1 walks the columns and records the path in
 * three masks + one byte per column, pass 2 turns the masks into runs.  nv1[i] = ~(V1 | stop), v0[i] = V0 (high
 * dwords: bit 31-j <-> pattern character j); stop has the bit of row jlim. */
static int lane_tb(const uint64_t *V1, const uint64_t *V0, int m, int TBL, size_t *tu, size_t *pu, run_sink *out, lane_stats *ls)
{
    const uint32_t jlim = (uint32_t)(m < TBL ? m : TBL);    /* j < m && j < W-O, genasm_cpu.cpp:307-310 */
    const uint32_t stop = 0x80000000u >> jlim;
    uint32_t j = 0, ti = 0, nDm = 0, Xm = 0, nIm = 0;
    uint8_t ilen[32];
    for (int i = 0; i < TBL; i++) {
        const uint32_t nv1 = ~((uint32_t)(V1[i] >> 32) | stop), v0 = (uint32_t)(V0[i] >> 32);
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
    /* A lane that has not reached row jlim was alive in every column.  One that has stopped in the column after its last
     * step that was not a deletion (a diagonal step took it there) or in the column of its last insertion run (if that
     * run took it there); from then on it reads "deletion, no insertion" (the stop row).  (genasm_lane_kernel.hip) */
    const uint32_t notD = nDm << nsh, Xraw = Xm << nsh, Draw = ~notD, Im = ~nIm << nsh;
    ti = lane_alive_columns(notD, Im, j, jlim, (uint32_t)TBL);
    const uint32_t A = ti ? ~(0xffffffffu >> ti) : 0u;
    const uint32_t D = Draw & A, X = Xraw & A;
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
 * that holds an edit.  mbase + c = matches pending when column c is reached (the window's own: every window closes with
 * its END byte, edit_stream.h); an insertion at c leaves none at c, a deletion/substitution none at c + 1. */
typedef struct { uint8_t *p; size_t cap, n; uint32_t mbase; } edit_sink;
static void es_put(edit_sink *o, uint32_t b) { if (o->n < o->cap) o->p[o->n] = (uint8_t)b; o->n++; }

static int lane_tb_edits(const uint64_t *V1, const uint64_t *V0, int m, int TBL, size_t *tu, size_t *pu, edit_sink *out, lane_stats *ls)
{
    const uint32_t jlim = (uint32_t)(m < TBL ? m : TBL);
    const uint32_t stop = 0x80000000u >> jlim;
    uint32_t j = 0, ti = 0, nDm = 0, Xm = 0, nIm = 0;
    uint8_t ilen[32];
    for (int i = 0; i < TBL; i++) {
        const uint32_t nv1 = ~((uint32_t)(V1[i] >> 32) | stop), v0 = (uint32_t)(V0[i] >> 32);
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
    const uint32_t notD = nDm << nsh, Xraw = Xm << nsh, Draw = ~notD, Im = ~nIm << nsh;
    ti = lane_alive_columns(notD, Im, j, jlim, (uint32_t)TBL);
    const uint32_t A = ti ? ~(0xffffffffu >> ti) : 0u;
    const uint32_t D = Draw & A, X = Xraw & A;
    const int edits = (int)(j - ti + 2u * (unsigned)__builtin_popcount(D) + (unsigned)__builtin_popcount(X));
    uint32_t E = D | X | Im;
    while (E) {
        const unsigned c = lp_ffbh32(E);
        const uint32_t bit = 0x80000000u >> c;
        uint32_t t = out->mbase + c;
        E &= ~bit;
        if (Im & bit) {
            es_put(out, 0x80u | t);                        /* (t < W-O <= 31: the window's own matches) */
            for (uint32_t q = 1; q < ilen[c]; q++) es_put(out, 0x80u);
            t = 0;
            out->mbase = 0u - c;
        }
        if ((D | X) & bit) {
            es_put(out, ((X & bit) ? 0x40u : 0xC0u) | t);
            out->mbase = ~c;
        }
    }
    /* the window ends (format version 2): the matches since its last edit, then the mark */
    es_put(out, out->mbase + ti);
    out->mbase = 0;
    *tu = ti; *pu = j;
    return edits;
}

int lane_align_edits(const uint8_t *text, size_t text_len, const uint8_t *read, size_t read_len, int W, int O,
                     uint8_t *stream, size_t cap, size_t *n_bytes, long long *edit_distance, lane_stats *ls)
{
    if (W < 2 || W > 64 || O < 0 || O >= W || W - O > 31) return GO_ERR_PARAMS;
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
    if (W < 2 || W > 64 || O < 0 || O >= W) return GO_ERR_PARAMS;
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

/* ------------------------------------------------------------------------------------------------------------
 * W > 64 (genasm_lane_mw_kernel.hip): the same difference-vector table with multi-word vectors.  A pattern vector
 * has NW = ceil(W/64) words, word 0 the MOST significant (bit 63-k of word w <-> pattern character 64 w + k); the
 * traceback only looks at rows j <= W-O, so a table row keeps the top RW = (W-O)/64 + 1 words.  Two passes as in
 * lane_tb, on RW-word masks.
 * ---------------------------------------------------------------------------------------------------------- */
#define MW_MAXNW 4
#define MW_MAXRW 4
typedef struct { uint64_t w[MW_MAXRW]; } mw_row;

static mw_row mwr_shl(mw_row a, unsigned s, int RW)            /* towards word 0 */
{
    mw_row r = { {0} };
    for (int k = 0; k < RW; k++) {
        unsigned src = k + s / 64, b = s % 64;
        uint64_t v = 0;
        if ((int)src < RW) v = a.w[src] << b;
        if (b && (int)src + 1 < RW) v |= a.w[src + 1] >> (64 - b);
        r.w[k] = v;
    }
    return r;
}
static mw_row mwr_shr1(mw_row a, int RW)
{
    mw_row r = { {0} };
    for (int k = 0; k < RW; k++) r.w[k] = (a.w[k] >> 1) | (k ? a.w[k - 1] << 63 : 0);
    return r;
}
static unsigned mwr_clz(mw_row a, int RW)
{
    unsigned n = 0;
    for (int k = 0; k < RW; k++) { if (a.w[k]) return n + (unsigned)__builtin_clzll(a.w[k]); n += 64; }
    return n;
}
static mw_row mwr_bit(unsigned c, int RW) { mw_row r = { {0} }; if ((int)(c / 64) < RW) r.w[c / 64] = 0x8000000000000000ull >> (c % 64); return r; }
static int mwr_test(mw_row a, unsigned c) { return (int)((a.w[c / 64] >> (63 - c % 64)) & 1); }
static unsigned mwr_pop(mw_row a, int RW) { unsigned n = 0; for (int k = 0; k < RW; k++) n += (unsigned)__builtin_popcountll(a.w[k]); return n; }
static int mwr_any(mw_row a, int RW) { for (int k = 0; k < RW; k++) if (a.w[k]) return 1; return 0; }

int lane_align_codes_mw(const uint8_t *text, size_t text_len, const uint8_t *read, size_t read_len, int W, int O,
                        go_run *runs, size_t cap, size_t *n_runs, long long *edit_distance, lane_stats *ls)
{
    if (W < 2 || W > 256 || O < 1 || O >= W) return GO_ERR_PARAMS;
    const int NW = (W + 63) / 64, TBL = W - O, RW = TBL / 64 + 1;
    run_sink out = { runs, cap, 0, 0 };
    size_t tpos = 0, rpos = 0; long long total = 0;
    static _Thread_local mw_row NV1[256], V0[256];
    while (rpos < read_len) {
        const int n = (int)(text_len - tpos < (size_t)W ? text_len - tpos : (size_t)W);
        const int m = (int)(read_len - rpos < (size_t)W ? read_len - rpos : (size_t)W);
        const int jlim = m < TBL ? m : TBL;
        const mw_row stop = mwr_bit((unsigned)jlim, RW);
        /* reversed pattern, left-aligned over NW words; valid = the top m bits */
        uint64_t Rlo[MW_MAXNW] = {0}, Rhi[MW_MAXNW] = {0}, valid[MW_MAXNW] = {0};
        for (int k = 0; k < 64 * NW; k++) {
            const uint8_t qc = k < m ? read[rpos + k] : (uint8_t)((k * 7 + 3) & 3);
            Rlo[k / 64] |= (uint64_t)(qc & 1) << (63 - k % 64);
            Rhi[k / 64] |= (uint64_t)(qc >> 1) << (63 - k % 64);
            if (k < m) valid[k / 64] |= 1ull << (63 - k % 64);
        }
        uint64_t Pv[MW_MAXNW], Mv[MW_MAXNW] = {0};
        for (int w = 0; w < NW; w++) Pv[w] = valid[w];
        for (int i = 64 * NW - 1; i >= 0; i--) {
            const uint8_t tc = i < n ? text[tpos + i] : (uint8_t)((i * 5 + 1) & 3);
            const uint64_t sl = 0ull - (uint64_t)(tc & 1), sh = 0ull - (uint64_t)(tc >> 1);
            uint64_t Eq[MW_MAXNW], Xv[MW_MAXNW], Xh[MW_MAXNW], Ph[MW_MAXNW], Mh[MW_MAXNW], Pvn[MW_MAXNW], Mvn[MW_MAXNW];
            unsigned carry = 0;
            for (int w = NW - 1; w >= 0; w--) {                 /* the add: carries run from the last word to word 0 */
                Eq[w] = (i < n ? ~((Rlo[w] ^ sl) | (Rhi[w] ^ sh)) : 0ull) | ~valid[w];
                Xv[w] = Eq[w] | Mv[w];
                const uint64_t t = Eq[w] & Pv[w];
                const unsigned __int128 s = (unsigned __int128)t + Pv[w] + carry;
                carry = (unsigned)(s >> 64);
                Xh[w] = (((uint64_t)s) ^ Pv[w]) | Eq[w];
                Ph[w] = Mv[w] | ~(Xh[w] | Pv[w]);
                Mh[w] = Pv[w] & Xh[w];
            }
            for (int w = 0; w < NW; w++) {
                const uint64_t ph1 = (Ph[w] << 1) | (w + 1 < NW ? Ph[w + 1] >> 63 : 0);
                const uint64_t mh1 = (Mh[w] << 1) | (w + 1 < NW ? Mh[w + 1] >> 63 : 0);
                Pvn[w] = mh1 | ~(Xv[w] | ph1);
                Mvn[w] = ph1 & Xv[w];
            }
            if (i < TBL)
                for (int r = 0; r < RW; r++) {
                    NV1[i].w[r] = ~((Pvn[r] | Ph[r]) | stop.w[r]);
                    V0[i].w[r] = (Pvn[r] | ~(Ph[r] | Xh[r])) | stop.w[r];
                }
            for (int w = 0; w < NW; w++) { Pv[w] = Pvn[w]; Mv[w] = Mvn[w]; }
            ls->columns++;
        }
        /* pass 1 */
        unsigned j = 0;
        mw_row nDm = { {0} }, Xm = { {0} }, nIm = { {0} };
        uint8_t ilen[256];
        for (int i = 0; i < TBL; i++) {
            mw_row x;
            for (int r = 0; r < RW; r++) x.w[r] = NV1[i].w[r] | ~V0[i].w[r] | stop.w[r];
            x = mwr_shl(x, j, RW);
            const unsigned ni = mwr_clz(x, RW);
            ilen[i] = (uint8_t)ni;
            nIm = mwr_shl(nIm, 1, RW); nIm.w[RW - 1] |= x.w[0] >> 63;
            j += ni;
            const mw_row nt1 = mwr_shl(NV1[i], j, RW), t0 = mwr_shl(V0[i], j, RW);
            nDm = mwr_shl(nDm, 1, RW); nDm.w[RW - 1] |= nt1.w[0] >> 63;
            Xm = mwr_shl(Xm, 1, RW); Xm.w[RW - 1] |= t0.w[0] >> 63;
            j += (unsigned)(nt1.w[0] >> 63);
            ls->tb_columns++;
        }
        const unsigned nsh = 64u * (unsigned)RW - (unsigned)TBL;
        mw_row Draw = mwr_shl(nDm, nsh, RW), Xraw = mwr_shl(Xm, nsh, RW), Im, dead, A, D, X, B, E;
        for (int r = 0; r < RW; r++) { Draw.w[r] = ~Draw.w[r]; Im.w[r] = ~nIm.w[r]; }
        Im = mwr_shl(Im, nsh, RW);                               /* (complemented before the shift: nothing below column TBL-1) */
        const mw_row lim = mwr_bit((unsigned)TBL, RW);
        for (int r = 0; r < RW; r++) dead.w[r] = (Draw.w[r] & Xraw.w[r]) | lim.w[r];
        const unsigned ti = mwr_clz(dead, RW);
        for (int r = 0; r < RW; r++) {                       /* A = the top ti bits */
            const int lo = 64 * r;
            A.w[r] = (int)ti >= lo + 64 ? ~0ull : ((int)ti <= lo ? 0ull : ~(~0ull >> (ti - (unsigned)lo)));
        }
        for (int r = 0; r < RW; r++) { D.w[r] = Draw.w[r] & A.w[r]; X.w[r] = Xraw.w[r] & A.w[r]; }
        const mw_row D1 = mwr_shr1(D, RW), X1 = mwr_shr1(X, RW);
        for (int r = 0; r < RW; r++) B.w[r] = ((D.w[r] ^ D1.w[r]) | (X.w[r] ^ X1.w[r]) | Im.w[r] | (r == 0 ? 0x8000000000000000ull : 0)) & A.w[r];
        total += (long long)(j - ti + 2u * mwr_pop(D, RW) + mwr_pop(X, RW));
        for (int r = 0; r < RW; r++) E.w[r] = B.w[r] | Im.w[r];
        while (mwr_any(E, RW)) {
            const unsigned c = mwr_clz(E, RW);
            if (mwr_test(Im, c)) sink_push(&out, 'I', ilen[c]);
            E.w[c / 64] &= ~(0x8000000000000000ull >> (c % 64));
            unsigned nx = mwr_clz(E, RW);
            if (nx > ti) nx = ti;
            if (mwr_test(B, c)) sink_push(&out, mwr_test(D, c) ? 'D' : (mwr_test(X, c) ? 'X' : '='), nx - c);
        }
        ls->windows++;
        tpos += ti; rpos += j;
    }
    *n_runs = out.n; *edit_distance = total;
    return out.overflow ? GO_ERR_CAPACITY : GO_OK;
}

/* ------------------------------------------------------------------------------------------------------------
 * 32 <= W-O <= 63, W <= 128 (genasm_lane_band_kernel.hip): the traceback rows are one 64-bit word, too many bits for
 * a table in registers — but a walk that starts at (0, 0) stays near the diagonal.  Column i keeps the 32 rows
 * lo_i .. lo_i + 31, lo_i = clamp(i - 16, 0, 32), of both words (a table of 2 (W-O) dwords, as many registers as
 * the 31-column kernel's); the walk records, per column, whether it was above the band on entry or ran off its lower
 * end inside an insertion run.  A lane for which that happened in a column it was alive in has the window redone on
 * the full rows (lane_tb_full below = the multi-word walk with one word).  Rows >= jlim all carry the stop mark: a
 * live walk never gets past row jlim, and a finished lane reads "stop" wherever the band has moved to.
 * ---------------------------------------------------------------------------------------------------------- */
#define BAND_UP 16
static unsigned band_lo(int i) { int lo = i - BAND_UP; return (unsigned)(lo < 0 ? 0 : (lo > 32 ? 32 : lo)); }
static uint32_t band_ext(uint64_t w, int i) { return (uint32_t)(w >> (32u - band_lo(i))); }

typedef struct { uint64_t nDm, Xm, nIm; unsigned j; uint8_t ilen[64]; } walk_out;

/* full rows: what genasm_lane_mw_kernel's first pass does with RW = 1 */
static void lane_walk_full(const uint64_t *V1, const uint64_t *V0, int TBL, unsigned jlim, walk_out *o)
{
    const uint64_t stop = 0x8000000000000000ull >> jlim;
    unsigned j = 0;
    uint64_t nDm = 0, Xm = 0, nIm = 0;
    for (int i = 0; i < TBL; i++) {
        const uint64_t nv1 = ~(V1[i] | stop), v0 = V0[i] | stop;
        const uint64_t x = (nv1 | ~v0 | stop) << j;
        const unsigned ni = (unsigned)lp_clz64(x);
        o->ilen[i] = (uint8_t)ni;
        nIm = (nIm << 1) | (x >> 63);
        j += ni;
        const uint64_t nt1 = nv1 << j, t0 = v0 << j;
        nDm = (nDm << 1) | (nt1 >> 63);
        Xm = (Xm << 1) | (t0 >> 63);
        j += (unsigned)(nt1 >> 63);
    }
    o->nDm = nDm; o->Xm = Xm; o->nIm = nIm; o->j = j;
}

/* the band: returns 1 when the lane left it in a column it was alive in */
static int lane_walk_band(const uint64_t *V1, const uint64_t *V0, int TBL, unsigned jlim, walk_out *o)
{
    const uint64_t S = ~0ull >> jlim;                         /* rows >= jlim */
    uint32_t j = 0;
    uint64_t nDm = 0, Xm = 0, nIm = 0, Fm = 0;
    for (int i = 0; i < TBL; i++) {
        const uint32_t sb = band_ext(S, i);
        const uint32_t nv1 = ~(band_ext(V1[i], i) | sb), v0 = band_ext(V0[i], i) | sb;     /* (what the table holds) */
        const uint32_t jr = j - band_lo(i);                   /* negative: above the band, > 31: below it */
        /* a 64-bit shift (v_lshlrev_b64 takes the count modulo 64) of the word in the upper half: a row outside the
         * band shifts everything out, and so does an insertion run that reaches the band's lower end */
        const uint32_t x = (uint32_t)((((uint64_t)(nv1 | ~v0 | sb) << 32) << (jr & 63u)) >> 32);
        const uint32_t ni = lp_ffbh32(x);                     /* 0xffffffff: outside the band */
        o->ilen[i] = (uint8_t)ni;
        Fm = (Fm << 1) | (ni >> 31);
        nIm = (nIm << 1) | (x >> 31);
        j += ni;
        const uint32_t sh = (j - band_lo(i)) & 31u;
        const uint32_t nt1 = nv1 << sh, t0 = v0 << sh;
        nDm = (nDm << 1) | (nt1 >> 31);
        Xm = (Xm << 1) | (t0 >> 31);
        j += nt1 >> 31;
    }
    o->nDm = nDm; o->Xm = Xm; o->nIm = nIm; o->j = j;
    const unsigned nsh = 64u - (unsigned)TBL;
    const uint64_t Draw = ~(nDm << nsh), Xraw = Xm << nsh;
    const unsigned ti = (unsigned)lp_clz64((Draw & Xraw) | (0x8000000000000000ull >> TBL));
    const uint64_t alive_in = ti >= 63 ? ~0ull : ~(~0ull >> (ti + 1));      /* columns 0 .. ti: the lane entered them alive */
    /* what a finished lane reads once the band has moved past its row is arbitrary: its row is jlim (the stop mark is
     * the only place that reads "deletion and substitution"), and an insertion run can only start in a column it entered alive */
    if (ti < (unsigned)TBL) o->j = jlim;
    o->nIm |= ~(alive_in >> nsh);
    return ((Fm << nsh) & alive_in) != 0;
}

typedef struct lane_stats_band { uint64_t windows, columns, tb_columns, escapes; } lane_stats_band;

int lane_align_codes_band(const uint8_t *text, size_t text_len, const uint8_t *read, size_t read_len, int W, int O,
                          go_run *runs, size_t cap, size_t *n_runs, long long *edit_distance, lane_stats_band *ls)
{
    const int NW = (W + 63) / 64, TBL = W - O;
    if (W < 2 || W > 128 || O < 1 || TBL < 32 || TBL > 63) return GO_ERR_PARAMS;
    run_sink out = { runs, cap, 0, 0 };
    size_t tpos = 0, rpos = 0; long long total = 0;
    uint64_t V1[64], V0[64];
    while (rpos < read_len) {
        const int n = (int)(text_len - tpos < (size_t)W ? text_len - tpos : (size_t)W);
        const int m = (int)(read_len - rpos < (size_t)W ? read_len - rpos : (size_t)W);
        const unsigned jlim = (unsigned)(m < TBL ? m : TBL);
        uint64_t Rlo[2] = {0}, Rhi[2] = {0}, valid[2] = {0};
        for (int k = 0; k < 64 * NW; k++) {
            const uint8_t qc = k < m ? read[rpos + k] : (uint8_t)((k * 7 + 3) & 3);
            Rlo[k / 64] |= (uint64_t)(qc & 1) << (63 - k % 64);
            Rhi[k / 64] |= (uint64_t)(qc >> 1) << (63 - k % 64);
            if (k < m) valid[k / 64] |= 1ull << (63 - k % 64);
        }
        uint64_t Pv[2], Mv[2] = {0};
        for (int w = 0; w < NW; w++) Pv[w] = valid[w];
        for (int i = 64 * NW - 1; i >= 0; i--) {
            const uint8_t tc = i < n ? text[tpos + i] : (uint8_t)((i * 5 + 1) & 3);
            const uint64_t sl = 0ull - (uint64_t)(tc & 1), sh = 0ull - (uint64_t)(tc >> 1);
            uint64_t Eq[2], Xv[2], Xh[2], Ph[2], Mh[2], Pvn[2], Mvn[2];
            unsigned carry = 0;
            for (int w = NW - 1; w >= 0; w--) {
                Eq[w] = (i < n ? ~((Rlo[w] ^ sl) | (Rhi[w] ^ sh)) : 0ull) | ~valid[w];
                Xv[w] = Eq[w] | Mv[w];
                const uint64_t t = Eq[w] & Pv[w];
                const unsigned __int128 s = (unsigned __int128)t + Pv[w] + carry;
                carry = (unsigned)(s >> 64);
                Xh[w] = (((uint64_t)s) ^ Pv[w]) | Eq[w];
                Ph[w] = Mv[w] | ~(Xh[w] | Pv[w]);
                Mh[w] = Pv[w] & Xh[w];
            }
            for (int w = 0; w < NW; w++) {
                const uint64_t ph1 = (Ph[w] << 1) | (w + 1 < NW ? Ph[w + 1] >> 63 : 0);
                const uint64_t mh1 = (Mh[w] << 1) | (w + 1 < NW ? Mh[w + 1] >> 63 : 0);
                Pvn[w] = mh1 | ~(Xv[w] | ph1);
                Mvn[w] = ph1 & Xv[w];
            }
            if (i < TBL) { V1[i] = Pvn[0] | Ph[0]; V0[i] = Pvn[0] | ~(Ph[0] | Xh[0]); }
            for (int w = 0; w < NW; w++) { Pv[w] = Pvn[w]; Mv[w] = Mvn[w]; }
            ls->columns++;
        }
        walk_out wo;
        if (lane_walk_band(V1, V0, TBL, jlim, &wo)) {
            ls->escapes++;
            lane_walk_full(V1, V0, TBL, jlim, &wo);
        }
        ls->tb_columns += (uint64_t)TBL;
        /* pass 2 on one-word masks (as lane_align_codes_mw with RW = 1) */
        const unsigned nsh = 64u - (unsigned)TBL;
        const uint64_t Draw = ~(wo.nDm << nsh), Xraw = wo.Xm << nsh, Im = ~wo.nIm << nsh;
        const unsigned ti = (unsigned)lp_clz64((Draw & Xraw) | (0x8000000000000000ull >> TBL));
        const uint64_t A = ti ? ~(~0ull >> ti) : 0ull;        /* (ti <= 63) */
        const uint64_t D = Draw & A, X = Xraw & A;
        const uint64_t B = ((D ^ (D >> 1)) | (X ^ (X >> 1)) | Im | 0x8000000000000000ull) & A;
        total += (long long)(wo.j - ti + 2u * (unsigned)__builtin_popcountll(D) + (unsigned)__builtin_popcountll(X));
        uint64_t E = B | Im;
        while (E) {
            const unsigned c = (unsigned)lp_clz64(E);
            const uint64_t bit = 0x8000000000000000ull >> c;
            if (Im & bit) sink_push(&out, 'I', wo.ilen[c]);
            E &= ~bit;
            unsigned nx = (unsigned)lp_clz64(E);
            if (nx > ti) nx = ti;
            if (B & bit) sink_push(&out, (D & bit) ? 'D' : ((X & bit) ? 'X' : '='), nx - c);
        }
        ls->windows++;
        tpos += ti; rpos += wo.j;
    }
    *n_runs = out.n; *edit_distance = total;
    return out.overflow ? GO_ERR_CAPACITY : GO_OK;
}

/* ------------------------------------------------------------------------------------------------------------
 * Round 6 experiment: the window's table computed in a 32-ROW BAND around the main diagonal (W = 64, W-O <= 31).
 * The traceback keeps d == D[i][j] and only ever asks whether a neighbour's value is d - 1; every cell it visits or
 * asks about lies within d_w of the main diagonal (d_w = D[0][0], the window's distance) and its best completion
 * deviates by no more than its own value, so if d_w <= B all of it happens inside |i - j| <= B.  A band-restricted
 * computation gives over-estimates outside and the exact values inside (an over-estimate can only turn a true
 * "== d - 1" into a miss, never invent one), and its D[0][0] is itself an over-estimate: d_band <= B proves the window
 * safe.  Column i (swept 63 .. 0) keeps rows i - 15 .. i + 16 as ONE dword (bit 31 <-> row i - 15): the band moves up
 * one row per column, so the two shifts of the full recurrence become one (Hyyro's diagonal form), the 64-bit add a
 * 32-bit add, and every v_bitop3 pair a single one: 10 instead of 19 instructions per column.  Eq of the column is the
 * 64-bit word of the text character shifted to the band (a compile-time shift in the unrolled kernel).
 * d_band = 64 - (number of columns whose main-diagonal step is free), read from bit 17 of Xh | Mv in every column.
 * Windows with n < 64 or m < 64 (a pair's last) and windows with d_band > BAND32_SAFE take the full table.
 * ---------------------------------------------------------------------------------------------------------- */
#define BAND32_UP 15            /* rows above the diagonal (j < i) */
#define BAND32_DOWN 16          /* rows below it */
#define BAND32_SAFE 14

typedef struct lane_stats_band32 { uint64_t windows, banded, escapes, mismatching_safe_windows; uint64_t hist[66]; } lane_stats_band32;

/* -> d_band; V1[i], V0[i] (i < TBL) in the full 64-bit row alignment of lane_dc, zero outside the band */
static int lane_dc_band32(const uint8_t *t, const uint8_t *q, int TBL, uint64_t *V1, uint64_t *V0)
{
    uint64_t Tlo = 0, Thi = 0, Plo = 0, Phi = 0;
    for (int k = 0; k < 64; k++) {
        Tlo |= (uint64_t)(t[k] & 1) << k; Thi |= (uint64_t)(t[k] >> 1) << k;
        Plo |= (uint64_t)(q[k] & 1) << k; Phi |= (uint64_t)(q[k] >> 1) << k;
    }
    const uint64_t Rlo = lp_brev64(Plo), Rhi = lp_brev64(Phi);
    /* band of column c (c = 64 is the boundary column): full-word bits s_c .. s_c + 31, s_c = 63 - c - BAND32_DOWN;
     * bits below bit 0 (rows > 63) are neutral — Eq = 1, Pv = Mv = 0, like the bits below a short pattern */
    uint32_t pv, mv = 0;
    {   /* boundary column 64: rows 49 .. 63 exist (+1 each), rows 64 .. 80 do not */
        const int s = 63 - 64 - BAND32_DOWN;                        /* -17 */
        pv = (uint32_t)(~0ull << (-s));                             /* full word of ones, moved up by 17 */
    }
    int free_diagonals = 0;
    for (int i = 63; i >= 0; i--) {
        const uint64_t sl = 0ull - ((Tlo >> i) & 1), sh = 0ull - ((Thi >> i) & 1);
        const uint64_t Eq64 = ~((Rlo ^ sl) | (Rhi ^ sh));
        const int s_old = 63 - (i + 1) - BAND32_DOWN;               /* the band of column i + 1: where pv / mv live */
        uint32_t eq;
        if (s_old >= 0) eq = (uint32_t)(Eq64 >> s_old);
        else eq = (uint32_t)((Eq64 << (-s_old)) | ((1ull << (-s_old)) - 1ull));     /* rows past 63: Eq = 1 */
        const uint32_t xv = eq | mv;
        const uint32_t xh = (((eq & pv) + pv) ^ pv) | eq;
        const uint32_t ph = mv | ~(xh | pv);
        const uint32_t mh = pv & xh;
        /* the main diagonal's step (i + 1, i + 1) -> (i, i): row i sits BAND32_UP + 1 rows below the old band's top row i + 1 - 15 ... */
        free_diagonals += (int)(((xh | mv) >> (31 - (BAND32_UP - 1))) & 1u);     /* row i of the old band: bit 31 - (i - (i + 1 - 15)) = 17 */
        const uint32_t xvs = xv >> 1;
        const uint32_t pvn = mh | ~(xvs | ph);
        const uint32_t mvn = ph & xvs;
        if (i < TBL) {
            /* table words in the NEW band's coordinates (one bit down from the old one's): Ph, Xh of row j sit one bit lower there */
            const uint32_t v1 = pvn | (ph >> 1) , v0 = pvn | ~((ph | xh) >> 1);
            const int s_new = 63 - i - BAND32_DOWN;                  /* >= 17 for i <= 30 */
            V1[i] = (uint64_t)v1 << s_new;
            V0[i] = (uint64_t)(v0 & ~0u) << s_new;
            /* rows below the band (bits under s_new) read "match", rows above it were shifted out */
        }
        pv = pvn; mv = mvn;
    }
    return 64 - free_diagonals;
}

int lane_align_codes_band32(const uint8_t *text, size_t text_len, const uint8_t *read, size_t read_len, int W, int O,
                            go_run *runs, size_t cap, size_t *n_runs, long long *edit_distance, lane_stats_band32 *ls)
{
    if (W != 64 || O < 33 || O >= W) return GO_ERR_PARAMS;
    run_sink out = { runs, cap, 0, 0 };
    size_t ti = 0, ri = 0; long long total = 0;
    const int TBL = W - O;
    uint64_t V1[64], V0[64], B1[64], B0[64];
    lane_stats dummy = {0, 0, 0};
    while (ri < read_len) {
        size_t n = text_len - ti < (size_t)W ? text_len - ti : (size_t)W;
        size_t m = read_len - ri < (size_t)W ? read_len - ri : (size_t)W;
        size_t tu, pu;
        int banded = 0;
        ls->windows++;
        if (n == 64 && m == 64) {
            const int d_band = lane_dc_band32(text + ti, read + ri, TBL, B1, B0);
            ls->hist[d_band < 0 ? 0 : (d_band > 65 ? 65 : d_band)]++;
            if (d_band <= BAND32_SAFE) banded = 1; else ls->escapes++;
        }
        if (banded) {
            /* (the proto also runs the full table and counts the safe windows on which the two walks differ: must stay 0) */
            run_sink o1 = { runs + out.n, cap > out.n ? cap - out.n : 0, 0, 0 };
            size_t tu2, pu2;
            lane_dc(text + ti, (int)n, read + ri, (int)m, TBL, V1, V0, &dummy);
            go_run tmp[80]; run_sink o2 = { tmp, 80, 0, 0 };
            const int e2 = lane_tb(V1, V0, (int)m, TBL, &tu2, &pu2, &o2, &dummy);
            const int e1 = lane_tb(B1, B0, (int)m, TBL, &tu, &pu, &o1, &dummy);
            int same = e1 == e2 && tu == tu2 && pu == pu2 && o1.n == o2.n;
            for (size_t k = 0; same && k < o1.n && k < o1.cap; k++) same = runs[out.n + k].count == tmp[k].count && runs[out.n + k].op == tmp[k].op;
            if (!same) ls->mismatching_safe_windows++;
            out.n += o1.n; out.overflow |= o1.overflow;
            total += e1;
            ls->banded++;
        } else {
            lane_dc(text + ti, (int)n, read + ri, (int)m, TBL, V1, V0, &dummy);
            total += lane_tb(V1, V0, (int)m, TBL, &tu, &pu, &out, &dummy);
        }
        ti += tu; ri += pu;
    }
    *n_runs = out.n; *edit_distance = total;
    return out.overflow ? GO_ERR_CAPACITY : GO_OK;
}
