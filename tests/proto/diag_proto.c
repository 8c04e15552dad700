/*
 * diag_proto.c — CPU prototype (not shipped, not part of the oracle) of the diagonal-major GenASM
 * window: entries are kept per diagonal (delta = j - i) with one bit per text position, the in-row
 * match chain becomes a carry chain solved with one 64-bit addition, and the traceback walks a
 * diagonal with count-leading-zeros.  Used to validate the arithmetic of the HIP kernel's diagonal
 * path against the oracle before porting it.  Includes the oracle's source for its static window
 * functions (fallback for windows the diagonal path does not cover).
 */
#include "../../oracle/genasm_oracle.c"

#define NDIAG 32
#define DOFF 16

typedef struct proto_stats { uint64_t diag_windows, fallback_windows, rows; } proto_stats;

static uint64_t brev64(uint64_t v)
{
    uint64_t r = 0;
    for (int k = 0; k < 64; k++) if ((v >> k) & 1) r |= 1ull << (63 - k);
    return r;
}

/* returns window distance or -1 (not covered); fills H[d][x] (high dwords) for d < dist */
static int diag_dc(const uint8_t *t, const uint8_t *q, int m, int max_rows, uint32_t H[][NDIAG], proto_stats *ps)
{
    uint64_t Tlo = 0, Thi = 0, Plo = 0, Phi = 0;
    for (int k = 0; k < 64; k++) {
        Tlo |= (uint64_t)(t[k] & 1) << k; Thi |= (uint64_t)(t[k] >> 1) << k;
        /* beyond the pattern's end the planes hold whatever follows in memory: use a fixed non-zero filler here */
        uint8_t qc = k < m ? q[k] : (uint8_t)((k * 7 + 3) & 3);
        Plo |= (uint64_t)(qc & 1) << k; Phi |= (uint64_t)(qc >> 1) << k;
    }
    Tlo = brev64(Tlo); Thi = brev64(Thi); Plo = brev64(Plo); Phi = brev64(Phi);   /* bit p = char 63-p */
    uint64_t mt[NDIAG], A[NDIAG], S[NDIAG], An[NDIAG], Sn[NDIAG];
    int thr[NDIAG];
    for (int x = 0; x < NDIAG; x++) {
        int dl = x - DOFF;
        int e = dl + (64 - m);                      /* a pattern of m < 64 characters shifts every boundary role by 64-m diagonals */
        uint64_t slo = dl >= 0 ? Plo << dl : Plo >> -dl, shi = dl >= 0 ? Phi << dl : Phi >> -dl;
        uint64_t lowm = e <= 0 ? ~0ull : (e >= 64 ? 0ull : ~0ull << e);     /* j < m  */
        uint64_t highm = dl >= 0 ? ~0ull : ~0ull >> -dl;                      /* j >= 0 */
        mt[x] = ~((Tlo ^ slo) | (Thi ^ shi)) & lowm & highm;
        thr[x] = e <= 0 ? -e : 1 << 20;
        A[x] = 0;
        S[x] = (e >= 1 && e <= 64) ? 1ull << (e - 1) : 0;      /* row "-1": only the forced pattern-end cells */
    }
    for (int d = 0; d <= max_rows; d++) {
        for (int x = 0; x < NDIAG; x++) {
            uint64_t y = S[x] | (x + 1 < NDIAG ? A[x + 1] : 0) | (x > 0 ? S[x - 1] : 0);
            uint64_t u = d >= thr[x];
            uint64_t U = y | mt[x], V = y;
            uint64_t sum = U + V + u;
            uint64_t a = y | (mt[x] & (sum ^ U ^ V));
            An[x] = a;
            Sn[x] = (a << 1) | u;
        }
        ps->rows++;
        memcpy(A, An, sizeof A); memcpy(S, Sn, sizeof S);
        if (A[DOFF] >> 63) return d;
        if (d < max_rows) for (int x = 0; x < NDIAG; x++) H[d][x] = (uint32_t)(A[x] >> 32);
    }
    return -1;
}

static int clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }

/* traceback over the diagonal rows; TBL = W - O <= 31 */
static int diag_tb(uint32_t H[][NDIAG], int dist, int TBL, int m, size_t *tu, size_t *pu, run_sink *out)
{
    int i = 0, j = 0, d = dist;
    char cur = 0; unsigned cur_len = 0;
#define EMIT(op_, n_) do { if ((n_)) { if (cur == (op_)) cur_len += (n_); else { if (cur_len) sink_push(out, cur, cur_len); cur = (op_); cur_len = (n_); } } } while (0)
    const int jlim = m < TBL ? m : TBL;            /* j < m && j < W-O, genasm_cpu.cpp:307-310 */
    while (i < TBL && j < jlim) {
        int x = j - i + DOFF;
        uint32_t E = 0, hi = 0, hd = 0, hs = 0;
        if (d > 0) {
            hi = H[d - 1][x + 1];            /* ins: a(i, j+1)   */
            hd = H[d - 1][x - 1] << 1;       /* del: a(i+1, j)   */
            hs = H[d - 1][x] << 1;           /* sub: a(i+1, j+1) */
            E = (hi | hd | hs) & (0xffffffffu >> i);
        }
        int i2 = clz32(E);
        int run = i2 - i;
        int lim = (TBL - i) < (jlim - j) ? (TBL - i) : (jlim - j);
        if (run >= lim) { EMIT('=', (unsigned)lim); i += lim; j += lim; break; }
        EMIT('=', (unsigned)run); i += run; j += run;
        uint32_t bit = 0x80000000u >> i;
        if (hi & bit) { EMIT('I', 1u); j++; }
        else if (hd & bit) { EMIT('D', 1u); i++; }
        else { EMIT('X', 1u); i++; j++; }
        d--;
    }
    if (cur_len) sink_push(out, cur, cur_len);
    *tu = (size_t)i; *pu = (size_t)j;
    return dist - d;
}

int proto_align_codes(const uint8_t *text, size_t text_len, const uint8_t *read, size_t read_len, int O, int max_rows,
                      go_run *runs, size_t cap, size_t *n_runs, long long *edit_distance, proto_stats *ps)
{
    const int W = 64;
    static _Thread_local go_scratch_64 scratch;
    static _Thread_local uint32_t H[64][NDIAG];
    run_sink out = { runs, cap, 0, 0 };
    size_t ti = 0, ri = 0; long long total = 0;
    const size_t limit = (size_t)(W - O);
    while (ri < read_len) {
        size_t n = text_len - ti < (size_t)W ? text_len - ti : (size_t)W;
        size_t m = read_len - ri < (size_t)W ? read_len - ri : (size_t)W;
        size_t tu, pu; int dist = -1;
        if (n == 64 && limit <= 31) dist = diag_dc(text + ti, read + ri, (int)m, max_rows, H, ps);
        if (dist >= 0) {
            ps->diag_windows++;
            total += diag_tb(H, dist, (int)limit, (int)m, &tu, &pu, &out);
        } else {
            ps->fallback_windows++;
            dist = distance_sweep_64(text + ti, n, read + ri, m, W, &scratch, NULL);
            total += traceback_64(&scratch, n, m, dist, limit, &tu, &pu, &out, NULL);
        }
        ti += tu; ri += pu;
    }
    *n_runs = out.n; *edit_distance = total;
    return out.overflow ? GO_ERR_CAPACITY : GO_OK;
}
