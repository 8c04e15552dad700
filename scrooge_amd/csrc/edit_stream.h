// edit_stream.h — the transfer format for CIGARs ("edit stream"), version 2: window ends travel with the edits.
//
// A pair's runs (scrg_run, flushed per window, genasm_cpu.cpp:304-305, 400-403) hold the sequence of alignment
// operations and the places where a window ended.  One byte per EDIT and one per WINDOW carries both:
//
//     byte = op << 6 | len        op 1 'X', 2 'I', 3 'D' : `len` matches, then that edit
//                                 op 0, len <= 62         : `len` matches, then the window ENDS (its run in progress is
//                                                           flushed, :400-403; the next byte belongs to the next window)
//                                 op 0, len == 63 (0x3F)  : 63 matches and nothing else (only W-O > 63 has such stretches)
//
// in alignment order; every window of the pair ends with its END byte, the last one too.  Canonical form (what the
// encoders emit): P matches before an edit or a window end are P / 63 bytes 0x3F followed by the byte with len = P % 63.
// A 10 kb read at 10 % error and W-O = 31 is ~1.3 KB (968 edits + 323 windows) instead of ~2140 runs x 2 bytes.
//
// (Version 1, rounds 3-5, sent the edits only — the window ends are a pure function of the operations, :307-310 — and the
// receiver REPLAYED the window loop: ~90 VALU instructions per byte and per window in every lane of the decoder, 0.31 ms
// per 100 k pairs.  With the ends on the wire a byte is decoded by itself: what it adds to the run list depends on the
// byte before it only.)
//
// replay_edit_stream() restores the runs bit for bit (host: scrg_edit_stream_to_runs) and, given W-O, also checks that
// every window ends where the reference's loop ends it; decode_lane_step() is the per-byte state machine
// decode_edits_kernel runs in every lane; encode_runs() is the way there from a run list (it replays the window loop:
// the run list does not say where a window ended when the runs on both sides differ).
#pragma once

#include <stdint.h>

#include "genasm_kernels.h"

namespace scrg {

constexpr uint32_t EDIT_OP_NONE = 0, EDIT_OP_X = 1, EDIT_OP_I = 2, EDIT_OP_D = 3;
constexpr uint32_t EDIT_MORE = 0x3F;             // 63 matches, nothing else
constexpr uint32_t EDIT_MORE_MATCHES = 63;

SCRG_HD inline uint32_t edit_code_of_char(uint32_t op)
{
    // '=' 0x3D, 'X' 0x58, 'I' 0x49, 'D' 0x44: bits (4, 2) = 3, 2, 0, 1 -> code 0, 1, 2, 3
    const uint32_t idx = ((op >> 3) & 2u) | ((op >> 2) & 1u);
    return (0x1Eu >> (2u * idx)) & 3u;
}
SCRG_HD inline uint32_t edit_char_of_code(uint32_t code) { return (0x4449583Du >> (8u * code)) & 0xffu; }

// Walks one pair's stream and calls sink(op_char, count) for every run, in order.  Returns the number of runs, or
// ~0ull if the stream is not the alignment of a read of this length: it does not end with a window end, the characters
// it places are not read_len, a run would be longer than 255 — and, with L = W-O > 0, a window does not end exactly
// where the reference's loop `while (j < m && i < W-O && j < W-O)` (genasm_cpu.cpp:307-310) ends it.  L = 0: the
// window geometry is not looked at (what the device decoder checks).
template <typename Sink>
SCRG_HD inline uint64_t replay_edit_stream(const uint8_t* s, uint64_t n_bytes, uint64_t read_len, uint32_t L, Sink&& sink)
{
    uint64_t n_runs = 0, placed = 0, pend = 0;
    uint64_t i = 0, j = 0;             // text / read characters of the window so far
    uint32_t cur = 0;
    uint64_t cur_len = 0;
    bool ended = true;                 // the byte before was a window end (or there was none)
    bool bad = false;
    auto flush = [&]() {
        if (cur_len) { sink(cur, cur_len); n_runs++; }
        cur = 0;
        cur_len = 0;
    };
    auto matches = [&](uint64_t t) {
        if (!t) return;
        flush();
        cur = '=';
        cur_len = t;
        if (t > 255) bad = true;
        if (L && (i + t > L || j + t > L || placed >= read_len)) bad = true;
        i += t; j += t; placed += t;
    };
    for (uint64_t k = 0; k < n_bytes; k++) {
        const uint32_t b = s[k], op = b >> 6, len = b & 63u;
        ended = false;
        if (b == EDIT_MORE) { pend += EDIT_MORE_MATCHES; continue; }
        matches(pend + len);
        pend = 0;
        if (op == EDIT_OP_NONE) {
            // the window ends: with L given, it must be full (:309-310) or the read used up (:308, j == m)
            if (L && !(i == L || j == L || placed == read_len)) bad = true;
            if (i == 0 && j == 0) bad = bad || L != 0;          // (an empty window: padding, never emitted by an encoder)
            flush();
            i = j = 0;
            ended = true;
            continue;
        }
        // an edit: the window must still be open on both sides
        if (L && (i >= L || j >= L || placed >= read_len)) bad = true;      // (:308: the loop ends with the read, no deletion follows it)
        const uint32_t c = edit_char_of_code(op);
        if (c == cur) {
            if (++cur_len > 255) bad = true;
        } else {
            flush();
            cur = c;
            cur_len = 1;
        }
        if (op != EDIT_OP_D) { j++; placed++; }
        if (op != EDIT_OP_I) i++;
    }
    if (bad || !ended || pend != 0 || placed != read_len) return ~0ull;
    return n_runs;
}

// The way there: the stream of a run list.  get(r) -> run r as count | op_char << 8; put(byte).  The window loop is
// replayed (L = W-O: a window ends as soon as it has taken L text or L read characters, or when the runs are used up),
// so a run may be cut — the align kernels never hand over such a run, a caller's own CIGAR may.  Returns the number of
// bytes, ~0ull for an operation that is not one of = X I D.
template <typename Get, typename Put>
SCRG_HD inline uint64_t encode_runs(uint64_t n_runs, uint32_t L, Get&& get, Put&& put)
{
    uint64_t bytes = 0;
    uint32_t i = 0, j = 0, pend = 0;
    bool open = false;
    auto out = [&](uint32_t code) {
        for (; pend >= EDIT_MORE_MATCHES; pend -= EDIT_MORE_MATCHES) { put((uint8_t)EDIT_MORE); bytes++; }
        put((uint8_t)(code << 6 | pend));
        bytes++;
        pend = 0;
    };
    for (uint64_t r = 0; r < n_runs; r++) {
        const uint32_t run = get(r), op = run >> 8;
        uint32_t cnt = run & 0xffu;
        if (op != '=' && op != 'X' && op != 'I' && op != 'D') return ~0ull;
        while (cnt) {
            if (op == '=') {
                uint32_t t = cnt;
                if (L - i < t) t = L - i;
                if (L - j < t) t = L - j;
                pend += t; i += t; j += t; cnt -= t;
            } else {
                out(edit_code_of_char(op));
                if (op != 'D') j++;
                if (op != 'I') i++;
                cnt--;
            }
            open = true;
            if (i == L || j == L) {
                out(EDIT_OP_NONE);
                i = j = 0;
                open = false;
            }
        }
    }
    if (open) out(EDIT_OP_NONE);
    return bytes;
}

// ---------------------------------------------------------------------------------------------------------------
// The decoder as a branch-free STATE MACHINE, one stream byte per step: what decode_edits_kernel runs in every lane (one
// pair per lane, 64 pairs per wavefront), compiled for the host too (scrg_edit_stream_to_runs_lane) so that the CPU
// tests can hold this very code against replay_edit_stream() on every golden fixture.
//   * runs are handed to `put(k, word)` speculatively: slot k holds run k (count | op << 8) and may be rewritten until
//     run k + 1 starts (both slots a step might touch are always written: put() must tolerate k = n, a free slot);
//   * conditions are 0 / ~0 masks and every update is arithmetic on them: the 64 lanes of a wavefront are at 64
//     different places of their streams, so a branch would be taken by some lane every time;
//   * a zero byte (a window end after no matches) changes nothing when the byte before it was a window end as well:
//     the decoder pads a lane's stream with zeros in front and behind.
//   * the state is kept in the form the GPU wants it: q is the BYTE offset of the next free run slot (2 x the runs started,
//     plus wherever the caller's ring starts); conditions are 0 / ~0 masks made by arithmetic (a sign bit smeared by
//     v_ashrrev_i32) and used through v_bitop3_b32 — NOT compare + v_cndmask_b32: a conditional move that reads VCC issues
//     at a seventh of the rate of the logic and add instructions on gfx950 (profiles/r03_valu_issue_rates.txt: 12.3 against
//     1.7 cycles at four wavefronts per SIMD).  38 instructions per byte, 32 of them full rate.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t es_neg_mask(uint32_t x)              // ~0 iff (int32)x < 0
{
    uint32_t r;
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(r) : "v"(x));
    return r;
}
// truth tables: bit index = a * 4 + b * 2 + c (genasm_device.h: bitop3_table)
__device__ __forceinline__ uint32_t es_sel(uint32_t a, uint32_t b, uint32_t m) { return __builtin_amdgcn_bitop3_b32(a, b, m, 0xE4); }        // (a & m) | (b & ~m)
__device__ __forceinline__ uint32_t es_andn(uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(a, b, b, 0x30); }                    // a & ~b
__device__ __forceinline__ uint32_t es_and_or(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xEA); }      // (a & b) | c
__device__ __forceinline__ uint32_t es_a_nb_c(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x20); }      // a & ~b & c
__device__ __forceinline__ uint32_t es_or_xor(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xF6); }      // a | (b ^ c)
__device__ __forceinline__ uint32_t es_opw(uint32_t e)
{
    // byte 1 = byte e of "\0XID", byte 0 = 1: one v_perm_b32 (selector bytes: 0x0C = zero, 4 + e = byte e of the first operand, 0 = byte 0 of the second)
    return __builtin_amdgcn_perm(0x44495800u, 1u, 0x0C0C0400u + (e << 8));
}
__device__ __forceinline__ uint32_t es_bit(uint32_t word, uint32_t at) { return __builtin_amdgcn_ubfe(word, at, 1); }
#else
SCRG_HD inline uint32_t es_neg_mask(uint32_t x) { return (uint32_t)((int32_t)x >> 31); }
SCRG_HD inline uint32_t es_sel(uint32_t a, uint32_t b, uint32_t m) { return (a & m) | (b & ~m); }
SCRG_HD inline uint32_t es_andn(uint32_t a, uint32_t b) { return a & ~b; }
SCRG_HD inline uint32_t es_and_or(uint32_t a, uint32_t b, uint32_t c) { return (a & b) | c; }
SCRG_HD inline uint32_t es_a_nb_c(uint32_t a, uint32_t b, uint32_t c) { return a & ~b & c; }
SCRG_HD inline uint32_t es_or_xor(uint32_t a, uint32_t b, uint32_t c) { return a | (b ^ c); }
SCRG_HD inline uint32_t es_opw(uint32_t e) { return (((0x44495800u >> ((e << 3) & 31u)) & 0xffu) << 8) | 1u; }        // 'X', 'I', 'D' for e = 1, 2, 3
SCRG_HD inline uint32_t es_bit(uint32_t word, uint32_t at) { return (word >> at) & 1u; }
#endif

constexpr uint32_t DEC_PREV_NONE = 0x40000000u;          // "no edit run to join": no key (below) has this bit (and key ^ prev stays below 2^31: the comparison looks at a sign)

struct DecodeLane {
    uint32_t pend;              // matches of 0x3F bytes waiting for the byte that closes the stretch
    uint32_t cur, prev;         // the run started last as a word (count | op << 8); the edit it consists of if the next edit byte may join it, else DEC_PREV_NONE
    uint32_t q, q0;             // byte offset of the next free run slot; where the pair's first run went
    uint32_t placed;            // read characters placed so far
    uint32_t over;              // bits 8.. set: some run was longer than 255
};

SCRG_HD inline void decode_lane_init(DecodeLane& s, uint32_t q0)
{
    s.pend = s.cur = s.placed = s.over = 0;
    s.prev = DEC_PREV_NONE;
    s.q = s.q0 = q0;
}
SCRG_HD inline uint32_t decode_lane_runs(const DecodeLane& s) { return (s.q - s.q0) >> 1; }

// true when the bytes seen so far are a whole stream for a read of this length (last: the last byte of the stream, 0 for
// an empty one): the counterpart of replay_edit_stream's last line with L = 0
SCRG_HD inline bool decode_lane_clean(const DecodeLane& s, uint32_t last, uint32_t read_len)
{
    return s.pend == 0 && (last >> 6) == 0 && last != EDIT_MORE && s.placed == read_len && (s.over >> 8) == 0;
}

// Streams off a wire may be long enough to wrap the 32-bit count of placed characters (64 per byte, 2^30 bytes): a count that
// has passed 2^31 — no read is that long — marks the pair for good.  The device decoder calls this once per 16-byte block.
SCRG_HD inline void decode_lane_guard(DecodeLane& s) { s.over |= (s.placed >> 31) << 8; }

// put(at, word): the run whose slot is at byte offset `at`
template <typename Put>
SCRG_HD inline void decode_lane_step(DecodeLane& s, const uint32_t b, Put&& put)
{
    constexpr uint32_t EQW = (uint32_t)'=' << 8;
    const uint32_t e = b >> 6, len = b & 63u;
    const uint32_t moreM = es_neg_mask((b ^ EDIT_MORE) - 1u);        // b == 0x3F
    // ---- the matches in front of the edit / the window end: always a run of their own
    const uint32_t tot = s.pend + len;
    s.pend = tot & moreM;
    const uint32_t t = es_andn(tot, moreM);
    const uint32_t tnzM = es_neg_mask(0u - t);                       // (t < 2^31)
    put(s.q, EQW | t);                                               // (t == 0: a free slot, rewritten by the next run)
    const uint32_t q1 = s.q - tnzM - tnzM;                           // + 2 after a run of matches
    // ---- the edit: joins the run of the same edit directly before it (no match, no window end in between)
    const uint32_t key = (t << 2) | e;                               // == prev only for the same edit with no match in front (prev is 1, 2, 3 or DEC_PREV_NONE)
    const uint32_t sameM = es_neg_mask((key ^ s.prev) - 1u);
    s.cur = es_sel(s.cur + 1u, es_opw(e), sameM);                    // (no edit: nobody looks at it again — prev becomes none — and it goes to a free slot)
    put(q1 + sameM + sameM, s.cur);                                  // run n - 1 grows, or run n starts (or a free slot is written)
    const uint32_t editM = es_neg_mask(63u - b);                     // b >= 64
    s.q = q1 + es_a_nb_c(editM, sameM, 2u);                          // + 2 after an edit that started a run
    // a window end (e = 0) closes the run: e - 1 has bit 30 then, and only then; 0x3F leaves things as they are
    s.prev = es_sel(s.prev, es_and_or(e - 1u, DEC_PREV_NONE, e), moreM);
    s.placed = s.placed + len + es_bit(6u, e);                       // matches (0x3F: its 63), + the read character of X and I
    // (`tot`, not `t`: a stretch of 0x3F bytes that has grown past 255 is reported at once and for good — the sum it becomes
    // when the stretch closes can only be larger — so a stream long enough to wrap the 32-bit `pend` cannot pass as clean.
    // `placed` grows by at most 64 per byte: decode_lane_guard(), called at least every 2^24 steps, catches it above 2^31.)
    s.over |= tot;
    s.over = es_or_xor(s.over, s.cur - 1u, s.cur);                   // a count of 0 — 256 of the same edit in a row — borrows from the letter
#if defined(__HIP_DEVICE_COMPILE__)
    // (the two sums are kept up step by step: left alone, the compiler adds up the sixteen steps of an unrolled block at its
    // end, in a tree, and keeps every step's len, e and t in registers until then — 150 VGPRs)
    asm volatile("" : "+v"(s.placed), "+v"(s.over));
#endif
}

hipError_t launch_encode_edits(uint64_t n_pairs, uint32_t W, uint32_t O, const scrg_pair_desc* d_pairs, const uint16_t* d_runs,
                               const uint32_t* d_n_runs, uint8_t* d_stream, uint64_t stream_cap, uint64_t* d_off,
                               uint32_t* d_len, uint64_t* d_total, hipStream_t s);
hipError_t launch_decode_edits(uint64_t n_pairs, const uint8_t* d_stream, uint64_t stream_bytes,
                               const uint64_t* d_off, const uint32_t* d_len, const uint64_t* d_read_len,
                               uint64_t read_len_stride, const uint64_t* d_dense_off, uint16_t* d_dense, uint64_t dense_cap,
                               uint32_t* d_n_runs, uint32_t* d_bad, void* sort_ws, size_t sort_temp_bytes, hipStream_t s);
size_t decode_sort_temp_bytes(uint64_t n_pairs);
bool decode_by_wavefront(uint64_t n_pairs, uint64_t stream_bytes);     // which of the two decoders launch_decode_edits takes

}  // namespace scrg
