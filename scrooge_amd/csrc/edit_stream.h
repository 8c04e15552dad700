// edit_stream.h — the transfer format for CIGARs ("edit stream") and its window replay.
//
// A pair's runs (scrg_run, flushed per window, genasm_cpu.cpp:304-305, 400-403) hold two things: the sequence of
// alignment operations and the places where a window ended.  The second is a pure function of the first: a
// window's traceback runs `while (j < m && i < W-O && j < W-O)` (genasm_cpu.cpp:307-310) over text index i and
// pattern index j, m = min(W, read length left) (:415-417), and the next window starts where it stopped.  So only
// the operations travel, one byte per EDIT:
//
//     byte = op << 6 | len        op 1 'X', 2 'I', 3 'D': `len` matches, then that edit
//                                 op 0              : `len + 1` matches and no edit (a match stretch > 63)
//
// in alignment order; the matches after the last edit are implied by the read length.  Canonical form (what
// the encoders emit): a stretch of P matches before an edit is P >> 6 bytes 0x3F followed by the edit byte
// with len = P & 63.  A 10 kb read at 10 % error is ~1.0 KB instead of ~2140 runs x 2 bytes.
//
// replay_edit_stream() restores the runs bit for bit, window breaks included (host: scrg_edit_stream_to_runs);
// decode_lane_step() below is the same replay as a state machine, the form decode_edits_kernel runs per lane.
#pragma once

#include <stdint.h>

#include "genasm_kernels.h"

namespace scrg {

constexpr uint32_t EDIT_OP_NONE = 0, EDIT_OP_X = 1, EDIT_OP_I = 2, EDIT_OP_D = 3;

SCRG_HD inline uint32_t edit_code_of_char(uint32_t op)
{
    // '=' 0x3D, 'X' 0x58, 'I' 0x49, 'D' 0x44: bits (4, 2) = 3, 2, 0, 1 -> code 0, 1, 2, 3
    const uint32_t idx = ((op >> 3) & 2u) | ((op >> 2) & 1u);
    return (0x1Eu >> (2u * idx)) & 3u;
}
SCRG_HD inline uint32_t edit_char_of_code(uint32_t code) { return (0x4449583Du >> (8u * code)) & 0xffu; }

// Walks one pair's stream and calls sink(op_char, count) for every run, in order.  Returns the number of runs,
// or ~0ull if the stream does not describe an alignment of a read of this length (bytes left over, or the
// read overrun).
template <typename Sink>
SCRG_HD inline uint64_t replay_edit_stream(const uint8_t* s, uint64_t n_bytes, uint64_t read_len, uint32_t W, uint32_t O,
                                           Sink&& sink)
{
    const uint64_t limit = W - O;
    uint64_t ri = 0, k = 0, n_runs = 0;
    uint64_t pend_m = 0;               // matches still to place
    uint32_t pend_e = EDIT_OP_NONE;    // the edit after them
    bool tail = false;                 // stream used up: the rest of the read matches
    while (ri < read_len) {
        const uint64_t left = read_len - ri;
        const uint64_t m = left < W ? left : W;
        uint64_t i = 0, j = 0;
        uint32_t cur = 0;
        uint64_t cur_len = 0;
        auto push = [&](uint32_t op, uint64_t t) {
            if (op == cur) {
                cur_len += t;
            } else {
                if (cur_len) { sink(cur, cur_len); n_runs++; }
                cur = op;
                cur_len = t;
            }
        };
        while (j < m && i < limit && j < limit) {
            if (pend_m == 0 && pend_e == EDIT_OP_NONE) {
                if (k < n_bytes) {
                    const uint32_t b = s[k++];
                    pend_e = b >> 6;
                    pend_m = (b & 63u) + (pend_e == EDIT_OP_NONE ? 1u : 0u);
                } else {
                    if (tail) return ~0ull;
                    tail = true;
                    pend_m = read_len - ri - j;
                }
            }
            if (pend_m) {
                uint64_t room = m - j;
                if (limit - i < room) room = limit - i;
                if (limit - j < room) room = limit - j;
                const uint64_t t = pend_m < room ? pend_m : room;
                push('=', t);
                i += t; j += t; pend_m -= t;
            } else {
                const uint32_t e = pend_e;
                pend_e = EDIT_OP_NONE;
                push(edit_char_of_code(e), 1);
                if (e != EDIT_OP_D) j++;
                if (e != EDIT_OP_I) i++;
            }
        }
        if (cur_len) { sink(cur, cur_len); n_runs++; }
        ri += j;
        if (i == 0 && j == 0) return ~0ull;      // cannot happen for W > O; guards the loop
    }
    if (k != n_bytes || pend_m != 0 || pend_e != EDIT_OP_NONE) return ~0ull;
    return n_runs;
}

// ---------------------------------------------------------------------------------------------------------------
// The same replay as a branch-free STATE MACHINE, one step at a time: what decode_edits_kernel runs in every lane (one
// pair per lane, 64 pairs per wavefront), compiled for the host too (scrg_edit_stream_to_runs_lane) so that the CPU
// tests can hold this very code against replay_edit_stream() above on every golden fixture.  Differences in form, not
// in result:
//   * one step places the matches of the pending stream byte that fit the window, then its edit if the window is not
//     full, then closes the window if it is — instead of one event per loop trip;
//   * bytes with op 0 ("len + 1 matches, no edit") are accumulated into the match count of the next edit byte before
//     anything is placed, so a match run is never continued by a later step: the only merge left is an edit joining
//     the run of the same edit directly before it (no match, no window break in between);
//   * runs are handed to `put(k, word)` speculatively: slot k holds run k (count | op << 8) and may be rewritten until
//     run k + 1 starts (both slots a step might touch are always written: put() must tolerate k = n, a free slot);
//   * conditions are 0 / ~0 masks and every update is arithmetic on them: the 64 lanes of a wavefront are at 64
//     different places of their streams, so a branch would be taken by some lane every time — straight-line code issues
//     at the VALU rate (the first version of this step, with nested conditionals, ran at 24 cycles per instruction).
// A step consumes at most one stream byte: the caller hands in the byte at `pos` and is told whether it was taken.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t es_neg_mask(uint32_t x)              // ~0 iff (int32)x < 0
{
    uint32_t r;
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t es_sel(uint32_t a, uint32_t b, uint32_t m)   // bits of a where m is set, else b
{
    return __builtin_amdgcn_bitop3_b32(a, b, m, 0xE4);           // (a & m) | (b & ~m); truth table bit index = a * 4 + b * 2 + m (genasm_device.h: bitop3_table)
}
#else
inline uint32_t es_neg_mask(uint32_t x) { return (uint32_t)((int32_t)x >> 31); }
inline uint32_t es_sel(uint32_t a, uint32_t b, uint32_t m) { return (a & m) | (b & ~m); }
#endif
SCRG_HD inline uint32_t es_nz_mask(uint32_t x) { return es_neg_mask(0u - x); }      // ~0 iff x != 0, for x < 2^31
SCRG_HD inline uint32_t es_min(uint32_t a, uint32_t b) { return a < b ? a : b; }

struct DecodeLane {
    uint32_t L;                 // W - O: a window consumes at most L text and L read characters (genasm_cpu.cpp:309-310)
    uint32_t pos, end;          // next stream byte, end of the stream
    uint32_t m, e;              // matches pending, the edit after them (EDIT_OP_NONE: no edit byte has arrived yet)
    uint32_t tailM;             // mask: stream used up, the rest of the read matches
    uint32_t R;                 // read characters not placed yet
    uint32_t ri, rj;            // window: text / read characters it can still take (rj starts at min(R, L))
    uint32_t cur, prev_e;       // run n - 1 as a word; the edit it consists of if the next edit may join it, else 0
    uint32_t n;                 // runs started
    uint32_t aliveM;            // mask: the read is not finished
};

SCRG_HD inline void decode_lane_init(DecodeLane& s, uint32_t W, uint32_t O, uint32_t pos, uint32_t end, uint32_t read_len)
{
    s.L = W - O;
    s.pos = pos;
    s.end = end;
    s.m = s.e = s.tailM = 0;
    s.R = read_len;
    s.ri = s.L;
    s.rj = read_len < s.L ? read_len : s.L;
    s.cur = s.prev_e = s.n = 0;
    s.aliveM = read_len != 0 ? ~0u : 0u;
}

// true when the pair is finished: every byte used, nothing pending (the counterpart of replay_edit_stream's last line).
// (A stream that overruns its read leaves matches or an edit pending, or bytes unused; one that ends early is completed by
// the tail rule below — the read's remaining characters match — so a lane never waits for input that does not come.)
SCRG_HD inline bool decode_lane_clean(const DecodeLane& s) { return s.pos == s.end && s.m == 0 && s.e == EDIT_OP_NONE; }

struct DecodeNoTake {
    SCRG_HD void operator()(uint32_t) const {}
};

// on_take(takeM) is called as soon as it is known whether the byte at pos is consumed (takeM = ~0) or not (0): the GPU
// decoder asks for the byte of the NEXT step there, a whole step before it is looked at.
template <typename Put, typename OnTake = DecodeNoTake>
SCRG_HD inline uint32_t decode_lane_step(DecodeLane& s, const uint32_t bn0, Put&& put, OnTake&& on_take = OnTake())      // -> 1 if the byte at pos was consumed
{
    constexpr uint32_t EQW = (uint32_t)'=' << 8;
    // ---- the next stream byte, if nothing is pending (an edit byte completes the pending item, an op-0 byte only adds matches)
    const uint32_t fetchM = ~(es_nz_mask(s.e) | s.tailM) & s.aliveM;
    const uint32_t hasM = es_neg_mask(s.pos - s.end);                // pos < end
    const uint32_t takeM = fetchM & hasM;
    on_take(takeM);
    const uint32_t e_new = bn0 >> 6;
    s.m += ((bn0 & 63u) + ((bn0 - 64u) >> 31)) & takeM;              // len, + 1 for op 0
    s.e = es_sel(e_new, s.e, takeM);
    s.pos -= takeM;
    // the stream is used up (once per pair): the matches after the last edit are implied by the read length (op-0 bytes
    // pending are part of them; more of them than the read has left means the stream overruns the read: m then stays
    // non-zero to the end and the pair is reported).  From here on m >= R, so the lane ends exactly when its read does.
    const uint32_t tail_now = fetchM & ~hasM;
    s.m = es_sel(s.m < s.R ? s.R : s.m, s.m, tail_now);
    s.tailM |= tail_now;
    const uint32_t readyM = es_nz_mask(s.e) | s.tailM;
    // ---- the matches that fit the window: always a new run
    const uint32_t t = es_min(es_min(s.m, s.ri), s.rj) & readyM;
    s.m -= t;
    s.ri -= t;
    s.rj -= t;
    s.R -= t;
    const uint32_t tnzM = es_nz_mask(t);
    const uint32_t wm = EQW | t;
    put(s.n, wm);                                                    // (t == 0: a free slot, rewritten by the next run)
    s.cur = es_sel(wm, s.cur, tnzM);
    s.n -= tnzM;
    s.prev_e &= ~tnzM;
    // ---- the edit, if all its matches are placed and the window is not full
    const uint32_t doM = es_nz_mask(es_min(es_min(s.e, s.ri), s.rj)) & ~es_nz_mask(s.m);
    const uint32_t sameM = ~es_nz_mask(s.e ^ s.prev_e);
    const uint32_t mergeM = doM & sameM, newM = doM & ~sameM;
#if defined(__HIP_DEVICE_COMPILE__)
    // byte 1 = byte e of "\0XID", byte 0 = 1: one v_perm_b32 (selector bytes: 0x0C = zero, 4 + e = byte e of the first operand, 0 = byte 0 of the second)
    const uint32_t opw = __builtin_amdgcn_perm(0x44495800u, 1u, 0x0C0C0400u + (s.e << 8));
#else
    const uint32_t opw = (((0x44495800u >> ((s.e << 3) & 31u)) & 0xffu) << 8) | 1u;        // 'X', 'I', 'D' for e = 1, 2, 3 (bit field, shift-or)
#endif
    s.cur = es_sel(s.cur + 1u, es_sel(opw, s.cur, newM), mergeM);
    put(s.n + mergeM, s.cur);                                        // run n - 1 grows, or run n starts (or nothing changes)
    s.n -= newM;
    s.prev_e = es_sel(s.e, s.prev_e, doM);
    s.ri -= s.e & doM & 1u;                                          // X and D consume a text character,
    const uint32_t jstep = (6u >> s.e) & doM & 1u;                   // X and I a read character
    s.rj -= jstep;
    s.R -= jstep;
    s.e &= ~doM;
    // ---- the window is full (genasm_cpu.cpp:307-310): the next one starts where it stopped, its run is flushed (:400-403)
    const uint32_t endM = ~es_nz_mask(es_min(s.ri, s.rj)) & s.aliveM;
    s.prev_e &= ~endM;
    s.ri = es_sel(s.L, s.ri, endM);
    s.rj = es_sel(es_min(s.R, s.L), s.rj, endM);
    s.aliveM &= es_nz_mask(s.R);                                     // (R = 0 ends the window too: rj <= R)
    return takeM & 1u;
}

hipError_t launch_encode_edits(uint64_t n_pairs, const scrg_pair_desc* d_pairs, const uint16_t* d_runs,
                               const uint32_t* d_n_runs, uint8_t* d_stream, uint64_t stream_cap, uint64_t* d_off,
                               uint32_t* d_len, uint64_t* d_total, hipStream_t s);
hipError_t launch_decode_edits(uint64_t n_pairs, uint32_t W, uint32_t O, const uint8_t* d_stream, uint64_t stream_bytes,
                               const uint64_t* d_off, const uint32_t* d_len, const uint64_t* d_read_len,
                               uint64_t read_len_stride, const uint64_t* d_dense_off, uint16_t* d_dense, uint64_t dense_cap,
                               uint32_t* d_n_runs, uint32_t* d_bad, void* sort_ws, size_t sort_temp_bytes, hipStream_t s);
size_t decode_sort_temp_bytes(uint64_t n_pairs);

}  // namespace scrg
