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
// The same replay as a STATE MACHINE, one step at a time: what decode_edits_kernel runs in every lane (one pair per
// lane), compiled for the host too (scrg_edit_stream_to_runs_lane) so that the CPU tests can hold it against
// replay_edit_stream() above on every golden fixture.  Differences in form, not in result:
//   * one step places the matches of the pending stream byte that fit the window, then its edit if the window is not
//     full, then closes the window if it is — instead of one event per loop trip;
//   * bytes with op 0 ("len + 1 matches, no edit") are accumulated into the match count of the next edit byte before
//     anything is placed, so a match run is never continued by a later step: the only merge left is an edit joining
//     the run of the same edit directly before it (no match, no window break in between);
//   * runs are handed to `put(k, word)` speculatively: slot k holds run k (count | op << 8) and may be rewritten until
//     run k + 1 starts; `n` counts the runs started.
// `peek()` returns the stream byte at `pos` (only called while pos < end), `advance()` moves on.
struct DecodeLane {
    uint32_t L;                 // W - O: a window consumes at most L text and L read characters (genasm_cpu.cpp:309-310)
    uint32_t pos, end;          // next stream byte, end of the stream
    uint32_t m, e;              // matches pending, the edit after them (EDIT_OP_NONE: none yet)
    uint32_t ready;             // the pending byte is complete (an edit byte has arrived, or the stream is used up)
    uint32_t tail;              // stream used up: the rest of the read matches
    uint32_t left;              // read characters from the start of the current window to the end of the read
    uint32_t jl, ri, rj;        // window: read limit min(left, L); text / read characters it can still take
    uint32_t cur, prev_e;       // run n - 1 as a word; the edit it consists of if the next edit may join it, else 0
    uint32_t n;                 // runs started
    uint32_t alive, bad;
};

SCRG_HD inline void decode_lane_init(DecodeLane& s, uint32_t W, uint32_t O, uint32_t pos, uint32_t end, uint32_t read_len)
{
    s.L = W - O;
    s.pos = pos;
    s.end = end;
    s.m = s.e = s.ready = s.tail = 0;
    s.left = read_len;
    s.jl = read_len < s.L ? read_len : s.L;
    s.ri = s.L;
    s.rj = s.jl;
    s.cur = s.prev_e = s.n = 0;
    s.alive = read_len != 0;
    s.bad = 0;
}

// true when the pair is finished: every byte used, nothing pending (the counterpart of replay_edit_stream's last line)
SCRG_HD inline bool decode_lane_clean(const DecodeLane& s) { return !s.bad && s.pos == s.end && s.m == 0 && s.e == EDIT_OP_NONE; }

template <typename Peek, typename Advance, typename Put>
SCRG_HD inline void decode_lane_step(DecodeLane& s, Peek&& peek, Advance&& advance, Put&& put)
{
    if (!s.alive) return;
    if (!s.ready) {
        if (s.pos < s.end) {
            const uint32_t b = peek();
            advance();
            s.pos++;
            const uint32_t e = b >> 6;
            s.m += (b & 63u) + (e == EDIT_OP_NONE ? 1u : 0u);
            s.e = e;
            s.ready = e != EDIT_OP_NONE;
        } else {
            // the matches after the last edit are implied by the read length (with op-0 bytes pending: they are part of them)
            const uint32_t rest = s.left - (s.jl - s.rj);
            if (s.m <= rest) s.m = rest;          // (else: the stream overruns the read; m stays non-zero to the end: bad)
            s.tail = s.ready = 1;
        }
    }
    if (s.ready) {
        uint32_t t = s.m < s.ri ? s.m : s.ri;
        t = t < s.rj ? t : s.rj;
        if (t) {
            s.m -= t;
            s.ri -= t;
            s.rj -= t;
            s.cur = ((uint32_t)'=' << 8) | t;
            put(s.n, s.cur);
            s.n++;
            s.prev_e = 0;
        }
        if (s.m == 0 && s.e != EDIT_OP_NONE && s.ri != 0 && s.rj != 0) {
            if (s.e == s.prev_e) {
                s.cur += 1;
                put(s.n - 1, s.cur);
            } else {
                s.cur = (edit_char_of_code(s.e) << 8) | 1u;
                put(s.n, s.cur);
                s.n++;
            }
            s.prev_e = s.e;
            s.ri -= s.e != EDIT_OP_I ? 1u : 0u;
            s.rj -= s.e != EDIT_OP_D ? 1u : 0u;
            s.e = EDIT_OP_NONE;
            s.ready = 0;
        }
    }
    if (s.ri == 0 || s.rj == 0) {
        // the window is full (genasm_cpu.cpp:307-310): the next one starts where it stopped, its run is flushed (:400-403)
        s.left -= s.jl - s.rj;
        s.prev_e = 0;
        s.jl = s.left < s.L ? s.left : s.L;
        s.ri = s.L;
        s.rj = s.jl;
        if (s.left == 0) s.alive = 0;
    } else if (s.tail && s.m == 0) {
        s.bad = 1;                                // nothing left to place and the read is not finished: not an alignment of this read
        s.alive = 0;
    }
}

hipError_t launch_encode_edits(uint64_t n_pairs, const scrg_pair_desc* d_pairs, const uint16_t* d_runs,
                               const uint32_t* d_n_runs, uint8_t* d_stream, uint64_t stream_cap, uint64_t* d_off,
                               uint32_t* d_len, uint64_t* d_total, hipStream_t s);
hipError_t launch_decode_edits(uint64_t n_pairs, uint32_t W, uint32_t O, const uint8_t* d_stream, uint64_t stream_bytes,
                               const uint64_t* d_off, const uint32_t* d_len, const uint64_t* d_read_len,
                               uint64_t read_len_stride, const uint64_t* d_dense_off, uint16_t* d_dense, uint32_t* d_n_runs,
                               uint32_t* d_bad, hipStream_t s);

}  // namespace scrg
