// genasm_device.h — device helpers shared by the gfx950 aligner kernels
// (genasm_kernels.hip: W <= 64, one 64-bit word per entry; genasm_kernel_multiword.hip: 64 < W <= 256).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scrg {

// ----------------------------------------------------------------------------
// small device helpers
// ----------------------------------------------------------------------------

// lane i <- lane i+1 across the whole wave (DPP wave_shl:1, full rate, no LDS)
__device__ __forceinline__ uint32_t dpp_from_next(uint32_t v)
{
    uint32_t r;   // lane 63 has no source and keeps an undefined value; it is a slot's last lane, which never uses it
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ uint64_t dpp_from_next64(uint64_t v)
{
    uint32_t lo = dpp_from_next((uint32_t)v);
    uint32_t hi = dpp_from_next((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// lane i <- lane i-1 (DPP wave_shr:1); lane 0 keeps an undefined value
__device__ __forceinline__ uint32_t dpp_from_prev(uint32_t v)
{
    uint32_t r;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ uint64_t dpp_from_prev64(uint64_t v)
{
    uint32_t lo = dpp_from_prev((uint32_t)v);
    uint32_t hi = dpp_from_prev((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// a + b + carry-in, where the carry-in of lane l is bit l of the SGPR mask cin (v_addc_co_u32 x2)
__device__ __forceinline__ uint64_t add64_cin(uint64_t a, uint64_t b, uint64_t cin)
{
    uint32_t lo, hi;
    uint64_t c1, c2;
    asm("v_addc_co_u32 %0, %1, %2, %3, %4" : "=v"(lo), "=s"(c1) : "v"((uint32_t)a), "v"((uint32_t)b), "s"(cin));
    asm("v_addc_co_u32 %0, %1, %2, %3, %4" : "=v"(hi), "=s"(c2) : "v"((uint32_t)(a >> 32)), "v"((uint32_t)(b >> 32)), "s"(c1));
    return ((uint64_t)hi << 32) | lo;
}

// 64-bit add / shift-by-one-and-add in ONE instruction (v_lshl_add_u64, gfx940+): half the issue cost of a
// v_addc pair and no carry through SGPRs
__device__ __forceinline__ uint64_t add64(uint64_t a, uint64_t b)
{
    uint64_t r;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint64_t shl1_add64(uint64_t a, uint64_t b)
{
    uint64_t r;
    asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// truth table of a 3-input function for v_bitop3_b32 (bit index = a*4 + b*2 + c)
template <typename F> constexpr int bitop3_table(F f)
{
    int tt = 0;
    for (int k = 0; k < 8; k++)
        if (f((k >> 2) & 1, (k >> 1) & 1, k & 1) & 1) tt |= 1 << k;
    return tt;
}

__device__ __forceinline__ uint64_t ones_shl(int d)
{
    // bitvector::ones() << d with the reference's ">= width gives zero" rule
    // (src/bitvector.hpp:116-122)
    return d >= 64 ? 0ull : (~0ull << d);
}

// 64 bases starting at base offset p of a planar array: returns the low-bit
// plane in .x and the high-bit plane in .y (bit k <-> base p+k)
struct Planes { uint64_t lo, hi; };
__device__ __forceinline__ Planes load_window(const uint64_t* __restrict__ seq, uint64_t p)
{
    const uint64_t w = p >> 5;
    const uint32_t s = (uint32_t)p & 31u;
    const uint64_t a = seq[w], b = seq[w + 1], c = seq[w + 2];
    const uint32_t l0 = (uint32_t)a, l1 = (uint32_t)b, l2 = (uint32_t)c;
    const uint32_t h0 = (uint32_t)(a >> 32), h1 = (uint32_t)(b >> 32), h2 = (uint32_t)(c >> 32);
    Planes r;
    r.lo = (uint64_t)__builtin_amdgcn_alignbit(l1, l0, s) | ((uint64_t)__builtin_amdgcn_alignbit(l2, l1, s) << 32);
    r.hi = (uint64_t)__builtin_amdgcn_alignbit(h1, h0, s) | ((uint64_t)__builtin_amdgcn_alignbit(h2, h1, s) << 32);
    return r;
}

// The same for a sequence whose consecutive words are `stride` words apart (lane-interleaved groups,
// scrg_pack_planar_groups): base k of the sequence at offset `off` lives in word off/32 + ((off%32 + k)/32)*stride.
__device__ __forceinline__ Planes load_window_strided(const uint64_t* __restrict__ seq, uint64_t off, uint32_t k, uint32_t stride)
{
    const uint32_t inner = ((uint32_t)off & 31u) + k;
    const uint64_t w = (off >> 5) + (uint64_t)(inner >> 5) * stride;
    const uint32_t s = inner & 31u;
    const uint64_t a = seq[w], b = seq[w + stride], c = seq[w + 2u * stride];
    const uint32_t l0 = (uint32_t)a, l1 = (uint32_t)b, l2 = (uint32_t)c;
    const uint32_t h0 = (uint32_t)(a >> 32), h1 = (uint32_t)(b >> 32), h2 = (uint32_t)(c >> 32);
    Planes r;
    r.lo = (uint64_t)__builtin_amdgcn_alignbit(l1, l0, s) | ((uint64_t)__builtin_amdgcn_alignbit(l2, l1, s) << 32);
    r.hi = (uint64_t)__builtin_amdgcn_alignbit(h1, h0, s) | ((uint64_t)__builtin_amdgcn_alignbit(h2, h1, s) << 32);
    return r;
}

// Minus-strand pairs (scrg_params.stranded, SCRG_READ_REVCOMP): the 64-bit word `w` (characters 64 w .. 64 w + 63) of the window at
// read_idx of a read's REVERSE COMPLEMENT, already REVERSED and left-aligned (bit 63-k <-> character 64 w + k) as the tables want
// it, from the read's one packed (forward) copy: the inverted 64 bases that END at len - read_idx - 64 w — complement = both
// planes inverted, and reading the bases backwards IS the reversal — moved up when fewer than 64 are left (what is below the
// pattern is masked by the caller's `valid` word).  A word beyond the pattern returns garbage the caller never looks at.
__device__ __forceinline__ Planes revcomp_pattern_word(const uint64_t* __restrict__ seq, uint64_t read_off, uint32_t read_len, uint32_t read_idx,
                                                       uint32_t w, uint32_t stride)
{
    const uint32_t left = read_len - read_idx;
    const uint32_t end = left > 64u * w ? left - 64u * w : 0u;          // the word's forward window ends here (exclusive)
    const uint32_t at = end > 64u ? end - 64u : 0u, sh = (end >= 64u ? 0u : 64u - end) & 63u;
    const Planes f = load_window_strided(seq, read_off, at, stride);
    Planes r;
    r.lo = ~(f.lo << sh);
    r.hi = ~(f.hi << sh);
    return r;
}

// The same in two steps, so that the loads can be issued long before the words are needed: the three words a window
// can touch, then the funnel shifts.
struct WindowWords { uint64_t a, b, c; uint32_t s; };
__device__ __forceinline__ WindowWords load_window_words(const uint64_t* __restrict__ seq, uint64_t off, uint32_t k, uint32_t stride)
{
    const uint32_t inner = ((uint32_t)off & 31u) + k;
    const uint64_t w = (off >> 5) + (uint64_t)(inner >> 5) * stride;
    return WindowWords{seq[w], seq[w + stride], seq[w + 2u * stride], inner & 31u};
}
// The same from a sequence's FIRST WORD (a pointer, made once per pair) and the offset of its first base inside that word
// (0..31): per window one add, one shift, one and, one 64-bit multiply-add and two 64-bit adds — the divisions of the base
// offset are not repeated for every window.
__device__ __forceinline__ WindowWords load_window_words_at(const uint64_t* __restrict__ first_word, uint32_t in_word, uint32_t k, uint32_t stride)
{
    const uint32_t inner = in_word + k;
    const uint64_t* const w = first_word + (uint64_t)(inner >> 5) * stride;
    return WindowWords{w[0], w[stride], w[2u * stride], inner & 31u};
}
__device__ __forceinline__ Planes window_planes(const WindowWords& v)
{
    const uint32_t l0 = (uint32_t)v.a, l1 = (uint32_t)v.b, l2 = (uint32_t)v.c;
    const uint32_t h0 = (uint32_t)(v.a >> 32), h1 = (uint32_t)(v.b >> 32), h2 = (uint32_t)(v.c >> 32);
    Planes r;
    r.lo = (uint64_t)__builtin_amdgcn_alignbit(l1, l0, v.s) | ((uint64_t)__builtin_amdgcn_alignbit(l2, l1, v.s) << 32);
    r.hi = (uint64_t)__builtin_amdgcn_alignbit(h1, h0, v.s) | ((uint64_t)__builtin_amdgcn_alignbit(h2, h1, v.s) << 32);
    return r;
}

// a - b, 0 if that is negative: one v_sub_u32 with the clamp bit ("a > b ? a - b : 0" would be a compare and a v_cndmask on VCC)
__device__ __forceinline__ uint32_t sub_sat_u32(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_sub_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Conditions as 0 / ~0 masks.  Written as asm / intrinsics so that the optimiser cannot turn them
// back into v_cmp + v_cndmask (both half rate on gfx950; v_ashrrev, v_sub and v_bitop3 are full rate).
__device__ __forceinline__ uint32_t neg_mask(uint32_t x)      // ~0 iff (int32)x < 0
{
    uint32_t r;
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t nz_mask(uint32_t x)       // ~0 iff x != 0, for x < 2^31
{
    return neg_mask(0u - x);
}
template <int TT> __device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
}

// minimum of v over the G lanes of a slot, returned in every lane.  G <= 16: butterfly of
// DPP-modified v_min_u32 (no LDS traffic, no SALU); wider slots finish with xor-shuffles.
template <int CTRL> __device__ __forceinline__ uint32_t dpp_min(uint32_t v)
{
    uint32_t r;
    if (CTRL == 0xB1) asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    else if (CTRL == 0x4E) asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    else if (CTRL == 0x141) asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    else asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}
template <int G> __device__ __forceinline__ uint32_t slot_min(uint32_t v)
{
    v = dpp_min<0xB1>(v);                    // quad_perm:[1,0,3,2]
    v = dpp_min<0x4E>(v);                    // quad_perm:[2,3,0,1]
    if (G >= 8) v = dpp_min<0x141>(v);       // row_half_mirror
    if (G >= 16) v = dpp_min<0x140>(v);      // row_mirror
    if (G >= 32) { const uint32_t o = (uint32_t)__shfl_xor((int)v, 16); v = o < v ? o : v; }
    if (G >= 64) { const uint32_t o = (uint32_t)__shfl_xor((int)v, 32); v = o < v ? o : v; }
    return v;
}

// 64-bit shift left by one as ONE v_lshlrev_b64 (quarter-rate class, like any 32-bit shift on
// gfx950); written as asm so the compiler does not split it into lshl + alignbit (two of them)
__device__ __forceinline__ uint64_t shl1(uint64_t v)
{
    uint64_t r;
    asm("v_lshlrev_b64 %0, 1, %1" : "=v"(r) : "v"(v));
    return r;
}

__device__ __forceinline__ uint64_t brev64(uint64_t v)
{
    return ((uint64_t)__builtin_bitreverse32((uint32_t)v) << 32) | __builtin_bitreverse32((uint32_t)(v >> 32));
}

constexpr uint64_t leader_mask(int g)
{
    uint64_t m = 0;
    for (int s = 0; s < 64 / g; s++) m |= 1ull << (s * g);
    return m;
}


}  // namespace scrg
