// row_ops.h — multi-word bit rows of the W > 64 kernels (genasm_lane_mw_kernel.hip: Row<RW>; genasm_kernel_multiword.hip:
// BV<NW>).  The counterpart of the reference's N-bit bitvector (src/bitvector.hpp:45-48: N/32 elements, shifts carried
// across the elements :124-139, has_one_at / single_one_at :150-190), in the mirrored layout the kernels use: word 0 is
// the MOST significant, position c of a row is bit 63 - c % 64 of word c / 64, so that "the next event" is one
// count-leading-zeros.  A reference vector of `bits` bits sits top-aligned in a row: reference bit i <-> position
// bits-1-i, the reference's shift_l(n) (which drops what leaves the top) <-> row_shl(n).
//
// Plain C++: the header compiles for the host as well, so that tests/test_row_ops.py checks these very functions with
// g++ against the known answers of the reference's bitvector tests (src/bitvector_test.cu:22-118).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SCRG_ROW_FN __host__ __device__ __forceinline__
#else
#define SCRG_ROW_FN static inline
#endif

namespace scrg {

constexpr uint64_t TOP = 0x8000000000000000ull;

SCRG_ROW_FN uint32_t clz64_mw(uint64_t v) { return v ? (uint32_t)__builtin_clzll(v) : 64u; }

// An RW-word row, word 0 the most significant; column / pattern row c <-> bit 63 - c % 64 of word c / 64.
template <int RW> struct Row {
    uint64_t w[RW];
};
template <int RW> SCRG_ROW_FN Row<RW> row_zero()
{
    Row<RW> r;
#pragma unroll
    for (int k = 0; k < RW; k++) r.w[k] = 0;
    return r;
}
template <int RW> SCRG_ROW_FN Row<RW> row_shl(const Row<RW>& a, uint32_t s)       // towards word 0, s < 64 RW
{
    Row<RW> r;
    const uint32_t ws = s >> 6, b = s & 63u;
#pragma unroll
    for (int k = 0; k < RW; k++) {
        uint64_t hi = 0, lo = 0;
#pragma unroll
        for (int q = 0; q < RW; q++) {
            if ((uint32_t)q == (uint32_t)k + ws) hi = a.w[q];
            if ((uint32_t)q == (uint32_t)k + ws + 1u) lo = a.w[q];
        }
        r.w[k] = b ? ((hi << b) | (lo >> (64u - b))) : hi;
    }
    return r;
}
template <int RW> SCRG_ROW_FN Row<RW> row_shl1_in(const Row<RW>& a, uint64_t in)   // << 1, `in` enters at the bottom
{
    Row<RW> r;
#pragma unroll
    for (int k = 0; k < RW; k++) r.w[k] = (a.w[k] << 1) | (k + 1 < RW ? a.w[k + 1] >> 63 : in);
    return r;
}
template <int RW> SCRG_ROW_FN Row<RW> row_shr1(const Row<RW>& a)
{
    Row<RW> r;
#pragma unroll
    for (int k = 0; k < RW; k++) r.w[k] = (a.w[k] >> 1) | (k ? a.w[k - 1] << 63 : 0ull);
    return r;
}
template <int RW> SCRG_ROW_FN uint32_t row_clz(const Row<RW>& a)
{
    uint32_t n = 0;
    bool done = false;
#pragma unroll
    for (int k = 0; k < RW; k++) {
        const uint32_t c = clz64_mw(a.w[k]);
        if (!done) n += c;
        done = done || a.w[k] != 0;
    }
    return n;
}
template <int RW> SCRG_ROW_FN Row<RW> row_bit(uint32_t c)
{
    Row<RW> r;
#pragma unroll
    for (int k = 0; k < RW; k++) r.w[k] = (c >> 6) == (uint32_t)k ? TOP >> (c & 63u) : 0ull;
    return r;
}
template <int RW> SCRG_ROW_FN bool row_test(const Row<RW>& a, uint32_t c)
{
    uint64_t v = 0;
#pragma unroll
    for (int k = 0; k < RW; k++) v |= (c >> 6) == (uint32_t)k ? a.w[k] : 0ull;
    return ((v >> (63u - (c & 63u))) & 1ull) != 0;
}
template <int RW> SCRG_ROW_FN Row<RW> row_top(uint32_t t)       // the top t bits set
{
    Row<RW> r;
#pragma unroll
    for (int k = 0; k < RW; k++) {
        const uint32_t lo = 64u * (uint32_t)k;
        r.w[k] = t >= lo + 64u ? ~0ull : (t <= lo ? 0ull : ~(~0ull >> (t - lo)));
    }
    return r;
}
template <int RW> SCRG_ROW_FN bool row_any(const Row<RW>& a)
{
    uint64_t v = 0;
#pragma unroll
    for (int k = 0; k < RW; k++) v |= a.w[k];
    return v != 0;
}
template <int RW> SCRG_ROW_FN uint32_t row_pop(const Row<RW>& a)
{
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < RW; k++) n += (uint32_t)__builtin_popcountll(a.w[k]);
    return n;
}

// An NW-word GenASM bitvector of the row-sweep kernel: w[0] holds characters 0..63 (character j at bit 63-j), w[1] 64..127, ...
template <int NW> struct BV {
    uint64_t w[NW];
};
template <int NW> SCRG_ROW_FN BV<NW> bv_fill(uint64_t x)
{
    BV<NW> r;
#pragma unroll
    for (int i = 0; i < NW; i++) r.w[i] = x;
    return r;
}
template <int NW> SCRG_ROW_FN BV<NW> bv_shl1(const BV<NW>& v)
{
    BV<NW> r;
#pragma unroll
    for (int i = 0; i < NW - 1; i++) r.w[i] = (v.w[i] << 1) | (v.w[i + 1] >> 63);
    r.w[NW - 1] = v.w[NW - 1] << 1;
    return r;
}

}  // namespace scrg
