// Host build of scrooge_amd/csrc/row_ops.h (g++): C entry points over Row<RW> / BV<NW> for tests/test_row_ops.py.
#include <stdint.h>
#include "row_ops.h"

using namespace scrg;

template <int RW> static Row<RW> load(const uint64_t* p)
{
    Row<RW> r;
    for (int k = 0; k < RW; k++) r.w[k] = p[k];
    return r;
}
template <int RW> static void store(const Row<RW>& r, uint64_t* p)
{
    for (int k = 0; k < RW; k++) p[k] = r.w[k];
}

// op: 0 row_shl(a, s)  1 row_shl1_in(a, s & 1)  2 row_shr1(a)  3 row_bit(s)  4 row_top(s)  5 bv_shl1(a)
//     returns: 6 row_clz(a)  7 row_test(a, s)  8 row_any(a)  9 row_pop(a)
template <int RW> static uint32_t run(int op, const uint64_t* a, uint32_t s, uint64_t* out)
{
    const Row<RW> x = load<RW>(a);
    switch (op) {
    case 0: store<RW>(row_shl<RW>(x, s), out); return 0;
    case 1: store<RW>(row_shl1_in<RW>(x, s & 1u), out); return 0;
    case 2: store<RW>(row_shr1<RW>(x), out); return 0;
    case 3: store<RW>(row_bit<RW>(s), out); return 0;
    case 4: store<RW>(row_top<RW>(s), out); return 0;
    case 5: {
        BV<RW> v;
        for (int k = 0; k < RW; k++) v.w[k] = a[k];
        v = bv_shl1<RW>(v);
        for (int k = 0; k < RW; k++) out[k] = v.w[k];
        return 0;
    }
    case 6: return row_clz<RW>(x);
    case 7: return row_test<RW>(x, s) ? 1u : 0u;
    case 8: return row_any<RW>(x) ? 1u : 0u;
    case 9: return row_pop<RW>(x);
    }
    return 0xffffffffu;
}

extern "C" uint32_t row_op(int rw, int op, const uint64_t* a, uint32_t s, uint64_t* out)
{
    switch (rw) {
    case 1: return run<1>(op, a, s, out);
    case 2: return run<2>(op, a, s, out);
    case 3: return run<3>(op, a, s, out);
    case 4: return run<4>(op, a, s, out);
    }
    return 0xffffffffu;
}
