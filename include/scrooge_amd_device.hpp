// scrooge_amd_device.hpp — the device-side pieces of the library that a hipcc-compiled CALLER may use directly.
//
// One thing lives here: the packing of ASCII bases into the reference's 2-bit byte layout — 4 bases per byte, the
// first base of each quad in bits 7..6, the tail byte zero-padded (/root/reference/src/genasm_gpu.cu:631-669) — as the
// device function that BOTH the library's own kernel (scrooge_amd/csrc/seq_kernels.hip: ascii_to_twobit_kernel, behind
// scrg_ascii_to_twobit) and the reference-named kernel genasm_gpu::ascii_to_twobit_strings
// (include/compat/genasm_gpu.hpp, src/genasm_gpu.hpp:9) are made of.  hipcc only; gfx950 code, no other target.
#pragma once

#if !defined(__HIPCC__)
#error "scrooge_amd_device.hpp is device code: compile with hipcc (host compilers use scrooge_amd.h / scrooge_amd.hpp)"
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scrooge_amd {
namespace device {

// One output byte of a 2-bit string: bases 4*quad .. 4*quad+3 of `ascii` (only those below `len`), base k of the quad in
// bits 7-2k..6-2k.  A/a -> 0, C/c -> 1, G/g -> 2, T/t -> 3 (src/genasm_gpu.cu:631-638); any other byte counts in *bad
// and packs as the two bits (c >> 1) & 3 Gray-decoded (the reference asserts there: the caller decides what an invalid
// base means — scrg_ascii_to_twobit reports SCRG_ERR_BAD_BASE, the reference-named kernel asserts like the reference).
__device__ __forceinline__ uint8_t twobit_quad(const char* __restrict__ ascii, uint64_t len, uint64_t quad, uint32_t* bad)
{
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t p = 4 * quad + (uint64_t)k;
        uint32_t code = 0;
        if (p < len) {
            const uint32_t c = (uint8_t)ascii[p];
            const uint32_t u = c & 0xdfu;                                   // upper case
            if (!(u == 'A' || u == 'C' || u == 'G' || u == 'T')) (*bad)++;
            const uint32_t x = (c >> 1) & 3u;                               // A 0, C 1, T 2, G 3: Gray code of the base index
            code = x ^ (x >> 1);
        }
        out |= code << (6 - 2 * k);
    }
    return (uint8_t)out;
}

// All ceil(len / 4) bytes of one string, the threads of the calling workgroup side by side (thread t: bytes t, t + blockDim.x, …:
// consecutive threads write consecutive bytes and read consecutive dwords).  Returns this thread's count of invalid bases.
__device__ __forceinline__ uint32_t twobit_string(long long len, const char* __restrict__ ascii, char* __restrict__ twobit)
{
    uint32_t bad = 0;
    const uint64_t n = len > 0 ? (uint64_t)len : 0u, nbytes = (n + 3) / 4;
    for (uint64_t b = threadIdx.x; b < nbytes; b += blockDim.x) twobit[b] = (char)twobit_quad(ascii, n, b, &bad);
    return bad;
}

}  // namespace device
}  // namespace scrooge_amd
