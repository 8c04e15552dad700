// Microbenchmark: does a wave64 VALU instruction whose EXEC mask has an empty 32-lane half issue in half the time on gfx950?
// (If it did, wavefronts of 32 pairs — twice as many of them — would hide latencies better for the same lane-cycles.)
// Build: hipcc --offload-arch=gfx950 -O3 -o half_wave half_wave.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define ITER 4096
template <int MODE>      // 0: all 64 lanes, 1: lanes 0..31, 2: lanes 32..63, 3: even lanes (32 active, both halves)
__global__ void __launch_bounds__(64) k(uint32_t* out, uint64_t* cyc) {
    uint32_t a0=threadIdx.x+1,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19;
    uint32_t b = out[threadIdx.x & 3] | 1, c = b * 7 + 1;
    const bool on = MODE == 0 || (MODE == 1 && threadIdx.x < 32) || (MODE == 2 && threadIdx.x >= 32) || (MODE == 3 && (threadIdx.x & 1) == 0);
    uint64_t t0 = __builtin_readcyclecounter();
    if (on) {
        for (int i = 0; i < ITER; i++) {
#define B(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0x96\n"
#define A(n) "v_and_b32 %" #n ", %" #n ", %8\n"
            asm volatile(B(0) A(1) B(2) A(3) B(4) A(5) B(6) A(7) B(0) A(1) B(2) A(3) B(4) A(5) B(6) A(7)
                : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b), "v"(c));
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0^a1^a2^a3^a4^a5^a6^a7;
    if (threadIdx.x == 0 || threadIdx.x == 32) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    uint32_t* out; uint64_t* cyc;
    const char* names[4] = {"all 64 lanes", "lanes 0..31", "lanes 32..63", "even lanes"};
    for (int wps : {1, 2, 4, 8}) {
        int blocks = cus * 4 * wps;
        hipMalloc(&out, (size_t)blocks * 64 * 4); hipMalloc(&cyc, (size_t)blocks * 8);
        hipMemset(out, 0, (size_t)blocks * 64 * 4);
        for (int m = 0; m < 4; m++) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            auto launch = [&]() {
                if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, cyc);
                if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, cyc);
                if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, cyc);
                if (m == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, cyc);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            double insts = (double)ITER * 16;
            printf("%d wave(s)/SIMD  %-14s wall %.3f ms  -> %.2f ns per wave instruction per SIMD\n", wps, names[m], ms, ms * 1e6 / (insts * wps));
        }
        hipFree(out); hipFree(cyc);
    }
    return 0;
}
