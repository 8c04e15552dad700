// d2h_rate.hip — how fast device -> host copies (and host -> device ones, at the end) run with (a) hipHostMalloc memory at page-aligned addresses and sizes that are
// multiples of 256 (what the host entry points' staging uses) and (b) ordinary (2 MB aligned, huge-page advised) memory made
// known to HIP with hipHostRegister, at arbitrary even offsets and exact sizes (what writing a chunk's runs straight into the
// result arrays would need); and what registering costs.   build: hipcc --offload-arch=gfx950 -O2 -o d2h_rate d2h_rate.hip
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t cap = 320u << 20;
    char* d; CK(hipMalloc(&d, cap)); CK(hipMemset(d, 7, cap));
    char* a; double t0 = now(); CK(hipHostMalloc(&a, cap, hipHostMallocDefault)); printf("hipHostMalloc %zu MB: %.2f ms\n", cap >> 20, (now() - t0) * 1e3);
    char* b = (char*)aligned_alloc(2u << 20, cap); madvise(b, cap, MADV_HUGEPAGE);
    t0 = now(); memset(b, 1, cap); printf("first touch of %zu MB: %.2f ms\n", cap >> 20, (now() - t0) * 1e3);
    t0 = now(); CK(hipHostRegister(b, cap, hipHostRegisterPortable)); printf("hipHostRegister %zu MB: %.2f ms\n", cap >> 20, (now() - t0) * 1e3);
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t sizes[] = {1u << 20, 23u << 20, 55u << 20, 200u << 20};
    const size_t offs[] = {0, 2, 4098, 1234562};
    for (size_t sz : sizes) {
        for (int mode = 0; mode < 5; mode++) {
            char* dst = mode == 0 ? a : b + offs[mode - 1];
            const size_t n = mode == 0 ? ((sz + 255) & ~(size_t)255) : sz + (mode > 1 ? 2 : 0);       // odd-ish exact sizes for the registered target
            const char* src = d + (mode >= 3 ? offs[mode - 1] % 256 : 0);                                // mode 3, 4: source misaligned like the target
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0, s));
                CK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, s));
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("%6.1f MB -> %s +%-8zu: %.3f ms  %.1f GB/s\n", n / 1048576.0, mode == 0 ? "hipHostMalloc " : "registered    ", mode == 0 ? (size_t)0 : offs[mode - 1], best, n / best / 1e6);
        }
    }
    for (size_t sz : sizes) {            // the other direction: host -> device from both kinds of memory
        for (int mode = 0; mode < 2; mode++) {
            const char* src = mode == 0 ? a : b + 4098;
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0, s));
                CK(hipMemcpyAsync(d, src, sz, hipMemcpyHostToDevice, s));
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("%6.1f MB <- %s: %.3f ms  %.1f GB/s\n", sz / 1048576.0, mode == 0 ? "hipHostMalloc " : "registered +4098", best, sz / best / 1e6);
        }
    }
    {   // registering memory that has not been touched: the pages are faulted in by the registration, on one thread
        char* c = (char*)aligned_alloc(2u << 20, cap); madvise(c, cap, MADV_HUGEPAGE);
        t0 = now(); CK(hipHostRegister(c, cap, hipHostRegisterPortable)); printf("hipHostRegister %zu MB untouched: %.2f ms\n", cap >> 20, (now() - t0) * 1e3);
        CK(hipHostUnregister(c)); free(c);
    }
    t0 = now(); CK(hipHostUnregister(b)); printf("hipHostUnregister: %.2f ms\n", (now() - t0) * 1e3);
    return 0;
}
