// Dependent-chain latency (cycles per instruction, one wave per SIMD) of the ops on the traceback's
// critical path, gfx950.  Each kernel is one serial chain of 16 x ITER instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define ITER 2048
#define R16(X) X X X X X X X X X X X X X X X X
#define KERN(NAME, ASM, ...)                                                       \
__global__ void __launch_bounds__(64) NAME(uint32_t* out, uint64_t* cyc) {         \
    uint32_t a = threadIdx.x + 1, b = out[threadIdx.x & 3] | 1, c = b * 7 + 3;    \
    uint64_t a64 = a; (void)a64; (void)c;                                          \
    __shared__ uint32_t sh[256]; sh[threadIdx.x] = (threadIdx.x * 4) & 255; sh[threadIdx.x+64]=0; sh[threadIdx.x+128]=4; sh[threadIdx.x+192]=8; \
    uint64_t t0 = __builtin_readcyclecounter();                                    \
    for (int i = 0; i < ITER; i++) { asm volatile(R16(ASM) __VA_ARGS__); }         \
    uint64_t t1 = __builtin_readcyclecounter();                                    \
    out[blockIdx.x * 64 + threadIdx.x] = a + (uint32_t)a64;                        \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                               \
}
KERN(k_xor, "v_xor_b32 %0, %0, %1\n", : "+v"(a) : "v"(b))
KERN(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c\n", : "+v"(a) : "v"(b), "v"(c))
KERN(k_lshl, "v_lshlrev_b32 %0, 1, %0\nv_or_b32 %0, %0, %1\n", : "+v"(a) : "v"(b))
KERN(k_lshl64, "v_lshlrev_b64 %0, 1, %0\n", : "+v"(a64) :)
KERN(k_align, "v_alignbit_b32 %0, %0, %0, %1\n", : "+v"(a) : "v"(b))
KERN(k_ffbh, "v_ffbh_u32 %0, %0\nv_or_b32 %0, %0, %1\n", : "+v"(a) : "v"(b))
KERN(k_min, "v_min_u32 %0, %0, %1\n", : "+v"(a) : "v"(b))
KERN(k_cnd, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n", : "+v"(a) : "v"(b) : "s20", "s21")
KERN(k_cmpcnd, "v_cmp_lt_u32 vcc, %0, %1\nv_cndmask_b32 %0, %0, %1, vcc\n", : "+v"(a) : "v"(b) : "vcc")
KERN(k_dpp, "s_nop 1\nv_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n", : "+v"(a) :)
KERN(k_mindpp, "s_nop 1\nv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n", : "+v"(a) :)
KERN(k_ashr, "v_sub_u32 %0, %1, %0\nv_ashrrev_i32 %0, 31, %0\n", : "+v"(a) : "v"(b))
KERN(k_ldsrd, "ds_read_b32 %0, %0\ns_waitcnt lgkmcnt(0)\n", : "+v"(a) :)
KERN(k_ldswr_rd, "ds_write_b32 %1, %0\nds_read_b32 %0, %1\ns_waitcnt lgkmcnt(0)\n", : "+v"(a) : "v"(c & 0xfc))
KERN(k_cmp_branchless, "v_cmp_ne_u32 vcc, 0, %0\ns_and_b64 s[20:21], vcc, exec\nv_cndmask_b32_e64 %0, %1, %0, s[20:21]\n", : "+v"(a) : "v"(b) : "vcc", "s20", "s21")
KERN(k_readlane, "v_readfirstlane_b32 s20, %0\nv_add_u32 %0, s20, %0\n", : "+v"(a) : : "s20")
typedef void (*kfn)(uint32_t*, uint64_t*);
struct E { const char* n; kfn f; int per; };
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int blocks = p.multiProcessorCount * 4;
    uint32_t* out; uint64_t* cyc; hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8); hipMemset(out, 0, blocks * 64 * 4);
    std::vector<E> es = {{"v_xor (dep)", k_xor, 1}, {"v_bitop3 (dep)", k_bitop3, 1}, {"v_lshlrev_b32 + v_or", k_lshl, 2}, {"v_lshlrev_b64 (dep)", k_lshl64, 1},
        {"v_alignbit (dep)", k_align, 1}, {"v_ffbh + v_or", k_ffbh, 2}, {"v_min_u32 (dep)", k_min, 1}, {"v_cndmask e64 (dep)", k_cnd, 1}, {"v_cmp + v_cndmask", k_cmpcnd, 2},
        {"s_nop1 + v_mov_dpp (dep)", k_dpp, 1}, {"s_nop1 + v_min_u32_dpp", k_mindpp, 1}, {"v_sub + v_ashr", k_ashr, 2}, {"ds_read_b32 (addr dep)", k_ldsrd, 1},
        {"ds_write + ds_read + wait", k_ldswr_rd, 1}, {"v_cmp + s_and + v_cndmask", k_cmp_branchless, 3}, {"v_readfirstlane + v_add", k_readlane, 2}};
    printf("one wave per SIMD; shader cycles per chain link (s_memtime), link = the instruction group named\n");
    for (auto& e : es) {
        hipLaunchKernelGGL(e.f, dim3(blocks), dim3(64), 0, 0, out, cyc); hipDeviceSynchronize();
        hipLaunchKernelGGL(e.f, dim3(blocks), dim3(64), 0, 0, out, cyc); hipDeviceSynchronize();
        std::vector<uint64_t> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        printf("%-30s %.1f cycles per link (%d instr)\n", e.n, avg / (ITER * 16.0), e.per);
    }
    return 0;
}
