// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD) of the integer/logic
// VALU ops the GenASM kernel is made of, on gfx950.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITER 4096
#define OPS_PER_ITER 16

#define KERNEL(NAME, BODY)                                                        \
__global__ void __launch_bounds__(64) NAME(uint32_t* out, uint64_t* cyc) {        \
    uint32_t a0=threadIdx.x+1,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; \
    uint32_t b = out[threadIdx.x & 3] | 1, c = b * 7 + 1;                         \
    uint64_t t0 = __builtin_readcyclecounter();                                   \
    for (int i = 0; i < ITER; i++) { BODY BODY }                                  \
    uint64_t t1 = __builtin_readcyclecounter();                                   \
    out[blockIdx.x * 64 + threadIdx.x] = a0^a1^a2^a3^a4^a5^a6^a7;                 \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                              \
}

#define A8(INS) \
  asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b), "v"(c));

#define I_AND(n)   "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_OR3(n)   "v_or3_b32 %" #n ", %" #n ", %8, %9\n"
#define I_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 1, %8\n"
#define I_ALIGN(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 31\n"
#define I_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define I_BITOP3(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0x80\n"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_CNDMASK64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\n"
#define I_ANDS(n) "v_and_b32 %" #n ", s20, %" #n "\n"
#define I_MIN(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define I_FFBH(n) "v_ffbh_u32 %" #n ", %" #n "\n"
#define I_NOT(n) "v_not_b32 %" #n ", %" #n "\n"
#define I_BFEU(n) "v_bfe_u32 %" #n ", %" #n ", 3, 2\n"
#define I_ASHR(n) "v_ashrrev_i32 %" #n ", 31, %" #n "\n"
#define I_LSHR(n) "v_lshrrev_b32 %" #n ", %8, %" #n "\n"
#define I_MINDPP(n) "v_min_u32_dpp %" #n ", %8, %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_SUBREV(n) "v_subrev_u32 %" #n ", %8, %" #n "\n"
#define I_MAXI(n) "v_max_i32 %" #n ", %" #n ", %8\n"
#define I_SUB(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 1, %8\n"
#define I_MBCNT(n) "v_mbcnt_lo_u32_b32 %" #n ", %8, %" #n "\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define I_ADD(n)   "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_FMA(n)   "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_DPP(n)   "v_mov_b32_dpp %" #n ", %8 wave_shl:1 row_mask:0xf bank_mask:0xf\n"
#define I_DPPROW(n) "v_mov_b32_dpp %" #n ", %8 row_shl:1 row_mask:0xf bank_mask:0xf\n"
#define I_LSHL(n)  "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define I_BFE(n)   "v_bfe_i32 %" #n ", %" #n ", 3, 1\n"
#define I_XOR(n)   "v_xor_b32 %" #n ", %" #n ", %8\n"
#define I_BFREV(n) "v_bfrev_b32 %" #n ", %" #n "\n"
#define I_MOV(n)   "v_mov_b32 %" #n ", %8\n"
#define I_CMP(n)   "v_cmp_lt_i32 vcc, %" #n ", %8\n"

KERNEL(k_and, A8(I_AND))
KERNEL(k_or3, A8(I_OR3))
KERNEL(k_lshlor, A8(I_LSHLOR))
KERNEL(k_align, A8(I_ALIGN))
KERNEL(k_andor, A8(I_ANDOR))
KERNEL(k_bitop3, A8(I_BITOP3))
KERNEL(k_cndmask, A8(I_CNDMASK))
KERNEL(k_add, A8(I_ADD))
KERNEL(k_cnd64, asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20","s21"); A8(I_CNDMASK64))
KERNEL(k_ands, asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20","s21"); A8(I_ANDS))
KERNEL(k_sub, A8(I_SUB))
KERNEL(k_min, A8(I_MIN))
KERNEL(k_ffbh, A8(I_FFBH))
KERNEL(k_not, A8(I_NOT))
KERNEL(k_bfeu, A8(I_BFEU))
KERNEL(k_ashr, A8(I_ASHR))
KERNEL(k_lshr, A8(I_LSHR))
KERNEL(k_mindpp, A8(I_MINDPP))
KERNEL(k_maxi, A8(I_MAXI))
KERNEL(k_lshladd, A8(I_LSHLADD))
KERNEL(k_mbcnt, A8(I_MBCNT))
KERNEL(k_perm, A8(I_PERM))
KERNEL(k_fma, A8(I_FMA))
KERNEL(k_dpp, A8(I_DPP))
KERNEL(k_dpprow, A8(I_DPPROW))
KERNEL(k_lshl, A8(I_LSHL))
KERNEL(k_bfe, A8(I_BFE))
KERNEL(k_xor, A8(I_XOR))
KERNEL(k_bfrev, A8(I_BFREV))
KERNEL(k_mov, A8(I_MOV))
KERNEL(k_cmp, A8(I_CMP))

// 64-bit shift: 4 independent 64-bit accumulators
__global__ void __launch_bounds__(64) k_lshl64(uint32_t* out, uint64_t* cyc) {
    uint64_t a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; i++) {
#define L64(n) "v_lshlrev_b64 %" #n ", 1, %" #n "\n"
        asm volatile(L64(0) L64(1) L64(2) L64(3) L64(4) L64(5) L64(6) L64(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        asm volatile(L64(0) L64(1) L64(2) L64(3) L64(4) L64(5) L64(6) L64(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// carry ops: v_addc_co_u32 (VOP3, carry in/out in SGPR pairs), 8 independent accumulators, carry-in from one fixed pair
#define I_ADDC(n) "v_addc_co_u32 %" #n ", s[22:23], %" #n ", %8, s[20:21]\n"
KERNEL(k_addc, asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20","s21","s22","s23"); A8(I_ADDC))
// 64-bit add as a carry pair: lo writes s[22:23], hi consumes it (4 independent 64-bit accumulators = 8 instructions)
__global__ void __launch_bounds__(64) k_addc_pair(uint32_t* out, uint64_t* cyc) {
    uint32_t a0=threadIdx.x+1,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19;
    uint32_t b = out[threadIdx.x & 3] | 1;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; i++) {
#define P2(l, h, s) "v_addc_co_u32 %" #l ", " s ", %" #l ", %8, s[20:21]\n" "v_addc_co_u32 %" #h ", " s ", %" #h ", %8, " s "\n"
#define PAIRS P2(0,1,"s[22:23]") P2(2,3,"s[24:25]") P2(4,5,"s[26:27]") P2(6,7,"s[28:29]")
        asm volatile("s_mov_b64 s[20:21], 0x5555\n" PAIRS PAIRS
            : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b) : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29");
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0^a1^a2^a3^a4^a5^a6^a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// v_lshl_add_u64: 64-bit add in one instruction, 8 independent 64-bit accumulators
__global__ void __launch_bounds__(64) k_lshladd64(uint32_t* out, uint64_t* cyc) {
    uint64_t a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint64_t b = out[threadIdx.x & 3] | 1;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; i++) {
#define LA64(n) "v_lshl_add_u64 %" #n ", %" #n ", 0, %8\n"
        asm volatile(LA64(0) LA64(1) LA64(2) LA64(3) LA64(4) LA64(5) LA64(6) LA64(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        asm volatile(LA64(0) LA64(1) LA64(2) LA64(3) LA64(4) LA64(5) LA64(6) LA64(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// v_cmp writing an SGPR pair (VOP3)
#define I_CMP64(n) "v_cmp_lt_i32 s[22:23], %" #n ", %8\n"
KERNEL(k_cmp64, A8(I_CMP64))
// v_add_co_u32 (VOP2, carry-out to vcc)
#define I_ADDCO(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %8\n"
KERNEL(k_addco, A8(I_ADDCO))
// ds_write_b32 / ds_read_b32 issue cost (address from accumulator, small range)
// dependent chain of v_and_b32 (one accumulator)
__global__ void __launch_bounds__(64) k_and_dep(uint32_t* out, uint64_t* cyc) {
    uint32_t a0 = threadIdx.x + 1, b = out[threadIdx.x & 3] | 1;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; i++) {
#define D1 "v_xor_b32 %0, %0, %1\n"
        asm volatile(D1 D1 D1 D1 D1 D1 D1 D1 D1 D1 D1 D1 D1 D1 D1 D1 : "+v"(a0) : "v"(b));
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*kfn)(uint32_t*, uint64_t*);
struct Entry { const char* name; kfn f; };

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    std::vector<Entry> es = {{"v_and_b32", k_and}, {"v_xor_b32", k_xor}, {"v_or3_b32", k_or3}, {"v_lshl_or_b32", k_lshlor}, {"v_alignbit_b32", k_align},
        {"v_and_or_b32", k_andor}, {"v_bitop3_b32", k_bitop3}, {"v_cndmask_b32", k_cndmask}, {"v_add_u32", k_add}, {"v_cndmask_e64 sgpr", k_cnd64}, {"v_and_b32 sgpr", k_ands}, {"v_sub_u32", k_sub}, {"v_min_u32", k_min}, {"v_ffbh_u32", k_ffbh}, {"v_not_b32", k_not}, {"v_bfe_u32", k_bfeu}, {"v_ashrrev_i32", k_ashr}, {"v_lshrrev_b32 vgpr", k_lshr}, {"v_min_u32_dpp", k_mindpp}, {"v_max_i32", k_maxi}, {"v_lshl_add_u32", k_lshladd}, {"v_mbcnt_lo", k_mbcnt}, {"v_perm_b32", k_perm}, {"v_fma_f32", k_fma},
        {"v_mov_dpp wave_shl", k_dpp}, {"v_mov_dpp row_shl", k_dpprow}, {"v_lshlrev_b32", k_lshl}, {"v_bfe_i32", k_bfe}, {"v_bfrev_b32", k_bfrev},
        {"v_mov_b32", k_mov}, {"v_cmp_lt_i32", k_cmp}, {"v_lshlrev_b64", k_lshl64}, {"v_addc_co_u32 vop3", k_addc}, {"v_addc pair (64b add)", k_addc_pair},
        {"v_lshl_add_u64", k_lshladd64}, {"v_cmp -> sgpr pair", k_cmp64}, {"v_add_co_u32 vcc", k_addco}, {"v_xor_b32 dependent", k_and_dep}};
    uint32_t* out; uint64_t* cyc;
    for (int wps : {1, 2, 4}) {      // waves per SIMD
        int blocks = cus * 4 * wps;
        hipMalloc(&out, (size_t)blocks * 64 * 4); hipMalloc(&cyc, (size_t)blocks * 8);
        hipMemset(out, 0, (size_t)blocks * 64 * 4);
        printf("--- %d wave(s) per SIMD (%d single-wave workgroups)\n", wps, blocks);
        for (auto& e : es) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(64), 0, 0, out, cyc);   // warm
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(64), 0, 0, out, cyc);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            std::vector<uint64_t> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto v : h) avg += v; avg /= blocks;
            double insts = (double)ITER * OPS_PER_ITER;
            // shader cycles counted by the wavefronts themselves (s_memtime); the clock = those cycles / the wall time of the launch.
            // cycles per instruction per SIMD = a wavefront's cycles / (its instructions x the wavefronts sharing its SIMD)
            printf("%-22s wall %.3f ms  clock %.2f GHz  %.2f cyc/inst/SIMD  (%.2f cyc/inst/wave)\n", e.name, ms,
                   avg / (ms * 1e6), avg / (insts * wps), avg / insts);
        }
        hipFree(out); hipFree(cyc);
    }
    return 0;
}
