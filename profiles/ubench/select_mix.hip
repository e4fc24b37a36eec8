// The VALU instruction mix of k_bs_select's slice loop as a register-only loop (round 6): what does one wave64 instruction of THAT
// mix occupy a SIMD for, at the kernel's four waves per SIMD?  The mix follows the opcode histogram of the compiled kernel
// (hipcc -S of sketch_bs.hip, k_bs_select<12, 11>: 2165 VALU instructions, 47 % of them of the slow class of profiles/ubench/README.md
// -- v_lshl_add_u64 / v_lshl_add_u32 address arithmetic, v_cmp_*, v_lshlrev_b32_sdwa, v_alignbit_b32, v_min_u32, v_lshl_or_b32, v_ffbl_b32):
// per 32 instructions 5 v_add_u32, 5 v_mov_b32, 3 v_lshl_add_u64, 2 v_bitop3_b32, 1 v_lshlrev_b32_sdwa, 4 v_cmp, 2 v_cndmask, ... on
// four independent chains.  Beside it the same loop with the fast-class instructions only and with the slow-class ones only.
// Build + run: hipcc --offload-arch=gfx950 -O3 select_mix.hip -o select_mix && ./select_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 2048
#define MIX32                                                                                                   \
    "v_add_u32 %0, %0, %8\n v_mov_b32 %1, %0\n v_lshl_add_u64 %4, %4, 2, %5\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\n"       \
    "v_add_u32 %1, %1, %9\n v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %3, %3, %1, vcc\n v_mov_b32 %2, %3\n"                     \
    "v_lshlrev_b32_sdwa %0, %10, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32 %2, %2, %0\n" \
    "v_lshl_add_u64 %5, %5, 3, %6\n v_and_b32 %3, %3, %8\n v_cmp_ne_u32 vcc, %1, %9\n v_mov_b32 %0, %2\n"                        \
    "v_lshl_add_u32 %1, %1, 2, %8\n v_sub_u32 %2, %2, %9\n v_add_u32 %3, %3, %1\n v_mov_b64 %6, %7\n"                            \
    "v_cmp_gt_u32 vcc, %2, %8\n v_cndmask_b32 %0, %0, %3, vcc\n v_xor_b32 %1, %1, %9\n v_min_u32 %2, %2, %8\n"                   \
    "v_alignbit_b32 %3, %3, %0, %10\n v_mov_b32 %1, %2\n v_add_u32 %0, %0, %3\n v_lshl_add_u64 %7, %7, 0, %4\n"                  \
    "v_bitop3_b32 %1, %1, %8, %9 bitop3:0x1e\n v_cmp_eq_u32 vcc, %0, %9\n v_lshrrev_b32 %2, 3, %2\n v_mov_b32 %3, %0\n"          \
    "v_lshl_or_b32 %0, %0, 4, %8\n v_ffbl_b32 %1, %1\n"
#define FAST32                                                                                                  \
    "v_add_u32 %0, %0, %8\n v_mov_b32 %1, %0\n v_add_u32 %2, %2, %9\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\n"               \
    "v_add_u32 %1, %1, %9\n v_xor_b32 %3, %3, %8\n v_cndmask_b32 %3, %3, %1, vcc\n v_mov_b32 %2, %3\n"                         \
    "v_and_b32 %0, %0, %8\n v_add_u32 %2, %2, %0\n v_sub_u32 %1, %1, %9\n v_and_b32 %3, %3, %8\n"                              \
    "v_or_b32 %0, %0, %9\n v_mov_b32 %0, %2\n v_add_u32 %1, %1, %8\n v_sub_u32 %2, %2, %9\n"                                   \
    "v_add_u32 %3, %3, %1\n v_mov_b32 %0, %3\n v_xor_b32 %1, %1, %9\n v_cndmask_b32 %0, %0, %3, vcc\n"                         \
    "v_xor_b32 %1, %1, %9\n v_and_b32 %2, %2, %8\n v_lshrrev_b32 %3, 1, %3\n v_mov_b32 %1, %2\n"                               \
    "v_add_u32 %0, %0, %3\n v_or_b32 %2, %2, %9\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0x1e\n v_not_b32 %0, %0\n"                \
    "v_lshrrev_b32 %2, 3, %2\n v_mov_b32 %3, %0\n v_add_u32 %0, %0, %8\n v_sub_u32 %1, %1, %9\n"
#define SLOW32                                                                                                  \
    "v_lshl_add_u64 %4, %4, 2, %5\n v_cmp_lt_u32 vcc, %0, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_min_u32 %2, %2, %8\n"          \
    "v_alignbit_b32 %3, %3, %0, %10\n v_lshl_add_u64 %5, %5, 3, %6\n v_cmp_ne_u32 vcc, %1, %9\n v_lshl_or_b32 %0, %0, 4, %8\n"  \
    "v_ffbl_b32 %1, %1\n v_lshlrev_b32_sdwa %2, %10, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"    \
    "v_cmp_gt_u32 vcc, %2, %8\n v_lshl_add_u64 %6, %6, 0, %7\n v_min_u32 %3, %3, %9\n v_lshl_add_u32 %0, %0, 1, %9\n"           \
    "v_alignbit_b32 %1, %1, %2, %10\n v_cmp_eq_u32 vcc, %0, %9\n v_lshl_add_u64 %7, %7, 1, %4\n v_lshl_or_b32 %2, %2, 2, %8\n"   \
    "v_lshlrev_b32 %3, 1, %3\n v_cmp_lt_u32 vcc, %3, %8\n v_lshl_add_u32 %1, %1, 3, %8\n v_min_u32 %0, %0, %9\n"                \
    "v_alignbit_b32 %2, %2, %3, %10\n v_lshl_add_u64 %4, %4, 2, %6\n v_cmp_ne_u32 vcc, %2, %9\n v_ffbl_b32 %3, %3\n"            \
    "v_lshl_or_b32 %1, %1, 4, %9\n v_lshl_add_u32 %2, %2, 2, %9\n v_cmp_gt_u32 vcc, %1, %8\n v_min_u32 %3, %3, %8\n"
#define KERNEL(NAME, BODY)                                                                                      \
    __global__ __launch_bounds__(1024) void NAME(unsigned *out, unsigned seed)                                  \
    {                                                                                                           \
        unsigned a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = seed | 1, f = seed ^ 0x1234567, g = 5;   \
        unsigned long long qa = a, qb = b, qc = c, qd = d;                                                      \
        for (int it = 0; it < N_IT; ++it)                                                                       \
            asm volatile(BODY : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(qa), "+v"(qb), "+v"(qc), "+v"(qd) : "v"(e), "v"(f), "v"(g) : "vcc"); \
        out[blockIdx.x * 1024 + threadIdx.x] = a ^ b ^ c ^ d ^ (unsigned)(qa ^ qb ^ qc ^ qd);                   \
    }
KERNEL(k_mix, MIX32)
KERNEL(k_fast, FAST32)
KERNEL(k_slow, SLOW32)
template <class K>
static double run(K kern, int blocks, unsigned *d_out)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 0, 0, d_out, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 0, 0, d_out, 2u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}
int main()
{
    unsigned *d;
    hipMalloc(&d, 1024u * 1024u * 4u);
    printf("# register-only loops of 32 wave64 VALU instructions x %d iterations, blocks of 16 waves, cycles at 2.4 GHz per instruction per SIMD\n", N_IT);
    printf("# %-28s %10s %10s %10s\n", "loop", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
    const char *names[] = {"k_bs_select's mix (47 % slow)", "fast-class only", "slow-class only"};
    for (int k = 0; k < 3; ++k) {
        double cyc[3];
        int q = 0;
        for (int blocks : {64, 128, 256}) {  // 16 waves per block on 256 CUs: 1, 2, 4 waves per SIMD
            const double ms = k == 0 ? run(k_mix, blocks, d) : k == 1 ? run(k_fast, blocks, d) : run(k_slow, blocks, d);
            const double instr_per_simd = (double)blocks * 16 * N_IT * 32 / 1024.0;
            cyc[q++] = ms * 1e-3 * 2.4e9 / instr_per_simd;
        }
        printf("  %-28s %10.2f %10.2f %10.2f\n", names[k], cyc[0], cyc[1], cyc[2]);
    }
    return 0;
}
