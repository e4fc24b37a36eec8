// Instruction-throughput microbenchmark for the integer ops of the ntHash step on gfx950 (evidence for DESIGN.md).
// Each kernel runs N_IT iterations of 32 copies of ONE instruction on 4 independent register chains; reports
// cycles per wave-instruction per SIMD at W waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define N_IT 4096

#define REP8(X) X X X X X X X X
#define BODY32(A, B, C, D) REP8(A B C D)

#define KERNEL(NAME, I0, I1, I2, I3)                                                              \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed)                     \
    {                                                                                             \
        unsigned a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = a + 7;                  \
        unsigned e = seed | 1, f = seed ^ 0x1234567;                                              \
        unsigned long long qa = a, qb = b, qc = c, qd = d;                                        \
        for (int it = 0; it < N_IT; ++it) {                                                       \
            asm volatile(BODY32(I0, I1, I2, I3)                                                   \
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(qa), "+v"(qb), "+v"(qc), "+v"(qd) \
                         : "v"(e), "v"(f), "s"(seed));                                            \
        }                                                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ (unsigned)(qa ^ qb ^ qc ^ qd);      \
    }

KERNEL(k_xor, "v_xor_b32 %0, %0, %8\n", "v_xor_b32 %1, %1, %8\n", "v_xor_b32 %2, %2, %8\n", "v_xor_b32 %3, %3, %8\n")
KERNEL(k_and, "v_and_b32 %0, %0, %8\n", "v_and_b32 %1, %1, %8\n", "v_and_b32 %2, %2, %8\n", "v_and_b32 %3, %3, %8\n")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0\n", "v_lshlrev_b32 %1, 1, %1\n", "v_lshlrev_b32 %2, 1, %2\n", "v_lshlrev_b32 %3, 1, %3\n")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %8, 31\n", "v_alignbit_b32 %1, %1, %8, 31\n", "v_alignbit_b32 %2, %2, %8, 31\n", "v_alignbit_b32 %3, %3, %8, 31\n")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %8, %9 bitop3:0x1e\n", "v_bitop3_b32 %1, %1, %8, %9 bitop3:0x1e\n", "v_bitop3_b32 %2, %2, %8, %9 bitop3:0x1e\n", "v_bitop3_b32 %3, %3, %8, %9 bitop3:0x1e\n")
KERNEL(k_or3, "v_or3_b32 %0, %0, %8, %9\n", "v_or3_b32 %1, %1, %8, %9\n", "v_or3_b32 %2, %2, %8, %9\n", "v_or3_b32 %3, %3, %8, %9\n")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %8\n", "v_lshl_or_b32 %1, %1, 1, %8\n", "v_lshl_or_b32 %2, %2, 1, %8\n", "v_lshl_or_b32 %3, %3, 1, %8\n")
KERNEL(k_add64, "v_lshl_add_u64 %4, %4, 0, %5\n", "v_lshl_add_u64 %5, %5, 0, %6\n", "v_lshl_add_u64 %6, %6, 0, %7\n", "v_lshl_add_u64 %7, %7, 0, %4\n")
KERNEL(k_add32, "v_add_u32 %0, %0, %8\n", "v_add_u32 %1, %1, %8\n", "v_add_u32 %2, %2, %8\n", "v_add_u32 %3, %3, %8\n")
KERNEL(k_addco, "v_add_co_u32 %0, vcc, %0, %8\n", "v_addc_co_u32 %1, vcc, %1, %9, vcc\n", "v_add_co_u32 %2, vcc, %2, %8\n", "v_addc_co_u32 %3, vcc, %3, %9, vcc\n")
KERNEL(k_cmp, "v_cmp_gt_u32 vcc, %10, %0\n", "v_cmp_gt_u32 vcc, %10, %1\n", "v_cmp_gt_u32 vcc, %10, %2\n", "v_cmp_gt_u32 vcc, %10, %3\n")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 2, 30\n", "v_bfe_u32 %1, %1, 2, 30\n", "v_bfe_u32 %2, %2, 2, 30\n", "v_bfe_u32 %3, %3, 2, 30\n")
KERNEL(k_mov, "v_mov_b32 %0, %8\n", "v_mov_b32 %1, %8\n", "v_mov_b32 %2, %8\n", "v_mov_b32 %3, %8\n")
// round 3: the instructions of the bit-sliced ring filter (hash_bs) and candidates for it
KERNEL(k_perm, "v_perm_b32 %0, %0, %8, %9\n", "v_perm_b32 %1, %1, %8, %9\n", "v_perm_b32 %2, %2, %8, %9\n", "v_perm_b32 %3, %3, %8, %9\n")
KERNEL(k_xnor, "v_xnor_b32 %0, %0, %8\n", "v_xnor_b32 %1, %1, %8\n", "v_xnor_b32 %2, %2, %8\n", "v_xnor_b32 %3, %3, %8\n")
KERNEL(k_not, "v_not_b32 %0, %0\n", "v_not_b32 %1, %1\n", "v_not_b32 %2, %2\n", "v_not_b32 %3, %3\n")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %8, %0\n", "v_bcnt_u32_b32 %1, %8, %1\n", "v_bcnt_u32_b32 %2, %8, %2\n", "v_bcnt_u32_b32 %3, %8, %3\n")
KERNEL(k_bfi, "v_bfi_b32 %0, %8, %0, %9\n", "v_bfi_b32 %1, %8, %1, %9\n", "v_bfi_b32 %2, %8, %2, %9\n", "v_bfi_b32 %3, %8, %3, %9\n")
KERNEL(k_bitop3s, "v_bitop3_b32 %0, %0, %8, %10 bitop3:0x8e\n", "v_bitop3_b32 %1, %1, %8, %10 bitop3:0x8e\n", "v_bitop3_b32 %2, %2, %8, %10 bitop3:0x8e\n", "v_bitop3_b32 %3, %3, %8, %10 bitop3:0x8e\n")
KERNEL(k_max, "v_max_u32 %0, %0, %8\n", "v_max_u32 %1, %1, %8\n", "v_max_u32 %2, %2, %8\n", "v_max_u32 %3, %3, %8\n")
KERNEL(k_add3, "v_add3_u32 %0, %0, %8, %9\n", "v_add3_u32 %1, %1, %8, %9\n", "v_add3_u32 %2, %2, %8, %9\n", "v_add3_u32 %3, %3, %8, %9\n")
KERNEL(k_sdwa, "v_mov_b32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n", "v_mov_b32_sdwa %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n", "v_mov_b32_sdwa %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n", "v_mov_b32_sdwa %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n")
KERNEL(k_lshr, "v_lshrrev_b32 %0, 1, %0\n", "v_lshrrev_b32 %1, 1, %1\n", "v_lshrrev_b32 %2, 1, %2\n", "v_lshrrev_b32 %3, 1, %3\n")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %8, %9\n", "v_mad_u32_u24 %1, %1, %8, %9\n", "v_mad_u32_u24 %2, %2, %8, %9\n", "v_mad_u32_u24 %3, %3, %8, %9\n")

// LDS: one ds_read_b128 + wait per "step", table of 20 x 16 B, random entry per lane
__global__ __launch_bounds__(256) void k_ds128(unsigned *out, unsigned seed)
{
    __shared__ uint4 tab[20];
    if (threadIdx.x < 20) tab[threadIdx.x] = make_uint4(threadIdx.x, seed, 3, 4);
    __syncthreads();
    unsigned idx = (threadIdx.x * 7 + seed) & 15, acc = 0;
    for (int it = 0; it < N_IT * 8; ++it) {
        uint4 t = tab[idx];
        acc ^= t.x ^ t.w;
        idx = (idx + t.y + acc) & 15;  // dependent address: exposes latency
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <class K>
static double run(K kern, int blocks, unsigned *d_out)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 2u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    unsigned *d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * 4);
    const double ghz = 2.4;  // nominal; GRBM_GUI_ACTIVE showed 2.38 GHz for the hash kernel
    struct { const char *name; void (*k)(unsigned *, unsigned); int per_it; } ks[] = {
        {"v_xor_b32", k_xor, 32}, {"v_and_b32", k_and, 32}, {"v_lshlrev_b32", k_lshl, 32}, {"v_alignbit_b32", k_alignbit, 32},
        {"v_bitop3_b32", k_bitop3, 32}, {"v_or3_b32", k_or3, 32}, {"v_lshl_or_b32", k_lshl_or, 32}, {"v_lshl_add_u64", k_add64, 32},
        {"v_add_u32", k_add32, 32}, {"v_add_co/addc pair", k_addco, 32}, {"v_cmp_gt_u32", k_cmp, 32}, {"v_bfe_u32", k_bfe, 32},
        {"v_mov_b32", k_mov, 32}, {"v_perm_b32", k_perm, 32}, {"v_xnor_b32", k_xnor, 32}, {"v_not_b32", k_not, 32},
        {"v_bcnt_u32_b32", k_bcnt, 32}, {"v_bfi_b32", k_bfi, 32}, {"v_bitop3_b32 (sgpr src2)", k_bitop3s, 32},
        {"v_max_u32", k_max, 32}, {"v_add3_u32", k_add3, 32}, {"v_mov_b32_sdwa", k_sdwa, 32}, {"v_lshrrev_b32", k_lshr, 32},
        {"v_mad_u32_u24", k_mad24, 32}};
    printf("%-22s %10s %10s %10s\n", "instruction", "cyc@1w/SIMD", "cyc@2w", "cyc@4w");
    for (auto &e : ks) {
        double c[3];
        int wps[3] = {1, 2, 4};
        for (int i = 0; i < 3; ++i) {
            int blocks = 256 * wps[i];  // 256 CUs x (4 waves per block = 1 wave per SIMD) x wps
            double ms = run(e.k, blocks, d_out);
            double instr_per_simd = (double)N_IT * e.per_it * wps[i];
            c[i] = ms * 1e-3 * ghz * 1e9 / instr_per_simd;
        }
        printf("%-22s %10.2f %10.2f %10.2f\n", e.name, c[0], c[1], c[2]);
    }
    for (int wps : {1, 2, 4, 8}) {
        double ms = run(k_ds128, 256 * wps, d_out);
        printf("ds_read_b128 dependent: %d waves/SIMD: %.1f cycles per read per wave (latency-bound chain)\n", wps,
               ms * 1e-3 * ghz * 1e9 / (N_IT * 8.0));
    }
    hipFree(d_out);
    return 0;
}
