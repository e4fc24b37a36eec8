// Which instruction could do the filter's left shifts (gen/bs_gen.py: chains of v_add_u32 x, x -- 16 + 8 of them per register pair in the
// transposes' stages 16 and 8) without slowing the stream?  One candidate per four instructions among fast-class ones (the mix that
// showed "a slow-class instruction slows the whole stream", profiles/ubench/README.md), cycles per wave64 instruction per SIMD at 2.4 GHz.
// Build + run: hipcc --offload-arch=gfx950 -O3 shift_candidates.hip -o shift_candidates && ./shift_candidates
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 4096
#define FAST3 "v_xor_b32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_and_b32 %2, %2, %4\n"
#define REP8(X) X X X X X X X X
#define KERNEL(NAME, CAND)                                                                                      \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed)                                   \
    {                                                                                                           \
        unsigned a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = (a + 7) & 0xFFFF, e = seed | 1, f = seed ^ 0x1234567; \
        for (int it = 0; it < N_IT; ++it)                                                                       \
            asm volatile(REP8(FAST3 CAND) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "s"(seed));    \
        out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d;                                                    \
    }
KERNEL(k_fast, "v_or_b32 %3, %3, %5\n")
KERNEL(k_lshl, "v_lshlrev_b32 %3, 1, %3\n")
KERNEL(k_mul_u24, "v_mul_u32_u24 %3, 0x100, %3\n")
KERNEL(k_mul_i24, "v_mul_i32_i24 %3, 0x100, %3\n")
KERNEL(k_mul_hi_u24, "v_mul_hi_u32_u24 %3, %3, %4\n")
KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %3, %3, %4\n")
KERNEL(k_lshl_add, "v_lshl_add_u32 %3, %3, 1, %4\n")
KERNEL(k_add_lshl, "v_add_lshl_u32 %3, %3, %4, 1\n")
KERNEL(k_lshl_b16, "v_lshlrev_b16 %3, 1, %3\n")
KERNEL(k_mul_lo_u16, "v_mul_lo_u16 %3, %3, %4\n")
KERNEL(k_pk_lshl, "v_pk_lshlrev_b16 %3, 1, %3\n")
KERNEL(k_mul_f32, "v_mul_f32 %3, 2.0, %3\n")
KERNEL(k_ldexp, "v_ldexp_f32 %3, %3, 1\n")
KERNEL(k_perm, "v_perm_b32 %3, %3, %4, %5\n")
KERNEL(k_alignbyte, "v_alignbyte_b32 %3, %3, %4, 2\n")
KERNEL(k_swap, "v_swap_b32 %3, %2\n")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_mad_u16, "v_mad_u16 %3, %3, %4, %5\n")
KERNEL(k_cvt_pk, "v_cvt_pk_u16_u32 %3, %3, %4\n")
KERNEL(k_sad, "v_sad_u32 %3, %3, %4, %5\n")
KERNEL(k_add_u32_self, "v_add_u32 %3, %3, %3\n")
template <class K>
static double run(K kern, int blocks, unsigned *d_out)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 2u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}
#define ROW(NAME, K)                                                                        \
    {                                                                                       \
        const double ms = run(K, 512, d);                                                   \
        printf("  %-22s %6.2f\n", NAME, ms * 1e-3 * 2.4e9 / ((double)512 * 4 * N_IT * 32 / 1024.0)); \
    }
int main()
{
    unsigned *d;
    hipMalloc(&d, 512u * 256u * 4u);
    printf("# one candidate in four among v_xor / v_add_u32 / v_and (32 instructions per iteration), two waves per SIMD: cycles per instruction per SIMD\n");
    ROW("v_or_b32 (all fast)", k_fast) ROW("v_add_u32 x, x", k_add_u32_self) ROW("v_lshlrev_b32", k_lshl) ROW("v_mul_u32_u24", k_mul_u24) ROW("v_mul_i32_i24", k_mul_i24)
    ROW("v_mul_hi_u32_u24", k_mul_hi_u24) ROW("v_mul_lo_u32", k_mul_lo_u32) ROW("v_lshl_add_u32", k_lshl_add) ROW("v_add_lshl_u32", k_add_lshl)
    ROW("v_lshlrev_b16", k_lshl_b16) ROW("v_mul_lo_u16", k_mul_lo_u16) ROW("v_pk_lshlrev_b16", k_pk_lshl) ROW("v_mul_f32", k_mul_f32) ROW("v_ldexp_f32", k_ldexp)
    ROW("v_perm_b32", k_perm) ROW("v_alignbyte_b32", k_alignbyte) ROW("v_swap_b32", k_swap) ROW("v_mov_b32_dpp", k_mov_dpp) ROW("v_mad_u16", k_mad_u16)
    ROW("v_cvt_pk_u16_u32", k_cvt_pk) ROW("v_sad_u32", k_sad)
    return 0;
}
