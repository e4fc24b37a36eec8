#!/usr/bin/env python3
"""Generates issue_bench.hip: what limits a long straight-line stream of full-rate VALU on gfx950?
(1) register banks: v_bitop3_b32 d, d, a, b with the three sources in different / equal banks (register index mod 4);
(2) code size: the same instruction mix unrolled to 2 .. 64 KB inside a loop (instruction cache: 64 KB per two CUs);
(3) dependent issue: chains of length 1 / 2 / 4 / 8 independent streams.
Usage: python gen_issue_bench.py > issue_bench.hip; hipcc --offload-arch=gfx950 -O3 issue_bench.hip -o issue_bench"""
import sys

out = []
w = out.append
w("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdlib>\n")
kernels = []


def kernel(name, body_lines, iters, clob):
    """body_lines: asm lines over physical registers"""
    txt = "".join(f'        "{l}\\n"\n' for l in body_lines)
    cl = ", ".join(f'"{c}"' for c in clob)
    w(f"__global__ __launch_bounds__(256) void {name}(unsigned *out, unsigned seed)\n{{\n")
    w(f"    unsigned it = {iters};\n    asm volatile(\n")
    w('        "s_mov_b32 s40, %0\\n"\n        "s_mov_b64 s[42:43], %2\\n"\n')
    for r in clob:
        if r.startswith('v'):
            w(f'        "v_mov_b32 {r}, %1\\n"\n')
    w('        "L_%=:\\n"\n')
    w(txt)
    w('        "s_sub_u32 s40, s40, 1\\n"\n        "s_cmp_lg_u32 s40, 0\\n"\n        "s_cbranch_scc1 L_%=\\n"\n')
    w(f'        : : "s"(it), "v"(seed), "s"(out) : {cl}, "s40", "s41", "s42", "s43", "vcc", "scc", "memory");\n')
    w("    out[blockIdx.x * 256 + threadIdx.x] = seed;\n}\n")
    kernels.append((name, len(body_lines) * iters))


regs = [f"v{i}" for i in range(32, 160)]
# (1) banks
for tag, pick in (("bank_diff", lambda i: (32 + 4 * i, 97 + 4 * (i % 8), 130 + 4 * (i % 6))),   # banks 0,1,2
                  ("bank_2same", lambda i: (32 + 4 * i, 96 + 4 * (i % 8), 130 + 4 * (i % 6))),  # banks 0,0,2
                  ("bank_3same", lambda i: (32 + 4 * i, 96 + 4 * (i % 8), 128 + 4 * (i % 6)))):  # banks 0,0,0
    body = []
    for rep in range(8):
        for i in range(16):
            d, a, b = pick(i)
            body.append(f"v_bitop3_b32 v{d}, v{d}, v{a}, v{b} bitop3:0x96")
    kernel(f"k_{tag}", body, 4096, regs)
# VOP2 xor: banks
for tag, pick in (("xor_diff", lambda i: (32 + 4 * i, 97 + 4 * (i % 8))), ("xor_same", lambda i: (32 + 4 * i, 96 + 4 * (i % 8)))):
    body = []
    for rep in range(8):
        for i in range(16):
            d, a = pick(i)
            body.append(f"v_xor_b32 v{d}, v{d}, v{a}")
    kernel(f"k_{tag}", body, 4096, regs)
# (1b) destination banks: dst is another register than the sources
def pat(name, fmt, regs_of):
    body = []
    for rep in range(8):
        for i in range(16):
            body.append(fmt.format(*regs_of(i)))
    kernel(name, body, 4096, regs)
B3 = "v_bitop3_b32 v{0}, v{1}, v{2}, v{3} bitop3:0x96"
# sources in banks 0,1,2; dst: same register as src0 / other register of bank 0 / bank 3 / bank 1 / bank 2
pat("k_d_eq_s0", B3, lambda i: (32 + 4 * i, 32 + 4 * i, 97 + 4 * (i % 8), 130 + 4 * (i % 6)))
pat("k_d_bank_s0", B3, lambda i: (36 + 4 * ((i + 5) % 16), 32 + 4 * i, 97 + 4 * (i % 8), 130 + 4 * (i % 6)))
pat("k_d_bank_free", B3, lambda i: (35 + 4 * i, 32 + 4 * i, 97 + 4 * (i % 8), 130 + 4 * (i % 6)))
pat("k_d_bank_s1", B3, lambda i: (33 + 4 * i, 32 + 4 * i, 97 + 4 * (i % 8), 130 + 4 * (i % 6)))
pat("k_d_bank_s2", B3, lambda i: (34 + 4 * i, 32 + 4 * i, 97 + 4 * (i % 8), 130 + 4 * (i % 6)))
X2 = "v_xor_b32 v{0}, v{1}, v{2}"
pat("k_x_d_eq_s0", X2, lambda i: (32 + 4 * i, 32 + 4 * i, 97 + 4 * (i % 8)))
pat("k_x_d_bank_s0", X2, lambda i: (36 + 4 * ((i + 5) % 16), 32 + 4 * i, 97 + 4 * (i % 8)))
pat("k_x_d_bank_s1", X2, lambda i: (33 + 4 * i, 32 + 4 * i, 97 + 4 * (i % 8)))
pat("k_x_d_free", X2, lambda i: (34 + 4 * i, 32 + 4 * i, 97 + 4 * (i % 8)))
pat("k_x_s_samebank", X2, lambda i: (34 + 4 * i, 32 + 4 * i, 96 + 4 * (i % 8)))
S1 = "v_lshrrev_b32 v{0}, 4, v{1}"
pat("k_shr_same", S1, lambda i: (32 + 4 * i, 32 + 4 * i))
pat("k_shr_bank", S1, lambda i: (36 + 4 * ((i + 5) % 16), 32 + 4 * i))
pat("k_shr_other", S1, lambda i: (33 + 4 * i, 32 + 4 * i))
S2 = "v_lshlrev_b32 v{0}, 4, v{1}"
pat("k_shl_other", S2, lambda i: (33 + 4 * i, 32 + 4 * i))
pat("k_add_other", "v_add_u32 v{0}, v{1}, v{1}", lambda i: (33 + 4 * i, 32 + 4 * i))
pat("k_mov_other", "v_mov_b32 v{0}, v{1}", lambda i: (33 + 4 * i, 32 + 4 * i))
pat("k_or_other", "v_or_b32 v{0}, v{1}, v{2}", lambda i: (34 + 4 * i, 32 + 4 * i, 97 + 4 * (i % 8)))
pat("k_b3_sgpr", "v_bitop3_b32 v{0}, v{1}, v{0}, s41 bitop3:0x8e", lambda i: (32 + 4 * i, 97 + 4 * (i % 8)))
pat("k_b3_2src", "v_bitop3_b32 v{0}, v{1}, v{2}, v{2} bitop3:0x8e", lambda i: (32 + 4 * i, 97 + 4 * (i % 8), 130 + 4 * (i % 6)))
pat("k_alignbit", "v_alignbit_b32 v{0}, v{1}, v{2}, 1", lambda i: (34 + 4 * i, 32 + 4 * i, 97 + 4 * (i % 8)))

# (1c) one "slow" instruction among three full-rate ones (v_lshrrev_b32 / v_xor_b32 / v_bitop3_b32): what does it really cost?
def mixed(name, slow_fmt, filler="v_lshrrev_b32 v{0}, 1, v{0}"):
    body = []
    for rep in range(8):
        for i in range(16):
            r = 32 + 4 * i
            if i % 4 == 3:
                body.append(slow_fmt.format(r, 97 + 4 * (i % 8), 130 + 4 * (i % 6), r + 1))
            else:
                body.append(filler.format(r, 97 + 4 * (i % 8), 130 + 4 * (i % 6), r + 1))
    kernel(name, body, 4096, regs)
mixed("k_mix_none", "v_lshrrev_b32 v{0}, 1, v{0}")
mixed("k_mix_lshl", "v_lshlrev_b32 v{0}, 4, v{0}")
mixed("k_mix_lshl1", "v_lshlrev_b32 v{0}, 1, v{0}")
mixed("k_mix_alignbit", "v_alignbit_b32 v{0}, v{0}, v{1}, 28")
mixed("k_mix_perm", "v_perm_b32 v{0}, v{0}, v{1}, s41")
mixed("k_mix_permv", "v_perm_b32 v{0}, v{0}, v{1}, v{2}")
mixed("k_mix_sdwa", "v_mov_b32_sdwa v{0}, v{0} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
mixed("k_mix_sdwa_or", "v_or_b32_sdwa v{0}, v{0}, v{1} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
mixed("k_mix_bfe", "v_bfe_u32 v{0}, v{0}, 4, 8")
mixed("k_mix_bcnt", "v_bcnt_u32_b32 v{0}, v{1}, v{0}")
mixed("k_mix_b3sgpr", "v_bitop3_b32 v{0}, v{0}, v{1}, s41 bitop3:0x96")
mixed("k_mix_lshl_add", "v_lshl_add_u32 v{0}, v{0}, 4, v{1}")
mixed("k_mix_lshl_or", "v_lshl_or_b32 v{0}, v{0}, 4, v{1}")
mixed("k_mix_mul24", "v_mul_u32_u24 v{0}, v{0}, v{1}")
mixed("k_mix_mullo", "v_mul_lo_u32 v{0}, v{0}, v{1}")
mixed("k_mix_cmp", "v_cmp_gt_u32 vcc, v{0}, v{1}")
mixed("k_mix_pk_lshl", "v_pk_lshlrev_b16 v{0}, 4, v{0}")
mixed("k_mix_lshl64", "v_lshlrev_b64 v[{0}:{3}], 4, v[{0}:{3}]")
mixed("k_mix_lshl_x", "v_lshlrev_b32 v{0}, 4, v{0}", "v_xor_b32 v{0}, v{0}, v{1}")
mixed("k_mix_lshl_b3", "v_lshlrev_b32 v{0}, 4, v{0}", "v_bitop3_b32 v{0}, v{0}, v{1}, v{2} bitop3:0x96")
mixed("k_mix_ashr", "v_ashrrev_i32 v{0}, 4, v{0}")
mixed("k_mix_addco", "v_add_co_u32 v{0}, vcc, v{0}, v{1}")
mixed("k_mix_dpp", "v_mov_b32_dpp v{0}, v{1} row_shr:1 row_mask:0xf bank_mask:0xf")
mixed("k_mix_readlane", "v_readlane_b32 s42, v{0}, 3")

# (1d) how long does a slow instruction slow the stream down?  S slow (v_lshlrev_b32) then F fast (v_lshrrev_b32), repeated
def runs(name, S, F, slow="v_lshlrev_b32 v{0}, 4, v{0}", fast="v_lshrrev_b32 v{0}, 1, v{0}"):
    body = []
    i = 0
    while len(body) < 512:
        for _ in range(S):
            body.append(slow.format(32 + 4 * (i % 16), 97 + 4 * (i % 8), 130 + 4 * (i % 6)))
            i += 1
        for _ in range(F):
            body.append(fast.format(32 + 4 * (i % 16), 97 + 4 * (i % 8), 130 + 4 * (i % 6)))
            i += 1
    kernel(name, body, 1024, regs)
for S, F in ((1, 1), (1, 3), (1, 7), (1, 15), (1, 31), (1, 63), (2, 6), (4, 12), (8, 24), (16, 48), (32, 96), (4, 28), (16, 112)):
    runs(f"k_run_{S}_{F}", S, F)
runs("k_runb3_4_28", 4, 28, fast="v_bitop3_b32 v{0}, v{0}, v{1}, v{2} bitop3:0x96")
runs("k_runperm_4_28", 4, 28, slow="v_perm_b32 v{0}, v{0}, v{1}, s41")
runs("k_runperm_16_48", 16, 48, slow="v_perm_b32 v{0}, v{0}, v{1}, s41")
mixed("k_mix_bfrev", "v_bfrev_b32 v{0}, v{0}")
mixed("k_mix_lshr64", "v_lshrrev_b64 v[{0}:{3}], 4, v[{0}:{3}]")
mixed("k_mix_sub", "v_sub_u32 v{0}, v{0}, v{1}")
mixed("k_mix_xad", "v_xad_u32 v{0}, v{0}, v{1}, v{2}")
mixed("k_mix_max", "v_max_u32 v{0}, v{0}, v{1}")
mixed("k_mix_cndmask", "v_cndmask_b32 v{0}, v{0}, v{1}, vcc")
mixed("k_mix_ffbh", "v_ffbh_u32 v{0}, v{0}")
mixed("k_mix_addf", "v_add_f32 v{0}, v{0}, v{1}")
mixed("k_mix_fma", "v_fma_f32 v{0}, v{0}, v{1}, v{2}")
mixed("k_mix_pkadd16", "v_pk_add_u16 v{0}, v{0}, v{1}")
mixed("k_mix_and_or", "v_and_or_b32 v{0}, v{0}, v{1}, v{2}")
mixed("k_mix_or3", "v_or3_b32 v{0}, v{0}, v{1}, v{2}")
mixed("k_mix_lshl_lit", "v_lshlrev_b32 v{0}, v{1}, v{0}")
mixed("k_mix_mulf", "v_mul_f32 v{0}, v{0}, v{1}")
mixed("k_mix_dsread", "ds_read_b32 v{0}, v{1}")

mixed("k_mix_pack", "v_pack_b32_f16 v{0}, v{0}, v{1}")
mixed("k_mix_pack_hi", "v_pack_b32_f16 v{0}, v{0}, v{1} op_sel:[1,1,0]")
mixed("k_mix_cvtpk_u16", "v_cvt_pk_u16_u32 v{0}, v{0}, v{1}")
mixed("k_mix_cvt_f32_u32", "v_cvt_f32_u32 v{0}, v{0}")
mixed("k_mix_cvt_u32_f32", "v_cvt_u32_f32 v{0}, v{0}")
mixed("k_mix_cvt_ubyte1", "v_cvt_f32_ubyte1 v{0}, v{0}")
mixed("k_mix_cvt_pk_u8", "v_cvt_pk_u8_f32 v{0}, v{0}, v{1}, v{2}")
mixed("k_mix_ldexp", "v_ldexp_f32 v{0}, v{0}, v{1}")
mixed("k_mix_alignbyte", "v_alignbyte_b32 v{0}, v{0}, v{1}, 1")
mixed("k_mix_sad_u32", "v_sad_u32 v{0}, v{0}, v{1}, v{2}")
mixed("k_mix_med3", "v_med3_u32 v{0}, v{0}, v{1}, v{2}")
mixed("k_mix_min", "v_min_u32 v{0}, v{0}, v{1}")
mixed("k_mix_mbcnt", "v_mbcnt_lo_u32_b32 v{0}, v{1}, v{0}")
mixed("k_mix_subrev", "v_subrev_u32 v{0}, v{0}, v{1}")
mixed("k_mix_addc", "v_addc_co_u32 v{0}, vcc, v{0}, v{1}, vcc")
mixed("k_mix_mov_lit", "v_mov_b32 v{0}, 0x12345678")
mixed("k_mix_xor_lit", "v_xor_b32 v{0}, 0x12345678, v{0}")
mixed("k_mix_and_sgpr", "v_and_b32 v{0}, s41, v{0}")
mixed("k_mix_bfm", "v_bfm_b32 v{0}, v{0}, v{1}")
mixed("k_mix_lshlrev16", "v_lshlrev_b16 v{0}, 4, v{0}")
mixed("k_mix_add_u16", "v_add_u16 v{0}, v{0}, v{1}")
mixed("k_mix_mul_lo_u16", "v_mul_lo_u16 v{0}, v{0}, v{1}")
mixed("k_mix_mad_u16", "v_mad_u16 v{0}, v{0}, v{1}, v{2}")
mixed("k_mix_mul_i24", "v_mul_i32_i24 v{0}, v{0}, v{1}")
mixed("k_mix_fmac", "v_fmac_f32 v{0}, v{1}, v{2}")
mixed("k_mix_swap", "v_swap_b32 v{0}, v{1}")
mixed("k_mix_accvgpr", "v_accvgpr_write_b32 a{0}, v{1}")
mixed("k_mix_gload", "global_load_dword v{0}, v{1}, s[42:43]")
mixed("k_mix_writelane", "v_writelane_b32 v{0}, s41, 5")
mixed("k_mix_snop", "s_nop 0")
mixed("k_mix_salu", "s_add_u32 s41, s41, 1")

# (3) dependent chains
for ch in (1, 2, 4, 8):
    body = []
    for i in range(128):
        d = 32 + (i % ch)
        body.append(f"v_bitop3_b32 v{d}, v{d}, v100, v121 bitop3:0x96")
    kernel(f"k_chain_{ch}", body, 4096, regs)

w("""
template <class K> static double run(K kern, int blocks, unsigned *d_out)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1u); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 2u); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms;
}
int main()
{
    unsigned *d_out; hipMalloc(&d_out, 256 * 8 * 256 * 4);
    struct { const char *name; void (*k)(unsigned *, unsigned); double instr; } ks[] = {
""")
for name, n in kernels:
    w(f'        {{"{name}", {name}, {float(n)}}},\n')
w("""    };
    printf("%-16s %10s %10s %10s   (cycles per wave-instruction per SIMD at 2.4 GHz)\\n", "kernel", "1 w/SIMD", "2 w/SIMD", "3 w/SIMD");
    for (auto &e : ks) {
        double c[3];
        for (int wps = 1; wps <= 3; ++wps) {
            double ms = run(e.k, 256 * wps, d_out);
            c[wps - 1] = ms * 1e-3 * 2.4e9 / (e.instr * wps);
        }
        printf("%-16s %10.2f %10.2f %10.2f\\n", e.name, c[0], c[1], c[2]);
    }
    return 0;
}
""")
sys.stdout.write("".join(out))
