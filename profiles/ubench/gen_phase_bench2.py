#!/usr/bin/env python3
"""phase_bench2.hip: the chunk body of the bit-sliced ring filter (ntjoin_amd/csrc/gen/bs_gen.py) as loops over one chunk's
data with parts taken out: registers only, + scalar loads, + vector loads, + stores, warm-up / test / roll alone."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bs_gen_v2 as G  # noqa: E402   (the generator as it was when this was measured: planes from a pre-transposed copy)

g = G.Gen(32)
body = g.chunk()
VAL = ('xor', 'and', 'or', 'mov', 'bitop3', 'add', 'lshr')
variants = {}
variants['regs_only'] = [i for i in body if i[0] in VAL]
variants['regs+gload'] = [i for i in body if i[0] in VAL or i[0] == 'gload1' or (i[0] == 'waitcnt' and 'vmcnt' in i[1])]
variants['regs+gstore'] = [i for i in body if i[0] in VAL or i[0] == 'gstore1']
variants['all'] = list(body)
n_warm = next(k for k, i in enumerate(body) if i[0] == 'mov' and i[1] == g.le)  # first instruction of the first test
# productive part starts a few instructions earlier (addc / mov / loads of step 0); close enough for a split
variants['warmup_regs'] = [i for i in body[:n_warm] if i[0] in VAL][:-4]
prod = [i for i in body[n_warm:] if i[0] in VAL]
test = [i for i in prod if i[1] in (g.s, g.cy, g.le, g.ones) or (i[0] == 'or' and i[2] == g.le)]
variants['test_regs'] = test
variants['roll_regs'] = [i for i in prod if i not in test]
variants['gload_nowait'] = [i for i in body if i[0] in VAL or i[0] == 'gload1']
variants['all_nowait'] = [i for i in body if i[0] != 'waitcnt']
variants['store_only_end'] = [i for i in body if i[0] in VAL] + [i for i in body if i[0] == 'gstore1']
variants['no_sgpr_b3'] = [('bitop3', i[1], i[2], i[3], g.A[6], i[5]) if (i[0] == 'bitop3' and str(i[4]).startswith('s')) else i
                          for i in variants['regs_only']]
out = []
w = out.append
w("#include <hip/hip_runtime.h>\n#include <cstdio>\n")
clob = ", ".join(f'"{c}"' for c in g.clobbers())
names = []
for name, ins in variants.items():
    cname = name.replace('+', '_')
    reps = max(1, 300000 // len(ins))
    w(f"__global__ __launch_bounds__(256) void k_{cname}(unsigned *buf, unsigned seed)\n{{\n    unsigned it = {reps};\n")
    w("    const unsigned voff = (threadIdx.x & 63u) * 4u;\n    const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));\n    unsigned *base = buf + (size_t)wv * 8192u;\n    const unsigned blo = (unsigned)(size_t)base, bhi = (unsigned)((size_t)base >> 32);\n")
    w("    asm volatile(\n")
    for r in range(G.B0, G.VEND):
        w(f'        "v_mov_b32 v{r}, %[seed]\\n"\n')
    for k in range(4):
        w(f'        "s_add_u32 s{G.S_TN + 2 * k}, %[base], {hex(4096 * k)}\\n"\n        "s_addc_u32 s{G.S_TN + 2 * k + 1}, %[basehi], 0\\n"\n')
    for k in range(2):
        w(f'        "s_add_u32 s{G.S_OC + 2 * k}, %[base], {hex(16384 + 4096 * k)}\\n"\n        "s_addc_u32 s{G.S_OC + 2 * k + 1}, %[basehi], 0\\n"\n')
    for d in (G.S_QN,):
        w(f'        "s_add_u32 s{d}, %[base], 0x6000\\n"\n        "s_addc_u32 s{d + 1}, %[basehi], 0\\n"\n')
    for i in range(G.B_PLANES):
        w(f'        "s_mov_b32 s{G.S_CM + i}, {-1 if (164 >> i) & 1 else 0}\\n"\n')
    w(f'        "s_mov_b32 s{G.S_C}, %[it]\\n"\n        "L_%=:\\n"\n')
    for i in ins:
        if i[0] != 'comment':
            w(f'        "{G.to_asm(i)}\\n"\n')
    w(f'        "s_sub_u32 s{G.S_C}, s{G.S_C}, 1\\n"\n        "s_cmp_lg_u32 s{G.S_C}, 0\\n"\n        "s_cbranch_scc1 L_%=\\n"\n        "s_waitcnt vmcnt(0)\\n"\n')
    w(f'        : : [it] "s"(it), [seed] "v"(seed), [base] "s"(blo), [basehi] "s"(bhi), [voff] "v"(voff) : {clob});\n')
    w("}\n")
    nv = sum(1 for i in ins if i[0] in VAL)
    names.append((name, cname, nv, reps))
w("""
template <class K> static double run(K kern, int blocks, unsigned *d)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1u); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 2u); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms;
}
int main()
{
    unsigned *d; hipMalloc(&d, (size_t)1024 * 4 * 8192 * 4);
    hipMemset(d, 0x5A, (size_t)1024 * 4 * 8192 * 4);
    struct { const char *name; void (*k)(unsigned *, unsigned); double n_ins, reps; } ks[] = {
""")
for name, cname, n, reps in names:
    w(f'        {{"{name}", k_{cname}, {float(n)}, {float(reps)}}},\n')
w("""    };
    printf("%-14s %8s %14s %14s %14s   (SIMD cycles per pass at 2.4 GHz; per VALU instruction in brackets)\\n", "variant", "VALU", "1 w/SIMD", "2 w/SIMD", "3 w/SIMD");
    for (auto &e : ks) {
        printf("%-14s %8.0f", e.name, e.n_ins);
        for (int wps = 1; wps <= 3; ++wps) {
            double ms = run(e.k, 256 * wps, d);
            double cyc = ms * 1e-3 * 2.4e9 / (e.reps * wps);
            printf(" %8.0f (%4.2f)", cyc, cyc / e.n_ins);
        }
        printf("\\n");
    }
    return 0;
}
""")
sys.stdout.write("".join(out))
