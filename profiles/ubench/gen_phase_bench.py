#!/usr/bin/env python3
"""Generates phase_bench.hip: the phases of the generated bit-sliced chunk body (ntjoin_amd/csrc/gen/bs_gen.py), each as a
register-only loop (loads / stores / waits removed), timed at 1..3 waves per SIMD: where do the cycles of a chunk go?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bs_gen_v1 as G  # noqa: E402

g = G.Gen(32)


def cname(n):
    return n.replace('+', '_')


prog = g.build()
phases = {}
cur = 'head'
for ins in prog.ins:
    if ins[0] == 'comment' and ins[1].startswith('PHASE '):
        cur = ins[1][6:]
        continue
    if ins[0].startswith(('gload', 'gstore', 'wait', 'comment')):
        continue
    phases.setdefault(cur, []).append(ins)
# sub-phases of the productive part: test only / roll only
prod = phases['productive']
test_only = [i for i in prod if i[1] in (g.s, g.cy, g.le, g.ones) or (i[0] == 'or' and i[2] == g.le)]
roll_only = [i for i in prod if i not in test_only]
phases['prod_test'] = test_only
phases['prod_roll'] = roll_only
tr = phases['transpose_in']
for opn in ('bitop3', 'lshl', 'lshr', 'add'):
    phases['tr_' + opn] = [i for i in tr if i[0] == opn]
for opn in ('bitop3', 'xor', 'alignbit', 'lshr', 'and', 'or'):
    phases['roll_' + opn] = [i for i in roll_only if i[0] == opn]
phases['test_b3_vgpr'] = [i for i in test_only if i[0] == 'bitop3' and not str(i[4]).startswith('s')]
phases['test_b3_sgpr'] = [i for i in test_only if i[0] == 'bitop3' and str(i[4]).startswith('s')]
def tr_map(fn):
    out_ = []
    for i in tr:
        r = fn(i)
        if r is None:
            continue
        out_ += r if isinstance(r, list) else [r]
    return out_
phases['trV1_noshl'] = tr_map(lambda i: ('lshr',) + i[1:] if i[0] == 'lshl' else i)
phases['trV3_shifts'] = tr_map(lambda i: i if i[0] in ('lshl', 'lshr', 'add') else None)
phases['trV4_xor'] = tr_map(lambda i: ('xor', i[1], i[2], i[3]) if i[0] == 'bitop3' else i)
phases['trV6_nops'] = tr_map(lambda i: [i, ('nop', 0)])
phases['trV7_xor_noshl'] = tr_map(lambda i: ('xor', i[1], i[2], i[3]) if i[0] == 'bitop3' else (('lshr',) + i[1:] if i[0] == 'lshl' else i))
phases['trV8_b3_selfdst'] = tr_map(lambda i: ('bitop3', i[2], i[2], i[3], i[4], i[5]) if i[0] == 'bitop3' else None)
phases['tr+nop+warm'] = phases['transpose_in'] + [('nop', 0)] + phases['warmup']
phases['warm+tr'] = phases['warmup'] + phases['transpose_in']
phases['warm+warm'] = phases['warmup'] + phases['warmup']
phases['tr+tr'] = phases['transpose_in'] + phases['transpose_in']
phases['tr+warm'] = phases['transpose_in'] + phases['warmup']
phases['warm+prod'] = phases['warmup'] + phases['productive']
phases['prod+out'] = phases['productive'] + phases['out']
phases['all'] = [i for ph in ('head', 'transpose_in', 'warmup', 'productive', 'out') for i in phases[ph]]
out = []
w = out.append
w("#include <hip/hip_runtime.h>\n#include <cstdio>\n")
clob = ", ".join(f'"{c}"' for c in g.clobbers() if c != "memory")
names = []
for name, ins in phases.items():
    if name == 'head' or not ins:
        continue
    reps = max(1, 400000 // len(ins))
    w(f"__global__ __launch_bounds__(256) void k_{cname(name)}(unsigned *out, unsigned seed)\n{{\n    unsigned it = {reps}, n = 0;\n    const unsigned long long z = 0;\n")
    w("    asm volatile(\n")
    for r in range(G.B0, G.VEND):
        w(f'        "v_mov_b32 v{r}, %2\\n"\n')
    w('        "s_mov_b32 s90, %1\\n"\n        "L_%=:\\n"\n')
    for i in ins:
        w(f'        "{G.to_asm(i)}\\n"\n')
    w('        "s_sub_u32 s90, s90, 1\\n"\n        "s_cmp_lg_u32 s90, 0\\n"\n        "s_cbranch_scc1 L_%=\\n"\n')
    w(f'        : "=&v"(n) : "s"(it), "v"(seed), "s"(z), "s"(seed) : {clob}, "s90");\n')
    w("    out[blockIdx.x * 256 + threadIdx.x] = n;\n}\n")
    names.append((name, len(ins), reps))
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
    struct { const char *name; void (*k)(unsigned *, unsigned); double n_ins, reps; } ks[] = {
""")
for name, n, reps in names:
    w(f'        {{"{name}", k_{cname(name)}, {float(n)}, {float(reps)}}},\n')
w("""    };
    printf("%-14s %8s %12s %12s %12s   (SIMD cycles per pass of the phase at 2.4 GHz; per instruction in brackets)\\n", "phase", "instr", "1 w/SIMD", "2 w/SIMD", "3 w/SIMD");
    for (auto &e : ks) {
        printf("%-14s %8.0f", e.name, e.n_ins);
        for (int wps = 1; wps <= 3; ++wps) {
            double ms = run(e.k, 256 * wps, d_out);
            double cyc = ms * 1e-3 * 2.4e9 / (e.reps * wps);
            printf(" %7.0f (%4.2f)", cyc, cyc / e.n_ins);
        }
        printf("\\n");
    }
    return 0;
}
""")
sys.stdout.write("".join(out))
