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
    w('        "s_mov_b32 s40, %0\\n"\n')
    for r in clob:
        if r.startswith('v'):
            w(f'        "v_mov_b32 {r}, %1\\n"\n')
    w('        "L_%=:\\n"\n')
    w(txt)
    w('        "s_sub_u32 s40, s40, 1\\n"\n        "s_cmp_lg_u32 s40, 0\\n"\n        "s_cbranch_scc1 L_%=\\n"\n')
    w(f'        : : "s"(it), "v"(seed) : {cl}, "s40", "scc");\n')
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
# (2) code size: bitop3 (8 bytes each) on 62 rotating destinations, like the ring update
for kb in (2, 8, 16, 32, 48, 64, 96):
    n = kb * 1024 // 8
    body = []
    for i in range(n):
        d = 32 + (i % 62)
        a = 100 + (i * 7) % 12
        b = 120 + (i * 5) % 12
        body.append(f"v_bitop3_b32 v{d}, v{d}, v{a}, v{b} bitop3:0x96")
    kernel(f"k_code_{kb}k", body, max(1, 2048 * 1024 // n), regs)
# (2b) the same with 4-byte VOP2
for kb in (8, 32, 64):
    n = kb * 1024 // 4
    body = []
    for i in range(n):
        d = 32 + (i % 62)
        a = 100 + (i * 7) % 12
        body.append(f"v_xor_b32 v{d}, v{d}, v{a}")
    kernel(f"k_code2_{kb}k", body, max(1, 2048 * 1024 // n), regs)
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
