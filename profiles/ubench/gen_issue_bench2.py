#!/usr/bin/env python3
"""issue_bench2.hip: the "one slow instruction slows the whole stream" effect (issue_bench, k_run_*) against the number of
waves per SIMD.  Kernels use v32..v63 only, so up to 8 waves per SIMD fit.  S slow instructions, then F fast ones, repeated.
Usage: python gen_issue_bench2.py > issue_bench2.hip; hipcc --offload-arch=gfx950 -O3 issue_bench2.hip -o issue_bench2"""
import sys

out = []
w = out.append
w("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdlib>\n")
kernels = []
regs = [f"v{i}" for i in range(32, 64)]


def kernel(name, body_lines, iters):
    txt = "".join(f'        "{l}\\n"\n' for l in body_lines)
    cl = ", ".join(f'"{c}"' for c in regs)
    w(f"__global__ __launch_bounds__(256) void {name}(unsigned *out, unsigned seed)\n{{\n")
    w(f"    unsigned it = {iters};\n    asm volatile(\n")
    w('        "s_mov_b32 s40, %0\\n"\n')
    for r in regs:
        w(f'        "v_mov_b32 {r}, %1\\n"\n')
    w('        "L_%=:\\n"\n')
    w(txt)
    w('        "s_sub_u32 s40, s40, 1\\n"\n        "s_cmp_lg_u32 s40, 0\\n"\n        "s_cbranch_scc1 L_%=\\n"\n')
    w(f'        : : "s"(it), "v"(seed) : {cl}, "s40", "s41", "vcc", "scc");\n')
    w("    out[blockIdx.x * 256 + threadIdx.x] = seed;\n}\n")
    kernels.append((name, len(body_lines) * iters))


FAST = {"shr": "v_lshrrev_b32 v{0}, 1, v{0}", "xor": "v_xor_b32 v{0}, v{0}, v{1}",
        "b3": "v_bitop3_b32 v{0}, v{0}, v{1}, v{2} bitop3:0x96"}
SLOW = {"shl": "v_lshlrev_b32 v{0}, 4, v{0}", "perm": "v_perm_b32 v{0}, v{0}, v{1}, s41", "alignbit": "v_alignbit_b32 v{0}, v{0}, v{1}, 28"}


def runs(name, S, F, slow, fast, n=1024):
    body = []
    i = 0
    while len(body) < n:
        for kind, cnt in ((slow, S), (fast, F)):
            for _ in range(cnt):
                d = 32 + 4 * (i % 8)            # bank 0
                a = 33 + 4 * ((i * 3) % 8)      # bank 1
                b = 34 + 4 * ((i * 5) % 8)      # bank 2
                body.append(kind.format(d, a, b))
                i += 1
    kernel(name, body, 1024 * 1024 // len(body))


runs("k_fast_shr", 0, 16, SLOW["shl"], FAST["shr"])
runs("k_fast_xor", 0, 16, SLOW["shl"], FAST["xor"])
runs("k_fast_b3", 0, 16, SLOW["shl"], FAST["b3"])
runs("k_slow_shl", 16, 0, SLOW["shl"], FAST["shr"])
for S, F in ((1, 3), (1, 15), (1, 63), (1, 255), (1, 1023), (16, 1008), (64, 960)):
    runs(f"k_shl_{S}_{F}", S, F, SLOW["shl"], FAST["b3"], n=max(1024, S + F))
runs("k_perm_1_15", 1, 15, SLOW["perm"], FAST["b3"])
runs("k_perm_64_960", 64, 960, SLOW["perm"], FAST["b3"])
runs("k_align_1_63", 1, 63, SLOW["alignbit"], FAST["xor"])
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
    const int W[] = {1, 2, 3, 4, 5, 6, 8};
    printf("%-16s", "waves per SIMD:");
    for (int wps : W) printf(" %7d", wps);
    printf("   (cycles per wave-instruction per SIMD at 2.4 GHz)\\n");
    for (auto &e : ks) {
        printf("%-16s", e.name);
        for (int wps : W) {
            double ms = run(e.k, 256 * wps, d_out);
            printf(" %7.2f", ms * 1e-3 * 2.4e9 / (e.instr * wps));
        }
        printf("\\n");
    }
    return 0;
}
""")
sys.stdout.write("".join(out))
