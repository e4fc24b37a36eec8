// bs_bench.hip -- standalone check + timing of the bit-sliced ring filter (ntjoin_amd/csrc/bs_kernels.h) on random bases.
// Build: hipcc --offload-arch=gfx950 -O3 -I ntjoin_amd/csrc tools/bs_bench.hip -o tools/bin/bs_bench
// Usage: bs_bench [Mbp=3000] [tt=164]   (filter straight from the packed bases, 4 chunks checked against the direct formula, timings)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bs_kernels.h"

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(2);                                                                            \
        }                                                                                       \
    } while (0)

static const uint64_t SEED[4] = {0x3c8bfbb395c60474ULL, 0x3193c18562a02b4cULL, 0x20323ed082572324ULL, 0x295549f54be24456ULL};
static uint32_t rotl31(uint32_t x, unsigned n)
{
    n %= 31;
    return n ? ((x << n) | (x >> (31 - n))) & 0x7FFFFFFFu : x;
}
static inline uint32_t base_at(const std::vector<uint32_t> &w, uint64_t p) { return (p >> 4) < w.size() ? (w[p >> 4] >> (2 * (p & 15))) & 3u : 0u; }
static bool ref_bit(const std::vector<uint32_t> &w, uint64_t p, uint32_t tt, int b)
{
    uint32_t F = 0, R = 0;
    for (unsigned j = 0; j < 32; ++j) {
        const uint32_t c = base_at(w, p + j);
        F ^= rotl31((uint32_t)(SEED[c] >> 33), 31 - j);
        R ^= rotl31((uint32_t)(SEED[3 - c] >> 33), j);
    }
    const uint32_t low = 31 - b, St = ((F >> low) + (R >> low)) & ((1u << b) - 1u);
    return St <= tt || St >= (1u << b) - 2u;
}

// one wave that sleeps `iters` times 127 x 64 shader clocks and reports how long that took on the constant 100 MHz counter: the
// shader clock while the filter runs beside it (s_sleep counts shader clocks whatever else the SIMD does)
__global__ void k_clock_probe(uint64_t *out, uint32_t iters)
{
    const uint64_t w0 = wall_clock64();
    for (uint32_t i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
    const uint64_t w1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = w1 - w0;
}

int main(int argc, char **argv)
{
    const double mbp = argc > 1 ? atof(argv[1]) : 3000.0;
    const uint32_t tt = argc > 2 ? (uint32_t)atoi(argv[2]) : 164u;
    const uint32_t n_chunks = (uint32_t)(mbp * 1e6 / 65536.0) + 1;
    const uint64_t n_words = (uint64_t)n_chunks * 4096 - 1000;  // (the last chunk is ragged)
    std::vector<uint32_t> hp(n_words);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (auto &v : hp) v = (uint32_t)rnd();
    uint32_t *dp, *dEdge, *dO;
    CK(hipMalloc(&dp, n_words * 4));
    CK(hipMalloc(&dEdge, (size_t)2 * mxg::BS_EDGE_WORDS * 4));  // padded copies of the first and of the last chunk's words
    CK(hipMemset(dEdge, 0, (size_t)2 * mxg::BS_EDGE_WORDS * 4));
    CK(hipMalloc(&dO, ((size_t)n_chunks * mxg::BS_OUT_WORDS + mxg::BS_OUT_PAD) * 4));
    dO += mxg::BS_OUT_PAD;
    CK(hipMemcpy(dp, hp.data(), n_words * 4, hipMemcpyHostToDevice));
    const uint64_t tail_lo = (uint64_t)(n_chunks - 1) * mxg::BS_CHUNK_WORDS;  // the last (ragged) chunk: zero-padded copy
    CK(hipMemcpy(dEdge + 2, dp, (size_t)mxg::BS_CHUNK_WORDS * 4, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(dEdge + mxg::BS_EDGE_WORDS, dp + tail_lo - 2, (n_words - tail_lo + 2) * 4, hipMemcpyDeviceToDevice));
    const uint32_t *dHead = dEdge + 2, *dTail = dEdge + mxg::BS_EDGE_WORDS + 2;
    CK(hipMemset(dO - mxg::BS_OUT_PAD, 0xAB, ((size_t)n_chunks * mxg::BS_OUT_WORDS + mxg::BS_OUT_PAD) * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(mxg::k_hash_bs, dim3(512), dim3(256), 0, 0, dp, dHead, dTail, dO, 0u, n_chunks, tt, n_chunks - 1);
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    // ---- verify chunks 0, 1, the middle one and the last one
    std::vector<uint32_t> ho(mxg::BS_OUT_WORDS);
    uint64_t bad = 0, total = 0;
    for (uint32_t c : {0u, 1u, n_chunks / 2, n_chunks - 1}) {
        // the chunk writes the words [c * 2048 - 1, c * 2048 + 2047)
        CK(hipMemcpy(ho.data(), dO + (size_t)c * mxg::BS_OUT_WORDS - 1, mxg::BS_OUT_WORDS * 4, hipMemcpyDeviceToHost));
        for (uint32_t wi = 0; wi < 2048; ++wi) {
            const int64_t g = (int64_t)c * 2048 + wi - 1;
            if (g < 0) continue;
            uint32_t want = 0;
            for (uint32_t t = 0; t < 32; ++t)
                if (ref_bit(hp, (uint64_t)g * 32 + t, tt, HASH_BS_PLANES)) want |= 1u << t;
            if (ho[wi] != want && bad++ < 5) printf("MISMATCH chunk %u word %u: got %08x want %08x\n", c, wi, ho[wi], want);
            total += __builtin_popcount(want);
        }
    }
    printf("verify: %s (%llu candidates in 4 chunks)\n", bad ? "FAILED" : "ok", (unsigned long long)total);
    const double kmers = (double)n_chunks * 65536.0;
    if (argc > 3) {  // bs_bench Mbp tt probe: the shader clock with and without the filter running
        hipStream_t s2;
        CK(hipStreamCreate(&s2));
        uint64_t *dW, hW[8];
        CK(hipMalloc(&dW, sizeof(hW)));
        const uint32_t iters = 1500;
        for (int with : {0, 1, 0, 1}) {
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(k_clock_probe, dim3(8), dim3(64), 0, s2, dW, iters);
            if (with)
                for (int r = 0; r < 12; ++r)
                    hipLaunchKernelGGL(mxg::k_hash_bs, dim3(512), dim3(256), 0, 0, dp, dHead, dTail, dO, 0u, n_chunks, tt, n_chunks - 1);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hW, dW, sizeof(hW), hipMemcpyDeviceToHost));
            double lo = 1e30, hi = 0;
            for (uint64_t v : hW) {
                const double mhz = (double)iters * 127.0 * 64.0 / ((double)v / 100.0);  // clocks per us
                lo = mhz < lo ? mhz : lo;
                hi = mhz > hi ? mhz : hi;
            }
            printf("shader clock %s: %.0f - %.0f MHz over %.2f ms (8 probe waves)\n", with ? "with the filter running" : "on an idle GPU        ", lo, hi, (double)hW[0] / 100e3);
        }
    }
    for (int blocks : {256, 512, 768, 1024}) {
        hipLaunchKernelGGL(mxg::k_hash_bs, dim3(blocks), dim3(256), 0, 0, dp, dHead, dTail, dO, 0u, n_chunks, tt, n_chunks - 1);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 5;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mxg::k_hash_bs, dim3(blocks), dim3(256), 0, 0, dp, dHead, dTail, dO, 0u, n_chunks, tt, n_chunks - 1);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const double cyc_instr = ms * 1e-3 * 2.4e9 * 1024.0 / ((double)n_chunks * HASH_BS_VALU_PER_CHUNK);
        printf("k_hash_bs %4d blocks (%d waves/SIMD): %8.1f us  %7.1f Gbp/s  %.2f cycles per VALU instruction per SIMD  %.0f GB/s of bases read, %.0f GB/s read + written\n",
               blocks, blocks / 256, ms * 1e3, kmers / ms * 1e-6, cyc_instr, kmers * 0.25 / ms * 1e-6, kmers * 0.375 / ms * 1e-6);
    }
    return bad ? 1 : 0;
}
