// bs_bench.hip -- standalone check + timing of the generated bit-sliced ring filter (ntjoin_amd/csrc/hash_bs_k32.inc)
// on random packed bases.  Build: hipcc --offload-arch=gfx950 -O3 -I ntjoin_amd/csrc tools/bs_bench.hip -o bs_bench
// Usage: bs_bench [Mbp=461] [tt=164]      (verifies 3 chunks against the direct formula, then times variants)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hash_bs_k32.inc"
#ifdef WITH_NOPERM
#include "hash_bs_k32_noperm.inc"
#endif

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(2);                                                                            \
        }                                                                                       \
    } while (0)

template <int WPB, int VAR>
__global__ __launch_bounds__(64 * WPB) void k_hash_bs(const uint32_t *__restrict__ packed, const uint32_t *__restrict__ kvalid,
                                                      uint32_t *__restrict__ bitmap, uint32_t *__restrict__ cnt,
                                                      uint32_t n_chunks, uint32_t tt)
{
    // (few live VGPRs around the block: it owns v8..v167 and the kernel must stay at 168 for three waves per SIMD)
    __shared__ uint32_t acc[WPB];
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * WPB + wib)), n_waves = gridDim.x * WPB;
    const uint32_t off256 = lane * 256u, off128 = lane * 128u;
    if (lane == 0) acc[wib] = 0;
    for (uint32_t c = wave; c < n_chunks; c += n_waves) {  // (wave-uniform: the addresses stay in SGPRs)
        const uint32_t *pin = packed + (size_t)c * 4096u;
        const uint32_t *pkv = kvalid + (size_t)c * 2048u;
        uint32_t *pout = bitmap + (size_t)c * 2048u;
        uint32_t n;
        if (VAR == 0)
            asm volatile(HASH_BS_ASM : "=&v"(n) : "s"(pin), "s"(pkv), "s"(pout), "s"(tt), "v"(off256), "v"(off128) : HASH_BS_CLOBBERS);
#ifdef WITH_NOPERM
        else
            asm volatile(HASH_BSNP_ASM : "=&v"(n) : "s"(pin), "s"(pkv), "s"(pout), "s"(tt), "v"(off256), "v"(off128) : HASH_BSNP_CLOBBERS);
#endif
        atomicAdd(&acc[wib], n);  // (LDS operations of one wave complete in order)
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            cnt[c] = acc[wib];
            acc[wib] = 0;
        }
    }
}

static const uint64_t SEED[4] = {0x3c8bfbb395c60474ULL, 0x3193c18562a02b4cULL, 0x20323ed082572324ULL, 0x295549f54be24456ULL};
static uint32_t rotl31(uint32_t x, unsigned n)
{
    n %= 31;
    return n ? ((x << n) | (x >> (31 - n))) & 0x7FFFFFFFu : x;
}
static inline uint32_t base_at(const std::vector<uint32_t> &w, uint64_t p) { return (w[p >> 4] >> (2 * (p & 15))) & 3u; }
// the ring test of the 32-mer at position p (direct formula)
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

template <class K>
static float time_kernel(K kern, int blocks, int threads, size_t lds, const uint32_t *dp, const uint32_t *dk, uint32_t *db,
                         uint32_t *dc, uint32_t n_chunks, uint32_t tt, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, dp, dk, db, dc, n_chunks, tt);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, dp, dk, db, dc, n_chunks, tt);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const double mbp = argc > 1 ? atof(argv[1]) : 461.0;
    const uint32_t tt = argc > 2 ? (uint32_t)atoi(argv[2]) : 164u;
    const uint32_t n_chunks = (uint32_t)(mbp * 1e6 / 65536.0) + 1;
    const size_t n_words = (size_t)n_chunks * 4096 + 64;
    std::vector<uint32_t> hp(n_words), hk((size_t)n_chunks * 2048);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (auto &v : hp) v = (uint32_t)rnd();
    for (size_t i = 0; i < hk.size(); ++i) hk[i] = i < 3 * 2048 ? (uint32_t)rnd() | (uint32_t)rnd() : 0xFFFFFFFFu;
    uint32_t *dp, *dk, *db, *dc;
    CK(hipMalloc(&dp, n_words * 4));
    CK(hipMalloc(&dk, hk.size() * 4));
    CK(hipMalloc(&db, hk.size() * 4));
    CK(hipMalloc(&dc, (size_t)n_chunks * 4));
    CK(hipMemcpy(dp, hp.data(), n_words * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dk, hk.data(), hk.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(db, 0, hk.size() * 4));
    hipLaunchKernelGGL((k_hash_bs<4, 0>), dim3(256 * 3), dim3(256), 0, 0, dp, dk, db, dc, n_chunks, tt);
    CK(hipDeviceSynchronize());
    // ---- verify chunks 0, 1, 2 and the last one
    std::vector<uint32_t> hb(hk.size()), hc(n_chunks);
    CK(hipMemcpy(hb.data(), db, hb.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc.data(), dc, hc.size() * 4, hipMemcpyDeviceToHost));
    uint64_t bad = 0, total = 0;
    for (uint32_t c : {0u, 1u, 2u, n_chunks - 1}) {
        uint32_t cc = 0;
        for (uint32_t wi = 0; wi < 2048; ++wi) {
            uint32_t want = 0;
            for (uint32_t t = 0; t < 32; ++t)
                if (ref_bit(hp, (uint64_t)c * 65536 + wi * 32 + t, tt, 14)) want |= 1u << t;
            want &= hk[(size_t)c * 2048 + wi];
            const uint32_t got = hb[(size_t)c * 2048 + wi];
            if (got != want && bad++ < 5) printf("MISMATCH chunk %u word %u: got %08x want %08x\n", c, wi, got, want);
            cc += __builtin_popcount(want);
        }
        if (cc != hc[c]) {
            printf("COUNT MISMATCH chunk %u: got %u want %u\n", c, hc[c], cc);
            ++bad;
        }
        total += cc;
    }
    printf("verify: %s (%llu candidates in 4 chunks)\n", bad ? "FAILED" : "ok", (unsigned long long)total);
    // ---- timing: waves per SIMD by grid size (persistent waves), blocks of 1 / 4 waves
    const double kmers = (double)n_chunks * 65536.0;
    struct V { const char *name; float ms; };
    auto report = [&](const char *name, float ms, int wps) {
        const double cyc = ms * 1e-3 * 2.4e9 * 1024.0 * 64.0 / kmers;  // lane-cycles per k-mer at 2.4 GHz
        printf("%-34s %d waves/SIMD: %8.1f us  %6.1f Gbp/s  %5.2f SIMD-cycles per k-mer-lane  (%.0f GB/s of packed bases)\n", name, wps,
               ms * 1e3, kmers / ms * 1e-6, cyc, kmers * 0.25 / ms * 1e-6);
    };
    for (int wps : {1, 2, 3}) {
        float ms = time_kernel(k_hash_bs<4, 0>, 256 * wps, 256, 0, dp, dk, db, dc, n_chunks, tt, 5);
        report("perm transposes, 256-thread blocks", ms, wps);
    }
    for (int wps : {2, 3}) {
        float ms = time_kernel(k_hash_bs<1, 0>, 1024 * wps, 64, 0, dp, dk, db, dc, n_chunks, tt, 5);
        report("perm transposes, 64-thread blocks", ms, wps);
    }
    {
        // one chunk per wave (no persistence): grid = chunks
        float ms = time_kernel(k_hash_bs<1, 0>, (int)n_chunks, 64, 0, dp, dk, db, dc, n_chunks, tt, 5);
        report("one chunk per 64-thread block", ms, 0);
    }
#ifdef WITH_NOPERM
    for (int wps : {2, 3}) {
        float ms = time_kernel(k_hash_bs<4, 1>, 256 * wps, 256, 0, dp, dk, db, dc, n_chunks, tt, 5);
        report("shift transposes, 256-thread blocks", ms, wps);
    }
#endif
    return bad ? 1 : 0;
}
