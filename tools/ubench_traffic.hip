// ubench_traffic.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of this library's kernels,
// against byte counts that are known by construction (MI355X_MICROARCH.md, "HBM": only the wide coalesced streaming read is
// calibrated there -- FETCH_SIZE reports half of its bytes; "calibrate on a known byte count in your own access pattern").
//
//   k_stream_read16      every lane reads 16 consecutive bytes, the wave 1 KB, the grid the whole buffer once      (k_hash_bs's reads)
//   k_gather12           every lane reads 12 bytes at a pseudo-random 4-byte aligned address of a 4 GB buffer      (k_bs_select's packed bases,
//                        k_emit's and the join's gathers): one 64-byte piece of memory per request unless the 12 bytes straddle two (1/8 do)
//   k_stream_write16     every lane writes 16 consecutive bytes                                                  (k_hash_bs's bitmap, k_emit's output)
//   k_scatter4           every lane writes 4 bytes at a pseudo-random address                                   (k_pj_join_pipe's verdicts)
//   k_scatter16          every lane writes 16 bytes at a pseudo-random 16-byte aligned address                  (the join's records)
//
// The buffer (4 GB by default) is far larger than the 256 MB Infinity Cache, and every address is touched at most about once per launch,
// so what reaches the memory side is what the pattern needs.  Run under rocprofv3 --pmc (tools/pmc_calib.sh), one counter per pass.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_traffic tools/ubench_traffic.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);   \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

__global__ __launch_bounds__(256) void k_stream_read16(const uint4 *__restrict__ buf, uint64_t n16, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256u) {
        const uint4 v = buf[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

struct __attribute__((packed, aligned(4))) W3 {
    uint32_t a, b, c;
};
__global__ __launch_bounds__(256) void k_gather12(const uint32_t *__restrict__ buf, uint64_t n_words, uint64_t n_req, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_req; i += (uint64_t)gridDim.x * 256u) {
        const uint64_t w = mix(i * 2654435761ull + 17) % (n_words - 4);
        const W3 v = *reinterpret_cast<const W3 *>(buf + w);
        acc ^= v.a ^ v.b ^ v.c;
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(256) void k_stream_write16(uint4 *__restrict__ buf, uint64_t n16)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256u)
        buf[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

__global__ __launch_bounds__(256) void k_scatter4(uint32_t *__restrict__ buf, uint64_t n_words, uint64_t n_req)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_req; i += (uint64_t)gridDim.x * 256u)
        buf[mix(i * 2654435761ull + 5) % n_words] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void k_scatter16(uint4 *__restrict__ buf, uint64_t n16, uint64_t n_req)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_req; i += (uint64_t)gridDim.x * 256u)
        buf[mix(i * 2654435761ull + 9) % n16] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

int main(int argc, char **argv)
{
    const uint64_t gb = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4;
    const uint64_t bytes = gb << 30, n16 = bytes / 16, n_words = bytes / 4;
    const uint64_t n_req = argc > 2 ? strtoull(argv[2], nullptr, 10) : (32ull << 20);  // requests of the random patterns
    void *buf;
    uint32_t *sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    const dim3 grid(256 * 8), block(256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto timed = [&](const char *name, double known_bytes, auto launch) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"kernel\": \"%s\", \"known_bytes\": %.0f, \"ms\": %.4f, \"known_gbs\": %.1f}\n", name, known_bytes, ms, known_bytes / ms / 1e6);
    };
    for (int rep = 0; rep < 2; ++rep) {
        timed("k_stream_read16", (double)bytes, [&] { hipLaunchKernelGGL(k_stream_read16, grid, block, 0, 0, (const uint4 *)buf, n16, sink); });
        // 12 bytes at a 4-byte aligned address lie in one 64-byte piece unless they start in its last two words: 14 of 16 starts
        timed("k_gather12", (double)n_req * 64.0 * (1.0 + 2.0 / 16.0),
              [&] { hipLaunchKernelGGL(k_gather12, grid, block, 0, 0, (const uint32_t *)buf, n_words, n_req, sink); });
        timed("k_stream_write16", (double)bytes, [&] { hipLaunchKernelGGL(k_stream_write16, grid, block, 0, 0, (uint4 *)buf, n16); });
        timed("k_scatter4", (double)n_req * 4.0, [&] { hipLaunchKernelGGL(k_scatter4, grid, block, 0, 0, (uint32_t *)buf, n_words, n_req); });
        timed("k_scatter16", (double)n_req * 16.0, [&] { hipLaunchKernelGGL(k_scatter16, grid, block, 0, 0, (uint4 *)buf, n16, n_req); });
    }
    CHECK(hipFree(buf));
    return 0;
}
