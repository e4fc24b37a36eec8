// scan_kernels.h -- ordered stream-compaction building blocks shared by sketch.hip, graph.hip, paths.hip, dgraph.hip.
// Hot path: two-level counts (count_publish / count_prefix below): the producer kernel stores its per-block counts, the
// consumer kernel sums what precedes its tile and re-scans the tile in LDS (block_exclusive_256): no scan launch.
// Cold paths (dense sketch path, path extraction): k_count_n / k_tile_sum_u32 -> k_scan_sums (one block) -> consumer.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mxg {

constexpr int TILE_PER_THREAD = 4;
constexpr int TILE = 256 * TILE_PER_THREAD;
static_assert(TILE_PER_THREAD == 4, "flag loads below fetch one 32-bit word per thread");

// the thread's four flag bytes [base, base+4) as one word (base is a multiple of 4; bytes at or beyond n read as 0).
// The flag arrays are allocated with >= 16 bytes of slack, so the word load never leaves the allocation.
__device__ __forceinline__ uint32_t load_flags4(const uint8_t *__restrict__ flags, uint32_t base, uint32_t n)
{
    if (base >= n) return 0u;
    uint32_t v = *reinterpret_cast<const uint32_t *>(flags + base);
    const uint32_t valid = n - base;
    if (valid < 4) v &= (1u << (8 * valid)) - 1u;
    return v;
}
__device__ __forceinline__ uint32_t count_flags4(uint32_t v) { return (v & 1u) + ((v >> 8) & 1u) + ((v >> 16) & 1u) + ((v >> 24) & 1u); }

static __global__ __launch_bounds__(256) void k_count(const uint8_t *__restrict__ sel, uint32_t n, uint32_t *__restrict__ bsum)
{
    __shared__ uint32_t sh[256];
    uint32_t base = blockIdx.x * TILE + threadIdx.x * TILE_PER_THREAD;
    uint32_t c = count_flags4(load_flags4(sel, base, n));
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = sh[0];
}

__device__ __forceinline__ uint32_t wave_inclusive_u32(uint32_t v, uint32_t lane);

// inside a block of NW waves: exclusive prefix of per-thread counts `c`; afterwards sh[255] holds the block total
// (sh: >= 256 words, NW <= 16).  Wave scans by lane shuffles + NW wave totals through LDS: two barriers (a Hillis-Steele
// scan over LDS took 17).  All threads of the block must call.
template <int NW>
__device__ __forceinline__ uint32_t block_exclusive(uint32_t c, uint32_t *sh)
{
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t incl = wave_inclusive_u32(c, lane);
    if (lane == 63u) sh[wv] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (uint32_t u = 0; u < (uint32_t)NW; ++u) {
        const uint32_t t = sh[u];
        base += u < wv ? t : 0u;
        total += t;
    }
    if (threadIdx.x == 0) sh[255] = total;
    __syncthreads();  // sh[255] visible; nobody still reads sh[0..NW) when a later call overwrites it
    return base + incl - c;
}
__device__ __forceinline__ uint32_t block_exclusive_256(uint32_t c, uint32_t *sh) { return block_exclusive<4>(c, sh); }

// exclusive scan of bsum[0..n) in place by ONE 256-thread block (16 elements per thread per pass); total -> *total
static __global__ __launch_bounds__(256) void k_scan_sums(uint32_t *__restrict__ bsum, uint32_t n, uint64_t *__restrict__ total)
{
    __shared__ uint32_t sh[256];
    uint64_t carry = 0;
    for (uint32_t base = 0; base < n; base += 4096) {
        const uint32_t i0 = base + threadIdx.x * 16u;
        uint32_t v[16];
        uint32_t c = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            v[u] = i0 + u < n ? bsum[i0 + u] : 0;
            c += v[u];
        }
        uint32_t run = (uint32_t)carry + block_exclusive_256(c, sh);
        const uint32_t tile_total = sh[255];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (i0 + u < n) bsum[i0 + u] = run;
            run += v[u];
        }
        carry += tile_total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

// ---- u32 array exclusive scan (tile sums -> k_scan_sums -> downsweep) -------------------------------
static __global__ __launch_bounds__(256) void k_tile_sum_u32(const uint32_t *__restrict__ in, uint32_t n, uint32_t *__restrict__ bsum)
{
    __shared__ uint32_t sh[256];
    uint32_t base = blockIdx.x * TILE + threadIdx.x * TILE_PER_THREAD;
    uint32_t c = 0;
    for (int u = 0; u < TILE_PER_THREAD; ++u)
        if (base + u < n) c += in[base + u];
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = sh[0];
}

// out[i] = exclusive prefix of in[0..i)   (bsum already exclusive-scanned)
static __global__ __launch_bounds__(256) void k_tile_excl_u32(const uint32_t *__restrict__ in, uint32_t n,
                                                               const uint32_t *__restrict__ bsum, uint32_t *__restrict__ out)
{
    __shared__ uint32_t sh[256];
    uint32_t base = blockIdx.x * TILE + threadIdx.x * TILE_PER_THREAD;
    uint32_t v[TILE_PER_THREAD];
    uint32_t c = 0;
    for (int u = 0; u < TILE_PER_THREAD; ++u) {
        v[u] = base + u < n ? in[base + u] : 0;
        c += v[u];
    }
    uint32_t run = bsum[blockIdx.x] + block_exclusive_256(c, sh);
    for (int u = 0; u < TILE_PER_THREAD; ++u) {
        if (base + u < n) out[base + u] = run;
        run += v[u];
    }
}

// ---- ordered offsets without count / scan launches: two-level counts ------------------------------------------------
// A producer kernel whose block b yields c items stores cnt[b] = c and adds c to sup[b >> 8] (one fire-and-forget atomic
// per block: nobody waits for it).  The consumer kernel's block that needs "items before producer block q" sums
// sup[0 .. q>>8) and cnt[(q>>8)<<8 .. q): a few hundred words from L2 for one wave.  No scan kernel, no count kernel,
// no inter-block waiting.  sup must be zero before the producer starts (an earlier kernel or the batch's memset does it).
// (Measured and dropped on MI355X: a chained look-back scan -- ~100 dependent rounds for 10^4 tiny simultaneous tiles --
// and a "last block scans" ticket scheme -- 3 us of atomic round trips at the end of every 10 us block; agent-scope
// acquire/release FENCES in either cost an L2 write-back + invalidate each, because the eight XCDs' L2s are not
// coherent with one another: 100 us -> 670 us for the hash kernel.)
// Same-line atomics are served one after the other (~10 ns each on MI355X): every super-count sits on its own 128-byte
// line, so the adds of different groups proceed in parallel and one group's 256 adds cost ~2.5 us, overlapped with the
// rest of the kernel.  (All on one line: +46 us for the 6 104 adds of the hash kernel.)
constexpr uint32_t SUP_SHIFT = 8;    // 256 producer blocks per super-count
constexpr uint32_t SUP_STRIDE = 32;  // words between super-counts
__host__ __device__ __forceinline__ uint32_t sup_words(uint32_t n_blocks) { return ((n_blocks >> SUP_SHIFT) + 1u) * SUP_STRIDE; }
// thread 0 of producer block b
__device__ __forceinline__ void count_publish(uint32_t *cnt, uint32_t *sup, uint32_t b, uint32_t c)
{
    cnt[b] = c;
    if (c) atomicAdd(&sup[(b >> SUP_SHIFT) * SUP_STRIDE], c);
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v);
// all 64 lanes of ONE wave: number of items of producer blocks [0, q)
__device__ __forceinline__ uint32_t count_prefix(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ sup, uint32_t q)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t ns = q >> SUP_SHIFT;
    // five independent loads per lane, issued together (as accumulating loops they were up to five dependent round trips)
    const uint32_t b0 = (ns << SUP_SHIFT) + lane;
    const uint32_t s0 = lane < ns ? sup[lane * SUP_STRIDE] : 0u;
    const uint32_t v0 = b0 < q ? cnt[b0] : 0u, v1 = b0 + 64u < q ? cnt[b0 + 64u] : 0u;
    const uint32_t v2 = b0 + 128u < q ? cnt[b0 + 128u] : 0u, v3 = b0 + 192u < q ? cnt[b0 + 192u] : 0u;
    uint32_t acc = s0 + v0 + v1 + v2 + v3;
    // > 16384 producer blocks only (the slices of a 3 Gbp assembly: up to 650 super-counts): four loads issued together per
    // round -- one after the other they were up to ten dependent round trips in front of everything a block of k_emit does
    for (uint32_t i = lane + 64u; i < ns; i += 256u) {
        const uint32_t t0 = sup[i * SUP_STRIDE];
        const uint32_t t1 = i + 64u < ns ? sup[(i + 64u) * SUP_STRIDE] : 0u;
        const uint32_t t2 = i + 128u < ns ? sup[(i + 128u) * SUP_STRIDE] : 0u;
        const uint32_t t3 = i + 192u < ns ? sup[(i + 192u) * SUP_STRIDE] : 0u;
        acc += (t0 + t1) + (t2 + t3);
    }
    return wave_sum_u32(acc);
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_inclusive_u32(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)v, o, 64);
        if (lane >= (uint32_t)o) v += t;
    }
    return v;
}

}  // namespace mxg
