// scan_kernels.h -- ordered stream-compaction building blocks shared by sketch.hip and graph.hip.
// Pattern: k_count (flags -> per-tile counts) ; k_scan_sums (exclusive scan of tile counts, one block) ;
// then a consumer kernel re-scans its tile in LDS (tile_exclusive_rank) and writes in order.
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

// inside a 256-thread block: exclusive prefix of per-thread counts `c` (uses sh[256]); all threads must call
__device__ __forceinline__ uint32_t block_exclusive_256(uint32_t c, uint32_t *sh)
{
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t t = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    return sh[threadIdx.x] - c;
}

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

// ---- scan fused into the producing kernel: the block that finishes LAST scans the per-block counts ------------------
// Every block of the producer publishes its count and draws a ticket; the block that draws the last ticket knows all
// counts are in memory and turns them into exclusive offsets (+ total) for the consumer kernel: no separate count /
// scan launches, no spinning.
// MI355X has eight XCDs whose L2s are not coherent with each other, so an agent-scope release/acquire FENCE means an L2
// write-back + invalidate; one per block made the 100 us hash kernel take 670 us (and a look-back scan with
// acquire/release status words 8x slower end to end).  What is needed is much less: the counts are published with
// agent-scope RELAXED atomic stores (written through to memory), the publishing thread waits for its store to be
// acknowledged (s_waitcnt vmcnt(0)) before the block draws its ticket, and the last block reads them with agent-scope
// relaxed loads.  Everything else a block wrote is for the NEXT kernel and is made visible by the kernel boundary.
// Tickets are two-level (one counter per group of 64 blocks, then one top counter: ~10 ns per same-address atomic
// made a single counter cost 300 us for 35 k blocks) and self-resetting: whoever completes a counter zeroes it, so the
// counters (zeroed once at allocation) are ready for the next launch.  Blocks that do not take part simply never call.
__device__ __forceinline__ void publish_u32(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // acknowledged by the memory side before anything that follows
}
// counters: [0] top, [(1 + g) * LB_STRIDE] group g (64 blocks; one counter per 256 B so that the groups' atomics land on
// different lines / channels instead of queueing behind each other); block = index among the n_blocks taking part.
// Ends with a block barrier; block-uniform result: true for exactly one block, after all others have drawn.
constexpr uint32_t LB_STRIDE = 64;  // words
__device__ __forceinline__ bool last_block_ticket(uint32_t *counters, uint32_t block, uint32_t n_blocks)
{
    __shared__ uint32_t lb_last;
    __syncthreads();   // every publish_u32 of the block is complete
    if (threadIdx.x == 0) {
        const uint32_t g = block >> 6, n_groups = (n_blocks + 63u) >> 6;
        const uint32_t in_group = (g + 1u == n_groups) ? n_blocks - (g << 6) : 64u;
        uint32_t last = 0;
        uint32_t *gc = counters + (size_t)(1u + g) * LB_STRIDE;
        if (atomicAdd(gc, 1u) == in_group - 1u) {
            *gc = 0;
            if (atomicAdd(&counters[0], 1u) == n_groups - 1u) {
                counters[0] = 0;
                last = 1;
            }
        }
        lb_last = last;
    }
    __syncthreads();
    return lb_last != 0;
}
__device__ __forceinline__ uint32_t load_coherent_u32(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // bypasses this CU's (non-coherent) L1
}
// exclusive scan of in[0..n) into out[0..n) by ONE 256-thread block (all threads call); total -> *total (u64).
// `in` was written by other blocks of the same launch: read coherently.  in == out is allowed.
__device__ __forceinline__ void block_scan_counts(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *total2)
{
    __shared__ uint32_t sh[256];
    uint64_t carry = 0;
    for (uint32_t base = 0; base < n; base += 4096) {
        const uint32_t i0 = base + threadIdx.x * 16u;
        uint32_t v[16];
        uint32_t c = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            v[u] = i0 + u < n ? load_coherent_u32(in + i0 + u) : 0;
            c += v[u];
        }
        uint32_t run = (uint32_t)carry + block_exclusive_256(c, sh);
        const uint32_t tile_total = sh[255];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (i0 + u < n) out[i0 + u] = run;
            run += v[u];
        }
        carry += tile_total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        total2[0] = (uint32_t)carry;
        total2[1] = (uint32_t)(carry >> 32);
    }
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
