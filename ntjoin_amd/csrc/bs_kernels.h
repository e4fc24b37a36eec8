// bs_kernels.h -- the bit-sliced ring filter of the sketch stage (k = 32): layout kernel + the generated filter kernel.
// Included by sketch_bs.hip (the library) and tools/bs_bench.hip (standalone check / timing).
//
// Reference semantics: `indexlr`'s ntHash candidate test as restated in SURVEY.md App. A (reference ntJoin:204-205); what the
// filter lets through is a superset of {k-mers with fwd + rev < tau}; exact hashes are computed for what it lets through.
// Layouts (see gen/bs_gen.py for why): a chunk = 65 536 base positions = 64 lanes x 32 strips x 32 positions
//     T[chunk][t / 2][lane][2 (t & 1) + beta]  u32   bit s = bit beta of the base at chunk * 65536 + (32 lane + s) * 32 + t
//     Q[chunk][lane][beta]                     u32   bit t = bit beta of the base at chunk * 65536 + (32 lane - 1) * 32 + t
//     OUT[p / 32] bit p % 32                   u32   the 32-mer at position p passed the ring test (a plain bitmap; the word in
//                                                    front of OUT[0] is written too: BS_OUT_PAD words of padding)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "hash_bs_k32.inc"

namespace mxg {

#ifndef MXG_BS_CHUNK_DEFINED
#define MXG_BS_CHUNK_DEFINED
constexpr uint32_t BS_CHUNK = 65536;        // base positions per chunk
#endif
constexpr uint32_t BS_T_WORDS = 4096;       // u32 words of T per chunk (= the chunk's packed words)
constexpr uint32_t BS_Q_WORDS = 128;        // u32 words of Q per chunk
constexpr uint32_t BS_OUT_WORDS = 2048;     // u32 words of OUT per chunk
constexpr uint32_t BS_OUT_PAD = 4;          // words in front of OUT[0] (slot 0 of the first lane writes OUT[-1])

// 32 x 32 bit transpose in registers: afterwards a[i] bit s = (before) a[s] bit i
__device__ __forceinline__ void bs_transpose32(uint32_t (&a)[32])
{
#pragma unroll
    for (uint32_t j = 16, m = 0x0000FFFFu; j; j >>= 1, m ^= m << j) {
#pragma unroll
        for (uint32_t k = 0; k < 32; ++k) {
            if (k & j) continue;
            const uint32_t x = a[k], y = a[k + j];
            a[k] = (x & m) | ((y << j) & ~m);
            a[k + j] = ((x >> j) & m) | (y & ~m);
        }
    }
}

// the even bits of x, packed into the low 16 bits
__device__ __forceinline__ uint32_t bs_even_bits(uint32_t x)
{
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    return (x | (x >> 8)) & 0xFFFFu;
}

// packed bases (16 per u32, base i of a word in bits 2i..2i+1) -> T and Q.  One wave per chunk; lane L owns the chunk's
// words [64 L, 64 L + 64) = its 32 strips of 32 bases.  Words at or beyond n_words read as 0 (base A).
__global__ __launch_bounds__(64) void k_bs_transpose(const uint32_t *__restrict__ packed, uint64_t n_words, uint32_t *__restrict__ T,
                                                     uint32_t *__restrict__ Q, uint32_t c_lo, uint32_t n_chunks)
{
    const uint32_t c = c_lo + blockIdx.x, lane = threadIdx.x;
    if (c >= n_chunks) return;
    const uint64_t w0 = (uint64_t)c * BS_T_WORDS + lane * 64u;
    uint32_t a[32], b[32];  // word 0 / word 1 of the lane's strips
#pragma unroll
    for (uint32_t s = 0; s < 32; s += 2) {  // 16 bytes = two strips per load
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        const uint64_t w = w0 + 2u * s;
        if (w + 4 <= n_words) {
            q = *reinterpret_cast<const uint4 *>(packed + w);
        } else {
            if (w < n_words) q.x = packed[w];
            if (w + 1 < n_words) q.y = packed[w + 1];
            if (w + 2 < n_words) q.z = packed[w + 2];
        }
        a[s] = q.x; b[s] = q.y; a[s + 1] = q.z; b[s + 1] = q.w;
    }
    // the strip before the lane's first one: lane L - 1's last strip, for lane 0 the 8 bytes in front of the chunk
    uint32_t pa = (uint32_t)__shfl_up((int)a[31], 1, 64), pb = (uint32_t)__shfl_up((int)b[31], 1, 64);
    if (lane == 0) {
        pa = 0; pb = 0;
        if (c > 0) {
            const uint64_t w = (uint64_t)c * BS_T_WORDS - 2u;
            if (w < n_words) pa = packed[w];
            if (w + 1 < n_words) pb = packed[w + 1];
        }
    }
    bs_transpose32(a);
    bs_transpose32(b);
    // row i of the first matrix = bit i of the strips' word 0: t = i / 2, beta = i & 1; the second matrix: t = 16 + i / 2
    uint4 *Tc = reinterpret_cast<uint4 *>(T + (uint64_t)c * BS_T_WORDS);
#pragma unroll
    for (uint32_t t2 = 0; t2 < 8; ++t2) {
        Tc[t2 * 64u + lane] = make_uint4(a[4 * t2], a[4 * t2 + 1], a[4 * t2 + 2], a[4 * t2 + 3]);
        Tc[(8u + t2) * 64u + lane] = make_uint4(b[4 * t2], b[4 * t2 + 1], b[4 * t2 + 2], b[4 * t2 + 3]);
    }
    uint2 *Qc = reinterpret_cast<uint2 *>(Q + (uint64_t)c * BS_Q_WORDS);
    Qc[lane] = make_uint2(bs_even_bits(pa) | (bs_even_bits(pb) << 16), bs_even_bits(pa >> 1) | (bs_even_bits(pb >> 1) << 16));
}

// The filter.  Blocks of 256 threads = one wave per SIMD; the grid is sized for TWO waves per SIMD (an even number of waves
// per SIMD issues at 2.05 cycles per instruction, an odd one at 2.5-2.7: profiles/ubench), every wave takes the chunks
// c0, c0 + stride, ...  The body is generated (gen/bs_gen.py) and owns v8..v243 and s36..s82; the few values around it
// stay in v0..v7.
__global__ __launch_bounds__(256) void k_hash_bs(const uint32_t *__restrict__ T, const uint32_t *__restrict__ Q,
                                                 uint32_t *__restrict__ OUT, uint32_t c_lo, uint32_t c_hi, uint32_t tt)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
    const uint32_t stride = gridDim.x * 4u, c0 = c_lo + wave, voff = lane * 16u, voff8 = lane * 8u, voff128 = lane * 128u;
    asm volatile(HASH_BS_ASM
                 :
                 : [t] "s"(T), [p] "s"(Q), [o] "s"(OUT), [c0] "s"(c0), [n] "s"(c_hi), [stride] "s"(stride), [tt] "s"(tt), [voff] "v"(voff), [voff8] "v"(voff8), [voff128] "v"(voff128)
                 : HASH_BS_CLOBBERS);
}

}  // namespace mxg
