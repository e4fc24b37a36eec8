// bs_kernels.h -- the bit-sliced ring filter of the sketch stage (k = 32): the generated filter kernel.
// Included by sketch_bs.hip (the library) and tools/bs_bench.hip (standalone check / timing).
//
// Reference semantics: `indexlr`'s ntHash candidate test as restated in SURVEY.md App. A (reference ntJoin:204-205); what the
// filter lets through is a superset of {k-mers with fwd + rev < tau}; exact hashes are computed for what it lets through.
// A chunk = 65 536 base positions = 64 lanes x 32 strips x 32 positions = 4096 words of the 2-bit packed assembly, read as they
// are (lane L: words 64 L - 2 .. 64 L + 63) and turned into bit planes in registers (gen/bs_gen.py).  The assembly's first and
// last chunk are read from padded copies of their words (two zero words in front of chunk 0; bases behind the assembly read as
// A), so nothing is read outside the packed array.  BS_EDGE_WORDS words per copy: [2 words in front][4096 words][2 spare].
//     OUT[p / 32] bit p % 32   u32   the 32-mer at position p passed the ring test (a plain bitmap; the word in front of OUT[0]
//                                    is written too: BS_OUT_PAD words of padding)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#ifndef HASH_BS_INC_FILE  // (tools/bs_ablate.sh builds tools/bs_bench.hip against timing variants of the generated code)
#define HASH_BS_INC_FILE "hash_bs_k32.inc"
#endif
#include HASH_BS_INC_FILE

namespace mxg {

#ifndef MXG_BS_CHUNK_DEFINED
#define MXG_BS_CHUNK_DEFINED
constexpr uint32_t BS_CHUNK = 65536;        // base positions per chunk
#endif
constexpr uint32_t BS_CHUNK_WORDS = 4096;   // packed u32 words per chunk
constexpr uint32_t BS_EDGE_WORDS = 4100;    // a padded copy of a chunk's words: 2 in front (the kernel gets the address behind them)
constexpr uint32_t BS_OUT_WORDS = 2048;     // u32 words of OUT per chunk
constexpr uint32_t BS_OUT_PAD = 4;          // words in front of OUT[0] (slot 0 of the first lane writes OUT[-1])

// The filter.  Blocks of 256 threads = one wave per SIMD; the grid is sized for TWO waves per SIMD (an even number of waves
// per SIMD issues at 2.05 cycles per instruction, an odd one at 2.5-2.7: profiles/ubench), every wave takes the chunks
// c0, c0 + stride, ...  The body is generated (gen/bs_gen.py) and owns v8..v247 and s36..s87; the few values around it
// stay in v0..v7.  `head` / `tail` = the padded copies of chunk 0's / chunk c_tail's words (the address of their word 0: two
// more words lie in front of it).
__global__ __launch_bounds__(256) void k_hash_bs(const uint32_t *__restrict__ packed, const uint32_t *__restrict__ head,
                                                 const uint32_t *__restrict__ tail, uint32_t *__restrict__ OUT, uint32_t c_lo,
                                                 uint32_t c_hi, uint32_t tt, uint32_t c_tail)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
    const uint32_t stride = gridDim.x * 4u, c0 = c_lo + wave, voff256 = lane * 256u, voff128 = lane * 128u;
    // the transposes' stage 16 goes through LDS (gen/bs_gen.py: swap16_lds): a few KB per wave, the lane's own eight bytes of a slot
    // at 8 lane + 4 (lane / 16) -- the 32 lanes of one LDS cycle on 32 different banks
    __shared__ uint32_t bs_lds[4 * (HASH_BS_LDS_PER_WAVE / 4) + 4];
    const uint32_t vlds = (uint32_t)(uintptr_t)bs_lds + (threadIdx.x >> 6) * (uint32_t)HASH_BS_LDS_PER_WAVE + lane * 8u + (lane >> 4) * 4u;
    asm volatile(HASH_BS_ASM
                 :
                 : [t] "s"(packed), [p] "s"(tail), [hd] "s"(head), [o] "s"(OUT), [c0] "s"(c0), [n] "s"(c_hi), [stride] "s"(stride), [tt] "s"(tt), [ctail] "s"(c_tail),
                   [voff256] "v"(voff256), [voff128] "v"(voff128), [vlds] "v"(vlds),
                   [vco0] "v"(lane * 16u), [vco1] "v"(lane * 16u + 4096u), [vco2] "v"(lane * 16u + 8192u), [vco3] "v"(lane * 16u + 12288u)  // (gen/bs_gen.py --ablate coalesced only)
                 : HASH_BS_CLOBBERS);
}

}  // namespace mxg
