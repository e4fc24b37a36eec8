// nthash_dev.h -- device-side ntHash helpers shared by sketch.hip and sketch_bs.hip (semantics: SURVEY.md Appendix A.1-A.2,
// i.e. what `indexlr` of reference ntJoin:204-205 computes).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "mxg_internal.h"

namespace mxg {

// ------------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------------
struct H2 {
    uint32_t flo, fhi, rlo, rhi;
};

// one ntHash step:  fwd = srol(fwd) ^ t.xy ;  rev = sror(rev ^ t.zw)
// srol/sror = rotate the low 33 bits and the high 31 bits by one, each within itself.
__device__ __forceinline__ void nt_step(H2 &h, const uint4 t)
{
    uint32_t nlo = (h.flo << 1) | (h.fhi & 1u);                       // bit 32 -> bit 0
    uint32_t nhi = __builtin_amdgcn_alignbit(h.fhi, h.flo, 31);       // (fhi << 1) | (flo >> 31)
    nhi = (nhi & ~2u) | ((h.fhi >> 30) & 2u);                         // bit 63 -> bit 33
    h.flo = nlo ^ t.x;
    h.fhi = nhi ^ t.y;
    uint32_t xlo = h.rlo ^ t.z, xhi = h.rhi ^ t.w;
    h.rlo = __builtin_amdgcn_alignbit(xhi, xlo, 1);                   // (xlo >> 1) | (xhi << 31)
    h.rhi = ((xhi >> 1) & 0x7FFFFFFEu) | (xlo & 1u) | ((xhi & 2u) << 30);  // bit 0 -> bit 32, bit 33 -> bit 63
}

// 16 consecutive 2-bit bases starting at global base index `pos` (any alignment)
__device__ __forceinline__ uint32_t fetch16(const uint32_t *__restrict__ packed, uint64_t pos)
{
    uint64_t wi = pos >> 4;
    uint32_t sh = ((uint32_t)pos & 15u) * 2u;
    uint32_t lo = packed[wi], hi = packed[wi + 1];
    return __builtin_amdgcn_alignbit(hi, lo, sh);
}

template <int VARIANT>
__device__ __forceinline__ uint64_t canonical(const H2 &h)
{
    uint64_t f = ((uint64_t)h.fhi << 32) | h.flo, r = ((uint64_t)h.rhi << 32) | h.rlo;
    if (VARIANT == MXG_VARIANT_V1_MIN) return f <= r ? f : r;
    return f + r;
}

__device__ __forceinline__ bool is_forward(const H2 &h)
{
    uint64_t f = ((uint64_t)h.fhi << 32) | h.flo, r = ((uint64_t)h.rhi << 32) | h.rlo;
    return f <= r;
}

__device__ __forceinline__ uint64_t ext_hash(uint64_t h0, uint64_t mult)
{
    uint64_t t = h0 * mult;  // mult = 1 ^ (k * MULTISEED)
    return t ^ (t >> 27);
}

// locate strip s: run index lo with run_strip0[lo] <= s < run_strip0[lo+1]
__device__ __forceinline__ uint32_t find_run(const uint32_t *__restrict__ run_strip0, uint32_t lo, uint32_t hi, uint32_t s)
{
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (run_strip0[mid] <= s) lo = mid; else hi = mid;
    }
    return lo;
}

template <class T>
__device__ __forceinline__ void warm_up(H2 &h, const uint32_t *__restrict__ packed, uint64_t b, uint32_t k, const T *tab)
{
    for (uint32_t t = 0; t < k; t += 16) {  // k steps with no outgoing base
        uint32_t chunk = fetch16(packed, b + t);
        uint32_t n = min(16u, k - t);
        for (uint32_t u = 0; u < n; ++u) {
            nt_step(h, tab[16 + (chunk & 3u)]);
            chunk >>= 2;
        }
    }
}

// split rotation by 4 positions (one packed byte = 4 bases), by -4, and by a wave-uniform n
__device__ __forceinline__ void srol4(uint32_t &lo, uint32_t &hi)
{
    const uint32_t b32 = hi & 1u, W = hi >> 1;
    const uint32_t nlo = (lo << 4) | (b32 << 3) | (lo >> 29);
    const uint32_t nW = ((W << 4) | (W >> 27)) & 0x7FFFFFFFu;
    hi = (nW << 1) | ((lo >> 28) & 1u);
    lo = nlo;
}
__device__ __forceinline__ void sror4(uint32_t &lo, uint32_t &hi)
{
    const uint32_t b32 = hi & 1u, W = hi >> 1;
    const uint32_t nlo = (lo >> 4) | (b32 << 28) | (lo << 29);
    const uint32_t nW = (W >> 4) | ((W & 15u) << 27);
    hi = (nW << 1) | ((lo >> 3) & 1u);
    lo = nlo;
}
__device__ __forceinline__ void srol_var(uint32_t &lo, uint32_t &hi, uint32_t n)
{
    uint64_t V = ((uint64_t)(hi & 1u) << 32) | lo;
    uint32_t W = hi >> 1;
    const uint32_t a = n % 33u, b = n % 31u;
    if (a) V = ((V << a) | (V >> (33u - a))) & 0x1FFFFFFFFull;
    if (b) W = ((W << b) | (W >> (31u - b))) & 0x7FFFFFFFu;
    lo = (uint32_t)V;
    hi = (W << 1) | (uint32_t)(V >> 32);
}

// Hash state of a k-mer without k rolling steps.  With m = 4*(k/4) and v_q the q-th packed byte (4 bases),
//     F = XOR_q srol^{4(P-1-q)} f4[v_q]            f4[v] = XOR_u srol^{3-u} seed[c_u]      (make_init_tab)
//     R = srol^{k-m} XOR_q srol^{4q} r4[v_q]       r4[v] = XOR_u srol^{u}   seed'[c_u]
// both by Horner over the bytes in memory order: F <- srol^4(F) ^ f4[v_q];  T <- sror^4(T) ^ r4[v_q], and
// R = srol^{4(P-1) + k-m}(T).  One 256-entry table (4 KB, in LDS) for every k; then k%4 ordinary warm-up steps.
template <class T>
__device__ __forceinline__ void init_direct(H2 &h, const uint32_t *__restrict__ packed, uint64_t b, uint32_t k,
                                            const uint4 *byte_tab, const T *tab)
{
    const uint32_t P = k / 4;
    uint32_t flo = 0, fhi = 0, tlo = 0, thi = 0;
    for (uint32_t q = 0; q < P; q += 4) {  // 16 bases = 4 table bytes per fetch
        const uint32_t word = fetch16(packed, b + 4u * q);
        const uint32_t nb = min(4u, P - q);
        for (uint32_t u = 0; u < nb; ++u) {
            const uint4 e = byte_tab[(word >> (8 * u)) & 255u];
            srol4(flo, fhi);
            sror4(tlo, thi);
            flo ^= e.x; fhi ^= e.y; tlo ^= e.z; thi ^= e.w;
        }
    }
    const uint32_t rem = k - 4 * P;
    if (P) srol_var(tlo, thi, 4u * (P - 1u) + rem);
    h.flo = flo; h.fhi = fhi; h.rlo = tlo; h.rhi = thi;
    if (rem) {
        uint32_t chunk = fetch16(packed, b + 4u * P);
        for (uint32_t u = 0; u < rem; ++u) {
            nt_step(h, tab[16 + (chunk & 3u)]);
            chunk >>= 2;
        }
    }
}

// The same state from POSITION tables (make_init_tab, entries 256..): ptab[j][v] = {srol^{4(7-j)} f4[v], srol^{4j} r4[v]}, so a
// group of 8 bytes (32 bases) is 8 lookups and 32 XORs with no rotation at all; groups are combined by rotations of 32
// (k > 35 only) and the last t < 8 bytes use the tables j + 8 - t (forward) and j (reverse).  k = 32: one group, nothing else.
//     F = XOR_G srol^{4(P - 8(G+1))} F_G          R = srol^{k-m} XOR_G srol^{32 G} R_G
template <class T>
__device__ __forceinline__ void init_pos(H2 &h, const uint32_t *__restrict__ packed, uint64_t b, uint32_t k,
                                         const uint4 *ptab, const T *tab)
{
    const uint32_t P = k / 4;
    const uint32_t *pw = packed + (b >> 4);
    const uint32_t sh = ((uint32_t)b & 15u) * 2u;
    uint32_t flo = 0, fhi = 0, tlo = 0, thi = 0;
    uint32_t q = 0;
    for (; q + 8 <= P; q += 8) {
        const uint32_t w0 = pw[q >> 2], w1 = pw[(q >> 2) + 1], w2 = pw[(q >> 2) + 2];
        const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
        uint4 acc = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            const uint4 e0 = ptab[u * 256u + ((lo >> (8 * u)) & 255u)];
            const uint4 e1 = ptab[(4u + u) * 256u + ((hi >> (8 * u)) & 255u)];
            acc.x ^= e0.x ^ e1.x; acc.y ^= e0.y ^ e1.y; acc.z ^= e0.z ^ e1.z; acc.w ^= e0.w ^ e1.w;
        }
        if (q) {
            srol_var(flo, fhi, 32u);
            srol_var(acc.z, acc.w, 4u * q);
        }
        flo ^= acc.x; fhi ^= acc.y; tlo ^= acc.z; thi ^= acc.w;
    }
    const uint32_t t = P - q;  // bytes after the last full group
    if (t) {
        const uint32_t w0 = pw[q >> 2], w1 = pw[(q >> 2) + 1], w2 = pw[(q >> 2) + 2];
        const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
        uint32_t ax = 0, ay = 0, az = 0, aw = 0;
        for (uint32_t j = 0; j < t; ++j) {
            const uint32_t v = ((j < 4 ? lo : hi) >> (8u * (j & 3u))) & 255u;
            const uint4 ef = ptab[(j + 8u - t) * 256u + v], er = ptab[j * 256u + v];
            ax ^= ef.x; ay ^= ef.y; az ^= er.z; aw ^= er.w;
        }
        if (q) {
            srol_var(flo, fhi, 4u * t);
            srol_var(az, aw, 4u * q);
        }
        flo ^= ax; fhi ^= ay; tlo ^= az; thi ^= aw;
    }
    const uint32_t rem = k - 4 * P;
    if (rem) srol_var(tlo, thi, rem);
    h.flo = flo; h.fhi = fhi; h.rlo = tlo; h.rhi = thi;
    if (rem) {
        uint32_t chunk = fetch16(packed, b + 4u * P);
        for (uint32_t u = 0; u < rem; ++u) {
            nt_step(h, tab[16 + (chunk & 3u)]);
            chunk >>= 2;
        }
    }
}

// split rotation by 16 positions
__device__ __forceinline__ void srol16(uint32_t &lo, uint32_t &hi)
{
    const uint32_t b32 = hi & 1u, W = hi >> 1;
    const uint32_t nlo = (lo << 16) | (b32 << 15) | (lo >> 17);
    const uint32_t nW = ((W << 16) | (W >> 15)) & 0x7FFFFFFFu;
    hi = (nW << 1) | ((lo >> 16) & 1u);
    lo = nlo;
}

// k = 32 from HALF the position tables (16 KB): half[j][v] = {srol^{4(3-j)} f4[v], srol^{4j} r4[v]}, j = 0..3.  Bytes 0..3
// of the k-mer give Fa, Ra and bytes 4..7 give Fb, Rb through the same four tables;  F = srol^16(Fa) ^ Fb,
// R = Ra ^ srol^16(Rb).  8 lookups, 32 XORs and two rotations where init_direct walks 16 rotations by 4 (k_hash_sparse, once
// per strip, in the LDS the kernel reserves anyway).
__device__ __forceinline__ void init32_half(H2 &h, const uint32_t *__restrict__ packed, uint64_t b, const uint4 *half)
{
    const uint32_t *pw = packed + (b >> 4);
    const uint32_t sh = ((uint32_t)b & 15u) * 2u;
    const uint32_t w0 = pw[0], w1 = pw[1], w2 = pw[2];
    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
    uint4 a = make_uint4(0u, 0u, 0u, 0u), c = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
        const uint4 e0 = half[u * 256u + ((lo >> (8 * u)) & 255u)];
        const uint4 e1 = half[u * 256u + ((hi >> (8 * u)) & 255u)];
        a.x ^= e0.x; a.y ^= e0.y; a.z ^= e0.z; a.w ^= e0.w;
        c.x ^= e1.x; c.y ^= e1.y; c.z ^= e1.z; c.w ^= e1.w;
    }
    srol16(a.x, a.y);
    srol16(c.z, c.w);
    h.flo = a.x ^ c.x; h.fhi = a.y ^ c.y; h.rlo = a.z ^ c.z; h.rhi = a.w ^ c.w;
}

}  // namespace mxg
