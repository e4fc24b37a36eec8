// sketch_bs.hip -- the k = 32 route of the sketch stage (replaces `indexlr`, reference ntJoin:204-205; semantics SURVEY.md App. A).
//
//   k_hash_bs       (bs_kernels.h)  the bit-sliced ring filter over the whole assembly, straight from the 2-bit packed bases (bit
//                                   planes are made in registers): one bit per base position, "the 32-mer starting here may have
//                                   canonical hash < tau" (a superset)
//   k_bs_select     (here)          one WAVE per slice of 64 consecutive strips (lane = strip: H halo strips, T own strips, H halo
//                                   strips): the strips' bits of the bitmap -> the wave's LDS queue in position order -> exact 64-bit
//                                   hashes (position tables), entries >= tau dropped -> the window decision of k_resolve
//                                   (sketch.hip: L + R + 1 >= w) on the slice's OWN candidates, whose every possible neighbour
//                                   lies in the halo -> the selected ones laid out per slice for k_emit, candidate-free stretches
//                                   for k_gap_fix.  Nothing is counted, ordered or exchanged between slices: the candidates of the
//                                   assembly never exist as one array (k_bs_count / k_bs_reorder_w / k_resolve of sketch.hip remain
//                                   the route for what this kernel does not take).
// The halo: H strips hold at least w k-mers of the own strips' contig on either side, or reach that contig's end
// (bs_select_halo checks it against the run table).  So an own candidate sees every candidate within w - 1 k-mers, and one that
// has no candidate of its contig behind it inside the slice is followed by a candidate-free stretch of >= w k-mers for
// certain: where that stretch ends is found by walking on through the strips behind the slice (stretch_end).
#include <algorithm>
#include <cmath>
#include <mutex>

#include "bs_kernels.h"
#include "nthash_dev.h"
#include "scan_kernels.h"
#include "sketch_bs.h"

namespace mxg {

namespace {

// canonical hash (fwd + rev) of the 32-mer at base index b: position tables ptab[j][v] = {srol^{4(7-j)} f4[v], srol^{4j} r4[v]}
// (init_pos of nthash_dev.h at k = 32: 8 lookups, 32 XORs, no rotation)
// three consecutive words from a 4-byte aligned address in ONE request (the address unit is what this kernel keeps busiest)
struct __attribute__((packed, aligned(4))) Words3 {
    uint32_t w0, w1, w2;
};
struct __attribute__((packed, aligned(4))) Words4 {
    uint32_t w0, w1, w2, w3;
};
// byte B of x, times 16 (the byte offset of a 16-byte table entry): one SDWA shift instead of a field extract and a shift
template <int B>
__device__ __forceinline__ uint32_t byte_x16(const uint32_t x)
{
    uint32_t r;
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(B >= 0 && B < 4, "byte of a word");
    if (B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(4u), "v"(x));
    else if (B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(4u), "v"(x));
    else if (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(4u), "v"(x));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(4u), "v"(x));
#else
    r = ((x >> (8 * B)) & 255u) << 4;
#endif
    return r;
}
__device__ __forceinline__ uint32_t xor3(const uint32_t a, const uint32_t b, const uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);  // (one v_bitop3_b32; the compiler makes two v_xor_b32 of a ^ b ^ c)
#else
    return a ^ b ^ c;
#endif
}
// {forward lo, hi, reverse lo, hi}: the state nt_step (nthash_dev.h) rolls on
__device__ __forceinline__ uint4 hash32_state(const uint32_t w0, const uint32_t w1, const uint32_t w2, const uint32_t sh, const uint4 *ptab)
{
    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
    // the tables lie at offset 0 of the block's LDS (k_bs_select checks it), so an entry's address is its byte offset: the
    // table's 4096 j goes into the instruction's offset field, and nothing is added (through a generic pointer the compiler adds
    // the -- zero -- address of the dynamic LDS symbol to every one of them)
#if defined(__HIP_DEVICE_COMPILE__)
    (void)ptab;
    typedef const uint4 __attribute__((address_space(3))) *lds_u4;
    auto ent = [&](const uint32_t j, const uint32_t off) { return *(lds_u4)(uintptr_t)(off + j * 4096u); };
#else
    auto ent = [&](const uint32_t j, const uint32_t off) { return ptab[j * 256u + off / 16u]; };  // (host pass of the compiler: never called)
#endif
    const uint4 a0 = ent(0, byte_x16<0>(lo)), b0 = ent(4, byte_x16<0>(hi));
    uint4 acc = make_uint4(a0.x ^ b0.x, a0.y ^ b0.y, a0.z ^ b0.z, a0.w ^ b0.w);
    const uint4 a1 = ent(1, byte_x16<1>(lo)), b1 = ent(5, byte_x16<1>(hi));
    acc = make_uint4(xor3(acc.x, a1.x, b1.x), xor3(acc.y, a1.y, b1.y), xor3(acc.z, a1.z, b1.z), xor3(acc.w, a1.w, b1.w));
    const uint4 a2 = ent(2, byte_x16<2>(lo)), b2 = ent(6, byte_x16<2>(hi));
    acc = make_uint4(xor3(acc.x, a2.x, b2.x), xor3(acc.y, a2.y, b2.y), xor3(acc.z, a2.z, b2.z), xor3(acc.w, a2.w, b2.w));
    const uint4 a3 = ent(3, byte_x16<3>(lo)), b3 = ent(7, byte_x16<3>(hi));
    acc = make_uint4(xor3(acc.x, a3.x, b3.x), xor3(acc.y, a3.y, b3.y), xor3(acc.z, a3.z, b3.z), xor3(acc.w, a3.w, b3.w));
    return acc;
}
__device__ __forceinline__ uint64_t hash32_words(const uint32_t w0, const uint32_t w1, const uint32_t w2, const uint32_t sh, const uint4 *ptab)
{
    const uint4 acc = hash32_state(w0, w1, w2, sh, ptab);
    return (((uint64_t)acc.y << 32) | acc.x) + (((uint64_t)acc.w << 32) | acc.z);
}
__device__ __forceinline__ uint64_t hash32_pos(const uint32_t *__restrict__ packed, const uint64_t b, const uint4 *ptab)
{
    const Words3 v = *reinterpret_cast<const Words3 *>(packed + (b >> 4));
    return hash32_words(v.w0, v.w1, v.w2, ((uint32_t)b & 15u) * 2u, ptab);
}

// the bits of k-mers [32 j, 32 j + 32) of a strip of len k-mers whose first k-mer is base position b (sh = b % 32): LSB first
__device__ __forceinline__ uint32_t sel_bits(const uint32_t w_lo, const uint32_t w_hi, const uint32_t sh, const uint32_t len, const uint32_t j)
{
    const uint32_t bits = __builtin_amdgcn_alignbit(w_hi, w_lo, sh);
    const uint32_t k0 = 32u * j;
    const uint32_t nvalid = len > k0 ? min(len - k0, 32u) : 0u;
    return bits & (nvalid >= 32u ? 0xFFFFFFFFu : ((1u << nvalid) - 1u));
}

// a strip of the assembly's strip table as its lane holds it
struct StripRegs {
    uint64_t b;                // base position of its first k-mer
    uint32_t len, cg, k0, nk;  // k-mers, contig (~0: no strip), contig-local index of the first k-mer, k-mers of the contig
};
// sS = the strip's index x S (mod 2^32): the wave's first strip x S is scalar work, lane x S is computed once per kernel
__device__ __forceinline__ StripRegs strip_of(const BsSelParams &p, const bool in, const uint32_t sS, const uint32_t ri)
{
    StripRegs r;
    r.b = 0; r.len = 0; r.cg = 0xFFFFFFFFu; r.k0 = 0; r.nk = 0;
    if (in) {
        const RunX run = p.runx[ri];  // (32 bytes, aligned)
        const uint32_t j0 = sS - run.strip0S;  // k-mers of the run in front of the strip (the true value is below 2^32)
        r.len = min(p.S, run.n_kmers - j0);
        r.b = run.base_off + j0;
        r.cg = run.contig;
        r.k0 = run.kidx0 + j0;
        r.nk = run.nk;
    }
    return r;
}
// the same from a run entry that was requested earlier, whatever `in` says (the entry of a clamped index): nothing here waits for
// memory under a condition, so the compiler places the wait where the values are first used, not behind the request
__device__ __forceinline__ StripRegs strip_from_run(const BsSelParams &p, const bool in, const uint32_t sS, const RunX &run)
{
    const uint32_t j0 = sS - run.strip0S;
    StripRegs r;
    r.len = in ? min(p.S, run.n_kmers - j0) : 0u;
    r.b = in ? run.base_off + j0 : 0ull;
    r.cg = in ? run.contig : 0xFFFFFFFFu;
    r.k0 = in ? run.kidx0 + j0 : 0u;
    r.nk = in ? run.nk : 0u;
    return r;
}
__device__ __forceinline__ StripRegs load_strip(const BsSelParams &p, const int64_t s64)
{
    const bool in = s64 >= 0 && s64 < (int64_t)p.n_strips_asm;
    const uint32_t s = in ? (uint32_t)s64 : 0u;
    return strip_of(p, in, s * p.S, in ? p.strip_run[s] : 0u);
}

// inclusive prefix sum over the wave's 64 lanes in six DPP additions (row shifts inside the rows of 16 lanes, then the rows' last
// lanes broadcast to the rows behind them); the shuffle version of scan_kernels.h takes an LDS-crossbar round trip per step
__device__ __forceinline__ uint32_t wave_inclusive_dpp(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2 and 3
    return v;
}

// The kernel's own argument block, through a pointer the compiler cannot see through: the fields only the rare paths need
// (stretches, the global-memory regions, the final report) are then read where they are used -- one scalar load each -- instead of
// sitting in scalar registers for the whole kernel, where, with the rest of its loop-invariant values, they spilled into vector
// lanes (v_readlane / v_writelane: a tenth of the slice loop's vector instructions).
__device__ __forceinline__ const BsSelParams *sel_rare_params()
{
#if defined(__HIP_DEVICE_COMPILE__)
    const void *q = (const void *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return static_cast<const BsSelParams *>(q);
#else
    return nullptr;  // (host pass of the compiler: never called)
#endif
}
// A stretch longer than the stretch kernel holds (gap_nmax k-mers) is reported in pieces that overlap by one window: a piece after
// the first starts with the last window of the piece before it and is marked (SEL_GAP_DROP) to leave that window's arg-min out --
// the arg-min moves right with the window, so that is the only minimizer the two pieces can share (k_gap_fix's drop_idx; what
// k_stretch_tiles does between its tiles).  gap_nmax = 0: stretches are reported whole.
__device__ __forceinline__ void push_gap(uint4 *gaps, uint32_t gap_cap, uint32_t *ctrl, uint32_t nmax, uint32_t w, uint32_t c, uint32_t lo,
                                         uint32_t hi, uint32_t hint)
{
    uint32_t s = lo, flag = 0u;
    for (;;) {
        const uint32_t e = nmax && hi - s >= nmax ? s + nmax - 1u : hi;
        const uint32_t idx = atomicAdd(&ctrl[1], 1u);
        if (idx < gap_cap) gaps[idx] = make_uint4(c, s, e, hint | flag);
        if (e == hi) break;
        s = e - w + 1u;  // (the piece's last window; the next piece's own windows start one k-mer behind it)
        flag = SEL_GAP_DROP;
    }
}
__device__ __forceinline__ void sel_push_gap(const BsSelParams &, uint32_t c, uint32_t lo, uint32_t hi, uint32_t hint)
{
    const BsSelParams *q = sel_rare_params();
    push_gap(q->gaps, q->gap_cap, q->ctrl, q->gap_nmax, q->w, c, lo, hi, hint);
}

// The whole wave: contig c has no candidate from k-mer k_from up to the end of the strips the slice holds, and goes on behind
// them.  Walk on, 64 strips at a time, to the contig's first real candidate (exact hash < tau) or its end.  -> last k-mer of the stretch
__device__ __forceinline__ uint32_t stretch_end(const BsSelParams &p, const uint4 *ptab, const uint32_t lane, int64_t s_next, const uint32_t c,
                                                const uint32_t nk)
{
    const uint32_t nwords = (p.S + 31u) / 32u;
    for (;; s_next += 64) {
        const StripRegs r = load_strip(p, s_next + lane);
        const bool stop = r.cg != c;  // (behind the assembly's last strip: cg = ~0)
        uint32_t kfound = 0xFFFFFFFFu;
        if (!stop) {
            const uint32_t *bw = p.bm + (r.b >> 5);
            const uint32_t sh = (uint32_t)r.b & 31u;
            uint32_t w0 = bw[0];
            for (uint32_t j = 0; j < nwords && kfound == 0xFFFFFFFFu; ++j) {
                const uint32_t w1 = bw[j + 1u];
                uint32_t bits = sel_bits(w0, w1, sh, r.len, j);
                w0 = w1;
                for (; bits; bits &= bits - 1u) {
                    const uint32_t ju = 32u * j + (uint32_t)__builtin_ctz(bits);
                    if (hash32_pos(p.packed, r.b + ju, ptab) < p.tau) {
                        kfound = r.k0 + ju;
                        break;
                    }
                }
            }
        }
        const uint64_t mf = __ballot(kfound != 0xFFFFFFFFu), ms = __ballot(stop);
        const uint64_t ev = mf | ms;
        if (ev) {
            const int first = __builtin_ctzll(ev);
            if ((mf >> first) & 1ull) return (uint32_t)__builtin_amdgcn_readlane((int)kfound, first) - 1u;
            return nk - 1u;
        }
    }
}

constexpr uint32_t SEL_SI = 6;  // per-strip words a wave keeps in LDS: base position (2), contig, first k-mer, contig's k-mers, fold base

// bytes of LDS per wave (a multiple of 16): hashes | coordinates (raw queue before) | strip info | requests | counters
__host__ __device__ inline uint32_t sel_wave_lds(uint32_t qcap)
{
    return (qcap + 2u * SEL_PAD) * 12u + SEL_SI * 64u * 4u + SEL_REQ * 16u + SEL_REQ * 16u + 16u;
}

// What a wave knows about its slice while it works on it
struct SelCtx {
    const uint4 *ptab;
    uint32_t *si;      // strip info (LDS)
    uint4 *req;        // stretches that end behind the slice (LDS)
    uint32_t *misc;    // [0] number of requests, [1] the wave's overflow region, [2] number of stretches to sketch here
    uint32_t lane, sl, own_end;
    bool has_drop;     // pieces of records cut between shards are loaded (ctg_drop)
    int64_t s_first;
    uint32_t nreal, own_lo, own_hi;  // the list: real candidates, the own ones among them [own_lo, own_hi)
};

// The candidate list of a slice: the wave's LDS, or -- GLOB -- a region of global memory for a slice whose raw candidates
// outgrow the queue; entries -SEL_PAD .. cap + SEL_PAD - 1 of
//   le   before the hashes: raw queue, item = strip lane | k-mer of the strip << 6
//        behind them: e = (folded coordinate << 6) | strip lane of the real candidates (real: hash < tau), in order
//   lh   their hashes
// folded coordinate: position on one axis on which the slice's k-mers lie in order, k-mers of one contig at their distances
// and contigs >= w apart -- "same contig and within w - 1 k-mers" is one subtraction and one compare (on e itself: strip lanes
// rise with the coordinate, so e_a - e_b = 64 (x_a - x_b) + (lane_a - lane_b) and x_a - x_b = (e_a - e_b) >> 6).
template <bool GLOB>
__device__ __forceinline__ void sel_sync()
{
    if (GLOB) __threadfence_block();
    else __builtin_amdgcn_wave_barrier();  // (LDS operations of one wave complete in order)
}

// bits -> raw queue -> exact hashes -> the real candidates at the front of the list, in order (+ sentinels)
// bt[j]: the strip's bits of k-mers [32 j, 32 j + 32), LSB first (words beyond the strip: 0)
template <int NB, bool GLOB>
__device__ __forceinline__ void sel_collect(const BsSelParams &p, SelCtx &c, uint64_t *lh, uint32_t *le, const uint32_t cap,
                                            const uint32_t (&bt)[NB], const uint32_t at0, const uint32_t tot)
{
    const uint32_t lane = c.lane, H = p.H;
    const uint32_t qn = min(tot, cap);  // (tot <= cap: the caller chose the list for it)
    {
        uint32_t at = at0;
#pragma unroll
        for (uint32_t j = 0; j < (uint32_t)NB; ++j) {
            uint32_t bits = bt[j];
            const uint32_t item0 = lane | ((32u * j) << 6);
            for (; bits; bits &= bits - 1u, ++at) le[at] = item0 + ((uint32_t)__builtin_ctz(bits) << 6);
        }
    }
    const uint32_t *s_blo = c.si, *s_bhi = c.si + 64, *s_f = c.si + 320;
    sel_sync<GLOB>();
    c.nreal = c.own_lo = c.own_hi = 0;
    if (p.ablate == 2) return;  // (profiling)
    uint32_t nreal = 0, own_lo = 0, own_hi = 0;
    // four rounds of 64 candidates at a time: the rounds' packed bases are all requested before the first round's table lookups,
    // so a slice waits once for them, not once per round (a round's request depends on nothing but the queue)
    constexpr uint32_t NR = 4;
    for (uint32_t q0 = 0; q0 < qn; q0 += 64u * NR) {
        uint32_t item[NR];
        Words3 wv[NR];
        uint32_t shv[NR];
#pragma unroll
        for (uint32_t r = 0; r < NR; ++r) {
            const uint32_t q = q0 + 64u * r + lane;
            const bool act = q < qn;
            item[r] = act ? le[q] : 0u;
            const uint32_t l2 = item[r] & 63u;
            const uint64_t b = (((uint64_t)s_bhi[l2] << 32) | s_blo[l2]) + (item[r] >> 6);
            wv[r] = Words3{0u, 0u, 0u};
            if (act) wv[r] = *reinterpret_cast<const Words3 *>(p.packed + (b >> 4));
            shv[r] = ((uint32_t)b & 15u) * 2u;
        }
#pragma unroll
        for (uint32_t r = 0; r < NR; ++r) {
            if (q0 + 64u * r >= qn) break;  // (wave-uniform)
            const bool act = q0 + 64u * r + lane < qn;
            const uint32_t l2 = item[r] & 63u, ju = item[r] >> 6;
            const uint64_t h = act ? hash32_words(wv[r].w0, wv[r].w1, wv[r].w2, shv[r], c.ptab) : ~0ull;
            const bool real = act && h < p.tau;
            const uint64_t m = __ballot(real);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (real) {  // (in place: an entry lands at or in front of the item it came from, and the batch's items have all been read)
                le[nreal + rank] = ((s_f[l2] + ju) << 6) | l2;
                lh[nreal + rank] = h;
            }
            own_lo += (uint32_t)__popcll(__ballot(real && l2 < H));
            own_hi += (uint32_t)__popcll(__ballot(real && l2 < c.own_end));
            nreal += (uint32_t)__popcll(m);
        }
    }
    if (lane < SEL_PAD) {  // sentinels: out of every scan's reach (the folded coordinates start at 2 w), hashes that never block
        le[-1 - (int)lane] = 0u;
        lh[-1 - (int)lane] = ~0ull;
        le[nreal + lane] = 0xFFFFFFFFu;
        lh[nreal + lane] = ~0ull;
    }
    sel_sync<GLOB>();
    c.nreal = nreal; c.own_lo = own_lo; c.own_hi = own_hi;
}

// the window decision (k_resolve's, sketch.hip) on the own candidates, stretch detection, the selected ones out in order
template <bool GLOB>
__device__ __forceinline__ void sel_decide(const BsSelParams &p, SelCtx &c, const uint64_t *lh, const uint32_t *le, const StripRegs &sr,
                                           const uint32_t f, bool &flag)
{
    const uint32_t lane = c.lane, sl = c.sl, H = p.H, w = p.w, wm1 = w - 1u, nreal = c.nreal;
    const bool has_drop = c.has_drop;
    const uint32_t lim = 64u * wm1 + 63u;  // e-distance: within w - 1 k-mers
    const uint32_t *s_c = c.si + 128, *s_k0 = c.si + 192, *s_nk = c.si + 256, *s_f = c.si + 320;
    // the contig that leaves the slice at its far end (none: ~0) -- every other contig of the slice ends inside it
    const uint32_t cov_c = (uint32_t)__builtin_amdgcn_readlane((int)sr.cg, 63);
    const uint32_t cov_kend = (uint32_t)__builtin_amdgcn_readlane((int)(sr.k0 + sr.len), 63);
    const uint32_t cov_nk = (uint32_t)__builtin_amdgcn_readlane((int)sr.nk, 63);
    const bool cov_open = cov_c != 0xFFFFFFFFu && cov_kend < cov_nk;
    // a stretch from k_from on, reported by the candidate (or contig start) in front of it; the next real candidate is entry nx
    // a stretch of the slice: into the wave's list for k_sel_stretch (no room there, or that kernel is off: straight to k_gap_fix)
    // (where the list's entries go is decided when the slice is done: one branch here, no parameter read)
    auto push_req = [&](uint32_t cg, uint32_t lo, uint32_t hi) {
        const uint32_t q = atomicAdd(&c.misc[2], 1u);
        if (q < SEL_REQ) reinterpret_cast<uint4 *>(c.misc + 4)[q] = make_uint4(cg, lo, hi, 0u);
        else sel_push_gap(p, cg, lo, hi, sl * p.rk);
    };
    auto report = [&](uint32_t cg, uint32_t k_from, uint32_t nk, uint32_t nx) {
        bool same = false;
        uint32_t k2 = 0;
        if (nx < nreal) {
            const uint32_t e2 = le[nx], l2 = e2 & 63u;
            same = s_c[l2] == cg;
            k2 = s_k0[l2] + ((e2 >> 6) - s_f[l2]);
        }
        if (same) {
            if (k2 - k_from >= w) push_req(cg, k_from, k2 - 1u);
        } else if (nk - k_from >= w) {
            if (!(cov_open && cg == cov_c)) {
                push_req(cg, k_from, nk - 1u);
            } else {  // the stretch's end lies behind the slice: the wave walks there (below)
                const uint32_t at = atomicAdd(&c.misc[0], 1u);
                if (at < SEL_REQ) c.req[at] = make_uint4(cg, k_from, nk, 0u);
            }
        }
    };
    // L = k-mers on the left with hash >= this one's (ties: the rightmost of equals wins), R = on the right with hash > this
    // one's, both capped by the window and the contig: a window of w k-mers in which this k-mer is the rightmost minimum exists
    // iff L + R + 1 >= w.  Eight neighbours per step, requested at once and decided with selects -- as branches per neighbour
    // (what the compiler makes of short-circuit conditions) each one cost two dependent LDS round trips.  Distances are taken on
    // e itself (see above); the right scan runs only for candidates the left scan left room for, as far as that room needs.
    uint32_t n_sel = 0;
    for (uint32_t i0 = c.own_lo; i0 < c.own_hi; i0 += 64u) {
        const bool live = i0 + lane < c.own_hi;
        const uint32_t i = live ? i0 + lane : i0;
        const uint32_t e = le[i], l1 = e & 63u;
        const uint64_t h = lh[i];
        const uint32_t kx = s_k0[l1] + ((e >> 6) - s_f[l1]), cg = s_c[l1], nk = s_nk[l1];
        // (no running "done" flag inside a step, and no distances: on the left the nearest neighbour with a smaller hash is the one
        // with the LARGEST e among those with a smaller hash -- a maximum, which needs no order and no mask arithmetic in scalar
        // registers -- and it lies inside the window iff that e >= e - lim; the step's farthest neighbour says whether the
        // window's end has been passed.  Two and a half vector instructions per neighbour: compare, select, half a max3.)
        const uint32_t e_lo = e - lim;  // (e >= 128 w > lim: the folded coordinates start at 2 w)
        uint32_t al = 0u;
        bool ldone = !live;
        for (uint32_t t = 1; !ldone; t += SEL_PAD) {
            uint32_t ae[SEL_PAD];
            uint64_t ah[SEL_PAD];
#pragma unroll
            for (uint32_t u = 0; u < SEL_PAD; ++u) {
                ae[u] = le[(int)i - (int)t - (int)u];
                ah[u] = lh[(int)i - (int)t - (int)u];
            }
            static_assert(SEL_PAD == 8, "the scans look at eight neighbours per step");
            const uint32_t c0 = ah[0] < h ? ae[0] : 0u, c1 = ah[1] < h ? ae[1] : 0u, c2 = ah[2] < h ? ae[2] : 0u, c3 = ah[3] < h ? ae[3] : 0u;
            const uint32_t c4 = ah[4] < h ? ae[4] : 0u, c5 = ah[5] < h ? ae[5] : 0u, c6 = ah[6] < h ? ae[6] : 0u, c7 = ah[7] < h ? ae[7] : 0u;
            al = max(max(max(c0, c1), c2), max(max(max(c3, c4), c5), max(c6, c7)));
            ldone = al >= e_lo || ae[SEL_PAD - 1] < e_lo;
        }
        const uint32_t L = al >= e_lo ? ((e - al) >> 6) - 1u : min(kx, wm1);
        const uint32_t R0 = min(nk - 1u - kx, wm1);
        bool s = live && (L + R0 + 1u >= w);
        // blocked by a smaller-or-equal hash at an e up to this one (e < 2^31: the folded axis of a slice is short)
        const uint32_t e_hi = e + 64u * (wm1 - min(L, wm1)) + 63u;
        bool rdone = !(s && L < wm1);
        for (uint32_t t = 1; !rdone; t += SEL_PAD) {
            uint32_t be[SEL_PAD];
            uint64_t bh[SEL_PAD];
#pragma unroll
            for (uint32_t u = 0; u < SEL_PAD; ++u) {
                be[u] = le[i + t + u];
                bh[u] = lh[i + t + u];
            }
            const uint32_t n = 0xFFFFFFFFu;
            const uint32_t c0 = bh[0] <= h ? be[0] : n, c1 = bh[1] <= h ? be[1] : n, c2 = bh[2] <= h ? be[2] : n, c3 = bh[3] <= h ? be[3] : n;
            const uint32_t c4 = bh[4] <= h ? be[4] : n, c5 = bh[5] <= h ? be[5] : n, c6 = bh[6] <= h ? be[6] : n, c7 = bh[7] <= h ? be[7] : n;
            const uint32_t bl = min(min(min(c0, c1), c2), min(min(min(c3, c4), c5), min(c6, c7)));
            const bool blocked = bl <= e_hi;
            s = blocked ? false : s;
            rdone = blocked || be[SEL_PAD - 1] > e_hi;
        }
        // a piece of a record that starts with the halo of the shard before it: see k_resolve
        if (has_drop && s && kx <= wm1 && L == kx && sel_rare_params()->ctg_drop[cg]) s = false;
        const uint64_t bm = __ballot(s);
        if (s) {
            const uint32_t dst = n_sel + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
            if (dst < p.rk) p.cs[(size_t)sl * p.rk + dst] = make_uint4((uint32_t)h, (uint32_t)(h >> 32), kx, cg);
        }
        n_sel += (uint32_t)__popcll(bm);
        // candidate-free stretches: the candidate in front of a stretch reports it
        if (live && le[i + 1u] - e >= 64u * (w + 1u)) report(cg, kx + 1u, nk, i + 1u);
    }
    // ... and a contig's first k-mer reports the stretch in front of the contig's first candidate (contigs without any: all of it)
    if (lane >= H && lane < c.own_end && sr.len && sr.k0 == 0) {
        uint32_t lo = 0, hi = nreal;  // first real candidate at or behind this strip
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((le[mid] >> 6) < f) lo = mid + 1u; else hi = mid;
        }
        report(sr.cg, 0u, sr.nk, lo);
    }
    sel_sync<GLOB>();
    if (lane == 0) {
        if (n_sel > p.rk) flag = true;
        count_publish(p.cnt, p.sup, sl, min(n_sel, p.rk));
    }
    // ---- stretches that end behind the slice
    const uint32_t n_req = c.misc[0];
    if (n_req) {
        if (n_req > SEL_REQ) flag = true;
        for (uint32_t r = 0; r < min(n_req, SEL_REQ); ++r) {
            const uint4 rq = c.req[r];
            const uint32_t k_end = stretch_end(p, c.ptab, lane, c.s_first + 64, rq.x, rq.z);
            if (lane == 0 && k_end + 1u - rq.y >= w) push_req(rq.x, rq.y, k_end);
        }
        sel_sync<GLOB>();
        if (lane == 0) c.misc[0] = 0;
    }
    // ---- the slice's stretches leave for k_sel_stretch together (one reservation; each says which of how many it is, so that ONE
    // wave of that kernel takes all of a slice's stretches -- they go into one row)
    if (__builtin_expect(c.misc[2] != 0u, 0)) {
        const BsSelParams *pr = sel_rare_params();
        const uint32_t n_i = min(c.misc[2], SEL_REQ);
        const bool to_kernel = pr->inl_amax != 0u;  // (else: every stretch straight to k_gap_fix)
        uint32_t base = 0xFFFFFFFFu - SEL_REQ;
        if (to_kernel && lane == 0) base = atomicAdd(&pr->ctrl[15], n_i);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        uint32_t ln = lane;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(ln));  // (the lane's place among the requests is computed HERE: hoisted out of the slice loop it took a register there)
#endif
        if (ln < n_i) {
            const uint32_t *g = c.misc + 4u + 4u * ln;
            if (base + n_i <= pr->ireq_cap) pr->ireq[base + ln] = make_uint4(g[0], g[1], g[2], sl | (ln << 24) | (n_i << 27));
            else sel_push_gap(p, g[0], g[1], g[2], sl * pr->rk);  // (no room: the stretch kernels take them)
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) c.misc[2] = 0;
    }
}

}  // namespace

// NWC = bitmap words a lane keeps of its strip: S / 32 rounded up + 1, as whole 16-byte requests
// NWX = the strip's words of bits when they are known where the kernel is compiled (S = 32 NWX: the default strip of 320 k-mers
// has its own instance), else 0: "word j belongs to the strip" is then no select per word on a mask kept in scalar registers
template <int NWC, int NWX>
__global__ __launch_bounds__(1024) void k_bs_select(const BsSelParams p)
{
    static_assert(NWC % 4 == 0, "the strips' bitmap words are requested four at a time");
    static_assert(NWX < NWC, "one word more than the strip's bits is read (the bits are shifted into place)");
    constexpr int NB = NWX ? NWX : NWC - 1;  // words of bits a lane keeps
    extern __shared__ uint4 sel_lds[];
    uint4 *ptab = sel_lds;
    if ((uint32_t)(uintptr_t)sel_lds != 0u) __builtin_trap();  // (hash32_words reads the tables at LDS offset 0: no static LDS in this kernel)
    for (uint32_t i = threadIdx.x; i < 2048u; i += blockDim.x) ptab[i] = p.ptab[i];
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    unsigned char *wb = reinterpret_cast<unsigned char *>(sel_lds + 2048) + (size_t)wib * sel_wave_lds(p.qcap);
    uint64_t *lh = reinterpret_cast<uint64_t *>(wb);
    uint32_t *le = reinterpret_cast<uint32_t *>(lh + p.qcap + 2u * SEL_PAD);
    SelCtx c;
    c.ptab = ptab;
    c.si = le + p.qcap + 2u * SEL_PAD;
    c.req = reinterpret_cast<uint4 *>(c.si + SEL_SI * 64u);
    c.misc = reinterpret_cast<uint32_t *>(c.req + SEL_REQ);  // (the slice's stretches for k_sel_stretch: the SEL_REQ entries behind misc's four words)
    c.lane = lane;
    c.has_drop = p.ctg_drop != nullptr;
    c.nreal = c.own_lo = c.own_hi = 0;
    // the block's slices are handed out to its waves one by one (a ticket counter in LDS behind the waves' areas): a wave's 41
    // slices of its own differ by +-4 % in time from wave to wave and the launch waits for the slowest of 4096 (PMC, round 5: a
    // wave lived 0.82 of the launch on average); the block's 650 slices differ by +-1 % from block to block
    uint32_t *blk_ticket = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(sel_lds + 2048) + (size_t)nwv * sel_wave_lds(p.qcap));
    if (threadIdx.x == 0) *blk_ticket = nwv;  // (tickets 0 .. nwv - 1: the waves' first slices)
    if (lane == 0) {
        c.misc[0] = 0;
        c.misc[2] = 0;
        c.misc[1] = 0xFFFFFFFFu;  // this wave's region of global memory, once it has needed one (kept in LDS: a loop-carried scalar
                                  // for a path one slice in 10^5 takes cost every slice a wait -- the compiler's phi of it)
    }
    __syncthreads();
    const uint32_t S = p.S, H = p.H, T = p.T, w = p.w;
    const uint32_t nwords = NWX ? (uint32_t)NWX : (S + 31u) / 32u;
    const uint32_t stride = gridDim.x * nwv, laneS = lane * S;
    uint32_t own_cands = 0;
    bool flag = false;
    // (ticket t of the block = slice (t / waves) x stride + block x waves + t % waves: the slices a block's waves took in turn when
    // every wave had every stride-th slice; runs of consecutive slices per wave measured 4 % slower)
    auto slice_of = [&](uint32_t t) { return (t / nwv) * stride + blockIdx.x * nwv + (t % nwv); };
    uint32_t sl = slice_of(wib);
    const uint32_t sl_end = p.n_slices;
    // a slice's strips are looked up one slice ahead: the strip -> run table while the slice before is being set up, the run
    // itself while that slice's candidates are being decided (nothing there waits for memory)
    // (strips are counted in 32 bits, bs_select_geom: a first strip "in front of the assembly" wraps to a number no assembly has)
    auto first_strip = [&](uint32_t q) { return (int64_t)(p.strip_lo + q * T) - (int64_t)H; };
    StripRegs sr = sl < sl_end ? load_strip(p, first_strip(sl) + lane) : StripRegs{0, 0, 0xFFFFFFFFu, 0, 0};
    while (sl < sl_end) {
        uint32_t tk = 0;
        if (lane == 0) tk = atomicAdd(blk_ticket, 1u);
        const uint32_t sl_n = slice_of((uint32_t)__builtin_amdgcn_readfirstlane((int)tk));  // the wave's next slice
        const uint32_t s_own0 = p.strip_lo + sl * T;
        c.sl = sl;
        c.s_first = first_strip(sl);
        c.own_end = H + min(T, p.strip_hi - s_own0);  // lanes [H, own_end) hold the slice's own strips
        const uint32_t sn32 = p.strip_lo + sl_n * T - H + lane;
        const bool in_n = sl_n < sl_end && sn32 < p.n_strips_asm;
        const uint32_t s_n = in_n ? sn32 : 0u;
        const uint32_t ri_n = p.strip_run[s_n];  // (unconditional, from a clamped index: a request under a condition is waited for at once)
        const uint32_t sS_n = (sn32 - lane) * S + laneS;  // (wrong only where in_n is false)
        // the strip's words of the bitmap, four per request
        uint32_t wd[NWC];
        {
            const Words4 *bw = reinterpret_cast<const Words4 *>(p.bm + (sr.b >> 5));
#pragma unroll
            for (uint32_t u = 0; u < (uint32_t)NWC / 4u; ++u) {
                Words4 v{0u, 0u, 0u, 0u};
                if (sr.len && 4u * u <= nwords) v = bw[u];
                wd[4 * u] = v.w0; wd[4 * u + 1] = v.w1; wd[4 * u + 2] = v.w2; wd[4 * u + 3] = v.w3;
            }
        }
        // fold: k-mers of one contig at their distances, a contig border = w more; the first strip starts at 2 w
        // (wave_shr:1: the lane in front; lane 0 keeps `old`)
        const uint32_t len_p = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sr.len, 0x138, 0xf, 0xf, false);
        const uint32_t cg_p = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)sr.cg, 0x138, 0xf, 0xf, false);
        const uint32_t f = wave_inclusive_dpp(lane ? len_p + (sr.cg != cg_p ? w : 0u) : 2u * w);
        c.si[lane] = (uint32_t)sr.b;
        c.si[64u + lane] = (uint32_t)(sr.b >> 32);
        c.si[128u + lane] = sr.cg;
        c.si[192u + lane] = sr.k0;
        c.si[256u + lane] = sr.nk;
        c.si[320u + lane] = f;
        // the strip's bits, word by word (k-mers [32 j, 32 j + 32), LSB first): shifted into place, cut at the strip's length --
        // positions whose 32-mer crosses the run's end are set in the bitmap (the filter sees bases, not runs).  When every strip
        // of the slice is whole (nearly always) nothing needs cutting.
        uint32_t bt[NB];
        uint32_t cnt = 0;
        {
            const uint32_t sh = (uint32_t)sr.b & 31u;
            const bool whole = (NWX || (S & 31u) == 0) && __ballot(sr.len != 0 && sr.len != S) == 0;
            if (whole) {
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)NB; ++j) bt[j] = j < nwords ? __builtin_amdgcn_alignbit(wd[j + 1], wd[j], sh) : 0u;
            } else {
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)NB; ++j) bt[j] = j < nwords ? sel_bits(wd[j], wd[j + 1], sh, sr.len, j) : 0u;
            }
#pragma unroll
            for (uint32_t j = 0; j < (uint32_t)NB; ++j) cnt += (uint32_t)__popc(bt[j]);
        }
        const uint32_t incl = wave_inclusive_dpp(cnt);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        // the next slice's strips: their run indices have arrived while this slice's bitmap words travelled.  (Asking ahead for the
        // next slice's bitmap and base lines as well, or touching this slice's base lines while the bitmap words travel, was
        // measured: 1.6 x the kernel's HBM traffic -- the waves of an XCD hold more lines than its L2 -- for no time at all.)
        const RunX run_n = p.runx[ri_n];  // (in flight until the end of the slice)
        if (p.ablate == 1) {  // (profiling)
            if (lane == 0) count_publish(p.cnt, p.sup, sl, 0u);
            own_cands += tot != 0;
        } else if (tot <= p.qcap) {
            sel_collect<NB, false>(p, c, lh + SEL_PAD, le + SEL_PAD, p.qcap, bt, incl - cnt, tot);
            if (p.ablate == 2 || p.ablate == 3) {
                if (lane == 0) count_publish(p.cnt, p.sup, sl, 0u);
                own_cands += 1u;
            } else {
                sel_decide<false>(p, c, lh + SEL_PAD, le + SEL_PAD, sr, f, flag);
            }
        } else {
            const BsSelParams *q = sel_rare_params();
            if (lane == 0 && c.misc[1] == 0xFFFFFFFFu) c.misc[1] = atomicAdd(q->ovf_next, 1u);
            __builtin_amdgcn_wave_barrier();
            const uint32_t region = c.misc[1];
            if (region < q->n_ovf) {
                const uint32_t ocap = q->ovf_cap;
                const size_t o = (size_t)region * (ocap + 2u * SEL_PAD) + SEL_PAD;
                sel_collect<NB, true>(p, c, q->ovf_h + o, q->ovf_e + o, ocap, bt, incl - cnt, tot);
                sel_decide<true>(p, c, q->ovf_h + o, q->ovf_e + o, sr, f, flag);
            } else {  // no region left: the host redoes the batch
                flag = true;
                c.own_lo = c.own_hi = 0;
                if (lane == 0) count_publish(p.cnt, p.sup, sl, 0u);
            }
        }
        own_cands += c.own_hi - c.own_lo;
        __builtin_amdgcn_wave_barrier();  // the next slice reuses the wave's LDS
        sr = strip_from_run(p, in_n, sS_n, run_n);
        sl = sl_n;
    }
    const uint32_t own_w = own_cands;  // (wave-uniform)
    if (lane == 0) {
        const BsSelParams *q = sel_rare_params();
        if (own_w) atomicAdd(&q->cand_spread[((blockIdx.x * nwv + wib) & 63u) * 32u], own_w);
        if (flag) q->ctrl[6] = q->ctrl[13] = 1;  // ([13]: it was this kernel that gave up)
    }
}

namespace {

// ---------------------------------------------------------------------------------------------------------------
// The candidate-free stretches between two candidates of one slice: sketched one wave per slice (round 6): k_sel_stretch
// ---------------------------------------------------------------------------------------------------------------
// A stretch of n = w + a k-mers (all of one run: their bases are consecutive) has a + 1 windows, and with a < w every one of
// them holds the stretch's middle [a, w): window j = k-mers [j, a) of the front part, the middle, and [w, w + j) of the back part.
// So the windows' arg-mins are those of a sliding window of a + 1 entries over
//     X = front part (a entries) | M = the rightmost minimum of the middle | back part (a entries),
// window j = min(suffix minimum of X[j .. a], prefix minimum of X[a + 1 .. a + j]), the right one of equals (btllib rescans with
// <=): two scans over a + 1 <= 64 SEL_INL_R entries, a lane holding SEL_INL_R consecutive ones.  Stretches with more windows are taken in
// pieces that overlap by one window, a piece after the first leaving out the arg-min of the window it shares with the piece before
// it (the rule of sel_push_gap / k_gap_fix).  The hashes: a lane rolls over ceil(n / 64) consecutive k-mers from the direct formula
// at its first one.  The stretch's minimizers become entries of the slice's own row, between the candidate in front of the stretch
// and the one behind it: k_gap_fix / k_gap_post / the placing blocks of k_emit never see the stretch.  This is what lets the
// filter's threshold come down (fewer candidates per window, e^-c of them followed by a stretch).
struct InlBest {
    uint64_t h;
    uint32_t id;  // k-mer of the piece (~0: none)
};
// what the lane CTRL names holds (DPP; lanes without a source: "none")
template <int CTRL, int ROWS>
__device__ __forceinline__ InlBest inl_dpp(const InlBest v)
{
    InlBest o;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)v.h, CTRL, ROWS, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)(v.h >> 32), CTRL, ROWS, 0xf, false);
    o.h = ((uint64_t)hi << 32) | lo;
    o.id = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v.id, CTRL, ROWS, 0xf, false);
    return o;
}
// Inclusive scan over the lanes, from lane 0 up, of "the smaller hash, the RIGHT one of equals".  OWN_LEFT = false: the lanes hold
// the segments in order (what comes from a lower lane lies to the left: it wins only with a smaller hash); OWN_LEFT = true: the
// lanes hold them in REVERSE order (what comes from a lower lane lies to the right: it wins unless the own hash is smaller).
template <bool OWN_LEFT>
__device__ __forceinline__ InlBest inl_scan(InlBest v)
{
#define MXG_INL_STEP(CTRL, ROWS)                                                   \
    {                                                                              \
        const InlBest g = inl_dpp<CTRL, ROWS>(v);                                  \
        const bool take = OWN_LEFT ? (g.id != 0xFFFFFFFFu && !(v.h < g.h)) : (g.h < v.h); \
        if (take) v = g;                                                           \
    }
    MXG_INL_STEP(0x111, 0xf)  // row_shr:1
    MXG_INL_STEP(0x112, 0xf)  // row_shr:2
    MXG_INL_STEP(0x114, 0xf)  // row_shr:4
    MXG_INL_STEP(0x118, 0xf)  // row_shr:8
    MXG_INL_STEP(0x142, 0xa)  // row_bcast:15 -> rows 1 and 3
    MXG_INL_STEP(0x143, 0xc)  // row_bcast:31 -> rows 2 and 3
#undef MXG_INL_STEP
    return v;
}
__device__ __forceinline__ InlBest inl_lane(const InlBest v, const int src_lane)
{
    InlBest o;
    o.h = __shfl(v.h, src_lane);
    o.id = __shfl(v.id, src_lane);
    return o;
}

// all 64 lanes of one wave; X: 2 amax + 1 words of LDS, tmp: cap_t entries
// -> entries written to tmp (0: none; ~0: more than cap_t, the stretch goes to k_gap_fix)
__device__ __forceinline__ uint32_t stretch_sketch(const SelStretchParams &p, const uint4 *byte_tab, const uint4 *rtab, uint64_t *X, uint4 *tmp,
                                                   const uint32_t cap_t, const uint32_t cg, const uint32_t k_from, const uint32_t k_to,
                                                   const uint64_t b_from, const bool drop0)
{
    const uint32_t lane = threadIdx.x & 63u, w = p.w, amax = p.amax;
    const uint32_t *__restrict__ packed = p.packed;
    const uint32_t a_tot = k_to - k_from + 1u - w;
    uint32_t m = 0;
    for (uint32_t lo = 0;;) {
        const uint32_t a = min(a_tot - lo, amax), n = w + a;
        // ---- exact hashes: front and back part into X, the middle's minimum in registers
        const uint32_t L = (n + 63u) >> 6;
        const uint32_t j0 = lane * L, jn = j0 < n ? min(L, n - j0) : 0u;  // the lane's k-mers [j0, j0 + jn)
        uint64_t mh = ~0ull;
        uint32_t mj = 0xFFFFFFFFu;
        // (no branches: every k-mer is stored -- the middle's into a slot nobody reads -- and every k-mer is compared)
        auto take = [&](const uint32_t jj, const uint64_t h) {
            const bool mid = jj >= a && jj < w;
            const uint32_t slot = jj < a ? jj : (mid ? 2u * a + 1u : jj - w + a + 1u);
            X[slot] = h;
            const bool better = mid && h <= mh;
            mh = better ? h : mh;
            mj = better ? jj : mj;
        };
        const uint64_t bl = b_from + lo + j0;  // first base of the lane's first k-mer
        H2 st = {0u, 0u, 0u, 0u};
        // 32 rolls at a time from five words requested together (the lane's first k-mer from the first three: init_direct's
        // formula at k = 32); roll r takes k-mer r to k-mer r + 1: base r leaves, base r + 32 comes in.  Nothing is read behind
        // the word of the lane's last base.
        for (uint32_t q = 0; q < L; q += 32u) {  // (wave-uniform; q = 0 also when L = 1)
            const uint32_t rolls = q + 1u < jn && !(p.ablate & 1u) ? min(32u, jn - 1u - q) : 0u;
            const uint64_t bq = bl + q;
            const uint32_t *pw = packed + (bq >> 4);
            const uint32_t last = jn ? (uint32_t)(((bq + 31u + rolls) >> 4) - (bq >> 4)) : 0u;  // word of the last base needed (>= 1 when jn)
            uint32_t wd[5] = {0u, 0u, 0u, 0u, 0u};
            if (jn && (q == 0u || rolls)) {
                const Words3 v = *reinterpret_cast<const Words3 *>(pw);  // (last >= 1; word 2 exists: the k-mer's bases 32 .. reach it or the pad does)
                wd[0] = v.w0; wd[1] = v.w1; wd[2] = last >= 2u ? v.w2 : 0u;
                if (last >= 3u) wd[3] = pw[3];
                if (last >= 4u) wd[4] = pw[4];
            }
            const uint32_t sh = ((uint32_t)bq & 15u) * 2u;
            if (q == 0u && jn && !(p.ablate & 2u)) {
                const uint32_t lo16 = __builtin_amdgcn_alignbit(wd[1], wd[0], sh), hi16 = __builtin_amdgcn_alignbit(wd[2], wd[1], sh);
                uint32_t flo = 0, fhi = 0, tlo = 0, thi = 0;
#pragma unroll
                for (uint32_t u = 0; u < 8u; ++u) {
                    const uint4 e = byte_tab[((u < 4u ? lo16 : hi16) >> (8u * (u & 3u))) & 255u];
                    srol4(flo, fhi);
                    sror4(tlo, thi);
                    flo ^= e.x; fhi ^= e.y; tlo ^= e.z; thi ^= e.w;
                }
                srol_var(tlo, thi, 28u);
                st = {flo, fhi, tlo, thi};
                take(j0, canonical<MXG_VARIANT_V2_SUM>(st));
            }
#pragma unroll
            for (uint32_t half = 0; half < 2u; ++half) {
                const uint32_t out16 = __builtin_amdgcn_alignbit(wd[half + 1u], wd[half], sh);
                const uint32_t in16 = __builtin_amdgcn_alignbit(wd[half + 3u], wd[half + 2u], sh);
#pragma unroll
                for (uint32_t t = 0; t < 16u; ++t) {
                    if (16u * half + t < rolls) {
                        const uint32_t o = (out16 >> (2u * t)) & 3u, in = (in16 >> (2u * t)) & 3u;
                        nt_step(st, rtab[o * 4u + in]);
                        take(j0 + q + 16u * half + t + 1u, canonical<MXG_VARIANT_V2_SUM>(st));
                    }
                }
            }
        }
        // the middle's minimum, the rightmost of equals (lanes hold rising k-mers): X[a]
        InlBest M = inl_scan<false>(InlBest{mh, mj});  // (a lane without a middle k-mer: {~0, none}, never "smaller")
        M.h = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(M.h >> 32), 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)M.h, 63);
        M.id = (uint32_t)__builtin_amdgcn_readlane((int)M.id, 63);
        if (lane == 0) X[a] = M.h;
        __builtin_amdgcn_wave_barrier();
        if (p.ablate & 4u) return M.id == 12345u ? 1u : 0u;
        // ---- the windows: lane holds windows j = R lane + r
        const uint32_t R = (a + 64u) >> 6;  // ceil((a + 1) / 64) <= SEL_INL_R
        InlBest sa[SEL_INL_R], pc[SEL_INL_R];
        {   // suffix minima of X[j .. a]: a smaller hash on the left wins, an equal one does not
            InlBest run{~0ull, 0xFFFFFFFFu};
#pragma unroll
            for (int r = (int)SEL_INL_R - 1; r >= 0; --r) {
                const uint32_t j = lane * R + (uint32_t)r;
                if ((uint32_t)r < R && j <= a) {
                    const uint64_t x = X[j];
                    if (run.id == 0xFFFFFFFFu || x < run.h) run = InlBest{x, j < a ? j : M.id};
                }
                sa[r] = run;
            }
            // the lanes' minima in reverse lane order, scanned from lane 0 up = over the lanes to the right; then one lane on
            // (the lanes strictly to the right) and back into place
            const InlBest rev = inl_scan<true>(inl_lane(run, 63 - (int)lane));
            const InlBest carry = inl_lane(inl_dpp<0x138, 0xf>(rev), 63 - (int)lane);  // (wave_shr:1)
#pragma unroll
            for (uint32_t r = 0; r < SEL_INL_R; ++r)
                if (carry.id != 0xFFFFFFFFu && !(sa[r].id != 0xFFFFFFFFu && sa[r].h < carry.h)) sa[r] = carry;
        }
        {   // prefix minima of X[a + 1 .. a + j] (window j holds j k-mers of the back part): the right one of equals wins
            InlBest run{~0ull, 0xFFFFFFFFu};
            InlBest own[SEL_INL_R];
#pragma unroll
            for (uint32_t r = 0; r < SEL_INL_R; ++r) {
                const uint32_t j = lane * R + r;
                if (r < R && j >= 1u && j <= a) {
                    const uint64_t x = X[a + j];
                    if (run.id == 0xFFFFFFFFu || x <= run.h) run = InlBest{x, w + j - 1u};
                }
                own[r] = run;
            }
            const InlBest carry = inl_dpp<0x138, 0xf>(inl_scan<false>(run));  // (the lanes strictly to the left)
#pragma unroll
            for (uint32_t r = 0; r < SEL_INL_R; ++r) {
                pc[r] = own[r];
                if (carry.id != 0xFFFFFFFFu && (own[r].id == 0xFFFFFFFFu || carry.h < own[r].h)) pc[r] = carry;
            }
        }
        // ---- the windows' arg-mins, every one once, in order
        uint32_t ids[SEL_INL_R];
        uint64_t hs[SEL_INL_R];
#pragma unroll
        for (uint32_t r = 0; r < SEL_INL_R; ++r) {
            const bool back = pc[r].id != 0xFFFFFFFFu && !(sa[r].id != 0xFFFFFFFFu && sa[r].h < pc[r].h);
            ids[r] = back ? pc[r].id : sa[r].id;
            hs[r] = back ? pc[r].h : sa[r].h;
        }
        // the window in front of the lane's first one: the last window of the lane before (all of whose windows exist)
        uint32_t last = ids[0];
#pragma unroll
        for (uint32_t r = 1; r < SEL_INL_R; ++r)
            if (r < R) last = ids[r];
        uint32_t prev = (uint32_t)__shfl((int)last, (int)lane - 1);
        if (lane == 0) prev = 0xFFFFFFFFu;
        uint32_t cnt = 0;
        bool em[SEL_INL_R];
#pragma unroll
        for (uint32_t r = 0; r < SEL_INL_R; ++r) {
            const uint32_t j = lane * R + r;
            // (btllib never reports 2^64 - 1; a piece after the first: its first window belongs to the piece before it)
            // (the dropped arg-min stays `prev`: the windows behind the first that share it do not report it either)
            em[r] = r < R && j <= a && ids[r] != prev && hs[r] != ~0ull && !(j == 0u && (lo != 0u || drop0));
            if (r < R && j <= a) prev = ids[r];
            cnt += em[r] ? 1u : 0u;
        }
        const uint32_t incl = wave_inclusive_dpp(cnt);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (m + tot > cap_t) return 0xFFFFFFFFu;
        uint32_t at = m + incl - cnt;
#pragma unroll
        for (uint32_t r = 0; r < SEL_INL_R; ++r)
            if (em[r]) tmp[at++] = make_uint4((uint32_t)hs[r], (uint32_t)(hs[r] >> 32), k_from + lo + ids[r], cg);
        m += tot;
        __builtin_amdgcn_wave_barrier();  // (X is reused by the next piece)
        if (lo + a >= a_tot) break;
        lo += a;
    }
    return m;
}

}  // namespace

// One wave per slice that has stretches: the wave that meets a slice's first request takes them all, the last one first (the row's
// entries behind a stretch's place move up by the stretch's minimizers), and adds what it put in to the slice's counts.  A stretch's
// place in the row: behind the entries with a smaller (contig, k-mer) -- the row is in that order.  What this kernel does not take
// (k-mers of more than one run: invalid bases inside; more than SEL_INL_PIECES pieces; more than SEL_INL_TMP minimizers; a full
// row) goes on to k_gap_fix.
constexpr uint32_t SST_WAVES = 4;
__global__ __launch_bounds__(SST_WAVES * 64, 5) void k_sel_stretch(const SelStretchParams p)
{
    __shared__ uint4 byte_tab[256];
    __shared__ uint4 rtab[16];
    __shared__ uint64_t Xs[SST_WAVES][2 * 64 * SEL_INL_R];
    __shared__ uint4 tmps[SST_WAVES][SEL_INL_TMP];
    const uint32_t n_req = min(p.ctrl[15], p.ireq_cap);
    if (blockIdx.x * SST_WAVES >= n_req) return;
    for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) byte_tab[i] = p.byte_tab[i];
    if (threadIdx.x < 16u) rtab[threadIdx.x] = p.tab.e[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint64_t *X = Xs[wv];
    uint4 *tmp = tmps[wv];
    // The requests are handed out one by one: a wave's time goes with what its slices hold, and a launch that deals them out in
    // advance waits for the unluckiest wave.  64 ticket counters, each on a line of its own, each for every 64th request (adds to
    // one line are served one after the other, ~10 ns each: ONE counter for 5120 waves and 8000 requests took 130 us).
    // Everything a request needs is asked for in as few dependent round trips as there are: the slice's requests (whatever
    // their number: eight entries), then run table and row.
    uint32_t *const ticket = p.tickets + ((blockIdx.x * SST_WAVES + wv) & 63u) * 32u;
    for (;;) {
        uint32_t r = 0;
        if (lane == 0) r = atomicAdd(ticket, 1u);
        r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r) * 64u + ((blockIdx.x * SST_WAVES + wv) & 63u);
        if (r >= n_req) break;
        const uint4 mine = p.ireq[r + (lane & 7u)];  // (the array has eight entries to spare)
        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mine.w);
        if ((w0 >> 24) & 7u) continue;  // (not the slice's first request)
        if (p.ablate & 8u) continue;
        const uint32_t n_i = min(w0 >> 27, SEL_REQ), sl = w0 & 0xFFFFFFu;
        const bool has = lane < n_i;
        const uint64_t my_key = has ? (((uint64_t)mine.x << 32) | mine.y) + 1ull : 0ull;  // (+ 1: 0 = taken / no request)
        uint4 *row = p.cs + (size_t)sl * p.rk;
        uint32_t n_cur = p.cnt[sl];
        // lane q < n_i: the run of request q's first k-mer
        uint64_t my_b = 0;
        bool my_ok = false;
        if (has) {
            uint32_t lo = p.ctg_run0[mine.x], hi = p.ctg_run0[mine.x + 1u];
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if (p.runs[mid].kidx0 <= mine.y) lo = mid; else hi = mid;
            }
            const Run run = p.runs[lo];
            my_ok = mine.z < run.kidx0 + run.n_kmers && mine.z - mine.y + 1u >= p.w && mine.z - mine.y + 1u - p.w <= p.amax * SEL_INL_PIECES;
            my_b = run.base_off + (mine.y - run.kidx0);
        }
        // the row as the slice kernel left it (rows of up to 128 entries stay in registers: they are moved from there)
        n_cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_cur);
        uint4 e0 = make_uint4(0u, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu), e1 = e0;
        if (lane < n_cur) e0 = row[lane];
        if (lane + 64u < n_cur) e1 = row[lane + 64u];
        const bool in_regs = n_cur <= 128u && n_i == 1u;
        uint32_t added = 0;
        uint64_t left = my_key;
        for (uint32_t it = 0; it < n_i; ++it) {
            // the request with the largest key among those left
            uint64_t best = left;
#pragma unroll
            for (uint32_t d = 1; d < 8u; d <<= 1) {
                const uint64_t o = __shfl_xor(best, (int)d);
                best = o > best ? o : best;
            }
            best = __shfl(best, 0);
            const uint64_t holder = __ballot(left == best && best != 0ull);
            const int src = __builtin_ctzll(holder);
            if ((int)lane == src) left = 0ull;
            const uint32_t cg = (uint32_t)__shfl((int)mine.x, src), k_from = (uint32_t)__shfl((int)mine.y, src), k_to = (uint32_t)__shfl((int)mine.z, src);
            const uint64_t b_from = __shfl(my_b, src);
            const bool ok = __shfl((int)my_ok, src) != 0;
            const bool drop0 = k_from == 0u && p.ctg_drop && p.ctg_drop[cg];  // a piece of a record that starts with the halo of the shard before it (k_gap_fix)
            // the stretch's place: behind the row's entries in front of it (what later stretches of the slice put in lies behind it)
            const uint64_t key = ((uint64_t)cg << 32) | k_from;
            uint32_t below = ((((uint64_t)e0.w << 32) | e0.z) < key ? 1u : 0u) + ((((uint64_t)e1.w << 32) | e1.z) < key ? 1u : 0u);
            for (uint32_t e = lane + 128u; e < n_cur - added; e += 64u) {
                const uint4 q = row[e];
                below += ((((uint64_t)q.w << 32) | q.z) < key) ? 1u : 0u;
            }
            uint32_t m = ok ? stretch_sketch(p, byte_tab, rtab, X, tmp, SEL_INL_TMP, cg, k_from, k_to, b_from, drop0) : 0xFFFFFFFFu;
            if (m != 0xFFFFFFFFu && n_cur + m > p.rk) m = 0xFFFFFFFFu;
            if (m == 0xFFFFFFFFu) {
                if (lane == 0) push_gap(p.gaps, p.gap_cap, p.ctrl, p.gap_nmax, p.w, cg, k_from, k_to, sl * p.rk);
                continue;
            }
            if (m == 0 || (p.ablate & 32u)) continue;  // (32, profiling: the stretch sketched, its row left alone)
            const uint32_t at = wave_sum_u32(below);
            if (in_regs) {
                if (lane >= at && lane < n_cur) row[lane + m] = e0;
                if (lane + 64u >= at && lane + 64u < n_cur) row[lane + 64u + m] = e1;
            } else {
                for (uint32_t hi = n_cur; hi > at;) {  // from the top, 64 entries at a time: what a pass writes lies above what is still to be read
                    const uint32_t lo_ = hi > at + 64u ? hi - 64u : at;
                    const uint32_t x = lo_ + lane;
                    if (x < hi) {
                        const volatile uint64_t *sp = reinterpret_cast<const volatile uint64_t *>(row + x);
                        const uint64_t v0 = sp[0], v1 = sp[1];
                        row[x + m] = make_uint4((uint32_t)v0, (uint32_t)(v0 >> 32), (uint32_t)v1, (uint32_t)(v1 >> 32));
                    }
                    hi = lo_;
                }
            }
            for (uint32_t t = lane; t < m; t += 64u) row[at + t] = tmp[t];
            if (it + 1u < n_i) __threadfence_block();  // (a slice's next stretch: the wave's other lanes read what these wrote)
            n_cur += m;
            added += m;
        }
        if (added && lane == 0) {
            p.cnt[sl] = n_cur;
            atomicAdd(&p.sup[(sl >> SUP_SHIFT) * SUP_STRIDE], added);
        }
    }
}

int launch_sel_stretch(mxg_handle *h, const SelStretchParams &p, hipStream_t st)
{
    // (the requests are counted on the device: a grid for the i.i.d. expectation's order of magnitude, every wave walks on)
    hipLaunchKernelGGL(k_sel_stretch, dim3(1280), dim3(SST_WAVES * 64), 0, st, p);
    MXG_HIP(h, hipGetLastError());
    return MXG_OK;
}

size_t bs_select_lds(uint32_t qcap, uint32_t waves) { return (size_t)2048 * 16 + (size_t)waves * sel_wave_lds(qcap) + 16; }  // (+ the block's ticket counter)

// ---------------------------------------------------------------------------------------------------------------
// k_bs_select: geometry + launch
// ---------------------------------------------------------------------------------------------------------------
// Smallest H such that every H consecutive strips of one contig that do not hold the contig's last strip hold >= w k-mers: then
// the H strips on either side of a slice's own strips give every own candidate w k-mers of its contig -- or the contig's end.
// Only a run's last strip is shorter than S, so contigs of one run need H = ceil(w / S) and nothing else; the others are walked
// strip by strip around their run ends (the inside of a long run is skipped).
uint32_t bs_select_halo(const Assembly *a, uint32_t S, uint32_t w)
{
    const uint32_t H0 = (w + S - 1) / S;
    if (H0 > SEL_MAX_H) return 0;
    uint32_t H = H0;
    std::vector<uint32_t> lens;
    for (size_t c = 0; c + 1 < a->ctg_run0.size(); ++c) {
        const uint32_t r0 = a->ctg_run0[c], r1 = a->ctg_run0[c + 1];
        if (r1 - r0 < 2) continue;
        // the contig's strips, the inside of long runs cut down to 2 * SEL_MAX_H full strips (windows there hold H S >= w k-mers)
        lens.clear();
        for (uint32_t r = r0; r < r1; ++r) {
            const uint32_t nk = a->runs[r].n_kmers, n_full = nk / S, last = nk - n_full * S;
            for (uint32_t q = 0; q < std::min<uint32_t>(n_full, 2 * SEL_MAX_H); ++q) lens.push_back(S);
            if (last) lens.push_back(last);
        }
        if (lens.size() < 2) continue;
        lens.pop_back();  // (the contig's last strip: a window holding it reaches the contig's end)
        for (;;) {
            bool ok = true;
            uint64_t sum = 0;
            for (size_t q = 0; q < lens.size(); ++q) {
                sum += lens[q];
                if (q >= H) sum -= lens[q - H];
                if (q + 1 >= H && sum < w) {
                    ok = false;
                    break;
                }
            }
            if (ok) break;
            if (++H > SEL_MAX_H) return 0;
        }
    }
    return H;
}

uint32_t bs_select_inline_amax(uint32_t w)
{
    if (w < 16u) return 0u;
    return std::min<uint32_t>(64u * SEL_INL_R - 1u, w - 1u);
}

BsSelGeom bs_select_geom(uint32_t S, uint32_t H, uint32_t w, double frac, uint32_t n_strips, uint32_t qcap_force, uint32_t rk_force)
{
    BsSelGeom g{};
    g.ok = false;
    if (H == 0 || 2 * H >= 64 || S > 1024 || n_strips == 0) return g;
    g.H = H;
    g.T = 64 - 2 * H;
    g.n_slices = (n_strips + g.T - 1) / g.T;
    // raw candidates of a slice: what passes the ring test (the top-bits sum lets ~2 % more through than tau)
    const double raw = 64.0 * S * frac * 1.03;
    g.qcap = ((uint32_t)(raw * 1.3 + 6.0 * std::sqrt(raw) + 32.0) + 63u) / 64u * 64u;
    if (qcap_force) g.qcap = std::max<uint32_t>(64u, (qcap_force + 63u) / 64u * 64u);  // (test knob: slices that outgrow their queue)
    g.qcap = std::min<uint32_t>(g.qcap, 64u * S);
    const double sel = (double)g.T * S * 2.0 / (double)(w + 1);
    g.rk = ((uint32_t)(sel * 1.3 + 6.0 * std::sqrt(sel) + 24.0) + 31u) / 32u * 32u;
    if (rk_force) g.rk = std::max<uint32_t>(4u, rk_force);  // (test knob: slices with more selected candidates than their room -- the kernel then gives the slice up)
    // as many waves per block as fit beside the 32 KB of position tables in the CU's 160 KB of LDS (one block per CU)
    const size_t budget = 160 * 1024 - 2048 * 16 - 1024;
    g.waves = (uint32_t)std::min<size_t>(16, budget / sel_wave_lds(g.qcap));
    g.ovf_cap = 64u * S;
    g.n_ovf = 1024;
    g.lds = bs_select_lds(g.qcap, g.waves);
    g.ok = g.waves >= 4 && (uint64_t)g.n_slices * g.rk < (1ull << 31);
    return g;
}

int launch_bs_select(mxg_handle *h, const BsSelParams &p, const BsSelGeom &g, hipStream_t st)
{
    const uint32_t nwc = (p.S + 31u) / 32u + 1u;
    const uint32_t blocks = std::min<uint32_t>((g.n_slices + g.waves - 1) / g.waves, 256u);
    const dim3 grid(blocks), block(g.waves * 64u);
    // more than 64 KB of dynamic LDS must be asked for, and the attribute belongs to the (function, device) pair: once per
    // DEVICE, whichever handle or thread comes first (handles on several GPUs of one process, handles made by several threads)
    static std::mutex attr_mutex;
    constexpr int MXG_MAX_DEVICES = 64;
    static bool attr_set[MXG_MAX_DEVICES] = {};
    {
        std::lock_guard<std::mutex> lock(attr_mutex);
        const int dev = h->device;
        if (dev < 0 || dev >= MXG_MAX_DEVICES || !attr_set[dev]) {
            const void *fns[] = {reinterpret_cast<const void *>(&k_bs_select<12, 10>), reinterpret_cast<const void *>(&k_bs_select<8, 6>),
                                 reinterpret_cast<const void *>(&k_bs_select<8, 5>), reinterpret_cast<const void *>(&k_bs_select<8, 7>),
                                 reinterpret_cast<const void *>(&k_bs_select<12, 11>), reinterpret_cast<const void *>(&k_bs_select<16, 12>),
                                 reinterpret_cast<const void *>(&k_bs_select<16, 13>), reinterpret_cast<const void *>(&k_bs_select<16, 15>),
                                 reinterpret_cast<const void *>(&k_bs_select<20, 16>),
                                 reinterpret_cast<const void *>(&k_bs_select<8, 0>),
                                 reinterpret_cast<const void *>(&k_bs_select<12, 0>), reinterpret_cast<const void *>(&k_bs_select<16, 0>),
                                 reinterpret_cast<const void *>(&k_bs_select<20, 0>), reinterpret_cast<const void *>(&k_bs_select<36, 0>)};
            for (const void *f : fns) MXG_HIP(h, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (dev >= 0 && dev < MXG_MAX_DEVICES) attr_set[dev] = true;
        }
    }
    if (p.S == 320u) hipLaunchKernelGGL((k_bs_select<12, 10>), grid, block, g.lds, st, p);
    else if (p.S == 192u) hipLaunchKernelGGL((k_bs_select<8, 6>), grid, block, g.lds, st, p);
    else if (p.S == 160u) hipLaunchKernelGGL((k_bs_select<8, 5>), grid, block, g.lds, st, p);
    else if (p.S == 224u) hipLaunchKernelGGL((k_bs_select<8, 7>), grid, block, g.lds, st, p);
    else if (p.S == 352u) hipLaunchKernelGGL((k_bs_select<12, 11>), grid, block, g.lds, st, p);
    else if (p.S == 384u) hipLaunchKernelGGL((k_bs_select<16, 12>), grid, block, g.lds, st, p);
    else if (p.S == 416u) hipLaunchKernelGGL((k_bs_select<16, 13>), grid, block, g.lds, st, p);
    else if (p.S == 480u) hipLaunchKernelGGL((k_bs_select<16, 15>), grid, block, g.lds, st, p);
    else if (p.S == 512u) hipLaunchKernelGGL((k_bs_select<20, 16>), grid, block, g.lds, st, p);
    else if (nwc <= 8) hipLaunchKernelGGL((k_bs_select<8, 0>), grid, block, g.lds, st, p);
    else if (nwc <= 12) hipLaunchKernelGGL((k_bs_select<12, 0>), grid, block, g.lds, st, p);
    else if (nwc <= 16) hipLaunchKernelGGL((k_bs_select<16, 0>), grid, block, g.lds, st, p);
    else if (nwc <= 20) hipLaunchKernelGGL((k_bs_select<20, 0>), grid, block, g.lds, st, p);
    else hipLaunchKernelGGL((k_bs_select<36, 0>), grid, block, g.lds, st, p);
    MXG_HIP(h, hipGetLastError());
    return MXG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// per assembly: layout + filter
// ---------------------------------------------------------------------------------------------------------------
bool bs_possible(const mxg_handle *h, const Assembly *a)
{
    if (h->cfg.k != 32 || h->cfg.variant != MXG_VARIANT_V2_SUM || a->bs_impossible) return false;
    return a->packed_words > 0 && a->packed_words < (1ull << 36);
}

int bs_prepare(mxg_handle *h, Assembly *a)
{
    if (a->bs_ready) return MXG_OK;
    // the run table must be sorted by position (it is, for every loader; a caller's own packed layout might not be)
    for (size_t r = 1; r < a->runs.size(); ++r)
        if (a->runs[r].base_off < a->runs[r - 1].base_off + a->runs[r - 1].n_kmers) {
            a->bs_impossible = true;
            return MXG_OK;
        }
    const uint64_t n_pos = a->packed_words * 16ull + 64;  // (+ two strips: no k-mer of the assembly starts in the last one)
    const uint32_t n_chunks = (uint32_t)((n_pos + BS_CHUNK - 1) / BS_CHUNK);
    a->bs_chunks = n_chunks;
    // (+ 256 bytes behind it: the batch kernels request the words of a whole strip, up to 1024 positions + 2 words, before masking)
    MXG_HIP(h, a->d_bs_out.ensure(((size_t)n_chunks * BS_OUT_WORDS + BS_OUT_PAD) * 4 + 256));
    // padded copies of the first and of the last chunk's words (k_bs_edges fills them in front of every filter launch)
    MXG_HIP(h, a->d_bs_tail.ensure((size_t)2 * BS_EDGE_WORDS * 4));
    MXG_HIP(h, hipStreamSynchronize(h->stream));
    a->bs_ready = true;
    return MXG_OK;
}

// The filter reads whole chunks and, per lane, the two words in front of its 64: the assembly's first and last chunk are read from
// padded copies [head copy | tail copy], BS_EDGE_WORDS each -- zeros in front of the assembly, zeros (= base A) behind it.  Made
// anew for every sketch: a borrowed packed buffer (mxg_add_assembly_packed_device) may have been refilled since the last one.
__global__ __launch_bounds__(256) void k_bs_edges(const uint32_t *__restrict__ packed, uint64_t packed_words, uint32_t n_chunks,
                                                  uint32_t *__restrict__ edge)
{
    const uint64_t n_head = min(packed_words, (uint64_t)BS_CHUNK_WORDS);
    const uint64_t from = (uint64_t)(n_chunks - 1) * BS_CHUNK_WORDS - 2;  // (the tail copy: with the two words in front; n_chunks > 1)
    for (uint32_t i = threadIdx.x; i < BS_EDGE_WORDS; i += 256u) {
        uint32_t v = 0;
        if (blockIdx.x == 0) {
            if (i >= 2u && i - 2u < n_head) v = packed[i - 2u];
        } else if (n_chunks > 1 && i < BS_CHUNK_WORDS + 2u && from + i < packed_words) {
            v = packed[from + i];
        }
        edge[blockIdx.x * BS_EDGE_WORDS + i] = v;
    }
}

int bs_edges(mxg_handle *h, Assembly *a, hipStream_t st)
{
    hipLaunchKernelGGL(k_bs_edges, dim3(2), dim3(256), 0, st, a->d_packed, a->packed_words, a->bs_chunks, a->d_bs_tail.as<uint32_t>());
    MXG_HIP(h, hipGetLastError());
    return MXG_OK;
}

int bs_hash(mxg_handle *h, Assembly *a, uint32_t tau_hi, hipStream_t st)
{
    // tau = T * 2^33 with T = tau_hi / 2 on the top ring; the filter compares the top HASH_BS_PLANES bits of F + R with
    // tt = (T - 1) >> (31 - planes) (gen/bs_gen.py: reference_bits)
    const uint32_t T = tau_hi >> 1;
    const uint32_t tt = T ? (T - 1u) >> (31 - HASH_BS_PLANES) : 0u;
    // (alone on the GPU 1024 blocks over the 512 resident ones even out the tail: 478 us against 500 us at 3 Gbp; beside the other
    // stream's kernels the step is the same or better with 512.  MXG_BS_BLOCKS: sweep knob)
    const unsigned env_blocks = (unsigned)std::max<uint64_t>(1, knob_u64(h, "MXG_BS_BLOCKS", 512));
    const uint32_t blocks = std::min<uint32_t>((uint32_t)env_blocks, (a->bs_chunks + 3u) / 4u);  // two waves per SIMD are resident (see k_hash_bs)
    const uint32_t *head = a->d_bs_tail.as<uint32_t>() + 2, *tail = a->bs_chunks > 1 ? head + BS_EDGE_WORDS : head;
    hipLaunchKernelGGL(k_hash_bs, dim3(blocks), dim3(256), 0, st, a->d_packed, head, tail, a->d_bs_out.as<uint32_t>() + BS_OUT_PAD, 0u,
                       a->bs_chunks, tt, a->bs_chunks - 1u);
    MXG_HIP(h, hipGetLastError());
    return MXG_OK;
}

}  // namespace mxg
