// sketch.hip -- the minimizer-sketch stage on gfx950 (replaces `indexlr`, reference ntJoin:204-205).
//
// Semantics: for every record, over its VALID k-mers only, the rightmost
// arg-min of canonical ntHash in every window of w consecutive valid k-mers; distinct arg-mins in
// position order; printed hash = second ntHash value ext(min_hash).
//
// Stateless, data-parallel formulation (equivalent to btllib's stateful ring-buffer loop; argued in
// DESIGN.md, checked against the oracle): k-mer p (contig-local valid-k-mer index) is a minimizer iff
//     L(p) + R(p) + 1 >= w
//     L(p) = number of consecutive k-mers left of p with hash >= h(p)   (capped by p and w-1)
//     R(p) = number of consecutive k-mers right of p with hash >  h(p)  (capped by n-1-p and w-1)
// i.e. there is room for a window of w k-mers around p in which p is the rightmost minimum.
//
// Fast path (sparse candidates): only k-mers with min_hash < tau (tau = 2^64 * c / w, c = 18 expected per
// window) can be the minimum of a window that contains at least one of them, and k-mers >= tau never block
// a candidate.  So the window logic runs on ~1.8 % of the k-mers.  Windows that contain NO candidate lie
// inside a candidate-free stretch ("gap") of >= w k-mers; those stretches are detected exactly and re-done
// by the dense path (every k-mer a candidate) as stand-alone ranges; the two result sets are disjoint and
// their union is the exact sketch (low-complexity sequence simply degrades to the dense path).
//
// Kernels:
//   k_hash_sparse  one lane per strip of S consecutive k-mers of one valid run.  ntHash's split rotation never mixes
//                  the top 31 bits of a hash with its low 33, so the candidate filter rolls ONLY the two 31-bit rings
//                  (forward and reverse-complement): 3 VALU per ring per base, one ds_read_b64 of the 20-entry step
//                  table, and "is (fwd+rev) below tau" decided from the rings up to the unknown carry out of the low
//                  33 bits (a superset, exact for the min(fwd,rev) variant).  Captured k-mer indices go to the wave's
//                  own arena slice (8 B entries, no atomics).
//   k_reorder_w    arena entry -> full 64-bit hash of that k-mer (position tables: one lookup per packed byte, no
//                  rotation) -> its ordered slot (exclusive scan of per-strip counts + rank inside the strip), one wave
//                  per arena slice; entries whose exact hash turns out >= tau stay in the list and are skipped by
//                  k_resolve.  (k_reorder: the block-per-slice Horner version, kept for slices too large for the queues)
//   k_hash_dense   every k-mer is a candidate, written at its own index (DENSE_ONLY mode and gap fix-up)
//   k_resolve      one lane per candidate: nearest smaller / smaller-or-equal neighbour scan; gap detection
//   k_emit         ordered stream compaction into the sketch arrays (offsets from k_resolve's two-level counts; the
//                  dense path counts and scans with k_count_n / k_scan_sums)
//   k_merge        merge of the (small) gap sketch into the batch sketch by (record,pos)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>

#include "mxg_internal.h"
#include "nthash_dev.h"
#include "scan_kernels.h"
#include "sketch_bs.h"

namespace mxg {

enum Scratch {
    SC_CAND_H, SC_CAND_K, SC_CAND_C, SC_SEL, SC_BSUM, SC_CTRL, SC_ARENA, SC_STRIP_CNT, SC_STRIP_META,
    SC_GAPS, SC_WAVE_CNT, SC_ST_HASH, SC_ST_POS, SC_ST_REC, SC_ST_FWD, SC_G_HASH, SC_G_POS,
    SC_G_REC, SC_G_FWD, SC_V_RUNS, SC_V_STRIP0, SC_V_G0, SC_V_NK, SC_V_REC, SC_V_RUN0, SC_V_DROP, SC_CNT256, SC_WAVE_TOT,
    SC_GR_HASH, SC_GR_POS, SC_GR_REC, SC_GR_CNT, SC_GR_KEY, SC_GD_HASH, SC_GD_POS, SC_GD_REC,  // device-side stretch fix-up
    SC_CS_H, SC_CS_K, SC_CS_C,  // selected candidates per k_resolve block
    SC_GB_WORK,                 // (unused)
    SC_COUNT
};
static_assert(SC_COUNT <= 40, "scratch pool too small");

// ------------------------------------------------------------------------------------------------------
// dense hash kernel
// ------------------------------------------------------------------------------------------------------
struct DenseParams {
    const uint32_t *packed;
    const Run *runs;
    const uint32_t *run_strip0;  // [n_runs+1] exclusive prefix of strips per run
    const uint64_t *run_g0;      // [n_runs+1] exclusive prefix of k-mers per run
    uint32_t run_lo, run_hi;     // runs of this batch
    uint32_t strip_lo, strip_hi; // strips of this batch
    uint64_t g_base;             // global k-mer index of the batch's first k-mer
    uint32_t k;
    uint64_t *cand_h;            // dense arena, indexed by (global k-mer index - g_base)
    uint32_t *cand_k;
    uint32_t *cand_c;
    HashTab tab;
};

template <int S, int VARIANT>
__global__ __launch_bounds__(256) void k_hash_dense(const DenseParams p)
{
    __shared__ uint4 tab[20];
    if (threadIdx.x < 20) tab[threadIdx.x] = p.tab.e[threadIdx.x];
    __syncthreads();
    const uint32_t s = p.strip_lo + blockIdx.x * 256u + threadIdx.x;
    if (s >= p.strip_hi) return;
    const uint32_t lo = find_run(p.run_strip0, p.run_lo, p.run_hi, s);
    const Run run = p.runs[lo];
    const uint32_t j0 = (s - p.run_strip0[lo]) * (uint32_t)S;
    const uint32_t len = min((uint32_t)S, run.n_kmers - j0);
    const uint64_t b = run.base_off + j0;
    const uint64_t gi = p.run_g0[lo] + j0 - p.g_base;
    const uint32_t kidx = run.kidx0 + j0;
    const uint32_t k = p.k;

    H2 h = {0u, 0u, 0u, 0u};
    warm_up(h, p.packed, b, k, tab);
    p.cand_h[gi] = canonical<VARIANT>(h);
    p.cand_k[gi] = kidx;
    p.cand_c[gi] = run.contig;
#pragma unroll 1
    for (uint32_t blk = 0; blk < (uint32_t)S / 16; ++blk) {
        uint32_t cout = fetch16(p.packed, b + 16u * blk);
        uint32_t cin = fetch16(p.packed, b + k + 16u * blk);
#pragma unroll
        for (uint32_t u = 0; u < 16; ++u) {
            uint32_t idx = ((cout >> (2 * u)) & 3u) * 4u + ((cin >> (2 * u)) & 3u);
            nt_step(h, tab[idx]);
            uint32_t j = 1u + 16u * blk + u;
            if (j < len) {
                p.cand_h[gi + j] = canonical<VARIANT>(h);
                p.cand_k[gi + j] = kidx + j;
                p.cand_c[gi + j] = run.contig;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// sparse hash kernel
// ------------------------------------------------------------------------------------------------------
struct SparseParams {
    const uint32_t *packed;
    const Run *runs;
    const uint32_t *run_strip0;
    uint32_t run_lo, run_hi;
    uint32_t strip_lo, strip_hi;
    uint32_t k;
    uint32_t S;           // k-mers per strip (multiple of 16, <= 1024)
    uint32_t n_tiles;     // tiles of 256 strips; the (persistent) blocks of the grid share them
    uint32_t five;        // k = 32: the five-block loop (one word fetched per block); 0 = the general loop (MXG_HASH_FIVE=0)
    uint32_t tau_hi;      // candidate iff high word of min_hash < tau_hi (EVEN: tau_hi = 2T, T on the top 31 bits)
    uint2 *arena;         // wave w owns entries [w*wave_cap, (w+1)*wave_cap): {strip (rel.), j | seq<<10}
    uint32_t wave_cap;
    uint32_t *wave_cnt;   // [n_waves] ENTRIES each wave wrote to its slice (k_reorder reads that many)
    uint32_t *ctrl;       // [0] max over waves of their candidate count when it exceeds wave_cap (atomicMax): the host
                          //     then redoes the batch with that capacity
    uint32_t *strip_cnt;  // [n_strips] candidates per strip
    uint32_t *wave_tot;   // [n_waves] candidates per wave, and their super-counts (scan_kernels.h; zeroed with ctrl)
    uint32_t *wave_sup;
    uint32_t *strip_meta; // [n_strips] the strip's run (k_reorder derives contig, first k-mer index and base offset from it: 4
                          // bytes per strip written and read instead of 16 -- at 3 Gbp the strip tables were 0.6 GB of HBM
                          // traffic per step)
    const uint4 *init_tab; // byte table of init_direct (make_init_tab), 256 entries
    const uint32_t *strip_run;  // k = 32 route: the run of every strip of the assembly (k_strip_runs)
    HashTab tab;
};

// The top-31 rings.  A 64-bit ntHash value is two rings that srol/sror rotate separately: bits 0..32 and bits 33..63.
// With F, R the 31-bit top rings of the forward and reverse-complement hashes,
//     F' = rotl31(F) ^ Tf        R' = rotr31(R ^ Tr)            (Tf, Tr: top rings of the step-table entry)
// and the top 31 bits of fwd+rev are (F + R + c) mod 2^31 with c the (unknown here) carry out of the low 33 bits.
// For tau = T * 2^33:  fwd+rev < tau  =>  (F+R) mod 2^31 in {-1, 0, .., T-1}.  The kernel keeps the COMPLEMENTED rings
// (same recurrences: rotation and xor commute with complement), x = ~F in bits 30..0 and y = ~R in bits 31..1, because
// then that interval test is a single unsigned compare:  (x<<1) + y  >=  2^32 - 2T - 2.  (Bit 31 of x and bit 0 of y
// are don't-cares that the updates never propagate into the rings.)  min(fwd,rev) < tau is exactly F < T or R < T.
// (min variant: max(x<<1, y) >= 2^32 - 2T.)
// byte n (a constant after unrolling) of a register as a zero-extended value: one SDWA move
__device__ __forceinline__ uint32_t byte_of(uint32_t v, uint32_t n)
{
    uint32_t r;
    switch (n) {
    case 0: asm("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(r) : "v"(v)); break;
    case 1: asm("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(r) : "v"(v)); break;
    case 2: asm("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(r) : "v"(v)); break;
    default: asm("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(r) : "v"(v)); break;
    }
    return r;
}

// x2 = x << 1, produced by ring_double: the compiler would emit a shift, which issues at half the rate of an add
__device__ __forceinline__ uint32_t ring_double(uint32_t x)
{
    uint32_t x2;
    asm("v_add_u32_e32 %0, %1, %1" : "=v"(x2) : "v"(x));
    return x2;
}
__device__ __forceinline__ void ring_step(uint32_t &x, uint32_t x2, uint32_t &y, const uint2 t)
{
    x = (x2 | ((x >> 30) & 1u)) ^ t.x;                            // v_bfe, v_bitop3
    const uint32_t u = y ^ t.y;
    y = __builtin_amdgcn_alignbit(u >> 1, u, 1);                  // (u >> 1) | (u[1] << 31)
}

// Packed bases are read straight from HBM/L2 by each lane (one 32-bit word per 16 steps per stream, requested one
// block ahead).  Staging the wave's strips through LDS with coalesced row loads was measured and does not pay
// (profiles/r01_notes.md): the kernel is bound by VALU issue, not by memory.
//
// Capture: the 16 test results of a block are shifted into one register per lane (v_addc with the compare's carry:
// one VALU per step, no branch, no scalar mask bookkeeping, so a block is one straight-line basic block whose table
// reads the compiler can hoist).  Once per block the lanes with a non-zero history store ONE 8-byte entry
// {strip, bits | block<<16 | rank<<22} to the wave's own arena slice (slot = running count + rank among the storing
// lanes: no atomics, no LDS staging, nothing waits on the stores).  k_reorder expands the bits.
// ABL (profiling builds only): 1 = ring updates only, 2 = + test and history, 0 = full.
template <int VARIANT, int ABL = 0>
__global__ __launch_bounds__(256) void k_hash_sparse(const SparseParams p)
{
    __shared__ uint4 tab[20];    // full step table: only init_direct's k%4 remainder uses it
    __shared__ uint2 ring[16];   // top rings of the rolling entries (out<<2 | in): {Tf, Tr<<1}
    __shared__ uint4 btab[256];  // byte table of init_direct
    extern __shared__ uint4 half_tab[];  // k = 32 (p.five): 4 x 256 entries of init32_half, in the LDS that caps the residency
    const bool use_half = p.five != 0 && p.k == 32u;
    if (use_half) {
        for (uint32_t i = threadIdx.x; i < 1024u; i += 256u) {
            const uint32_t j = i >> 8, v = i & 255u;
            const uint4 f = p.init_tab[256u + (j + 4u) * 256u + v], r = p.init_tab[256u + j * 256u + v];
            half_tab[i] = make_uint4(f.x, f.y, r.z, r.w);
        }
    } else {
        btab[threadIdx.x] = p.init_tab[threadIdx.x];
    }
    if (threadIdx.x < 20) {
        const uint4 e = p.tab.e[threadIdx.x];
        tab[threadIdx.x] = e;
        if (threadIdx.x < 16) ring[threadIdx.x] = make_uint2(e.y >> 1, e.w & ~1u);
    }
    __syncthreads();
    const uint32_t S = p.S;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    __shared__ uint32_t wtot[4];
    // Block g works on the 256-strip tiles g, g + gridDim.x, ...; the grid has one block per tile.  (Round 2 tried n persistent
    // blocks per CU that share the tiles: a residency cap by grid size instead of by unused LDS.  Measured at
    // 3 Gbp: 1231 Gbp/s against 1304 -- with the CU's LDS free, the other stream's kernels move in beside the hash kernel
    // and both run slower than one after the other.)
#pragma unroll 1
    for (uint32_t vb = blockIdx.x; vb < p.n_tiles; vb += gridDim.x) {
    const uint32_t srel = vb * 256u + threadIdx.x;        // strip index relative to strip_lo
    const uint32_t s = p.strip_lo + srel;
    uint32_t len = 0;                                            // 0 for lanes beyond the batch: they stay alive
    uint64_t b = 0;                                              // (wave-wide steps) and never capture anything
    if (s < p.strip_hi) {
        const uint32_t lo = find_run(p.run_strip0, p.run_lo, p.run_hi, s);
        const Run run = p.runs[lo];
        const uint32_t j0 = (s - p.run_strip0[lo]) * S;
        len = min(S, run.n_kmers - j0);
        b = run.base_off + j0;
        p.strip_meta[srel] = lo;
    }
    const uint32_t k = p.k;
    const uint32_t thr = 0u - p.tau_hi - (VARIANT == MXG_VARIANT_V1_MIN ? 0u : 2u);
    const uint32_t wave_id = vb * 4u + wv;
    uint2 *const region = p.arena + (size_t)wave_id * p.wave_cap;  // this wave's private slice of the arena
    const uint32_t wave_cap = p.wave_cap;
    uint32_t cnt_w = 0;  // wave-uniform: entries this wave has written
    uint32_t seq = 0;    // this lane's candidates so far (rank inside its strip)
    uint32_t abl_acc = 0;  // (profiling builds only)

    uint32_t x, y;
    {
        H2 h = {0u, 0u, 0u, 0u};
        if (use_half) init32_half(h, p.packed, b, half_tab);
        else init_direct(h, p.packed, b, k, btab, tab);  // replaces k rolling warm-up steps per strip
        x = ~(h.fhi >> 1);
        y = ~h.rhi;
    }
    // byte offset of step u's ring entry = (out_u << 2 | in_u) << 3.  The two 16-base words of a block are interleaved
    // once into nibble streams (even steps, odd steps), so a step costs one shift and one mask.  The words of block
    // blk+1 are requested before block blk is hashed (software prefetch).
    const unsigned char *ringb = reinterpret_cast<const unsigned char *>(ring);
    const uint32_t *po = p.packed + (b >> 4), *pi = p.packed + ((b + k) >> 4);
    const uint32_t so = ((uint32_t)b & 15u) * 2u, si = ((uint32_t)(b + k) & 15u) * 2u;
    const uint32_t nblk = S / 16;
    // blocks that are complete in every lane of the wave (all of them but in the last wave of a run): no length mask
    uint32_t nfast = len / 16u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nfast = min(nfast, (uint32_t)__shfl_xor((int)nfast, o, 64));
    nfast = (uint32_t)__builtin_amdgcn_readfirstlane((int)nfast);
    uint32_t bits = 0;  // bit 15-u: k-mer 16*blk+u passed the ring test (never reset: the block's mask drops what is older)
    // the 16 steps of a block: cout / cin = the 16 outgoing / incoming bases
    auto steps16 = [&](const uint32_t cout, const uint32_t cin) {
        constexpr uint32_t M = 0x33333333u, B = 0x78787878u;
        const uint32_t ze = ((cout & M) << 2) | (cin & M);        // nibble v = (out, in) of step 2v
        const uint32_t zo = (cout & ~M) | ((cin >> 2) & M);       // nibble v = (out, in) of step 2v+1
        // one BYTE per step, already scaled to the entry's byte offset: a step then needs a single byte-select move
        // (v_mov_b32_sdwa, full rate) instead of a shift (half rate) and a mask.  q[2*(u&1) + ((u>>1)&1)] byte u>>2.
        const uint32_t q[4] = {(ze << 3) & B, (ze >> 1) & B, (zo << 3) & B, (zo >> 1) & B};
#pragma unroll
        for (uint32_t u = 0; u < 16; ++u) {
            // k-mer j = 16*blk + u is in (x, y): test, then roll to j+1 (the last roll of a strip is never looked at)
            const uint32_t x2 = ring_double(x);
            if (ABL != 1) {
                const uint32_t v = VARIANT == MXG_VARIANT_V1_MIN ? max(x2, y) : x2 + y;
                // bits = 2*bits + (v >= thr): the compare's carry goes straight into the add
                asm("v_cmp_le_u32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(v), "s"(thr) : "vcc");
            }
            const uint32_t off = byte_of(q[2u * (u & 1u) + ((u >> 1) & 1u)], u >> 2);
            ring_step(x, x2, y, *reinterpret_cast<const uint2 *>(ringb + off));
        }
    };
    // what a block caught: one arena entry per lane with a candidate
    auto store_entry = [&](const uint32_t blk, const uint32_t mine) {
        const uint64_t mask = __builtin_amdgcn_ballot_w64(mine != 0u);
        if (mask) {  // wave-uniform; almost always taken (64 lanes x 16 k-mers at ~1 %)
            if (mine) {
                const uint32_t slot = cnt_w + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                if (slot < wave_cap) region[slot] = make_uint2(srel, mine | (blk << 16) | (seq << 22));
                seq += (uint32_t)__popc(mine);
            }
            cnt_w += (uint32_t)__popcll(mask);
        }
    };
    auto capture_full = [&](const uint32_t blk) {
        if (ABL == 1) { abl_acc ^= x + y; return; }
        if (ABL == 2) { abl_acc += bits; return; }
        store_entry(blk, bits & 0xFFFFu);
    };
    auto capture = [&](const uint32_t blk) {
        if (ABL == 1) { abl_acc ^= x + y; return; }
        if (ABL == 2) { abl_acc += bits; return; }
        // k-mers at or beyond the strip's length (end of a run, lanes beyond the batch) do not count
        const uint32_t j0 = 16u * blk;
        const uint32_t nvalid = len > j0 ? min(len - j0, 16u) : 0u;
        store_entry(blk, bits & 0xFFFFu & ~(0xFFFFu >> nvalid));
    };
    uint32_t blk = 0;
    if (k == 32u && p.five) {
        // k = 32: the incoming bases are the outgoing bases two words later, so ONE word per block is fetched (one block
        // ahead) and five blocks are written out so that the five-word window needs no register moves
        uint32_t w0 = po[0], w1 = po[1], w2 = po[2], w3 = po[3], w4;
        const uint32_t *pw = po;  // (one pointer, constant offsets: one 64-bit add per five blocks)
#pragma unroll 1
        for (; blk + 5u <= nfast; blk += 5u, pw += 5) {
            w4 = pw[4];
            steps16(__builtin_amdgcn_alignbit(w1, w0, so), __builtin_amdgcn_alignbit(w3, w2, so));
            capture_full(blk);
            w0 = pw[5];
            steps16(__builtin_amdgcn_alignbit(w2, w1, so), __builtin_amdgcn_alignbit(w4, w3, so));
            capture_full(blk + 1u);
            w1 = pw[6];
            steps16(__builtin_amdgcn_alignbit(w3, w2, so), __builtin_amdgcn_alignbit(w0, w4, so));
            capture_full(blk + 2u);
            w2 = pw[7];
            steps16(__builtin_amdgcn_alignbit(w4, w3, so), __builtin_amdgcn_alignbit(w1, w0, so));
            capture_full(blk + 3u);
            w3 = pw[8];
            steps16(__builtin_amdgcn_alignbit(w0, w4, so), __builtin_amdgcn_alignbit(w2, w1, so));
            capture_full(blk + 4u);
        }
    }
    if (blk < nblk) {  // any k; the blocks the five-block loop left over
        uint32_t o0 = po[blk], o1 = po[blk + 1], i0 = pi[blk], i1 = pi[blk + 1];
#pragma unroll 1
        for (; blk < nblk; ++blk) {
            const uint32_t o2 = po[blk + 2], i2 = pi[blk + 2];       // next block's words; reads stay inside the padding
            const uint32_t cout = __builtin_amdgcn_alignbit(o1, o0, so);
            const uint32_t cin = __builtin_amdgcn_alignbit(i1, i0, si);
            o0 = o1; o1 = o2; i0 = i1; i1 = i2;
            steps16(cout, cin);
            capture(blk);
        }
    }
    if (ABL != 0) {
        if (abl_acc == 0x12345u) p.ctrl[3] = abl_acc;  // keep the ablated work alive
        return;
    }
    if (s < p.strip_hi) p.strip_cnt[srel] = seq;
    // candidates of the whole wave (the ordered arrays hold wave_cap per wave)
    const uint32_t tot = wave_sum_u32(seq);
    if (lane == 0) {
        p.wave_cnt[wave_id] = cnt_w;
        p.wave_tot[wave_id] = tot;  // k_reorder derives every wave's first ordered slot from these and the super-counts
        wtot[wv] = tot;
        if (tot > wave_cap) atomicMax(&p.ctrl[0], tot);  // overflow: the host redoes the batch with this capacity
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // one add per block: its four waves share a super-count (256 waves = 64 blocks)
        const uint32_t c = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        if (c) atomicAdd(&p.wave_sup[(wave_id >> SUP_SHIFT) * SUP_STRIDE], c);
    }
    __syncthreads();  // wtot is reused by the next tile
    }
}

// k = 32 route: k_hash_bs (bs_kernels.h) has written one bit per base position of the whole assembly -- bit p = "the 32-mer at
// base position p passed the ring test" -- so a batch needs no arena: k_bs_count counts the bits of every strip (the strips'
// k-mer ranges mask out positions whose k-mer would cross a run's end) and publishes the per-wave totals, k_bs_reorder_w reads
// the same words again (58 MB per 461 Mbp batch) and writes the ordered candidate arrays.  What the filter lets through beyond
// hash < tau (it compares the top 14 bits of the ring sum: ~2 % more) are entries whose exact hash turns out >= tau, which
// k_resolve treats as absent.
//
// the bits of k-mers [32 j, 32 j + 32) of a strip whose first k-mer is base position b (bw = bm + b / 32, sh = b % 32), LSB first
__device__ __forceinline__ uint32_t bs_strip_bits(const uint32_t w_lo, const uint32_t w_hi, const uint32_t sh, const uint32_t len,
                                                  const uint32_t j)
{
    const uint32_t bits = __builtin_amdgcn_alignbit(w_hi, w_lo, sh);
    const uint32_t k0 = 32u * j;
    const uint32_t nvalid = len > k0 ? min(len - k0, 32u) : 0u;
    return bits & (nvalid >= 32u ? 0xFFFFFFFFu : ((1u << nvalid) - 1u));
}

__global__ __launch_bounds__(256) void k_bs_count(const SparseParams p, const uint32_t *__restrict__ bm)
{
    const uint32_t S = p.S;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    __shared__ uint32_t wtot[4];
#pragma unroll 1
    for (uint32_t vb = blockIdx.x; vb < p.n_tiles; vb += gridDim.x) {
        const uint32_t srel = vb * 256u + threadIdx.x;
        const uint32_t s = p.strip_lo + srel;
        uint32_t len = 0;
        uint64_t b = 0;
        if (s < p.strip_hi) {
            const uint32_t lo = p.strip_run[s];
            const Run run = p.runs[lo];
            const uint32_t j0 = (s - p.run_strip0[lo]) * S;
            len = min(S, run.n_kmers - j0);
            b = run.base_off + j0;
        }
        const uint32_t wave_id = vb * 4u + wv;
        // eight words are requested at a time (one dependent load per word left the kernel waiting for memory)
        const uint32_t *bw = bm + (b >> 5);
        const uint32_t sh = (uint32_t)b & 31u;
        const uint32_t nwords = (S + 31u) / 32u;
        uint32_t w0 = bw[0], cnt = 0;
#pragma unroll 1
        for (uint32_t j0w = 0; j0w < nwords; j0w += 8u) {
            uint32_t wn[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) wn[u] = j0w + u < nwords ? bw[j0w + u + 1u] : 0u;
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                cnt += (uint32_t)__popc(bs_strip_bits(w0, wn[u], sh, len, j0w + u));
                w0 = wn[u];
            }
        }
        if (s < p.strip_hi) p.strip_cnt[srel] = cnt;
        const uint32_t tot = wave_sum_u32(cnt);
        if (lane == 0) {
            p.wave_tot[wave_id] = tot;
            wtot[wv] = tot;
            if (tot > p.wave_cap) atomicMax(&p.ctrl[0], tot);  // more than a queue / the ordered arrays hold: the host redoes the batch
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t c = wtot[0] + wtot[1] + wtot[2] + wtot[3];
            if (c) atomicAdd(&p.wave_sup[(wave_id >> SUP_SHIFT) * SUP_STRIDE], c);
        }
        __syncthreads();
    }
}

// arena entry -> full hashes -> ordered candidate slots: one block per wave slice, one thread per entry (1.15
// candidates per entry on average).  The 64-bit canonical hash of each captured k-mer comes from the direct formula
// (k/4 table lookups on the packed bases, which this block's 64 strips keep hot in L2).  An entry whose exact hash is
// >= tau (the ring test cannot see the carry out of the low 33 bits) keeps its slot; k_resolve treats it as absent.
struct ReorderParams {
    const uint2 *arena;
    const uint32_t *wave_cnt;
    uint32_t wave_cap, n_cap;
    uint32_t n_waves;            // slices; a block takes slices blockIdx.x, blockIdx.x + gridDim.x, ...
    const uint32_t *strip_cnt;   // [n_strips] candidates per strip
    uint32_t n_strips;
    const uint32_t *wave_tot;    // [n_waves] candidates per wave + super-counts (k_hash_sparse)
    const uint32_t *wave_sup;
    uint32_t *n_cand;            // ctrl[4..5]: total, written by the block of the last wave
    uint32_t queue_cap;          // candidates the LDS queue holds (= wave_cap), 0: slice too large, hash per entry
    const uint32_t *strip_meta;  // run of every strip (k_hash_sparse)
    const Run *runs;
    const uint32_t *run_strip0;
    uint32_t strip_lo, S;
    const uint32_t *packed;
    const uint4 *init_tab;
    uint32_t k;
    uint64_t *ch;
    uint32_t *ck, *cc;
    HashTab tab;
    const uint32_t *bm;          // k = 32 route (k_bs_reorder_w): the filter's bitmap instead of the arena
};

// 320 threads: a slice holds ~290 candidates (64 strips x 320 k-mers x 1.4 %), so one pass of the block hashes them
// all; with 256 threads a second, nearly empty pass doubled the block's lifetime (the kernel is latency-bound).
constexpr uint32_t RB = 320;
template <int VARIANT, int ABL = 0, uint32_t RBT = RB>
__global__ __launch_bounds__(RBT) void k_reorder(const ReorderParams p)
{
    __shared__ uint4 tab[20];
    __shared__ uint4 btab[256];     // byte table of init_direct
    __shared__ uint32_t spref[64];  // ordered slot of the first candidate of each of this wave's 64 strips
    __shared__ uint4 smeta[64];     // their {contig, first k-mer index, base offset}
    __shared__ uint32_t sh_last;    // candidates of the slice's last strip
    if (threadIdx.x < 256) btab[threadIdx.x] = p.init_tab[threadIdx.x];
    if (threadIdx.x < 20) tab[threadIdx.x] = p.tab.e[threadIdx.x];
    for (uint32_t wv = blockIdx.x; wv < p.n_waves; wv += gridDim.x) {  // (the tables above are loaded once per block)
    const uint2 *src = p.arena + (size_t)wv * p.wave_cap;
    // The kernel is a chain of dependent memory round trips, so everything that can be asked for at once is: the first
    // two entries of every thread (a slice holds about one per thread) are requested without waiting for the slice's
    // entry count (the slice is allocated in full; what lies beyond the count is masked afterwards).
    const uint32_t i0 = threadIdx.x, i1 = threadIdx.x + RBT;
    uint2 a0 = make_uint2(0u, 0u), a1 = make_uint2(0u, 0u);
    if (i0 < p.wave_cap) a0 = src[i0];
    if (i1 < p.wave_cap) a1 = src[i1];
    const uint32_t cnt = min(p.wave_cnt[wv], p.wave_cap);
    if (threadIdx.x < 64) {  // exclusive scan over the strips: the wave's first slot + prefix inside the wave
        const uint32_t s = wv * 64u + threadIdx.x;
        const bool in = s < p.n_strips;
        const uint32_t c = in ? p.strip_cnt[s] : 0u;
        if (in) {
            const uint32_t ri = p.strip_meta[s];
            const Run run = p.runs[ri];
            const uint32_t j0 = (p.strip_lo + s - p.run_strip0[ri]) * p.S;
            const uint64_t b = run.base_off + j0;
            smeta[threadIdx.x] = make_uint4(run.contig, run.kidx0 + j0, (uint32_t)b, (uint32_t)(b >> 32));
        }
        const uint32_t before = count_prefix(p.wave_tot, p.wave_sup, wv);
        const uint32_t incl = wave_inclusive_u32(c, threadIdx.x);
        spref[threadIdx.x] = before + incl - c;
        if (threadIdx.x == 63) sh_last = c;
        if (wv + 1 == p.n_waves && threadIdx.x == 63) {
            p.n_cand[0] = before + incl;
            p.n_cand[1] = 0;
        }
    }
    __syncthreads();
    // item = strip (lane of the hash kernel) | block << 6 | k-mer in the block << 12; dst = its ordered slot
    auto hash_item = [&](const uint32_t item, const uint32_t dst) {
        const uint32_t j0 = ((item >> 6) & 63u) * 16u, u = (item >> 12) & 15u;
        if (dst >= p.n_cap) return;  // beyond it only when a wave overflowed: the host redoes the batch
        const uint4 sm = smeta[item & 63u];
        H2 h = {0u, 0u, 0u, 0u};
        if (ABL == 1) {  // (profiling only) no hashing: what the rest of the kernel costs
            h.flo = sm.z + j0 + u; h.fhi = 0x00100000u; h.rlo = u; h.rhi = 0u;
        } else {
            init_direct(h, p.packed, (((uint64_t)sm.w << 32) | sm.z) + j0 + u, p.k, btab, tab);
        }
        p.ch[dst] = canonical<VARIANT>(h);
        p.ck[dst] = sm.y + j0 + u;
        p.cc[dst] = sm.x;
    };
    if (p.queue_cap) {
        // One thread per CANDIDATE: an entry holds 1.15 candidates on average but some lane of every wave holds 2 or 3,
        // so hashing per entry makes the whole wave walk the 8 table lookups 2-3 times.  The entries are expanded into
        // a queue in LDS, every candidate at its ORDERED position inside the slice (first slot of its strip + rank:
        // no scan), and the queue is hashed densely: thread q's candidate goes to slot base + q, so the stores of a
        // wave are contiguous and its reads of the packed bases are neighbours.
        extern __shared__ uint32_t queue[];
        const uint32_t base = spref[0];
        const uint32_t qn = min(spref[63] + sh_last - base, p.queue_cap);
        for (uint32_t e0 = 0; e0 < cnt; e0 += RBT) {
            const uint32_t i = e0 + threadIdx.x;
            uint2 a = e0 == 0 ? a0 : (e0 == RBT ? a1 : (i < cnt ? src[i] : make_uint2(0u, 0u)));
            if (i >= cnt) a = make_uint2(0u, 0u);
            uint32_t bits = a.y & 0xFFFFu;
            const uint32_t item0 = (a.x & 63u) | (((a.y >> 16) & 63u) << 6);
            uint32_t at = spref[a.x & 63u] - base + (a.y >> 22);
            for (; bits; ++at) {  // most significant bit = first k-mer of the block
                const uint32_t u = (uint32_t)__clz((int)bits) - 16u;
                bits &= ~(0x8000u >> u);
                if (at < qn) queue[at] = item0 + (u << 12);
            }
        }
        __syncthreads();
        for (uint32_t q = threadIdx.x; q < qn; q += RBT) hash_item(queue[q], base + q);
        __syncthreads();  // the next slice reuses spref / smeta / queue
        continue;
    }
    auto place = [&](const uint2 a) {  // (slices too large for the queue: one thread per entry)
        uint32_t bits = a.y & 0xFFFFu;
        const uint32_t item0 = (a.x & 63u) | (((a.y >> 16) & 63u) << 6);
        uint32_t dst = spref[a.x & 63u] + (a.y >> 22);
        for (; bits; ++dst) {
            const uint32_t u = (uint32_t)__clz((int)bits) - 16u;
            bits &= ~(0x8000u >> u);
            hash_item(item0 + (u << 12), dst);
        }
    };
    place(i0 < cnt ? a0 : make_uint2(0u, 0u));
    place(i1 < cnt ? a1 : make_uint2(0u, 0u));
    for (uint32_t i = threadIdx.x + 2u * RBT; i < cnt; i += RBT) place(src[i]);
    __syncthreads();
    }
}

// The same step with one WAVE per slice and position tables (init_pos): a block of RW_WAVES waves loads the 32 KB of tables
// once and its waves walk slices on their own -- no block barrier inside the loop, so the waves of a CU drift apart and
// cover one another's memory round trips.  Lane = strip while the slice is set up (its prefix and its run stay in that
// lane's registers; the candidates fetch them by lane shuffles), lane = candidate while hashing.  VALU per candidate falls
// from ~300 (Horner: two split rotations by 4 per byte) to ~100.
constexpr uint32_t RW_WAVES = 8;
template <int VARIANT>
__global__ __launch_bounds__(RW_WAVES * 64) void k_reorder_w(const ReorderParams p)
{
    extern __shared__ uint4 rw_lds[];  // [2048] position tables | RW_WAVES queues of queue_cap words
    __shared__ uint4 tab[20];
    uint4 *ptab = rw_lds;
    for (uint32_t i = threadIdx.x; i < 2048u; i += RW_WAVES * 64u) ptab[i] = p.init_tab[256u + i];
    if (threadIdx.x < 20) tab[threadIdx.x] = p.tab.e[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    uint32_t *queue = reinterpret_cast<uint32_t *>(rw_lds + 2048) + wib * p.queue_cap;
    for (uint32_t wv = blockIdx.x * RW_WAVES + wib; wv < p.n_waves; wv += gridDim.x * RW_WAVES) {
        const uint2 *src = p.arena + (size_t)wv * p.wave_cap;
        // everything that can be asked for at once is: three entries per lane (a slice holds ~180) before the count is known
        uint2 a0 = make_uint2(0u, 0u), a1 = a0, a2 = a0;
        if (lane < p.wave_cap) a0 = src[lane];
        if (lane + 64u < p.wave_cap) a1 = src[lane + 64u];
        if (lane + 128u < p.wave_cap) a2 = src[lane + 128u];
        const uint32_t cnt = min(p.wave_cnt[wv], p.wave_cap);
        const uint32_t s = wv * 64u + lane;
        const bool in = s < p.n_strips;
        const uint32_t c = in ? p.strip_cnt[s] : 0u;
        uint32_t m_c = 0, m_k = 0, m_blo = 0, m_bhi = 0;  // the strip's contig, first k-mer index, base offset
        if (in) {
            const uint32_t ri = p.strip_meta[s];
            const Run run = p.runs[ri];
            const uint32_t j0 = (p.strip_lo + s - p.run_strip0[ri]) * p.S;
            const uint64_t b = run.base_off + j0;
            m_c = run.contig; m_k = run.kidx0 + j0; m_blo = (uint32_t)b; m_bhi = (uint32_t)(b >> 32);
        }
        const uint32_t before = count_prefix(p.wave_tot, p.wave_sup, wv);
        const uint32_t incl = wave_inclusive_u32(c, lane);
        const uint32_t pref = before + incl - c;
        const uint32_t base = before, tot = (uint32_t)__shfl((int)incl, 63, 64);
        if (wv + 1 == p.n_waves && lane == 63u) {
            p.n_cand[0] = before + incl;
            p.n_cand[1] = 0;
        }
        const uint32_t qn = min(tot, p.queue_cap);
        for (uint32_t e0 = 0; e0 < cnt; e0 += 64u) {
            const uint32_t i = e0 + lane;
            uint2 a = e0 == 0 ? a0 : (e0 == 64u ? a1 : (e0 == 128u ? a2 : (i < cnt ? src[i] : make_uint2(0u, 0u))));
            if (i >= cnt) a = make_uint2(0u, 0u);
            uint32_t bits = a.y & 0xFFFFu;
            const uint32_t item0 = (a.x & 63u) | (((a.y >> 16) & 63u) << 6);
            uint32_t at = (uint32_t)__shfl((int)pref, (int)(a.x & 63u), 64) - base + (a.y >> 22);
            for (; bits; ++at) {  // most significant bit = first k-mer of the block
                const uint32_t u = (uint32_t)__clz((int)bits) - 16u;
                bits &= ~(0x8000u >> u);
                if (at < qn) queue[at] = item0 + (u << 12);
            }
        }
        __builtin_amdgcn_wave_barrier();  // (LDS operations of one wave complete in order)
        for (uint32_t q0 = 0; q0 < qn; q0 += 64u) {
            const uint32_t q = q0 + lane;
            const bool act = q < qn;
            const uint32_t item = act ? queue[q] : 0u;
            const int sl = (int)(item & 63u);
            const uint32_t sc = (uint32_t)__shfl((int)m_c, sl, 64), sk = (uint32_t)__shfl((int)m_k, sl, 64);
            const uint32_t sblo = (uint32_t)__shfl((int)m_blo, sl, 64), sbhi = (uint32_t)__shfl((int)m_bhi, sl, 64);
            const uint32_t dst = base + q;
            if (act && dst < p.n_cap) {  // beyond n_cap only when a wave overflowed: the host redoes the batch
                const uint32_t ju = ((item >> 6) & 63u) * 16u + ((item >> 12) & 15u);
                H2 h;
                init_pos(h, p.packed, (((uint64_t)sbhi << 32) | sblo) + ju, p.k, ptab, tab);
                p.ch[dst] = canonical<VARIANT>(h);
                p.ck[dst] = sk + ju;
                p.cc[dst] = sc;
            }
        }
        __builtin_amdgcn_wave_barrier();  // the next slice reuses the queue
    }
}

// k = 32 route: the same step from the filter's bitmap.  Lane = strip while the queue is filled (the strip's bits, its count from
// k_bs_count, its run), lane = candidate while hashing; queue item = strip | k-mer of the strip << 6.
template <int VARIANT>
__global__ __launch_bounds__(RW_WAVES * 64) void k_bs_reorder_w(const ReorderParams p)
{
    extern __shared__ uint4 rw_lds[];  // [2048] position tables | RW_WAVES queues of queue_cap words
    __shared__ uint4 tab[20];
    uint4 *ptab = rw_lds;
    for (uint32_t i = threadIdx.x; i < 2048u; i += RW_WAVES * 64u) ptab[i] = p.init_tab[256u + i];
    if (threadIdx.x < 20) tab[threadIdx.x] = p.tab.e[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    uint32_t *queue = reinterpret_cast<uint32_t *>(rw_lds + 2048) + wib * p.queue_cap;
    const uint32_t nwords = (p.S + 31u) / 32u;
    for (uint32_t wv = blockIdx.x * RW_WAVES + wib; wv < p.n_waves; wv += gridDim.x * RW_WAVES) {
        const uint32_t s = wv * 64u + lane;
        const bool in = s < p.n_strips;
        const uint32_t c = in ? p.strip_cnt[s] : 0u;
        uint32_t m_c = 0, m_k = 0, m_blo = 0, m_bhi = 0, len = 0;  // the strip's contig, first k-mer index, base offset, k-mers
        if (in) {
            const uint32_t ri = p.strip_meta[p.strip_lo + s];  // (here: the assembly's strip -> run table)
            const Run run = p.runs[ri];
            const uint32_t j0 = (p.strip_lo + s - p.run_strip0[ri]) * p.S;
            const uint64_t b = run.base_off + j0;
            m_c = run.contig; m_k = run.kidx0 + j0; m_blo = (uint32_t)b; m_bhi = (uint32_t)(b >> 32);
            len = min(p.S, run.n_kmers - j0);
        }
        const uint32_t before = count_prefix(p.wave_tot, p.wave_sup, wv);
        const uint32_t incl = wave_inclusive_u32(c, lane);
        const uint32_t base = before, tot = (uint32_t)__shfl((int)incl, 63, 64);
        if (wv + 1 == p.n_waves && lane == 63u) {
            p.n_cand[0] = before + incl;
            p.n_cand[1] = 0;
        }
        const uint32_t qn = min(tot, p.queue_cap);
        // every set bit -> the queue, at the strip's first slot + its rank inside the strip
        const uint32_t *bw = p.bm + ((((uint64_t)m_bhi << 32) | m_blo) >> 5);
        const uint32_t sh = m_blo & 31u;
        uint32_t at = incl - c;
        uint32_t w0 = in ? bw[0] : 0u;
#pragma unroll 1
        for (uint32_t j0w = 0; j0w < nwords; j0w += 8u) {
            uint32_t wn[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) wn[u] = in && j0w + u < nwords ? bw[j0w + u + 1u] : 0u;
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                uint32_t bits = bs_strip_bits(w0, wn[u], sh, len, j0w + u);
                w0 = wn[u];
                const uint32_t item0 = lane | ((32u * (j0w + u)) << 6);
                for (; bits; bits &= bits - 1u, ++at) {
                    const uint32_t t = (uint32_t)__builtin_ctz(bits);
                    if (at < qn) queue[at] = item0 + (t << 6);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();  // (LDS operations of one wave complete in order)
        for (uint32_t q0 = 0; q0 < qn; q0 += 64u) {
            const uint32_t q = q0 + lane;
            const bool act = q < qn;
            const uint32_t item = act ? queue[q] : 0u;
            const int sl = (int)(item & 63u);
            const uint32_t sc = (uint32_t)__shfl((int)m_c, sl, 64), sk = (uint32_t)__shfl((int)m_k, sl, 64);
            const uint32_t sblo = (uint32_t)__shfl((int)m_blo, sl, 64), sbhi = (uint32_t)__shfl((int)m_bhi, sl, 64);
            const uint32_t dst = base + q;
            if (act && dst < p.n_cap) {  // beyond n_cap only when a wave overflowed: the host redoes the batch
                const uint32_t ju = item >> 6;
                H2 h;
                init_pos(h, p.packed, (((uint64_t)sbhi << 32) | sblo) + ju, p.k, ptab, tab);
                p.ch[dst] = canonical<VARIANT>(h);
                p.ck[dst] = sk + ju;
                p.cc[dst] = sc;
            }
        }
        __builtin_amdgcn_wave_barrier();  // the next slice reuses the queue
    }
}

// ------------------------------------------------------------------------------------------------------
// resolve
// ------------------------------------------------------------------------------------------------------
struct ResolveParams {
    const uint64_t *ch;
    const uint32_t *ck, *cc;
    const uint32_t *n_ptr;  // number of candidates (device), clamped to n_cap
    uint32_t n_cap;
    uint32_t n_likely;      // hint (the count of the previous run of this batch, 0: none): blocks below it request their
                            // entries before the count has arrived
    const uint32_t *ovf;    // != 0: a wave overflowed its arena slice, candidate arrays are incomplete -> do nothing
    uint64_t tau;           // entries with hash >= tau are not candidates (ring-test false positives); dense: 2^64-1
    const uint32_t *ctg_nk;
    const uint8_t *ctg_drop;  // split load (host_io.cpp plan_pieces): contigs whose first window's minimizer is not reported
    uint32_t w;
    uint8_t *sel;
    // gap detection (sparse mode)
    uint32_t ctg_lo, ctg_hi;  // contigs of this batch
    uint4 *gaps;              // {contig, k_lo, k_hi, 0}
    uint32_t gap_cap;
    uint32_t *gap_count;
    // fused count (COUNT): minimizers per block of 256 candidates + super-counts (scan_kernels.h) for k_emit
    uint32_t *cnt256;         // [gridDim.x]
    uint32_t *sel_sup;        // zeroed with the control block
    // COUNT, optional: the selected candidates of block b laid end to end from entry b * 256 of these arrays {hash, k-mer
    // index, contig} instead of one flag per candidate.  About one candidate in ten is selected: k_emit then reads 16 bytes
    // per MINIMIZER in order, where gathering through the flags pulled a 64-byte sector for every 4-8 bytes it wanted
    // (80 MB fetched per launch at 3 Gbp to emit 7 MB).
    uint64_t *cs_h;
    uint32_t *cs_k, *cs_c;
};

// hint = index of the candidate that reports the stretch: the stretch lies right before or right after it, which tells
// k_emit's placement blocks in which k_resolve block to count the minimizers that precede the stretch
__device__ __forceinline__ void push_gap(const ResolveParams &p, uint32_t c, uint32_t lo, uint32_t hi, uint32_t hint)
{
    uint32_t idx = atomicAdd(p.gap_count, 1u);
    if (idx < p.gap_cap) p.gaps[idx] = make_uint4(c, lo, hi, hint);
}

// One lane per candidate, wave-cooperative for long scans.  sel[i] = 1 iff candidate i is a minimizer.
// Phase 1: each lane looks at its T1 nearest neighbours on its own (decides most candidates: for random
// hashes the nearest smaller element is O(1) away).  Phase 2: candidates still undecided are served one at
// a time by the whole wave, 64 neighbours per step (ballot + first set bit = nearest event), so a true
// minimizer's scan over up to w-1 k-mers costs ceil(w/64) steps instead of w dependent loads.

struct CoopCtx {
    const uint64_t *ch;
    const uint32_t *ck, *cc;
    uint32_t n, wm1;
};

// nearest event to the LEFT of candidate (bi,bh,bkx,bc) beyond distance-in-index `start`:
// returns (distance in k-mers of the nearest strictly-smaller element) or 0xFFFFFFFF if the scan ran out
__device__ __forceinline__ uint32_t coop_left(const CoopCtx &q, uint32_t lane, uint32_t bi, uint64_t bh, uint32_t bkx,
                                              uint32_t bc, uint32_t start)
{
    for (uint32_t base = start;; base += 64) {
        const uint32_t off = base + lane;
        bool inr = off <= bi;  // j = bi - off >= 0
        uint32_t dist = 0xFFFFFFFFu;
        if (inr) {
            const uint32_t j = bi - off;
            inr = (q.cc[j] & 0x7FFFFFFFu) == bc;
            if (inr) {
                dist = bkx - q.ck[j];
                inr = dist <= q.wm1;
            }
        }
        const bool blk = inr && q.ch[inr ? bi - off : 0] < bh;
        const uint64_t bm = __ballot(blk), sm = __ballot(!inr);
        const uint64_t ev = bm | sm;
        if (ev) {
            const int first = __builtin_ctzll(ev);
            if ((bm >> first) & 1ull) return (uint32_t)__builtin_amdgcn_readlane((int)dist, first);
            return 0xFFFFFFFFu;
        }
    }
}

// is there an element <= bh to the RIGHT within k-mer distance `need`, beyond distance-in-index `start`?
__device__ __forceinline__ bool coop_right_blocked(const CoopCtx &q, uint32_t lane, uint32_t bi, uint64_t bh, uint32_t bkx,
                                                   uint32_t bc, uint32_t start, uint32_t need)
{
    for (uint32_t base = start;; base += 64) {
        const uint32_t off = base + lane;
        const uint64_t j64 = (uint64_t)bi + off;
        bool inr = j64 < q.n;
        if (inr) {
            const uint32_t j = (uint32_t)j64;
            inr = (q.cc[j] & 0x7FFFFFFFu) == bc;
            if (inr) inr = (q.ck[j] - bkx) <= need;
        }
        const bool blk = inr && q.ch[inr ? (uint32_t)j64 : 0] <= bh;
        const uint64_t bm = __ballot(blk), sm = __ballot(!inr);
        const uint64_t ev = bm | sm;
        if (ev) {
            const int first = __builtin_ctzll(ev);
            return ((bm >> first) & 1ull) != 0;
        }
    }
}

constexpr int RH = 64;   // halo (candidates) staged on each side of a block's 256 candidates
constexpr int RP = 4;    // padding entries so that 4-wide neighbour groups never index outside the arrays
constexpr uint32_t RK = 256;  // candidates (= threads) per k_resolve block (measured: 128 -> 32.7 us, 256 -> 31.4, 512 -> 35.0)

template <bool GAPS, bool COUNT, int ABL = 0, int RHT = RH>
__global__ __launch_bounds__(RK) void k_resolve(const ResolveParams p)
{
    __shared__ uint64_t lh[RK + 2 * RHT + 2 * RP];
    __shared__ uint2 lkc[RK + 2 * RHT + 2 * RP];  // {k-mer index, contig}; contig = ~0 outside the candidate array
    constexpr uint32_t TOT = RK + 2 * RHT + 2 * RP, NST = (TOT + RK - 1) / RK;  // staged entries (per thread)
    const uint32_t i0 = blockIdx.x * RK;
    // Blocks that the previous run's count says will be in use request their entries before the candidate count has
    // arrived (the arrays hold n_cap entries; what lies beyond the count is masked below): one memory round trip for
    // the block instead of two.  The others (the grid covers n_cap, about twice the count) wait for it and mostly exit.
    const bool early = i0 + RK <= p.n_likely;  // (block-uniform)
    uint32_t n = 0, ovf = 0;
    if (!early) {
        n = min(*p.n_ptr, p.n_cap);
        ovf = *p.ovf;
        if (i0 >= n || ovf) return;
    }
    const uint32_t bound = early ? p.n_cap : n;
    uint64_t vh[NST];
    uint32_t vk[NST], vc[NST];
#pragma unroll
    for (uint32_t r = 0; r < NST; ++r) {
        const uint32_t e = threadIdx.x + RK * r;
        const int64_t g = (int64_t)i0 - RHT - RP + e;
        vh[r] = 0; vk[r] = 0; vc[r] = 0;
        if (e < TOT && g >= 0 && g < (int64_t)bound) {
            vh[r] = p.ch[g];
            vk[r] = p.ck[g];
            vc[r] = p.cc[g];
        }
    }
    if (early) {
        n = min(*p.n_ptr, p.n_cap);
        ovf = *p.ovf;
    }
#pragma unroll
    for (uint32_t r = 0; r < NST; ++r) {
        const uint32_t e = threadIdx.x + RK * r;
        const int64_t g = (int64_t)i0 - RHT - RP + e;
        if (e < TOT) {
            const bool in = g >= 0 && g < (int64_t)n;
            lh[e] = in ? vh[r] : 0ull;
            lkc[e] = in ? make_uint2(vk[r], vc[r] & 0x7FFFFFFFu) : make_uint2(0u, 0xFFFFFFFFu);
        }
    }
    if (i0 >= n || ovf) return;  // (block-uniform)
    __syncthreads();
    const uint32_t i = i0 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const bool live = i < n;
    const uint32_t li = RHT + RP + threadIdx.x;
    const uint64_t h = lh[li];
    const uint32_t kx = lkc[li].x;
    const uint32_t c = live ? lkc[li].y : 0u;
    const uint32_t nk = p.ctg_nk[c];
    const uint32_t w = p.w, wm1 = w - 1;
    CoopCtx q{p.ch, p.ck, p.cc, n, wm1};

    // The per-lane scans look at 4 neighbours per iteration with predicated selects: the loop control of a
    // one-neighbour-per-iteration divergent loop (exec-mask bookkeeping, ~535 SALU per wave) dominated this kernel.
    // ---- left: nearest strictly smaller (ties: rightmost wins) ----
    uint32_t L = min(kx, wm1);
    bool ldone = !live || ABL == 1;
    if (live && ABL != 1) {
        for (uint32_t t = 1; t <= (uint32_t)RHT; t += 4) {
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                const uint2 kc = lkc[li - t - u];
                const uint64_t hn = lh[li - t - u];
                const uint32_t d = kx - kc.x;
                const bool stop = (kc.y != c) || (d > wm1);
                const bool hit = !stop && (hn < h);
                L = (!ldone && hit) ? d - 1 : L;
                ldone = ldone || stop || hit;
            }
            if (ldone) break;
        }
    }
    for (uint64_t todo = __ballot(!ldone); todo; todo &= todo - 1) {
        const int src = __builtin_ctzll(todo);
        const uint32_t bi = (uint32_t)__builtin_amdgcn_readlane((int)i, src);
        const uint32_t bkx = (uint32_t)__builtin_amdgcn_readlane((int)kx, src);
        const uint32_t bc = (uint32_t)__builtin_amdgcn_readlane((int)c, src);
        const uint64_t bh = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(h >> 32), src) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)h, src);
        const uint32_t d = coop_left(q, lane, bi, bh, bkx, bc, RHT + 1);
        if ((int)lane == src && d != 0xFFFFFFFFu) L = d - 1;
    }
    // ---- right: any smaller-or-equal within the distance still needed ----
    const uint32_t R = min(nk - 1 - kx, wm1);
    bool s = live && (L + R + 1 >= w);  // enough room if nothing blocks on the right
    const uint32_t need = wm1 - min(L, wm1);  // need R >= need
    bool rdone = !(s && need > 0) || ABL == 1 || ABL == 2;
    if (!rdone) {
        for (uint32_t t = 1; t <= (uint32_t)RHT; t += 4) {
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                const uint2 kc = lkc[li + t + u];
                const uint64_t hn = lh[li + t + u];
                const uint32_t d = kc.x - kx;
                const bool stop = (kc.y != c) || (d > need);
                const bool hit = !stop && (hn <= h);
                s = (!rdone && hit) ? false : s;
                rdone = rdone || stop || hit;
            }
            if (rdone) break;
        }
    }
    for (uint64_t todo = __ballot(!rdone); todo; todo &= todo - 1) {
        const int src = __builtin_ctzll(todo);
        const uint32_t bi = (uint32_t)__builtin_amdgcn_readlane((int)i, src);
        const uint32_t bkx = (uint32_t)__builtin_amdgcn_readlane((int)kx, src);
        const uint32_t bc = (uint32_t)__builtin_amdgcn_readlane((int)c, src);
        const uint32_t bneed = (uint32_t)__builtin_amdgcn_readlane((int)need, src);
        const uint64_t bh = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(h >> 32), src) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)h, src);
        const bool blocked = coop_right_blocked(q, lane, bi, bh, bkx, bc, RHT + 1, bneed);
        if ((int)lane == src && blocked) s = false;
    }
    // A piece of a record that starts with the halo of the shard before it: the arg-min of its FIRST window (nothing
    // smaller on the left down to the contig's first k-mer, nothing smaller-or-equal up to k-mer w-1) is that shard's
    // last minimizer, not this one's.
    if (p.ctg_drop && s && kx <= wm1 && L == kx && p.ctg_drop[c]) s = false;
    // An entry >= tau is not a candidate at all (it never blocks anybody: every real candidate is < tau <= it).
    // btllib never reports min_hash == 2^64-1.
    const uint64_t tau = p.tau;
    const bool absent = h >= tau;
    const bool chosen = live && s && !absent && h != 0xFFFFFFFFFFFFFFFFull;
    if (live && !(COUNT && p.cs_h)) p.sel[i] = chosen ? 1 : 0;

    if (GAPS && live && ABL == 0) {
        // candidate-free stretches of >= w k-mers hold windows whose minimum is not a candidate.  Every real candidate
        // reports the stretch up to the next real one (and the first reports what precedes it); absent entries
        // (about one in 10^9) are stepped over.  lh[] outside the candidate array is 0, i.e. "not absent".
        if (!absent) {
            // the neighbours come from the staged copy (contig ~0: outside the array) unless absent entries intervene
            uint32_t pcg = lkc[li - 1].y;
            if (lh[li - 1] >= tau) {
                int64_t jp = (int64_t)i - 1;
                do --jp; while (jp >= 0 && p.ch[jp] >= tau);
                pcg = jp < 0 ? 0xFFFFFFFFu : (p.cc[jp] & 0x7FFFFFFFu);
            }
            const bool first = pcg != c;
            if (first) {
                uint32_t pc = pcg == 0xFFFFFFFFu ? p.ctg_lo : pcg + 1;
                for (; pc < c; ++pc) push_gap(p, pc, 0, p.ctg_nk[pc] - 1, i);  // contigs without any candidate
                if (kx >= w) push_gap(p, c, 0, kx - 1, i);
            }
            uint32_t ncg = lkc[li + 1].y, nx = lkc[li + 1].x;
            if (lh[li + 1] >= tau) {
                uint32_t jn = i + 1;
                do ++jn; while (jn < n && p.ch[jn] >= tau);
                ncg = jn < n ? (p.cc[jn] & 0x7FFFFFFFu) : 0xFFFFFFFFu;
                nx = jn < n ? p.ck[jn] : 0u;
            }
            const bool last = ncg != c;
            if (!last) {
                if (nx - kx - 1 >= w) push_gap(p, c, kx + 1, nx - 1, i);
            } else {
                if (nk - 1 - kx >= w) push_gap(p, c, kx + 1, nk - 1, i);
                if (ncg == 0xFFFFFFFFu)
                    for (uint32_t nc = c + 1; nc < p.ctg_hi; ++nc) push_gap(p, nc, 0, p.ctg_nk[nc] - 1, i);
            }
        } else if (i == 0) {  // nobody else speaks for a batch whose every entry is absent
            uint32_t jn = 1;
            while (jn < n && p.ch[jn] >= tau) ++jn;
            if (jn >= n)
                for (uint32_t pc = p.ctg_lo; pc < p.ctg_hi; ++pc) push_gap(p, pc, 0, p.ctg_nk[pc] - 1, i);
        }
    }
    if (COUNT) {
        if (p.cs_h) {
            __shared__ uint32_t wsel[RK / 64];
            const uint64_t bm = __ballot(chosen);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
            if (lane == 0) wsel[threadIdx.x >> 6] = (uint32_t)__popcll(bm);
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (uint32_t u = 0; u < RK / 64; ++u) {
                before += u < (threadIdx.x >> 6) ? wsel[u] : 0u;
                total += wsel[u];
            }
            if (chosen) {
                const uint32_t dst = blockIdx.x * RK + before + rank;
                p.cs_h[dst] = h;
                p.cs_k[dst] = kx;
                p.cs_c[dst] = c;
            }
            if (threadIdx.x == 0) count_publish(p.cnt256, p.sel_sup, blockIdx.x, total);
        } else {
            const uint32_t cn = (uint32_t)__syncthreads_count(chosen ? 1 : 0);
            if (threadIdx.x == 0) count_publish(p.cnt256, p.sel_sup, blockIdx.x, cn);
        }
    }
}

// k_count with the element count read from device memory
__global__ __launch_bounds__(256) void k_count_n(const uint8_t *__restrict__ sel, const uint32_t *__restrict__ n_ptr,
                                                 uint32_t n_cap, uint32_t *__restrict__ bsum)
{
    __shared__ uint32_t sh[256];
    const uint32_t n = min(*n_ptr, n_cap);
    uint32_t base = blockIdx.x * TILE + threadIdx.x * TILE_PER_THREAD;
    if (blockIdx.x * TILE >= n) {  // whole tile beyond the candidates
        if (threadIdx.x == 0) bsum[blockIdx.x] = 0;
        return;
    }
    uint32_t c = count_flags4(load_flags4(sel, base, n));
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = sh[0];
}

constexpr uint32_t GAP_DEV_MAX = 4096;  // stretches per batch the device route holds at least (defined here: k_emit places them);
// a batch's real capacity (`gcap` below) grows with its assembly up to GAP_DEV_CAP_MAX: repeat-rich genomes hold thirty times the
// stretches of i.i.d. sequence, and a first sketch -- no density is known yet -- must not run out of room and go round again
constexpr uint32_t GAP_DEV_CAP_MAX = 65536;
// Their minimizers wait in one pool per batch, each stretch's in a region of the size it needs (k_gap_fix reserves it with one
// add to ctrl[14]; r_start[stretch]).  Until round 4 every stretch had a region of 64 entries: a di- or trinucleotide run longer
// than w reports every second or third k-mer, and each such stretch went to the host and through the dense kernels.  A pool that
// runs out (a batch whose stretches hold more than four million minimizers) hands the stretch to the host like any other it
// cannot keep.
constexpr uint32_t GAP_DEV_POOL = 4u << 20;
constexpr uint32_t GAP_DEFER_MAX = 1024;  // stretches per batch that can be handed to the host (a batch's slice of a pinned array)
constexpr uint32_t EMIT_COMPACT_BLOCKS = 16;  // k_resolve blocks per k_emit tile on the sparse path (a power of two)
struct EmitParams {
    uint32_t ecb;        // compact mode: blocks (slices) of selected candidates per tile, a power of two <= 64
    const uint8_t *sel;
    const uint64_t *ch;
    const uint32_t *ck, *cc;
    const uint32_t *n_ptr;
    uint32_t n_cap;
    const uint32_t *ovf;   // see ResolveParams
    const uint32_t *bsum;  // dense path: exclusive offsets per 1024-tile (k_count_n + k_scan_sums); sparse path: nullptr,
    const uint32_t *cnt256, *sel_sup;  // ... offsets come from k_resolve's two-level counts
    uint32_t *n_sel;       // sparse path: ctrl[2..3], written by the tile that holds the last candidate
    uint32_t *n_out;       // sparse path, optional: a second device word that receives the total (fused sketch+graph call)
    uint32_t *host_ctrl;   // sparse path: pinned host copy of the control block (8 words), written by that tile too:
                           // no copy on the stream, the host reads it after the stream has drained
    const Run *runs;
    const uint32_t *ctg_run0, *ctg_rec;
    const uint4 *ctg_info; // {first run, runs, first run's pos0, record} per contig, or null (virtual contigs): ctg_run0 -> runs -> ctg_rec
                           // are three dependent round trips per minimizer; a contig of one run (no N inside) needs only this entry
    uint64_t mult;         // 1 ^ (k * MULTISEED)
    uint64_t out_base;     // where this batch starts in the output arrays
    uint64_t out_limit;    // capacity of the output arrays (entries at or beyond it are dropped: speculative emit)
    uint64_t *o_hash;
    uint32_t *o_pos, *o_rec;
    uint8_t *o_fwd;
    // batches of one assembly enqueued without a host sync in between: where this batch starts is the sum of the batches
    // before it, which is still on the device (null: 0); the tile holding the last candidate passes the sum on
    const uint64_t *base_in;
    uint64_t *base_out;
    // device-side fix-up of candidate-free stretches (k_gap_fix, k_gap_post run BEFORE this kernel): the batch's own
    // minimizers are written to their final places at once -- minimizer t of the batch goes to base + t + (minimizers in
    // the stretches before it; s_key / s_off of k_gap_post, searched once per tile unless a stretch falls inside it) -- and
    // GAP_DEV_MAX / 4 extra blocks at the end of the grid, one wave per stretch, put the stretches' minimizers between them:
    // stretch r goes to base + (own minimizers before it) + s_off[r].  No staging copy, no merge pass.
    uint32_t dev_gaps;
    uint32_t gcap, n_place; // stretches the batch's arrays hold; blocks at the front of the grid that place them (four each)
    uint32_t n_tiles;      // tiles of the candidate array = blocks that emit; blocks beyond them place stretches (dev_gaps)
    const uint64_t *s_key; const uint32_t *s_off, *s_src;
    const uint4 *gaps;
    const uint64_t *r_hash; const uint32_t *r_pos, *r_rec, *r_cnt, *r_start;
    // the selected candidates laid out per k_resolve block (ResolveParams::cs_*): replaces sel / ch / ck / cc
    const uint64_t *cs_h;
    const uint32_t *cs_k, *cs_c;
    const uint4 *cs_aos;   // ... or (behind k_bs_select) as ONE array of {hash lo, hash hi, k-mer index, contig}: cs_h is then only a flag
    // entries per producer block in the cs_* arrays (RK behind k_resolve; a slice of k_bs_select has its own) and, behind
    // k_bs_select, the number of entries to walk (slices x rk: the candidate count in *n_ptr is then only reported)
    uint32_t rk, n_fixed;
    const uint32_t *cand_spread;  // k_bs_select's candidate counters (64, 32 words apart) or null: then *n_ptr is the count
};

// number of keys < key in the sorted array keys[0..n)
__device__ __forceinline__ uint32_t lower_bound_u64(const uint64_t *keys, uint32_t n, uint64_t key)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_emit(const EmitParams p)
{
    __shared__ uint32_t sh[256];
    const uint32_t RKe = p.rk;  // entries per producer block
    const uint32_t n = p.n_fixed ? p.n_fixed : min(*p.n_ptr, p.n_cap);
    const uint32_t n_g_raw = p.ovf[1];
    if (*p.ovf || n == 0) {  // arena overflow (the host redoes the batch) or no candidate at all: only report
        if (p.host_ctrl && blockIdx.x == 0 && threadIdx.x < 16)
            p.host_ctrl[threadIdx.x] = threadIdx.x == 0 ? *p.ovf : (threadIdx.x == 1 ? n_g_raw : 0u);
        if (p.base_out && blockIdx.x == 0 && threadIdx.x == 0) *p.base_out = p.base_in ? *p.base_in : 0ull;
        return;
    }
    // stretches sketched on the device: count, their minimizers, "could not be finished here" (the host then redoes the batch)
    // (k_bs_select raises it too: slices beyond their queues / regions, more selected candidates than a slice's room)
    // (... and more stretches than this launch has blocks to place them: the grid follows what earlier sketches of the assembly met)
    const uint32_t flag = ((p.dev_gaps || p.n_fixed) ? p.ovf[6] : 0u) | ((p.dev_gaps && n_g_raw > 4u * p.n_place && n_g_raw <= p.gcap) ? 1u : 0u);
    const uint32_t n_g = p.dev_gaps && !flag && n_g_raw <= p.gcap ? n_g_raw : 0u;
    const uint32_t nB = n_g ? p.ovf[7] : 0u;
    const uint64_t obase = p.out_base + (p.base_in ? *p.base_in : 0ull), limit = p.out_limit;
    const uint32_t n_place = p.dev_gaps ? p.n_place : 0u;  // the grid's FIRST blocks: they start at once
    if (blockIdx.x < n_place) {  // placement of the stretches' minimizers: one wave per stretch
        const uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
        if (r >= n_g) return;
        const uint32_t g = p.s_src[r];
        const uint64_t key = p.s_key[r];
        const uint32_t nblk = (n + RKe - 1u) / RKe;
        const uint32_t b = min((p.gaps[g].w & ~SEL_GAP_DROP) / RKe, nblk - 1u);  // the block of the candidate that reported the stretch
        const uint32_t cb = p.cnt256[b];
        uint32_t below = 0;  // minimizers of block b in front of the stretch
        for (uint32_t e = lane; e < cb; e += 64u) {
            const uint32_t src = b * RKe + e;
            uint32_t ec, ek;
            if (p.cs_aos) {
                const uint4 q = p.cs_aos[src];
                ek = q.z; ec = q.w;
            } else {
                ec = p.cs_c[src]; ek = p.cs_k[src];
            }
            below += ((((uint64_t)ec << 32) | ek) < key) ? 1u : 0u;
        }
        const uint64_t o0 = obase + count_prefix(p.cnt256, p.sel_sup, b) + wave_sum_u32(below) + p.s_off[r];
        const uint32_t rc = p.r_cnt[g];
        for (uint32_t e = lane; e < rc && o0 + e < limit; e += 64u) {
            const size_t at = (size_t)p.r_start[g] + e;
            p.o_hash[o0 + e] = p.r_hash[at];
            p.o_pos[o0 + e] = p.r_pos[at];
            p.o_rec[o0 + e] = p.r_rec[at];
        }
        return;
    }
    // compact mode (the sparse path: k_resolve laid the selected candidates out per block): a tile = ECB blocks of k_resolve,
    // ~400 minimizers; flag mode (dense path): a tile = TILE candidates
    const uint32_t ECB = p.ecb;  // (16 behind k_resolve's blocks of 256 candidates; the slices of k_bs_select hold ~18 each: 32 or 64)
    const uint32_t tile = blockIdx.x - n_place;
    const uint32_t span = p.cs_h ? ECB * RKe : (uint32_t)TILE;
    if ((uint64_t)tile * span >= n) return;  // whole tile beyond the candidates
    // (the stretch keys this block will search in are requested before anything else: they go to LDS further down, and their round
    // trip passes under those of the block's offset)
    constexpr uint32_t EK = 2048;
    uint64_t kreg[EK / 256];
    const bool keys_in_lds = n_g && n_g <= EK;
    if (keys_in_lds) {
#pragma unroll
        for (uint32_t u = 0; u < EK / 256; ++u) {
            const uint32_t q = threadIdx.x + u * 256u;
            kreg[u] = q < n_g ? p.s_key[q] : 0ull;
        }
    }
    uint32_t base = tile * TILE + threadIdx.x * TILE_PER_THREAD;
    const uint32_t fl = p.cs_h ? 0u : load_flags4(p.sel, base, n);
    uint32_t c = count_flags4(fl);
    uint32_t before;
    __shared__ uint32_t spre[64 + 1];  // compact mode: the selected counts of the tile's blocks as a prefix
    if (p.bsum) {
        before = p.bsum[tile];
    } else {
        __shared__ uint32_t sh_before;
        if (threadIdx.x < 64) {
            const uint32_t bpt = span / RKe;  // k_resolve blocks per tile
            const uint32_t bef = count_prefix(p.cnt256, p.sel_sup, tile * bpt);
            if (threadIdx.x == 0) sh_before = bef;
            if (p.cs_h) {
                const uint32_t nblk = (n + RKe - 1u) / RKe, b = tile * ECB + threadIdx.x;
                const uint32_t cb = threadIdx.x < ECB && b < nblk ? p.cnt256[b] : 0u;
                const uint32_t incl = wave_inclusive_u32(cb, threadIdx.x);
                if (threadIdx.x < ECB) spre[threadIdx.x + 1] = incl;
                if (threadIdx.x == 0) spre[0] = 0;
            }
            if ((n - 1) / span == tile) {  // the tile holding the last candidate also reports the totals
                const uint32_t all = count_prefix(p.cnt256, p.sel_sup, (n + RKe - 1u) / RKe);
                const uint32_t n_report = p.cand_spread ? wave_sum_u32(p.cand_spread[threadIdx.x * 32u]) : *p.n_ptr;
                const uint64_t total = (uint64_t)all + nB;
                if (threadIdx.x == 0) {
                    p.n_sel[0] = all;
                    p.n_sel[1] = 0;
                    if (p.cand_spread) {  // (ctrl[4..5]: the candidate count, where the other route's reorder kernel leaves it)
                        p.n_sel[2] = n_report;
                        p.n_sel[3] = 0;
                    }
                    if (p.n_out) *p.n_out = (uint32_t)(obase + total);
                    if (p.base_out) *p.base_out = obase + total;
                }
                if (p.host_ctrl && threadIdx.x < 16) {  // layout: see HostCtrl
                    const uint32_t w = threadIdx.x;
                    p.host_ctrl[w] = w == 1 ? n_g_raw : w == 2 ? all : w == 3 ? flag : w == 4 ? n_report : w == 5 ? nB
                                   : w == 6 ? (uint32_t)total : w == 7 ? (uint32_t)(total >> 32)
                                   : w == 8 ? (uint32_t)obase : w == 9 ? (uint32_t)(obase >> 32)
                                   : (w == 10 && p.dev_gaps) ? p.ovf[10] : (w == 11 && p.dev_gaps && !flag) ? p.ovf[11] : w == 12 ? p.ovf[13]
                                   : (w == 15 && p.n_fixed) ? p.ovf[15] : 0u;  // ([15]: the stretches k_sel_stretch was asked for)
                }
            }
        }
        __syncthreads();
        before = sh_before;
    }
    // About one candidate in ten is selected: the selected ones' indices are first laid end to end in LDS, then consecutive
    // threads turn them into minimizers -- full waves of gathers (candidate arrays, run table) and contiguous stores instead
    // of a few lanes per wave.
    __shared__ uint16_t picked[TILE];
    uint32_t tile_total;
    if (p.cs_h) {
        tile_total = spre[ECB];
    } else {
        uint32_t l = block_exclusive_256(c, sh);
        tile_total = sh[255];
        for (int u = 0; u < TILE_PER_THREAD; ++u)
            if ((fl >> (8 * u)) & 1u) picked[l++] = (uint16_t)(threadIdx.x * TILE_PER_THREAD + u);
        __syncthreads();
    }
    auto item = [&](uint32_t r, uint64_t &hsh, uint32_t &kx, uint32_t &ctg) {  // minimizer r of the tile
        if (p.cs_h) {
            uint32_t u = 0;  // the block holding it: last u with spre[u] <= r
            for (uint32_t st = ECB / 2; st > 0; st >>= 1)
                if (spre[u + st] <= r) u += st;
            const uint32_t src = (tile * ECB + u) * RKe + (r - spre[u]);
            if (p.cs_aos) {
                const uint4 q = p.cs_aos[src];
                hsh = ((uint64_t)q.y << 32) | q.x;
                kx = q.z;
                ctg = q.w;
            } else {
                hsh = p.cs_h[src];
                kx = p.cs_k[src];
                ctg = p.cs_c[src];
            }
        } else {
            const uint32_t i = tile * TILE + picked[r];
            hsh = p.ch[i];
            kx = p.ck[i];
            ctg = p.cc[i] & 0x7FFFFFFFu;
        }
    };
    // A few hundred stretches among a million minimizers: almost every tile lies between two neighbouring stretches, so the
    // tile's first and last key are searched once and only a tile that straddles a stretch searches per minimizer.
    // (up to EK stretch keys are copied to LDS in one round trip: the searches then do not walk through L2)
    __shared__ uint64_t skeys[EK];
    __shared__ uint32_t lb_edge[2];
    const uint64_t *keys = p.s_key;
    if (keys_in_lds) {
#pragma unroll
        for (uint32_t u = 0; u < EK / 256; ++u) {
            const uint32_t q = threadIdx.x + u * 256u;
            if (q < n_g) skeys[q] = kreg[u];
        }
        keys = skeys;
        __syncthreads();
    }
    if (n_g && tile_total && (threadIdx.x == 0 || threadIdx.x == 64)) {
        uint64_t hsh; uint32_t kx, ctg;
        item(threadIdx.x ? tile_total - 1u : 0u, hsh, kx, ctg);
        lb_edge[threadIdx.x ? 1 : 0] = lower_bound_u64(keys, n_g, ((uint64_t)ctg << 32) | kx);
    }
    if (n_g) __syncthreads();
    const uint64_t o0 = obase + before;
    for (uint32_t r = threadIdx.x; r < tile_total; r += 256u) {
        uint32_t ctg, kx;
        uint64_t hsh;
        item(r, hsh, kx, ctg);
        uint64_t o = o0 + r;
        if (n_g) {
            uint32_t lb = lb_edge[0];
            if (lb_edge[1] != lb) lb = lower_bound_u64(keys, n_g, ((uint64_t)ctg << 32) | kx);
            o += p.s_off[lb];  // (s_off[n_g] = all of them)
        }
        if (o >= limit) continue;  // (speculative emit into arrays sized by an estimate)
        // contig-local valid-k-mer index -> base position, through the contig's run table
        uint32_t lo, hi, pos0, kidx0 = 0u, rec;
        if (p.ctg_info) {
            const uint4 ci = p.ctg_info[ctg];
            lo = ci.x; hi = ci.x + ci.y; pos0 = ci.z; rec = ci.w;
        } else {
            lo = p.ctg_run0[ctg]; hi = p.ctg_run0[ctg + 1];
            rec = p.ctg_rec[ctg];
            pos0 = 0u;
        }
        if (hi - lo > 1 || !p.ctg_info) {
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (p.runs[mid].kidx0 <= kx) lo = mid; else hi = mid;
            }
            pos0 = p.runs[lo].pos0;
            kidx0 = p.runs[lo].kidx0;
        }
        p.o_hash[o] = ext_hash(hsh, p.mult);  // (the strand byte is filled lazily by k_strand, only when somebody asks for it)
        p.o_pos[o] = pos0 + (kx - kidx0);
        p.o_rec[o] = rec;
    }
}

// forward[i] = (forward hash <= reverse-complement hash) of minimizer i's k-mer, recomputed from the bases with the
// k-step direct formula.  Only `--strand` output and host/device sketch views need it, so it runs on demand.
__global__ __launch_bounds__(256) void k_strand(const uint32_t *__restrict__ packed, const uint64_t *__restrict__ rec_base,
                                                const uint32_t *__restrict__ pos, const uint32_t *__restrict__ rec,
                                                uint64_t n, uint32_t k, const HashTab t, uint8_t *__restrict__ fwd)
{
    __shared__ uint4 tab[20];
    if (threadIdx.x < 20) tab[threadIdx.x] = t.e[threadIdx.x];
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    H2 h = {0u, 0u, 0u, 0u};
    warm_up(h, packed, rec_base[rec[i]] + pos[i], k, tab);
    fwd[i] = is_forward(h) ? 1 : 0;
}

// merge two (record,pos)-sorted sketches A (nA) and B (nB) with disjoint keys into O
struct MergeParams {
    const uint64_t *a_hash; const uint32_t *a_pos, *a_rec; const uint8_t *a_fwd; uint32_t nA;
    const uint64_t *b_hash; const uint32_t *b_pos, *b_rec; const uint8_t *b_fwd; uint32_t nB;
    uint64_t *o_hash; uint32_t *o_pos, *o_rec; uint8_t *o_fwd;
};

__device__ __forceinline__ uint32_t lower_bound_key(const uint32_t *rec, const uint32_t *pos, uint32_t n, uint64_t key)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        uint64_t km = ((uint64_t)rec[mid] << 32) | pos[mid];
        if (km < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_merge(const MergeParams p)
{
    uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < p.nA) {
        uint64_t key = ((uint64_t)p.a_rec[t] << 32) | p.a_pos[t];
        uint32_t d = t + lower_bound_key(p.b_rec, p.b_pos, p.nB, key);
        p.o_hash[d] = p.a_hash[t]; p.o_pos[d] = p.a_pos[t]; p.o_rec[d] = p.a_rec[t]; p.o_fwd[d] = p.a_fwd[t];
    } else if (t < p.nA + p.nB) {
        uint32_t u = t - p.nA;
        uint64_t key = ((uint64_t)p.b_rec[u] << 32) | p.b_pos[u];
        uint32_t d = u + lower_bound_key(p.a_rec, p.a_pos, p.nA, key);
        p.o_hash[d] = p.b_hash[u]; p.o_pos[d] = p.b_pos[u]; p.o_rec[d] = p.b_rec[u]; p.o_fwd[d] = p.b_fwd[u];
    }
}


// ------------------------------------------------------------------------------------------------------
// candidate-free stretches fixed up on the device
// ------------------------------------------------------------------------------------------------------
// A window without any candidate lies inside a stretch of >= w k-mers whose hashes are all >= tau (k_resolve finds every
// such stretch exactly).  The minimizers of the windows INSIDE a stretch are the stretch's own sketch as a stand-alone
// contig.  The host-driven route (process_gaps: virtual contigs through the dense pipeline) costs several host round trips
// per batch; at genome scale almost every batch of 10^9 k-mers holds a few stretches (about 0.3 per 10^9 k-mers at 18
// candidates per window, hundreds at 12), so here they are sketched by one block each without leaving the stream:
//   k_gap_fix    block = stretch: exact hashes of its k-mers into LDS, rightmost arg-min of every window through a sparse
//                table (log2 w doubling passes), distinct arg-mins compacted in order into the stretch's region
//   k_gap_post   one block: ranks the stretches by (contig, first k-mer); sorted keys + minimizers before each stretch
//   k_emit       (runs after them) writes the batch's own minimizers straight to their final places, shifted by the
//                stretch minimizers before each; a few extra blocks put the stretches' minimizers in between; reports
// Anything this route cannot hold -- more than GAP_DEV_MAX stretches, a stretch longer than GAP_DEV_NMAX k-mers or with more
// invalid bases inside than the block's words take, minimizers beyond the batch's pool -- is left out here: more than
// GAP_DEV_MAX stretches raise ctrl[6] and the host redoes the batch; single stretches are handed to the host (defer_stretch).
constexpr uint32_t GAP_DEV_NMAX = 4096;

struct GapFixParams {
    const uint4 *gaps;   // {contig, k_lo, k_hi, reporting candidate} in arrival order (k_resolve)
    uint32_t *ctrl;      // [1] stretches, [6] "host must redo", [10] k-mers hashed here
    const Run *runs;
    const uint32_t *ctg_run0, *ctg_rec;
    const uint8_t *ctg_drop;
    const uint32_t *packed;
    const uint4 *init_tab;
    uint32_t k, w;
    uint64_t mult;
    uint64_t *r_hash;    // [GAP_DEV_POOL]
    uint32_t *r_pos, *r_rec;
    uint32_t *r_cnt;     // [GAP_DEV_MAX]
    uint32_t *r_start;   // [GAP_DEV_MAX] the stretch's region of the pool
    uint32_t pool;       // entries of the pool (GAP_DEV_POOL; MXG_GAP_POOL: test knob)
    uint32_t gcap;       // stretches the per-stretch arrays hold
    uint64_t *r_key;     // [GAP_DEV_MAX] contig << 32 | k_lo
    uint4 *defer;        // [GAP_DEFER_MAX] (pinned host memory) stretches left to the host: ctrl[11] of them
    HashTab tab;
};

// A stretch this route cannot hold (longer than GAP_DEV_NMAX k-mers: satellite arrays, long low-complexity runs; too many
// invalid bases inside; the pool used up) contributes nothing here; the host sketches it afterwards through the dense pipeline
// and merges its minimizers into the assembly's sketch (Driver::merge_deferred).  One thread of the block calls this.
__device__ __forceinline__ void defer_stretch(const GapFixParams &p, const uint4 gp)
{
    const uint32_t at = atomicAdd(&p.ctrl[11], 1u);
    if (at < GAP_DEFER_MAX) p.defer[at] = gp;
    else p.ctrl[6] = 1;
}

// One stretch by one block of 256 threads.  The work arrays are the caller's: LDS for the common stretches (k_gap_fix, at
// most GAP_DEV_NSMALL k-mers: 23 KB per block, which fits beside the other stream's hash kernel -- blocks of 56 KB, nearly
// all with nothing to do, waited 100-160 us for a CU with room), global scratch for the rare long ones (k_gap_post's one
// block walks them before it ranks the stretches).  All early exits are block-uniform.
constexpr uint32_t GAP_FIX_BLOCKS = 768;   // blocks of k_gap_fix (they walk the batch's stretches): three of 51 KB per CU (tools/sweep_gap_blocks.sh)
constexpr uint32_t GAP_DEV_NSMALL = 4096;  // (= GAP_DEV_NMAX since round 3: the filter kernel of the k = 32 route uses no LDS, so a
                                           // 51 KB block finds room; the one-block walk of the longer stretches in k_gap_post took 45 us
                                           // in three batches of thirteen at configs[2]; 1792 and 23 KB before)
template <int VARIANT>
__device__ __forceinline__ void gap_fix_one(const GapFixParams &p, const uint32_t j, const uint32_t nmax, uint64_t *lh,
                                            uint16_t *lidx0, uint16_t *lidx1, uint32_t *selbits, uint32_t *lw, uint32_t *sh,
                                            const uint4 *tab, uint32_t *drop_idx)
{
    auto lidxc = [&](uint32_t which) { return which ? lidx1 : lidx0; };
    const uint4 gp = p.gaps[j];
    const uint32_t c = gp.x, klo = gp.y, khi = gp.z, n = khi - klo + 1u, w = p.w, k = p.k;
    if (threadIdx.x == 0) {
        p.r_key[j] = ((uint64_t)c << 32) | klo;
        p.r_cnt[j] = 0;
    }
    // the run holding k_lo; a stretch that is not inside one run (invalid bases in it) goes to the host
    uint32_t lo = p.ctg_run0[c], hi = p.ctg_run0[c + 1];
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (p.runs[mid].kidx0 <= klo) lo = mid; else hi = mid;
    }
    // (a stretch may run across invalid bases: its k-mers are the valid ones of several runs, its bases one range of the packed
    // array with the invalid ones in it -- taken as long as the range fits the block's words)
    const uint32_t r_end = p.ctg_run0[c + 1];
    uint32_t rl = lo;  // the run holding k_hi
    while (rl + 1 < r_end && khi >= p.runs[rl].kidx0 + p.runs[rl].n_kmers) ++rl;
    const Run run = p.runs[lo], runl = p.runs[rl];
    const uint64_t b_glob = run.base_off + (klo - run.kidx0);
    const uint64_t bspan = runl.base_off + (khi - runl.kidx0) + k - b_glob;  // bases from the first k-mer's first to the last one's last
    if (n > nmax || n < w || khi >= runl.kidx0 + runl.n_kmers || bspan > (uint64_t)nmax + 1024u) {
        if (threadIdx.x == 0) defer_stretch(p, gp);
        return;
    }
    for (uint32_t i = threadIdx.x; i < nmax / 32; i += 256) selbits[i] = 0;
    if (threadIdx.x == 0) *drop_idx = 0xFFFFFFFFu;
    __syncthreads();
    {
        const uint32_t n_words = (uint32_t)(((b_glob & 15u) + bspan + 15u) / 16u) + 1u;  // <= the size of lw: bspan <= NMAX + 1024
        for (uint32_t q = threadIdx.x; q < n_words; q += 256) lw[q] = p.packed[(b_glob >> 4) + q];
    }
    __syncthreads();
    const uint64_t b = b_glob & 15u;  // base index inside lw
    const uint32_t per = (n + 255u) / 256u, i0 = threadIdx.x * per, i1 = min(i0 + per, n);
    // the run of a k-mer of the stretch, walking on from the run the caller is in
    auto seek = [&](uint32_t &rr, Run &cr, uint32_t kk) {
        while (kk >= cr.kidx0 + cr.n_kmers) cr = p.runs[++rr];
    };
    if (i0 < n) {  // exact hashes: the direct formula once (and once more behind invalid bases), then rolling
        uint32_t rr = lo, kk = klo + i0;
        Run cr = run;
        seek(rr, cr, kk);
        uint64_t bb = cr.base_off + (kk - cr.kidx0) - b_glob + b;  // the k-mer's first base inside lw
        H2 h = {0u, 0u, 0u, 0u};
        warm_up(h, lw, bb, k, tab);  // (k rolling steps: no byte table in this kernel's LDS)
        lh[i0] = canonical<VARIANT>(h);
        for (uint32_t i = i0 + 1; i < i1; ++i) {
            ++kk;
            if (kk == cr.kidx0 + cr.n_kmers) {  // the next valid k-mer begins behind invalid bases
                cr = p.runs[++rr];
                bb = cr.base_off - b_glob + b;
                h = {0u, 0u, 0u, 0u};
                warm_up(h, lw, bb, k, tab);
            } else {
                const uint64_t go = bb, gi = go + k;
                const uint32_t o = (lw[go >> 4] >> (2u * ((uint32_t)go & 15u))) & 3u;
                const uint32_t in = (lw[gi >> 4] >> (2u * ((uint32_t)gi & 15u))) & 3u;
                nt_step(h, tab[o * 4u + in]);
                ++bb;
            }
            lh[i] = canonical<VARIANT>(h);
        }
    }
    for (uint32_t i = threadIdx.x; i < n; i += 256) lidx0[i] = (uint16_t)i;
    __syncthreads();
    // the smaller hash, the RIGHT one of equals (btllib rescans with <=)
    auto best = [&](uint32_t a, uint32_t c2) {
        const uint64_t ha = lh[a], hc = lh[c2];
        return (hc < ha || (hc == ha && c2 > a)) ? c2 : a;
    };
    uint32_t J = 0;
    while ((2u << J) <= w) ++J;  // 2^J <= w < 2^(J+1)
    uint32_t cur = 0;
    for (uint32_t lv = 0; lv < J; ++lv) {  // lidxc(cur)[i] = arg-min over [i, i + 2^lv) -> [i, i + 2^(lv+1))
        const uint32_t step = 1u << lv;
        for (uint32_t i = threadIdx.x; i < n; i += 256) {
            const uint32_t a = lidxc(cur)[i];
            lidxc(cur ^ 1)[i] = (uint16_t)(i + step < n ? best(a, lidxc(cur)[i + step]) : a);
        }
        cur ^= 1;
        __syncthreads();
    }
    const uint32_t span = 1u << J;
    // (... or the piece of a long stretch that starts with the last window of the piece before it: sel_push_gap, sketch_bs.hip)
    const bool drop = (klo == 0 && p.ctg_drop && p.ctg_drop[c]) || (gp.w & SEL_GAP_DROP) != 0;
    for (uint32_t s = threadIdx.x; s + w <= n; s += 256) {
        const uint32_t a = best(lidxc(cur)[s], lidxc(cur)[s + w - span]);
        if (lh[a] != 0xFFFFFFFFFFFFFFFFull) atomicOr(&selbits[a >> 5], 1u << (a & 31u));  // btllib never reports 2^64-1
        if (s == 0 && drop) *drop_idx = a;  // a piece's first window belongs to the shard before it (plan_pieces)
    }
    __syncthreads();
    if (threadIdx.x == 0 && *drop_idx != 0xFFFFFFFFu) selbits[*drop_idx >> 5] &= ~(1u << (*drop_idx & 31u));
    __syncthreads();
    uint32_t cnt = 0;
    for (uint32_t i = i0; i < i1 && i0 < n; ++i) cnt += (selbits[i >> 5] >> (i & 31u)) & 1u;
    uint32_t o = block_exclusive_256(cnt, sh);
    const uint32_t total = sh[255];
    if (threadIdx.x == 0) atomicAdd(&p.ctrl[10], n);
    if (threadIdx.x == 0) sh[0] = total ? atomicAdd(&p.ctrl[14], total) : 0u;  // (sh: the scan is done with it)
    __syncthreads();
    const uint32_t region = sh[0];
    __syncthreads();
    if (region + total > p.pool) {  // (block-uniform)
        if (threadIdx.x == 0) defer_stretch(p, gp);
        return;
    }
    if (threadIdx.x == 0) {
        p.r_cnt[j] = total;
        p.r_start[j] = region;
    }
    const uint32_t rec = p.ctg_rec[c];
    uint32_t rr = lo;
    Run cr = run;
    for (uint32_t i = i0; i < i1 && i0 < n; ++i)
        if ((selbits[i >> 5] >> (i & 31u)) & 1u) {
            const size_t at = (size_t)region + o++;
            seek(rr, cr, klo + i);
            p.r_hash[at] = ext_hash(lh[i], p.mult);
            p.r_pos[at] = cr.pos0 + (klo + i - cr.kidx0);
            p.r_rec[at] = rec;
        }
}

template <int VARIANT>
__global__ __launch_bounds__(256) void k_gap_fix(const GapFixParams p)
{
    // GAP_FIX_BLOCKS blocks walk the stretches (a block of 51 KB per possible stretch -- 2048, nearly all with nothing to do -- came
    // to the CUs in three rounds: 36 us per launch beside the other stream's kernels)
    const uint32_t n_g = p.ctrl[1];
    if (blockIdx.x >= n_g || n_g > p.gcap || p.ctrl[0]) return;
    __shared__ uint64_t lh[GAP_DEV_NSMALL];
    __shared__ uint16_t lidx[2][GAP_DEV_NSMALL];
    __shared__ uint32_t selbits[GAP_DEV_NSMALL / 32];
    __shared__ uint4 tab[20];
    __shared__ uint32_t sh[256];
    __shared__ uint32_t drop_idx;
    __shared__ uint32_t lw[GAP_DEV_NSMALL / 16 + 1024 / 16 + 4];  // the stretch's packed bases (+ k): every later read is local
    if (threadIdx.x < 20) tab[threadIdx.x] = p.tab.e[threadIdx.x];
    __syncthreads();
    for (uint32_t j = blockIdx.x; j < n_g; j += gridDim.x) {
        const uint4 g = p.gaps[j];
        if (g.z - g.y + 1u <= GAP_DEV_NSMALL) {
            gap_fix_one<VARIANT>(p, j, GAP_DEV_NSMALL, lh, lidx[0], lidx[1], selbits, lw, sh, tab, &drop_idx);
        } else if (threadIdx.x == 0) {  // longer than the block's arrays: left to the host (it contributes no minimizer here)
            p.r_key[j] = ((uint64_t)g.x << 32) | g.y;
            p.r_cnt[j] = 0;
            defer_stretch(p, g);
        }
        __syncthreads();  // the work arrays are reused
    }
}

struct GapPostParams {
    uint32_t *ctrl;  // [1] stretches, [6] flag, [7] <- minimizers found in them
    const uint32_t *r_cnt; const uint64_t *r_key;
    // the stretches in (contig, first k-mer) order: key, minimizers in the stretches before it ([n] = all), index of its region
    uint64_t *s_key; uint32_t *s_off, *s_src;
    uint32_t gcap;   // stretches the arrays hold
};

// The batch's stretches ranked by (contig, first k-mer): every stretch counts the stretches with a smaller key -- its rank (the
// keys are distinct) -- and, in the same pass, the minimizers those hold -- its offset: nothing is scanned, no block waits for
// another.  A block takes 16 stretches, sixteen threads each (a sixteenth of the other keys per thread), the other stretches' keys
// and counts pass through 6 KB of LDS 512 at a time (a block that wants more LDS than k_bs_select leaves free on a CU waits for
// one of its blocks to end).  One block of 256 threads ranking everything took 48 us for 950 stretches (a third of the batch's
// emit + stretch time at 2 x 10^9 k-mers per batch).
constexpr uint32_t GPB = 256, GP_PER = 16, GP_PARTS = GPB / GP_PER;  // (64 stretches x 4 threads until round 5: 30 us for 3000 stretches)
__global__ __launch_bounds__(GPB) void k_gap_post(const GapPostParams p)
{
    constexpr uint32_t CH = 512;
    static_assert(CH % GP_PARTS == 0, "a chunk of keys is dealt out evenly");
    __shared__ uint64_t keys[CH];
    __shared__ uint32_t cnts[CH];
    __shared__ uint32_t part_rank[GPB], part_off[GPB];
    const uint32_t n_g = p.ctrl[1];
    if (n_g == 0 || n_g > p.gcap || p.ctrl[0]) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            p.ctrl[7] = 0;
            if (n_g > p.gcap) p.ctrl[6] = 1;
        }
        return;
    }
    if (blockIdx.x * GP_PER >= n_g) return;
    const uint32_t i = blockIdx.x * GP_PER + (threadIdx.x & (GP_PER - 1u)), part = threadIdx.x / GP_PER;
    const bool on = i < n_g;
    const uint64_t mine = on ? p.r_key[i] : ~0ull;
    uint32_t rank = 0, off = 0, total = 0;
    for (uint32_t c0 = 0; c0 < n_g; c0 += CH) {
        const uint32_t cn = min(CH, n_g - c0);
        __syncthreads();
        for (uint32_t q = threadIdx.x; q < CH; q += GPB) {  // (beyond the stretches: keys no stretch lies behind)
            keys[q] = q < cn ? p.r_key[c0 + q] : ~0ull;
            cnts[q] = q < cn ? p.r_cnt[c0 + q] : 0u;
        }
        __syncthreads();
        // a thread's share is strided by the threads of its stretch: the 16 stretches of a part read the same word (one broadcast)
#pragma unroll 8
        for (uint32_t q = part; q < CH; q += GP_PARTS) {
            const bool less = keys[q] < mine;
            rank += less ? 1u : 0u;
            off += less ? cnts[q] : 0u;
        }
        if (blockIdx.x == 0)
            for (uint32_t q = threadIdx.x; q < cn; q += GPB) total += cnts[q];
    }
    part_rank[threadIdx.x] = rank;
    part_off[threadIdx.x] = off;
    __syncthreads();
    if (part == 0 && on) {
        const uint32_t t = threadIdx.x;
        uint32_t r = 0, o = 0;
#pragma unroll
        for (uint32_t u = 0; u < GP_PARTS; ++u) {
            r += part_rank[t + u * GP_PER];
            o += part_off[t + u * GP_PER];
        }
        p.s_key[r] = mine;
        p.s_src[r] = i;
        p.s_off[r] = o;
    }
    if (blockIdx.x == 0) {  // the sum of all counts
        __syncthreads();
        part_rank[threadIdx.x] = total;
        __syncthreads();
        for (uint32_t st = GPB / 2; st > 0; st >>= 1) {
            if (threadIdx.x < st) part_rank[threadIdx.x] += part_rank[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            p.ctrl[7] = part_rank[0];
            p.s_off[n_g] = part_rank[0];
        }
    }
}


// ------------------------------------------------------------------------------------------------------
// stretches of any length, any number of minimizers: tiles
// ------------------------------------------------------------------------------------------------------
// What k_gap_fix leaves to the host (defer_stretch) is the repeat structure of real genomes: satellite arrays (tens of
// thousands of k-mers none of which is a candidate), homopolymer / dinucleotide runs (every k-mer hashes alike, so every window
// reports its LAST k-mer: hundreds of minimizers in a row), stretches that span N gaps.  The dense pipeline would take them,
// but its window decision scans up to w neighbours per k-mer when hashes tie -- 30 ms for the 2 x 10^7 such k-mers of a
// repeat-rich Gbp.  Here a stretch is cut into tiles of ST_WIN window starts; a block hashes the tile's k-mers (+ w - 1
// behind them, + 1 in front) into LDS and takes every window's rightmost arg-min from a doubling table, O(log w) per window
// whatever the hashes.  The arg-min moves right monotonically with the window, so the only minimizer a tile can share with the
// tile before it is the arg-min of the window just in front: the tile computes that one too and leaves it out.
// Two passes over the same tiles: counts, then (offsets known) the minimizers themselves, in (contig, position) order.
constexpr uint32_t ST_WIN = 1024;                       // window starts per tile (49 KB of LDS per block at ST_WMAX)
constexpr uint32_t ST_WMAX = 2048;                      // largest w this route takes (the caller checks)
constexpr uint32_t ST_LDS = ST_WIN + ST_WMAX + 1;       // k-mers a tile holds

struct StretchTile {
    uint32_t contig, klo;    // the stretch: contig, its first k-mer (contig-local valid-k-mer index)
    uint32_t t0, nwin;       // this tile: first window start relative to klo, window starts
};
struct StretchParams {
    const StretchTile *tiles;
    uint32_t n_tiles;
    const Run *runs;
    const uint32_t *ctg_run0, *ctg_rec;
    const uint8_t *ctg_drop;
    const uint32_t *packed;
    uint32_t k, w;
    uint64_t mult;
    HashTab tab;
    uint32_t *cnt;           // [n_tiles] pass 0: minimizers per tile
    const uint32_t *off;     // [n_tiles] pass 1: where the tile's minimizers start
    uint64_t *o_hash;
    uint32_t *o_pos, *o_rec;
};

template <int VARIANT, int PASS>
__global__ __launch_bounds__(256) void k_stretch_tiles(const StretchParams p)
{
    __shared__ uint64_t lh[ST_LDS];
    __shared__ uint16_t lidx[2][ST_LDS];
    __shared__ uint32_t lpos[PASS ? ST_LDS : 1];  // base position (contig-local) of every k-mer
    __shared__ uint32_t selbits[(ST_LDS + 31) / 32];
    __shared__ uint4 tab[20];
    __shared__ uint32_t sh[256];
    __shared__ uint32_t excl[2];
    const StretchTile tl = p.tiles[blockIdx.x];
    const uint32_t w = p.w, k = p.k, c = tl.contig;
    const uint32_t front = tl.t0 ? 1u : 0u;                       // the window in front of the tile (for the exclusion)
    const uint32_t i_lo = tl.klo + tl.t0 - front;                 // first k-mer held (contig-local index)
    const uint32_t m = front + tl.nwin + w - 1u;                  // k-mers held: <= ST_LDS
    if (threadIdx.x < 20) tab[threadIdx.x] = p.tab.e[threadIdx.x];
    for (uint32_t i = threadIdx.x; i < (ST_LDS + 31) / 32; i += 256) selbits[i] = 0;
    if (threadIdx.x < 2) excl[threadIdx.x] = 0xFFFFFFFFu;
    __syncthreads();
    // exact hashes: every thread a run of consecutive k-mers -- the direct formula at its first k-mer and wherever a new run of
    // valid bases begins, rolling in between
    const uint32_t per = (m + 255u) / 256u, a0 = threadIdx.x * per, a1 = min(a0 + per, m);
    if (a0 < m) {
        uint32_t lo = p.ctg_run0[c], hi = p.ctg_run0[c + 1];
        const uint32_t ki = i_lo + a0;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (p.runs[mid].kidx0 <= ki) lo = mid; else hi = mid;
        }
        Run run = p.runs[lo];
        H2 h = {0u, 0u, 0u, 0u};
        bool fresh = true;
        for (uint32_t a = a0; a < a1; ++a) {
            const uint32_t kx = i_lo + a;
            if (kx >= run.kidx0 + run.n_kmers) {
                run = p.runs[++lo];
                fresh = true;
            }
            const uint64_t b = run.base_off + (kx - run.kidx0);
            if (fresh) {
                h = H2{0u, 0u, 0u, 0u};
                warm_up(h, p.packed, b, k, tab);
                fresh = false;
            } else {
                const uint64_t go = b - 1, gi = go + k;
                const uint32_t o = (p.packed[go >> 4] >> (2u * ((uint32_t)go & 15u))) & 3u;
                const uint32_t in = (p.packed[gi >> 4] >> (2u * ((uint32_t)gi & 15u))) & 3u;
                nt_step(h, tab[o * 4u + in]);
            }
            lh[a] = canonical<VARIANT>(h);
            if (PASS) lpos[a] = run.pos0 + (kx - run.kidx0);
        }
    }
    for (uint32_t i = threadIdx.x; i < m; i += 256) lidx[0][i] = (uint16_t)i;
    __syncthreads();
    auto best = [&](uint32_t a, uint32_t c2) {  // the smaller hash, the RIGHT one of equals (btllib rescans with <=)
        const uint64_t ha = lh[a], hc = lh[c2];
        return (hc < ha || (hc == ha && c2 > a)) ? c2 : a;
    };
    uint32_t J = 0;
    while ((2u << J) <= w) ++J;  // 2^J <= w < 2^(J+1)
    uint32_t cur = 0;
    for (uint32_t lv = 0; lv < J; ++lv) {  // lidx[cur][i] = arg-min over [i, i + 2^lv) -> [i, i + 2^(lv+1))
        const uint32_t step = 1u << lv;
        for (uint32_t i = threadIdx.x; i < m; i += 256) {
            const uint32_t a = lidx[cur][i];
            lidx[cur ^ 1][i] = (uint16_t)(i + step < m ? best(a, lidx[cur][i + step]) : a);
        }
        cur ^= 1;
        __syncthreads();
    }
    const uint32_t span = 1u << J;
    const bool drop = tl.klo == 0 && tl.t0 == 0 && p.ctg_drop && p.ctg_drop[c];
    for (uint32_t s = threadIdx.x; s < front + tl.nwin; s += 256) {
        const uint32_t a = best(lidx[cur][s], lidx[cur][s + w - span]);
        if (s < front) {
            excl[0] = a;  // the window in front of the tile: its arg-min is the previous tile's to report
        } else {
            if (lh[a] != 0xFFFFFFFFFFFFFFFFull) atomicOr(&selbits[a >> 5], 1u << (a & 31u));  // btllib never reports 2^64-1
            if (s == 0 && drop) excl[1] = a;  // a piece's first window belongs to the shard before it (plan_pieces)
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 && excl[threadIdx.x] != 0xFFFFFFFFu) atomicAnd(&selbits[excl[threadIdx.x] >> 5], ~(1u << (excl[threadIdx.x] & 31u)));
    __syncthreads();
    uint32_t cnt = 0;
    for (uint32_t i = a0; i < a1 && a0 < m; ++i) cnt += (selbits[i >> 5] >> (i & 31u)) & 1u;
    uint32_t o = block_exclusive_256(cnt, sh);
    if (PASS == 0) {
        if (threadIdx.x == 0) p.cnt[blockIdx.x] = sh[255];
        return;
    }
    const uint32_t rec = p.ctg_rec[c];
    const size_t base = p.off[blockIdx.x];
    for (uint32_t i = a0; i < a1 && a0 < m; ++i)
        if ((selbits[i >> 5] >> (i & 31u)) & 1u) {
            const size_t at = base + o++;
            p.o_hash[at] = ext_hash(lh[i], p.mult);
            p.o_pos[at] = lpos[i];
            p.o_rec[at] = rec;
        }
}

// Pinned host copy of a batch's control block (16 words), written by the batch's last kernel:
// [0] largest wave count if a wave overflowed its arena slice, [1] candidate-free stretches, [2] minimizers among the
// candidates, [3] "the device route could not finish the stretches", [4] candidates, [5] minimizers inside stretches,
// [6..7] minimizers of the batch, [8..9] where the batch starts in the assembly's sketch, [10] k-mers hashed by k_gap_fix,
// [11] stretches left to the host (their {contig, first, last k-mer} in the batch's slice of pinned_defer)
// ------------------------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------------------------
constexpr int S_DENSE = 128;

// MXG_SPARSE_S=<n> (environment) forces the strip length instead of the choice below.
// Strip length for the sparse kernel: one lane rolls S consecutive k-mers.  A strip costs one direct hash evaluation
// (k/4 table lookups) and 16 B of strip table, and every wave one slice of the candidate arena and one k_reorder block,
// so long strips make the stages after the hash kernel cheaper; but a lane's 16-base words are S/4 bytes apart, so
// long strips spread a wave's loads over more cache lines and leave fewer waves per SIMD on small inputs.  Measured on
// MI355X (tools/sweep_S.sh, sketch+graph, k=32 w=1000): 2 x 100 Mbp 569 / 578 / 584 / 588 / 556 Gbp/s and 2 x 1 Gbp
// 659 / 675 / 683 / 689 / 581 Gbp/s at S = 192 / 256 / 320 / 384 / 512; the hash kernel alone is fastest at 192-256.
static uint32_t choose_sparse_S(const mxg_handle *h, uint64_t total_kmers)
{
    const uint64_t e = knob_u64(h, "MXG_SPARSE_S", 0);
    if (e >= 16) return (uint32_t)std::min<uint64_t>(1024, (e + 15) / 16 * 16);
    // the k = 32 route's slice kernel wants w k-mers in a few strips (sketch_bs.h: SEL_MAX_H), whatever the input's size
    const bool sel_route = h->cfg.k == 32 && h->cfg.variant == MXG_VARIANT_V2_SUM && knob_u64(h, "MXG_BS", 1) != 0 &&
                           knob_u64(h, "MXG_BS_SELECT", 1) != 0;
    // (352 at eight candidates per window: ~186 raw candidates per slice = three rounds of 64 lanes in the slice kernel's hash and
    // decision phases, nearly full; tools/sweep_cS.sh, round 6: 2.60 / 2.58 / 2.68 ms per step at S = 320 / 352 / 384)
    if (sel_route && h->cfg.w > 64 * SEL_MAX_H) return knob_u64(h, "MXG_SEL_INLINE", 1) ? 352 : 320;
    const uint64_t lanes = 1024ull * 64;  // SIMDs x lanes
    // small inputs: keep at least ~4 waves per SIMD in flight
    uint64_t S = (total_kmers + lanes * 4 - 1) / (lanes * 4);
    S = (S + 15) / 16 * 16;
    S = std::min<uint64_t>(std::max<uint64_t>(S, 64), 320);
    // the slice kernel's work goes with the candidates of a slice (64 strips x S x c / w): shorter windows, shorter strips.  Measured at
    // w = 500 on configs[3] (tools/sweep_c3.sh; 12 Gbp, Gbp/s at S = 160 / 192 / 224 / 256 / 320): 1227 / 1313 / 1283 / 1256 / 1290
    if (sel_route && h->cfg.w >= 400 && S > 192) S = h->cfg.w >= 600 ? 256 : 192;
    return (uint32_t)S;
}
// batch sizes (whole records; a single record may exceed them).  Test knobs (environment, read per call):
// MXG_DENSE_BATCH_KMERS / MXG_SPARSE_BATCH_KMERS shrink the batches, MXG_WAVE_CAP forces the arena-overflow retry.
static uint64_t env_u64(const mxg_handle *h, const char *name, uint64_t dflt) { return knob_u64(h, name, dflt); }
#define DENSE_BATCH_KMERS env_u64(h, "MXG_DENSE_BATCH_KMERS", 96ull << 20)    /* dense arena = 16 B per k-mer */
#define SPARSE_BATCH_KMERS env_u64(h, "MXG_SPARSE_BATCH_KMERS", 512ull << 20) /* < 2^31 k-mers per batch; measured on MI355X
   (3 Gbp + 3 Gbp, k=32 w=1000): 985 / 968 / 931 Gbp/s at 512 Mi / 1 Gi / 2040 Mi k-mers per batch */
constexpr uint32_t GAP_CAP = 1u << 20;

static hipError_t grow_preserve(DevBuf &b, size_t used_bytes, size_t need_bytes, hipStream_t st)
{
    if (need_bytes <= b.bytes) return hipSuccess;
    void *np = nullptr;
    size_t want = need_bytes + need_bytes / 2 + 4096;
    hipError_t e = hipMalloc(&np, want);
    if (e != hipSuccess) return e;
    if (used_bytes) {
        e = hipMemcpyAsync(np, b.p, used_bytes, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            (void)hipFree(np);
            return e;
        }
    }
    if (b.p) (void)hipFree(b.p);
    b.p = np;
    b.bytes = want;
    return hipSuccess;
}

template <class T>
static int upload(mxg_handle *h, DevBuf &b, const std::vector<T> &v, hipStream_t st = nullptr)
{
    MXG_HIP(h, b.ensure(std::max<size_t>(v.size() * sizeof(T), 16)));
    if (!v.empty())
        MXG_HIP(h, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st ? st : h->stream));
    return MXG_OK;
}

// a set of contigs/runs to sketch: host vectors + their device copies
struct Tables {
    const std::vector<Run> *runs;
    const std::vector<uint32_t> *ctg_nk, *ctg_rec, *ctg_run0;
    const std::vector<uint32_t> *strip0_dense, *strip0_sparse;  // sparse may be null
    const std::vector<uint64_t> *g0;
    const Run *d_runs;
    const uint32_t *d_strip0_dense, *d_strip0_sparse, *d_ctg_nk, *d_ctg_rec, *d_ctg_run0;
    const uint4 *d_ctg_info = nullptr;      // EmitParams::ctg_info, or null (virtual contigs)
    const uint32_t *d_strip_run = nullptr;  // strip (sparse table) -> run, or null (virtual contigs)
    const uint64_t *d_g0;
    const uint8_t *d_ctg_drop = nullptr;  // split load: contigs whose first minimizer is not reported (k_resolve), else null
    const std::vector<Record> *recs;  // for base accounting (may be null)
};

struct OutArrays {
    DevBuf *hash, *pos, *rec, *fwd;
    uint64_t n = 0;
    uint64_t cap() const { return std::min<uint64_t>({hash->bytes / 8, pos->bytes / 4, rec->bytes / 4, fwd->bytes}); }
};

static int out_reserve(mxg_handle *h, OutArrays &o, uint64_t need, hipStream_t st)
{
    if (need <= o.cap()) return MXG_OK;
    MXG_HIP(h, grow_preserve(*o.hash, o.n * 8, need * 8, st));
    MXG_HIP(h, grow_preserve(*o.pos, o.n * 4, need * 4, st));
    MXG_HIP(h, grow_preserve(*o.rec, o.n * 4, need * 4, st));
    MXG_HIP(h, grow_preserve(*o.fwd, o.n, need, st));
    return MXG_OK;
}

static void build_strip_tables(const std::vector<Run> &runs, int S, std::vector<uint32_t> &strip0, bool *overflow)
{
    strip0.resize(runs.size() + 1);
    uint64_t s = 0;
    for (size_t r = 0; r < runs.size(); ++r) {
        strip0[r] = (uint32_t)s;
        s += (runs[r].n_kmers + S - 1) / S;
    }
    strip0[runs.size()] = (uint32_t)s;
    if (s >= (1ull << 32)) *overflow = true;
}

// pinned host copies of the batches' control blocks: 16 words each; the last slot belongs to the synchronous path
constexpr uint32_t PINNED_SLOTS = 1024;
static int ensure_pinned_ctrl(mxg_handle *h)
{
    if (!h->pinned_ctrl) MXG_HIP(h, hipHostMalloc((void **)&h->pinned_ctrl, (size_t)PINNED_SLOTS * 64));
    if (!h->pinned_defer) MXG_HIP(h, hipHostMalloc((void **)&h->pinned_defer, (size_t)PINNED_SLOTS * GAP_DEFER_MAX * 16));
    return MXG_OK;
}

struct Driver {
    mxg_handle *h;
    bool timing;
    hipStream_t st;   // stream this driver enqueues on
    int slot;         // which of the handle's two scratch sets it uses (two drivers can be in flight at once)
    explicit Driver(mxg_handle *h_, int slot_ = 0)
        : h(h_), timing((h_->cfg.flags & (MXG_FLAG_TIMING | MXG_FLAG_TIMING_FINE)) != 0), st(slot_ == 0 ? h_->stream : slot_ == 1 ? h_->stream2 : h_->stream_x[slot_ - 2]), slot(slot_)
    {
        fine = (h_->cfg.flags & MXG_FLAG_TIMING_FINE) != 0;
    }
    DevBuf &sc(int i) { return h->scratch[slot][i]; }
    // stretches the device route's per-stretch arrays hold for the batch being enqueued, and the blocks at the front of k_emit's
    // grid that place them, four stretches each (see gap_capacity)
    uint32_t gcap = GAP_DEV_MAX, n_place = GAP_DEV_MAX / 4;
    void set_gaps(uint32_t cap, uint32_t place, double expect = -1.0)
    {
        gcap = cap;
        n_place = place;
        gap_expect = expect;
    }
    double gap_expect = -1.0;  // stretches the plan expects of the batch for the stretch kernels (< 0: not known)

    // SC_CTRL holds the control block (16 words) followed by the two super-count arrays of the batch (scan_kernels.h):
    // per hash-kernel wave, then per k_resolve block; ctrl_bytes() of it are zeroed by the batch's one memset
    static constexpr uint32_t CTRL_WORDS = 16;
    uint32_t n_wave_sup = 0;
    uint32_t *wave_sup() { return sc(SC_CTRL).as<uint32_t>() + CTRL_WORDS; }
    uint32_t *sel_sup(uint32_t) { return wave_sup() + n_wave_sup; }
    static uint32_t n_sel_sup(uint32_t n_cap) { return sup_words((n_cap + RK - 1) / RK); }

    bool fine = false;  // MXG_FLAG_TIMING_FINE: a span per kernel
    int ev_next(int kind)  // fine timing only: close the running span and open one of `kind`
    {
        if (!timing || !fine) return MXG_OK;
        int rc = ev_end();
        return rc != MXG_OK ? rc : ev_begin(0, false, kind);
    }
    int ev_begin(uint64_t bases, bool is_hash, int kind = -1)
    {
        if (!timing) return MXG_OK;
        if (kind < 0) kind = is_hash ? 0 : (fine ? 2 : 1);
        while (h->ev_pool.size() < h->ev_used + 2) {
            hipEvent_t e;
            MXG_HIP(h, hipEventCreate(&e));
            h->ev_pool.push_back(e);
        }
        TimedSpan sp{h->ev_pool[h->ev_used], h->ev_pool[h->ev_used + 1], bases, is_hash, kind};
        h->ev_used += 2;
        MXG_HIP(h, hipEventRecord(sp.a, st));
        h->ev_spans.push_back(sp);
        return MXG_OK;
    }
    int ev_end()
    {
        if (!timing) return MXG_OK;
        MXG_HIP(h, hipEventRecord(h->ev_spans.back().b, st));
        return MXG_OK;
    }
    int collect()
    {
        // read back lazily; only keep the number of events in flight bounded
        return h->ev_spans.size() > 4096 ? flush_timers(h) : MXG_OK;
    }

    uint64_t batch_bases(const Tables &T, size_t c0, size_t c1) const
    {
        uint64_t b = 0;
        if (T.recs)
            for (size_t c = c0; c < c1; ++c) b += (*T.recs)[(*T.ctg_rec)[c]].len;
        return b;
    }

    // sparse path: resolve + gap detection + per-256 counts (SC_CNT256 + super-counts behind the control block) in ONE
    // launch; emit(..., true) turns them into offsets, places the minimizers and writes the total to ctrl[2..3]
    int resolve_count(const Tables &T, uint32_t n_cap, uint32_t ctg_lo, uint32_t ctg_hi, uint64_t tau, uint32_t n_likely)
    {
        if (!n_cap) return MXG_OK;
        MXG_HIP(h, sc(SC_SEL).ensure(std::max<uint32_t>(n_cap, 16)));
        const uint32_t blocks = ((grid_cand ? std::min(grid_cand, n_cap) : n_cap) + RK - 1) / RK;
        MXG_HIP(h, sc(SC_CNT256).ensure((size_t)blocks * 4 + 64));
        uint32_t *ctrl = sc(SC_CTRL).as<uint32_t>();
        ResolveParams rp;
        rp.ch = sc(SC_CAND_H).as<uint64_t>();
        rp.ck = sc(SC_CAND_K).as<uint32_t>();
        rp.cc = sc(SC_CAND_C).as<uint32_t>();
        rp.n_ptr = ctrl + 4;
        rp.n_cap = n_cap;
        rp.n_likely = 0;
        rp.ovf = ctrl;
        rp.tau = tau;
        rp.ctg_nk = T.d_ctg_nk;
        rp.ctg_drop = T.d_ctg_drop;
        rp.w = h->cfg.w;
        rp.sel = sc(SC_SEL).as<uint8_t>();
        rp.ctg_lo = ctg_lo;
        rp.ctg_hi = ctg_hi;
        rp.gaps = sc(SC_GAPS).as<uint4>();
        rp.gap_cap = GAP_CAP;
        rp.gap_count = ctrl + 1;
        rp.cnt256 = sc(SC_CNT256).as<uint32_t>();
        rp.sel_sup = sel_sup(n_cap);
        MXG_HIP(h, sc(SC_CS_H).ensure((size_t)blocks * RK * 8));
        MXG_HIP(h, sc(SC_CS_K).ensure((size_t)blocks * RK * 4));
        MXG_HIP(h, sc(SC_CS_C).ensure((size_t)blocks * RK * 4));
        rp.cs_h = sc(SC_CS_H).as<uint64_t>();
        rp.cs_k = sc(SC_CS_K).as<uint32_t>();
        rp.cs_c = sc(SC_CS_C).as<uint32_t>();
        rp.n_likely = std::min(n_likely, n_cap);
        static const int abl = getenv("MXG_ABLATE_RESOLVE") ? atoi(getenv("MXG_ABLATE_RESOLVE")) : 0;  // profiling only
        if (abl == 1)
            hipLaunchKernelGGL((k_resolve<true, true, 1>), dim3(blocks), dim3(RK), 0, st, rp);
        else if (abl == 2)
            hipLaunchKernelGGL((k_resolve<true, true, 2>), dim3(blocks), dim3(RK), 0, st, rp);
        else if (abl == 3)
            hipLaunchKernelGGL((k_resolve<true, true, 3>), dim3(blocks), dim3(RK), 0, st, rp);
        else if (env_u64(h, "MXG_RH", few_cand ? 32 : 64) == 32)  // halo of 32 candidates: enough for <= 12 per window (976 -> 985 Gbp/s)
            hipLaunchKernelGGL((k_resolve<true, true, 0, 32>), dim3(blocks), dim3(RK), 0, st, rp);
        else
            hipLaunchKernelGGL((k_resolve<true, true>), dim3(blocks), dim3(RK), 0, st, rp);
        MXG_HIP(h, hipGetLastError());
        return MXG_OK;
    }

    // resolve -> count -> scan over candidates already in SC_CAND_*; n candidates read from ctrl[4] (<= n_cap)
    template <bool GAPS>
    int resolve_and_count(const Tables &T, uint32_t n_cap, uint32_t ctg_lo, uint32_t ctg_hi, uint64_t tau)
    {
        const uint32_t n_tiles = (n_cap + TILE - 1) / TILE;
        MXG_HIP(h, sc(SC_SEL).ensure(std::max<uint32_t>(n_cap, 16)));
        MXG_HIP(h, sc(SC_BSUM).ensure((size_t)n_tiles * 4 + 16));
        uint32_t *ctrl = sc(SC_CTRL).as<uint32_t>();
        ResolveParams rp;
        rp.ch = sc(SC_CAND_H).as<uint64_t>();
        rp.ck = sc(SC_CAND_K).as<uint32_t>();
        rp.cc = sc(SC_CAND_C).as<uint32_t>();
        rp.n_ptr = ctrl + 4;
        rp.n_cap = n_cap;
        rp.n_likely = 0;
        rp.ovf = ctrl;
        rp.tau = tau;
        rp.ctg_nk = T.d_ctg_nk;
        rp.ctg_drop = T.d_ctg_drop;
        rp.w = h->cfg.w;
        rp.sel = sc(SC_SEL).as<uint8_t>();
        rp.ctg_lo = ctg_lo;
        rp.ctg_hi = ctg_hi;
        rp.gaps = sc(SC_GAPS).as<uint4>();
        rp.gap_cap = GAP_CAP;
        rp.gap_count = ctrl + 1;
        rp.cnt256 = nullptr;
        rp.sel_sup = nullptr;
        rp.cs_h = nullptr;
        rp.cs_k = rp.cs_c = nullptr;
        if (n_cap) {
            hipLaunchKernelGGL((k_resolve<GAPS, false>), dim3((n_cap + RK - 1) / RK), dim3(RK), 0, st, rp);
            hipLaunchKernelGGL(k_count_n, dim3(n_tiles), dim3(256), 0, st, rp.sel, rp.n_ptr, n_cap,
                               sc(SC_BSUM).as<uint32_t>());
        }
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, st, sc(SC_BSUM).as<uint32_t>(), n_tiles,
                           reinterpret_cast<uint64_t *>(ctrl + 2));
        MXG_HIP(h, hipGetLastError());
        return MXG_OK;
    }

    // offsets: SC_BSUM per 1024-tile (after resolve_and_count) or, fused = true, from SC_CNT256 (after resolve_count)
    uint32_t *n_out = nullptr;  // see EmitParams::n_out (set by sketch_assemblies for the fused call)
    bool few_cand = false;      // <= 12 candidates per window: smaller k_reorder blocks and k_resolve halos (set per batch)
    // Batches enqueued without a host sync: k_resolve and k_emit are launched for the EXPECTED number of candidates (+ 30 %)
    // instead of the arrays' capacity (about 2.7 x the expectation), so that half their blocks do not start just to find
    // nothing to do -- blocks that each hold a wave slot for a memory round trip beside the other stream's hash kernel.  A batch
    // with more candidates than that never reports (no tile holds its last candidate): the host redoes the assembly.
    uint32_t grid_cand = 0;     // 0: the capacity
    // how a batch hangs together with the batches before it and with the device-side stretch fix-up (see EmitParams)
    struct ChainIO {
        const uint64_t *base_in = nullptr;
        uint64_t *base_out = nullptr;
        bool dev_gaps = false;
        hipEvent_t wait = nullptr;  // recorded behind the previous batch of the assembly when that ran on the other stream
        uint32_t ecb = 0;           // slices per k_emit tile behind k_bs_select (0: the default)
    };
    int emit(const uint32_t *d_packed, const Tables &T, uint32_t n_cap, DevBuf &oh, DevBuf &op, DevBuf &orc, DevBuf &of,
             uint64_t out_base, bool fused = false, uint32_t *host_ctrl = nullptr, const ChainIO *io = nullptr,
             uint32_t rk = RK, uint32_t n_fixed = 0, const uint32_t *cand_spread = nullptr)
    {
        const uint64_t limit = std::min<uint64_t>({oh.bytes / 8, op.bytes / 4, orc.bytes / 4, of.bytes});
        if (!n_cap) return MXG_OK;
        EmitParams ep;
        ep.sel = sc(SC_SEL).as<uint8_t>();
        ep.ch = sc(SC_CAND_H).as<uint64_t>();
        ep.ck = sc(SC_CAND_K).as<uint32_t>();
        ep.cc = sc(SC_CAND_C).as<uint32_t>();
        ep.n_ptr = sc(SC_CTRL).as<uint32_t>() + 4;
        ep.ovf = sc(SC_CTRL).as<uint32_t>();
        ep.n_cap = n_cap;
        ep.bsum = fused ? nullptr : sc(SC_BSUM).as<uint32_t>();
        ep.cnt256 = fused ? sc(SC_CNT256).as<uint32_t>() : nullptr;
        ep.sel_sup = fused ? sel_sup(n_cap) : nullptr;
        ep.n_sel = sc(SC_CTRL).as<uint32_t>() + 2;
        ep.host_ctrl = host_ctrl;
        ep.n_out = host_ctrl ? n_out : nullptr;
        ep.runs = T.d_runs;
        ep.ctg_run0 = T.d_ctg_run0;
        ep.ctg_rec = T.d_ctg_rec;
        ep.ctg_info = T.d_ctg_info;
        ep.mult = 1ull ^ ((uint64_t)h->cfg.k * 0x90b45d39fb6da1faull);
        (void)d_packed;
        ep.out_base = out_base;
        ep.out_limit = limit;
        ep.o_hash = oh.as<uint64_t>();
        ep.o_pos = op.as<uint32_t>();
        ep.o_rec = orc.as<uint32_t>();
        ep.o_fwd = of.as<uint8_t>();
        ep.cs_h = fused ? sc(SC_CS_H).as<uint64_t>() : nullptr;  // (the sparse path's k_resolve laid the selected ones out per block)
        ep.cs_k = fused ? sc(SC_CS_K).as<uint32_t>() : nullptr;
        ep.cs_c = fused ? sc(SC_CS_C).as<uint32_t>() : nullptr;
        ep.cs_aos = fused && n_fixed ? sc(SC_CS_H).as<uint4>() : nullptr;  // (n_fixed: the batch went through k_bs_select)
        ep.base_in = io ? io->base_in : nullptr;
        ep.base_out = io ? io->base_out : nullptr;
        ep.dev_gaps = io && io->dev_gaps ? 1u : 0u;
        ep.gcap = gcap;
        ep.n_place = n_place;
        ep.rk = rk;
        ep.n_fixed = n_fixed;
        ep.cand_spread = cand_spread;
        const uint32_t n_grid = n_fixed ? n_fixed : (grid_cand ? std::min(grid_cand, n_cap) : n_cap);
        ep.ecb = EMIT_COMPACT_BLOCKS;
        if (rk < RK) {  // the slices of k_bs_select: ~18 minimizers each
            const uint64_t e = env_u64(h, "MXG_EMIT_ECB", io && io->ecb ? io->ecb : 16);
            ep.ecb = e >= 64 ? 64u : e >= 32 ? 32u : 16u;
        }
        ep.n_tiles = fused ? (n_grid + ep.ecb * rk - 1) / (ep.ecb * rk) : (n_grid + TILE - 1) / TILE;
        ep.s_key = nullptr;
        ep.s_off = ep.s_src = nullptr;
        ep.gaps = nullptr;
        ep.r_hash = nullptr;
        ep.r_pos = ep.r_rec = ep.r_cnt = ep.r_start = nullptr;
        uint32_t grid = ep.n_tiles;
        if (ep.dev_gaps) {
            ep.s_key = sc(SC_GD_HASH).as<uint64_t>();
            ep.s_off = sc(SC_GD_POS).as<uint32_t>();
            ep.s_src = sc(SC_GD_REC).as<uint32_t>();
            ep.gaps = sc(SC_GAPS).as<uint4>();
            ep.r_hash = sc(SC_GR_HASH).as<uint64_t>();
            ep.r_pos = sc(SC_GR_POS).as<uint32_t>();
            ep.r_rec = sc(SC_GR_REC).as<uint32_t>();
            ep.r_cnt = sc(SC_GR_CNT).as<uint32_t>();
            ep.r_start = ep.r_cnt + gcap;
            grid += n_place;
        }
        hipLaunchKernelGGL(k_emit, dim3(grid), dim3(256), 0, st, ep);
        MXG_HIP(h, hipGetLastError());
        return MXG_OK;
    }

    // ---- dense pipeline over every contig of T, appended to `out` ------------------------------------
    int dense_all(const uint32_t *d_packed, const Tables &T, OutArrays &out, bool count_as_hash)
    {
        const size_t n_ctg = T.ctg_rec->size();
        MXG_HIP(h, sc(SC_CTRL).ensure(64));
        size_t c0 = 0;
        while (c0 < n_ctg) {
            size_t c1 = c0;
            uint64_t nk = 0;
            while (c1 < n_ctg && (c1 == c0 || nk + (*T.ctg_nk)[c1] <= DENSE_BATCH_KMERS)) nk += (*T.ctg_nk)[c1++];
            if (nk >= (1ull << 31))
                return set_err(h, MXG_ELIMIT, "a candidate-free stretch / record of %llu k-mers exceeds the dense path's 2^31 limit",
                               (unsigned long long)nk);
            const uint32_t r_lo = (*T.ctg_run0)[c0], r_hi = (*T.ctg_run0)[c1];
            MXG_HIP(h, sc(SC_CAND_H).ensure(nk * 8));
            MXG_HIP(h, sc(SC_CAND_K).ensure(nk * 4));
            MXG_HIP(h, sc(SC_CAND_C).ensure(nk * 4));
            const uint32_t n_cand = (uint32_t)nk;
            uint32_t ctrl_init[8] = {0, 0, 0, 0, n_cand, 0, 0, 0};
            MXG_HIP(h, hipMemcpyAsync(sc(SC_CTRL).p, ctrl_init, 32, hipMemcpyHostToDevice, st));
            DenseParams hp;
            hp.packed = d_packed;
            hp.runs = T.d_runs;
            hp.run_strip0 = T.d_strip0_dense;
            hp.run_g0 = T.d_g0;
            hp.run_lo = r_lo;
            hp.run_hi = r_hi;
            hp.strip_lo = (*T.strip0_dense)[r_lo];
            hp.strip_hi = (*T.strip0_dense)[r_hi];
            hp.g_base = (*T.g0)[r_lo];
            hp.k = h->cfg.k;
            hp.cand_h = sc(SC_CAND_H).as<uint64_t>();
            hp.cand_k = sc(SC_CAND_K).as<uint32_t>();
            hp.cand_c = sc(SC_CAND_C).as<uint32_t>();
            hp.tab = h->tab;
            int rc = ev_begin(count_as_hash ? batch_bases(T, c0, c1) : 0, count_as_hash);
            if (rc != MXG_OK) return rc;
            const uint32_t n_strips = hp.strip_hi - hp.strip_lo;
            dim3 grid((n_strips + 255) / 256), block(256);
            if (!n_strips) {
                // (a batch of records without a single k-mer: nothing to hash)
            } else if (h->cfg.variant == MXG_VARIANT_V1_MIN)
                hipLaunchKernelGGL((k_hash_dense<S_DENSE, MXG_VARIANT_V1_MIN>), grid, block, 0, st, hp);
            else
                hipLaunchKernelGGL((k_hash_dense<S_DENSE, MXG_VARIANT_V2_SUM>), grid, block, 0, st, hp);
            if (count_as_hash && (rc = ev_end()) != MXG_OK) return rc;
            MXG_HIP(h, hipGetLastError());
            if ((rc = resolve_and_count<false>(T, n_cand, (uint32_t)c0, (uint32_t)c1, ~0ull)) != MXG_OK) return rc;
            if (!count_as_hash && (rc = ev_end()) != MXG_OK) return rc;
            uint32_t ctrl[4];
            MXG_HIP(h, hipMemcpyAsync(ctrl, sc(SC_CTRL).p, 16, hipMemcpyDeviceToHost, st));
            MXG_HIP(h, hipStreamSynchronize(st));
            const uint64_t total = (uint64_t)ctrl[2] | ((uint64_t)ctrl[3] << 32);
            if ((rc = out_reserve(h, out, out.n + total, st)) != MXG_OK) return rc;
            if ((rc = emit(d_packed, T, n_cand, *out.hash, *out.pos, *out.rec, *out.fwd, out.n)) != MXG_OK) return rc;
            out.n += total;
            h->stat_dense_kmers += nk;
            c0 = c1;
        }
        return MXG_OK;
    }

    // ---- gap fix-up: dense pipeline over candidate-free stretches, result in SC_G_* ---------------------
    int process_gaps(Assembly *a, const Tables &T, std::vector<uint4> &gaps, uint64_t *n_gap_mx)
    {
        std::sort(gaps.begin(), gaps.end(), [](const uint4 &x, const uint4 &y) {
            return x.x != y.x ? x.x < y.x : x.y < y.y;
        });
        // every stretch becomes a stand-alone virtual contig made of pieces of the real contig's runs
        std::vector<Run> vruns;
        std::vector<uint32_t> v_nk, v_rec, v_run0;
        std::vector<uint8_t> v_drop;
        for (size_t v = 0; v < gaps.size(); ++v) {
            const uint32_t c = gaps[v].x, klo = gaps[v].y, khi = gaps[v].z;
            if (a->any_drop) v_drop.push_back(a->ctg_drop[c] && klo == 0 ? 1 : 0);  // the stretch holds the contig's first window
            v_run0.push_back((uint32_t)vruns.size());
            v_nk.push_back(khi - klo + 1);
            v_rec.push_back((*T.ctg_rec)[c]);
            for (uint32_t r = (*T.ctg_run0)[c]; r < (*T.ctg_run0)[c + 1]; ++r) {
                const Run &run = (*T.runs)[r];
                uint64_t lo = std::max<uint64_t>(run.kidx0, klo);
                uint64_t hi = std::min<uint64_t>((uint64_t)run.kidx0 + run.n_kmers, (uint64_t)khi + 1);
                if (lo >= hi) continue;
                Run vr;
                vr.base_off = run.base_off + (lo - run.kidx0);
                vr.n_kmers = (uint32_t)(hi - lo);
                vr.contig = (uint32_t)v;
                vr.kidx0 = (uint32_t)(lo - klo);
                vr.pos0 = run.pos0 + (uint32_t)(lo - run.kidx0);
                vruns.push_back(vr);
            }
        }
        v_run0.push_back((uint32_t)vruns.size());
        std::vector<uint32_t> vstrip0;
        std::vector<uint64_t> vg0(vruns.size() + 1);
        bool ovf = false;
        build_strip_tables(vruns, S_DENSE, vstrip0, &ovf);
        if (ovf) return set_err(h, MXG_ELIMIT, "too many strips in gap fix-up");
        uint64_t g = 0;
        for (size_t r = 0; r < vruns.size(); ++r) {
            vg0[r] = g;
            g += vruns[r].n_kmers;
        }
        vg0[vruns.size()] = g;
        int rc;
        if ((rc = upload(h, sc(SC_V_RUNS), vruns, st)) != MXG_OK) return rc;
        if ((rc = upload(h, sc(SC_V_STRIP0), vstrip0, st)) != MXG_OK) return rc;
        if ((rc = upload(h, sc(SC_V_G0), vg0, st)) != MXG_OK) return rc;
        if ((rc = upload(h, sc(SC_V_NK), v_nk, st)) != MXG_OK) return rc;
        if ((rc = upload(h, sc(SC_V_REC), v_rec, st)) != MXG_OK) return rc;
        if ((rc = upload(h, sc(SC_V_RUN0), v_run0, st)) != MXG_OK) return rc;
        if (a->any_drop && (rc = upload(h, sc(SC_V_DROP), v_drop, st)) != MXG_OK) return rc;
        Tables V;
        V.runs = &vruns;
        V.ctg_nk = &v_nk;
        V.ctg_rec = &v_rec;
        V.ctg_run0 = &v_run0;
        V.strip0_dense = &vstrip0;
        V.strip0_sparse = nullptr;
        V.g0 = &vg0;
        V.d_runs = sc(SC_V_RUNS).as<Run>();
        V.d_strip0_dense = sc(SC_V_STRIP0).as<uint32_t>();
        V.d_strip0_sparse = nullptr;
        V.d_ctg_nk = sc(SC_V_NK).as<uint32_t>();
        V.d_ctg_rec = sc(SC_V_REC).as<uint32_t>();
        V.d_ctg_run0 = sc(SC_V_RUN0).as<uint32_t>();
        V.d_g0 = sc(SC_V_G0).as<uint64_t>();
        V.d_ctg_drop = a->any_drop ? sc(SC_V_DROP).as<uint8_t>() : nullptr;
        V.recs = nullptr;
        OutArrays og{&sc(SC_G_HASH), &sc(SC_G_POS), &sc(SC_G_REC), &sc(SC_G_FWD), 0};
        MXG_HIP(h, og.hash->ensure(1024 * 8));
        MXG_HIP(h, og.pos->ensure(1024 * 4));
        MXG_HIP(h, og.rec->ensure(1024 * 4));
        MXG_HIP(h, og.fwd->ensure(1024));
        if ((rc = dense_all(a->d_packed, V, og, false)) != MXG_OK) return rc;
        *n_gap_mx = og.n;
        return MXG_OK;
    }

    // ---- sparse pipeline -------------------------------------------------------------------------------------
    // control block (u32 words): [0] max wave count if a wave overflowed its arena slice, [1] gap count,
    // [2..3] number of selected candidates (u64), [4..5] number of candidates (u64)
    struct BatchGeom {
        size_t c0, c1;
        uint64_t nk;
        uint32_t r_lo, r_hi, strip_lo, strip_hi, n_strips, n_blocks, n_waves;
    };
    void batch_geom(const Tables &T, size_t c0, BatchGeom &g, uint64_t budget = 0) const
    {
        const size_t n_ctg = T.ctg_rec->size();
        if (!budget) budget = SPARSE_BATCH_KMERS;
        g.c0 = c0;
        g.c1 = c0;
        g.nk = 0;
        while (g.c1 < n_ctg && (g.c1 == c0 || g.nk + (*T.ctg_nk)[g.c1] <= budget)) g.nk += (*T.ctg_nk)[g.c1++];
        g.r_lo = (*T.ctg_run0)[g.c0];
        g.r_hi = (*T.ctg_run0)[g.c1];
        g.strip_lo = (*T.strip0_sparse)[g.r_lo];
        g.strip_hi = (*T.strip0_sparse)[g.r_hi];
        g.n_strips = g.strip_hi - g.strip_lo;
        g.n_blocks = (g.n_strips + 255) / 256;
        g.n_waves = g.n_blocks * 4;
    }
    uint64_t default_wave_cap(uint32_t S, double cand_frac) const
    {
        // every wave owns a slice of the arena: twice the expected candidates of 64 strips plus 6 sigma
        const double expect = 64.0 * S * cand_frac;
        uint64_t wave_cap = (uint64_t)(2.0 * expect + 6.0 * std::sqrt(expect)) + 64;
        wave_cap = std::max<uint64_t>(wave_cap, h->arena_cap_hint);
        wave_cap = std::min<uint64_t>(wave_cap, 64ull * S);  // a wave can never produce more
        if (h->arena_cap_hint == 0) wave_cap = env_u64(h, "MXG_WAVE_CAP", wave_cap);  // test knob
        return wave_cap;
    }
    // Enqueue one batch completely (hash -> order -> resolve+count -> speculative emit at out.n); the last kernel writes
    // the control block to `ctrl_host` (PINNED host memory); NO host sync.  *n_cap_out = capacity the candidate arrays were sized for.
    // grid of the hash kernel: one block per tile of 256 strips (a few persistent blocks per CU with an equal number of tiles each
    // were an experiment of round 2, measured and dropped)
    uint32_t sparse_grid(uint32_t n_tiles, uint64_t) const { return n_tiles; }
    // stretches sketched on the device, one block each (results wait in their regions for k_gap_post): reads SC_GAPS / ctrl[1]
    int enqueue_dev_gaps(Assembly *a, const Tables &T, const uint32_t *ctrl_host)
    {
        MXG_HIP(h, sc(SC_GR_HASH).ensure((size_t)GAP_DEV_POOL * 8));
        MXG_HIP(h, sc(SC_GR_POS).ensure((size_t)GAP_DEV_POOL * 4));
        MXG_HIP(h, sc(SC_GR_REC).ensure((size_t)GAP_DEV_POOL * 4));
        MXG_HIP(h, sc(SC_GR_CNT).ensure((size_t)gcap * 8));  // counts, then the regions' starts
        MXG_HIP(h, sc(SC_GR_KEY).ensure((size_t)gcap * 8));
        MXG_HIP(h, sc(SC_GD_HASH).ensure((size_t)(gcap + 1) * 8));  // the stretches in order: key,
        MXG_HIP(h, sc(SC_GD_POS).ensure((size_t)(gcap + 1) * 4));   // minimizers before,
        MXG_HIP(h, sc(SC_GD_REC).ensure((size_t)(gcap + 1) * 4));   // region
        GapFixParams gp;
        gp.gaps = sc(SC_GAPS).as<uint4>();
        gp.ctrl = sc(SC_CTRL).as<uint32_t>();
        gp.runs = T.d_runs;
        gp.ctg_run0 = T.d_ctg_run0;
        gp.ctg_rec = T.d_ctg_rec;
        gp.ctg_drop = T.d_ctg_drop;
        gp.packed = a->d_packed;
        gp.init_tab = h->d_init_tab.as<uint4>();
        gp.k = h->cfg.k;
        gp.w = h->cfg.w;
        gp.mult = 1ull ^ ((uint64_t)h->cfg.k * 0x90b45d39fb6da1faull);
        gp.r_hash = sc(SC_GR_HASH).as<uint64_t>();
        gp.r_pos = sc(SC_GR_POS).as<uint32_t>();
        gp.r_rec = sc(SC_GR_REC).as<uint32_t>();
        gp.r_cnt = sc(SC_GR_CNT).as<uint32_t>();
        gp.r_start = gp.r_cnt + gcap;
        gp.gcap = gcap;
        gp.pool = (uint32_t)std::min<uint64_t>(env_u64(h, "MXG_GAP_POOL", GAP_DEV_POOL), GAP_DEV_POOL);
        gp.r_key = sc(SC_GR_KEY).as<uint64_t>();
        // the batch's slice of the pinned list of deferred stretches goes with its pinned control block
        gp.defer = reinterpret_cast<uint4 *>(h->pinned_defer) + (size_t)((ctrl_host - h->pinned_ctrl) / 16) * GAP_DEFER_MAX;
        gp.tab = h->tab;
        // (no more blocks than the launch expects stretches -- what k_emit has placing blocks for: a block of 51 KB that finds
        // nothing to do still has to be brought to a CU, 22 us for 768 of them behind k_sel_stretch, which leaves few stretches over)
        const uint32_t gf_blocks = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(std::max<uint64_t>(env_u64(h, "MXG_GAP_FIX_BLOCKS", GAP_FIX_BLOCKS), 1), gcap),
                                                                gap_expect >= 0.0 ? std::max<uint64_t>(32, (uint64_t)(2.0 * gap_expect) + 16) : 4ull * n_place);
        if (h->cfg.variant == MXG_VARIANT_V1_MIN)
            hipLaunchKernelGGL(k_gap_fix<MXG_VARIANT_V1_MIN>, dim3(gf_blocks), dim3(256), 0, st, gp);
        else
            hipLaunchKernelGGL(k_gap_fix<MXG_VARIANT_V2_SUM>, dim3(gf_blocks), dim3(256), 0, st, gp);
        GapPostParams pp;
        pp.ctrl = sc(SC_CTRL).as<uint32_t>();
        pp.r_cnt = sc(SC_GR_CNT).as<uint32_t>();
        pp.r_key = sc(SC_GR_KEY).as<uint64_t>();
        pp.s_key = sc(SC_GD_HASH).as<uint64_t>();
        pp.s_off = sc(SC_GD_POS).as<uint32_t>();
        pp.s_src = sc(SC_GD_REC).as<uint32_t>();
        pp.gcap = gcap;
        // (blocks beyond the stretches there are leave at once: the grid follows what the launch may meet, like k_emit's placing blocks)
        hipLaunchKernelGGL(k_gap_post, dim3(std::max(1u, (std::min(gcap, 4u * n_place) + GP_PER - 1u) / GP_PER)), dim3(GPB), 0, st, pp);  // (rounded UP: 4 * n_place is any number once the placing blocks follow the density met)
        MXG_HIP(h, hipGetLastError());
        return MXG_OK;
    }

    int enqueue_sparse(Assembly *a, const Tables &T, const BatchGeom &g, uint64_t wave_cap, uint32_t tau_hi,
                       OutArrays &out, uint32_t *ctrl_host, uint32_t *n_cap_out, const ChainIO *io = nullptr,
                       uint32_t cand_hint = 0xFFFFFFFFu, const uint32_t *bs_bitmap = nullptr)
    {
        // MXG_TIMING_SAMPLE=n: event pairs around one batch in n only (four event records per batch cost 2 % of the step at
        // 3 Gbp; bench.py asks for one in four and scales the sums by the bases they cover)
        struct TimingGuard {
            bool &t; bool saved;
            TimingGuard(bool &t_, bool now) : t(t_), saved(t_) { t = now; }
            ~TimingGuard() { t = saved; }
        };
        const uint64_t t_sample = timing && !fine && io ? env_u64(h, "MXG_TIMING_SAMPLE", 1) : 1;
        TimingGuard timing_guard(timing, timing && (t_sample <= 1 || (h->timing_batches++ % t_sample) == 0));
        const uint32_t S = a->S_sparse;
        few_cand = (double)tau_hi / 4294967296.0 * (double)h->cfg.w <= 12.5;
        MXG_HIP(h, sc(SC_GAPS).ensure((size_t)GAP_CAP * 16));
        MXG_HIP(h, sc(SC_STRIP_CNT).ensure((size_t)g.n_strips * 4 + 16));
        MXG_HIP(h, sc(SC_STRIP_META).ensure((size_t)g.n_strips * 4 + 16));
        MXG_HIP(h, sc(SC_WAVE_CNT).ensure((size_t)g.n_waves * 4 + 16));
        MXG_HIP(h, sc(SC_WAVE_TOT).ensure((size_t)g.n_waves * 4 + 16));
        const uint64_t n_cap64 = (uint64_t)g.n_waves * wave_cap;
        if (n_cap64 >= (1ull << 32))
            return set_err(h, MXG_ELIMIT, "candidate arena would exceed 2^32 entries; use MXG_FLAG_DENSE_ONLY");
        const uint32_t n_cap = (uint32_t)n_cap64;
        *n_cap_out = n_cap;
        MXG_HIP(h, sc(SC_CAND_H).ensure((size_t)n_cap * 8));
        MXG_HIP(h, sc(SC_CAND_K).ensure((size_t)n_cap * 4));
        MXG_HIP(h, sc(SC_CAND_C).ensure((size_t)n_cap * 4));
        n_wave_sup = sup_words(g.n_waves);
        const size_t ctrl_bytes = ((size_t)CTRL_WORDS + n_wave_sup + n_sel_sup(n_cap)) * 4;
        MXG_HIP(h, sc(SC_CTRL).ensure(ctrl_bytes));
        MXG_HIP(h, hipMemsetAsync(sc(SC_CTRL).p, 0, ctrl_bytes, st));  // control block + both super-count arrays
        SparseParams sp;
        sp.packed = a->d_packed;
        sp.runs = T.d_runs;
        sp.run_strip0 = T.d_strip0_sparse;
        sp.run_lo = g.r_lo;
        sp.run_hi = g.r_hi;
        sp.strip_lo = g.strip_lo;
        sp.strip_hi = g.strip_hi;
        sp.k = h->cfg.k;
        sp.S = S;
        sp.five = env_u64(h, "MXG_HASH_FIVE", 1) != 0 ? 1u : 0u;
        // MXG_RING_SLACK=<percent> (test knob): the ring filter captures up to that many percent more k-mers than have
        // hash < tau, i.e. entries that k_reorder finds to be >= tau and k_resolve must treat as absent.  On real runs
        // such entries occur about once per 10^9 k-mers, so the tests force them.
        const uint32_t ring_slack = (uint32_t)env_u64(h, "MXG_RING_SLACK", 0);  // read per call, like the other knobs
        sp.tau_hi = ring_slack ? (uint32_t)std::min<uint64_t>(0x7FFFFFFEull, (uint64_t)tau_hi * (100 + ring_slack) / 100) & ~1u
                               : tau_hi;
        sp.wave_cap = (uint32_t)wave_cap;
        sp.wave_cnt = sc(SC_WAVE_CNT).as<uint32_t>();
        sp.ctrl = sc(SC_CTRL).as<uint32_t>();
        sp.strip_cnt = sc(SC_STRIP_CNT).as<uint32_t>();
        sp.strip_meta = sc(SC_STRIP_META).as<uint32_t>();
        sp.wave_tot = sc(SC_WAVE_TOT).as<uint32_t>();
        sp.wave_sup = wave_sup();
        // reorder geometry (needed first: the k = 32 route's reorder kernel exists in the one-wave-per-slice form only)
        const uint32_t queue_cap = sp.wave_cap <= 8192 ? sp.wave_cap : 0;
        const size_t q_lds = (size_t)queue_cap * 4;
        // slices per block of k_reorder: one (two to four, the block's byte table loaded once for all of them, measured slower in round 2)
        const uint32_t r_g = 1;
        const uint32_t r_grid = (g.n_waves + r_g - 1) / r_g;
        // one wave per slice + position tables when the queues of a block fit beside the 32 KB of tables
        const size_t w_lds = (size_t)2048 * 16 + (size_t)RW_WAVES * queue_cap * 4;
        const bool r_wave = queue_cap && w_lds + 512 <= 65536 && env_u64(h, "MXG_REORDER_W", 1) != 0;
        const uint32_t w_grid = std::min<uint32_t>((g.n_waves + RW_WAVES - 1) / RW_WAVES,
                                                   (uint32_t)env_u64(h, "MXG_REORDER_W_GRID", 256 * 3));
        if (!r_wave || !T.d_strip_run) bs_bitmap = nullptr;  // (a batch whose slices outgrow the LDS queues takes the rolling-hash kernel)
        if (!bs_bitmap) MXG_HIP(h, sc(SC_ARENA).ensure((size_t)n_cap * 8));
        sp.arena = sc(SC_ARENA).as<uint2>();
        int rc;
        sp.init_tab = h->d_init_tab.as<uint4>();
        sp.strip_run = T.d_strip_run;
        sp.tab = h->tab;
        if ((rc = bs_bitmap ? ev_begin(0, false) : ev_begin(batch_bases(T, g.c0, g.c1), true)) != MXG_OK) return rc;
        sp.n_tiles = g.n_blocks;
        dim3 grid(sparse_grid(g.n_blocks, g.nk)), block(256);
        static const int abl = getenv("MXG_ABLATE") ? atoi(getenv("MXG_ABLATE")) : 0;  // profiling only
        // Every lane keeps one 128-byte line of packed bases "open" for 32 block iterations (16 bases = 4 bytes per iteration),
        // so the waves resident on an XCD hold (waves x 64 x 128 B) of live lines.  At full occupancy that is more than the
        // XCD's 4 MB of L2 once the input no longer fits the caches behind it: measured at 3 Gbp (PMC FETCH_SIZE x 2), the
        // kernel fetched 336 MB for 115 MB of bases.  Dynamic LDS caps the residency at six blocks per CU (18 KB; five with
        // 24 KB: 1285-1308 Gbp/s against 1304-1331; none: 1257).  It also keeps the other stream's kernels from moving in
        // beside this one, which costs both more than it gains.  Small inputs (cache-resident) keep full occupancy.  At
        // k = 32 the first 16 KB of it hold init32_half's tables.
        size_t pad = (size_t)env_u64(h, "MXG_HASH_LDS", g.nk >= (256ull << 20) ? 18000 : 0);
        if (sp.five && sp.k == 32) pad = std::max<size_t>(pad, 16384);
        if (bs_bitmap)  // (k = 32 route: the filter has run over the whole assembly; only its bits are turned into entries)
            hipLaunchKernelGGL(k_bs_count, dim3(g.n_blocks), block, 0, st, sp, bs_bitmap);
        else if (h->cfg.variant == MXG_VARIANT_V1_MIN)
            hipLaunchKernelGGL((k_hash_sparse<MXG_VARIANT_V1_MIN>), grid, block, pad, st, sp);
        else if (abl == 1)
            hipLaunchKernelGGL((k_hash_sparse<MXG_VARIANT_V2_SUM, 1>), grid, block, pad, st, sp);
        else if (abl == 2)
            hipLaunchKernelGGL((k_hash_sparse<MXG_VARIANT_V2_SUM, 2>), grid, block, pad, st, sp);
        else
            hipLaunchKernelGGL((k_hash_sparse<MXG_VARIANT_V2_SUM>), grid, block, pad, st, sp);
        if ((rc = ev_end()) != MXG_OK) return rc;
        MXG_HIP(h, hipGetLastError());
        // order the candidates: exclusive scan of per-strip counts (total = number of candidates), then scatter
        if ((rc = ev_begin(0, false)) != MXG_OK) return rc;
        ReorderParams op;
        op.arena = sp.arena;
        op.wave_cnt = sp.wave_cnt;
        op.wave_cap = sp.wave_cap;
        op.n_cap = n_cap;
        op.n_waves = g.n_waves;
        op.strip_cnt = sp.strip_cnt;
        op.n_strips = g.n_strips;
        op.wave_tot = sp.wave_tot;
        op.wave_sup = sp.wave_sup;
        op.n_cand = sp.ctrl + 4;
        op.strip_meta = bs_bitmap ? T.d_strip_run : sp.strip_meta;
        op.runs = sp.runs;
        op.run_strip0 = sp.run_strip0;
        op.strip_lo = sp.strip_lo;
        op.S = sp.S;
        op.packed = sp.packed;
        op.init_tab = sp.init_tab;
        op.k = sp.k;
        op.ch = sc(SC_CAND_H).as<uint64_t>();
        op.ck = sc(SC_CAND_K).as<uint32_t>();
        op.cc = sc(SC_CAND_C).as<uint32_t>();
        op.tab = h->tab;
        op.bm = bs_bitmap;
        op.queue_cap = queue_cap;
        if (bs_bitmap)  // (bs_possible: V2, and enqueue_sparse's caller has checked that the queues fit)
            hipLaunchKernelGGL(k_bs_reorder_w<MXG_VARIANT_V2_SUM>, dim3(w_grid), dim3(RW_WAVES * 64), w_lds, st, op);
        else if (r_wave && h->cfg.variant == MXG_VARIANT_V1_MIN)
            hipLaunchKernelGGL(k_reorder_w<MXG_VARIANT_V1_MIN>, dim3(w_grid), dim3(RW_WAVES * 64), w_lds, st, op);
        else if (r_wave && !knob_set(h, "MXG_ABLATE_REORDER"))
            hipLaunchKernelGGL(k_reorder_w<MXG_VARIANT_V2_SUM>, dim3(w_grid), dim3(RW_WAVES * 64), w_lds, st, op);
        else if (h->cfg.variant == MXG_VARIANT_V1_MIN)
            hipLaunchKernelGGL(k_reorder<MXG_VARIANT_V1_MIN>, dim3(r_grid), dim3(RB), q_lds, st, op);
        else
        {
            static const int rabl = getenv("MXG_ABLATE_REORDER") ? atoi(getenv("MXG_ABLATE_REORDER")) : 0;  // profiling only
            if (rabl == 1)
                hipLaunchKernelGGL((k_reorder<MXG_VARIANT_V2_SUM, 1>), dim3(r_grid), dim3(RB), q_lds, st, op);
            else if (env_u64(h, "MXG_RB", few_cand ? 256 : RB) == 256)  // ~205 candidates per slice at 10 per window: one pass of 256 threads
                hipLaunchKernelGGL((k_reorder<MXG_VARIANT_V2_SUM, 0, 256>), dim3(r_grid), dim3(256), q_lds, st, op);
            else
                hipLaunchKernelGGL(k_reorder<MXG_VARIANT_V2_SUM>, dim3(r_grid), dim3(RB), q_lds, st, op);
        }
        MXG_HIP(h, hipGetLastError());
        // resolve + speculative emit straight into the output arrays (guarded by their capacity): on the common path
        // (no gap, no overflow) the batch then needs a single host sync
        if ((rc = ev_next(3)) != MXG_OK) return rc;
        grid_cand = 0;
        if (const uint64_t by_est = io ? env_u64(h, "MXG_GRID_BY_ESTIMATE", 1) : 0) {
            // (min(fwd, rev) < tau: either strand may pass)
            const uint64_t expect = (uint64_t)((double)g.nk * (double)sp.tau_hi / 4294967296.0) *
                                    (h->cfg.variant == MXG_VARIANT_V1_MIN ? 2u : 1u);
            const uint64_t hint = cand_hint != 0xFFFFFFFFu ? cand_hint : 0;
            uint64_t gc = std::max(expect, hint) * 13 / 10 + 16384;
            if (by_est == 2) gc = std::max<uint64_t>(RK, expect / 2);  // (test knob: too small on purpose)
            // a whole number of k_emit tiles, so that k_resolve (256 candidates per block) and k_emit cover the same candidates:
            // a tile that reports must have had all its blocks resolved
            constexpr uint64_t ET = (uint64_t)EMIT_COMPACT_BLOCKS * RK;
            grid_cand = (uint32_t)std::min<uint64_t>(n_cap, (gc + ET - 1) / ET * ET);
        }
        if ((rc = resolve_count(T, n_cap, (uint32_t)g.c0, (uint32_t)g.c1, (uint64_t)tau_hi << 32,
                                cand_hint != 0xFFFFFFFFu ? cand_hint : a->cand_hint)) != MXG_OK) return rc;
        const bool dev = io && io->dev_gaps;
        if (dev && (rc = enqueue_dev_gaps(a, T, ctrl_host)) != MXG_OK) return rc;
        if ((rc = ev_next(4)) != MXG_OK) return rc;
        // the batch before this one (same assembly, other stream) must have passed its count on
        if (io && io->wait) MXG_HIP(h, hipStreamWaitEvent(st, io->wait, 0));
        rc = emit(a->d_packed, T, n_cap, *out.hash, *out.pos, *out.rec, *out.fwd, out.n, true, ctrl_host, io);
        grid_cand = 0;
        if (rc != MXG_OK) return rc;
        return ev_end();
    }

    // ---- k = 32 route (sketch_bs.hip): the assembly's filter bitmap is in a->d_bs_out (bs_hash, enqueued by the caller on
    // some stream this one has been made to wait for); one batch = k_bs_select over the batch's strips -> stretches -> emit.
    BsSelGeom sel_geom(const Assembly *a, const BatchGeom &g, double frac) const
    {
        return bs_select_geom(a->S_sparse, a->sel_H, h->cfg.w, frac, g.n_strips, (uint32_t)env_u64(h, "MXG_SEL_QCAP", 0), (uint32_t)env_u64(h, "MXG_SEL_RK", 0));
    }
    // The slice kernel's control block cleared ahead of the filter that precedes it on this stream (the filter does not touch it):
    // filter and slice kernel then follow one another without a fill between them (two idle spells of ~6 us per assembly).
    bool ctrl_cleared = false;
    int clear_sel_ctrl(const BsSelGeom &b)
    {
        int rc;
        if ((rc = flush_emit(nullptr)) != MXG_OK) return rc;  // (this driver's scratch is about to be reused)
        const size_t ctrl_bytes = ((size_t)CTRL_WORDS + sup_words(b.n_slices) + 2 * 64 * 32) * 4;
        MXG_HIP(h, sc(SC_CTRL).ensure(ctrl_bytes));
        MXG_HIP(h, hipMemsetAsync(sc(SC_CTRL).p, 0, ctrl_bytes, st));
        ctrl_cleared = true;
        return MXG_OK;
    }
    int enqueue_sel(Assembly *a, const Tables &T, const BatchGeom &g, const BsSelGeom &b, uint32_t tau_hi, OutArrays &out,
                    uint32_t *ctrl_host, const ChainIO *io)
    {
        int rc;
        if ((rc = flush_emit(nullptr)) != MXG_OK) return rc;  // (this driver's scratch is about to be reused)
        MXG_HIP(h, sc(SC_GAPS).ensure((size_t)GAP_CAP * 16));
        n_wave_sup = 0;
        const size_t ctrl_bytes = ((size_t)CTRL_WORDS + sup_words(b.n_slices) + 2 * 64 * 32) * 4;  // + the candidate counters
        MXG_HIP(h, sc(SC_CTRL).ensure(ctrl_bytes));
        MXG_HIP(h, sc(SC_CNT256).ensure((size_t)b.n_slices * 4 + 64));
        const size_t n_ent = (size_t)b.n_slices * b.rk;
        MXG_HIP(h, sc(SC_CS_H).ensure(n_ent * 16));  // (one 16-byte entry per selected candidate: EmitParams::cs_aos)
        // regions for the slices that outgrow their LDS queue (SC_CAND_H / SC_CAND_K: this route has no candidate arrays)
        const size_t ovf_ent = (size_t)b.n_ovf * (b.ovf_cap + 2 * SEL_PAD);
        MXG_HIP(h, sc(SC_CAND_H).ensure(ovf_ent * 8));
        MXG_HIP(h, sc(SC_CAND_K).ensure(ovf_ent * 4));
        if (!ctrl_cleared) MXG_HIP(h, hipMemsetAsync(sc(SC_CTRL).p, 0, ctrl_bytes, st));
        ctrl_cleared = false;
        // (fine timing: the slice kernel is booked where the other route books count + reorder, the stretch kernels where it books
        // resolve + stretches)
        // (the slice kernel has a span of its own in either timing mode: bench.py's roofline object times it inside the timed region)
        if ((rc = ev_begin(0, false, 2)) != MXG_OK) return rc;
        BsSelParams bp{};
        bp.bm = a->d_bs_out.as<uint32_t>() + 4;  // (BS_OUT_PAD)
        bp.packed = a->d_packed;
        bp.runx = a->d_runx.as<RunX>();
        bp.strip_run = T.d_strip_run;
        bp.ctg_drop = T.d_ctg_drop;
        bp.ptab = h->d_init_tab.as<uint4>() + 256;
        bp.n_strips_asm = (*T.strip0_sparse)[T.runs->size()];
        bp.strip_lo = g.strip_lo;
        bp.strip_hi = g.strip_hi;
        bp.S = a->S_sparse;
        bp.H = b.H;
        bp.T = b.T;
        bp.n_slices = b.n_slices;
        bp.w = h->cfg.w;
        bp.tau = (uint64_t)tau_hi << 32;
        bp.qcap = b.qcap;
        bp.ovf_h = sc(SC_CAND_H).as<uint64_t>();
        bp.ovf_e = sc(SC_CAND_K).as<uint32_t>();
        bp.ovf_cap = b.ovf_cap;
        bp.n_ovf = b.n_ovf;
        bp.ovf_next = sc(SC_CTRL).as<uint32_t>() + 12;
        bp.rk = b.rk;
        bp.cs = sc(SC_CS_H).as<uint4>();
        bp.cnt = sc(SC_CNT256).as<uint32_t>();
        bp.sup = sel_sup(0);
        bp.gaps = sc(SC_GAPS).as<uint4>();
        // (the upper part of the stretch array holds the requests for k_sel_stretch, two entries each)
        bp.gap_cap = GAP_CAP - SEL_IREQ_CAP;
        bp.ireq = sc(SC_GAPS).as<uint4>() + bp.gap_cap;
        bp.ireq_cap = SEL_IREQ_CAP - 8u;  // (k_sel_stretch reads eight entries from any request on)
        // (pieces only where the host's route for what k_gap_fix hands over knows them: the tile kernel)
        bp.gap_nmax = (io && io->dev_gaps && h->cfg.w <= ST_WMAX && !knob_set(h, "MXG_STRETCH_DENSE") && !knob_set(h, "MXG_GAP_WHOLE")) ? GAP_DEV_NMAX : 0u;
        bp.ctrl = sc(SC_CTRL).as<uint32_t>();
        bp.cand_spread = sel_sup(0) + sup_words(b.n_slices);
        bp.ablate = (uint32_t)env_u64(h, "MXG_SEL_ABLATE", 0);  // (profiling only: stop every slice after phase n)
        // the stretches between two candidates of a slice go to k_sel_stretch (MXG_SEL_INLINE=0: all stretches through k_gap_fix)
        bp.inl_amax = knob_u64(h, "MXG_SEL_INLINE", 1) && b.n_slices < (1u << 24) ? bs_select_inline_amax(h->cfg.w) : 0u;
        if ((rc = launch_bs_select(h, bp, b, st)) != MXG_OK) return rc;
        // (the next assembly's filter may start here; MXG_STAGGER=2, an experiment: behind this batch's emit instead, below)
        const bool late = knob_u64(h, "MXG_STAGGER", 1) == 2;
        if (h->ev_sel_done[slot] && !late) MXG_HIP(h, hipEventRecord(h->ev_sel_done[slot], st));
        h->stat_sel_slices += b.n_slices;
        const bool dev = io && io->dev_gaps;
        if (timing && !fine) {  // the rest of the batch (stretches, emit) as one span
            if ((rc = ev_end()) != MXG_OK || (rc = ev_begin(0, false, 1)) != MXG_OK) return rc;
        }
        if ((rc = ev_next(3)) != MXG_OK) return rc;
        // (booked with the stretch kernels: the slice kernel's span is its own)
        if (bp.inl_amax) {
            SelStretchParams sp{};
            sp.packed = bp.packed;
            sp.runs = T.d_runs;
            sp.ctg_run0 = T.d_ctg_run0;
            sp.ctg_drop = T.d_ctg_drop;
            sp.byte_tab = h->d_init_tab.as<uint4>();
            sp.tab = h->tab;
            sp.w = bp.w;
            sp.amax = bp.inl_amax;
            sp.rk = bp.rk;
            sp.cs = bp.cs;
            sp.cnt = bp.cnt;
            sp.sup = bp.sup;
            sp.ireq = bp.ireq;
            sp.ireq_cap = bp.ireq_cap;
            sp.ctrl = bp.ctrl;
            sp.tickets = bp.cand_spread + 64 * 32;
            sp.ablate = (uint32_t)env_u64(h, "MXG_SST_ABLATE", 0);
            sp.gaps = bp.gaps;
            sp.gap_cap = bp.gap_cap;
            sp.gap_nmax = bp.gap_nmax;
            if ((rc = launch_sel_stretch(h, sp, st)) != MXG_OK) return rc;
        }
        if (dev && (rc = enqueue_dev_gaps(a, T, ctrl_host)) != MXG_OK) return rc;
        if (defer_emit && io && !io->wait && !io->base_in && !io->base_out) {
            // the emit is held back (flush_emit): the caller enqueues it behind the NEXT assembly's slice kernel -- enqueued here it
            // would start when that assembly's filter lets go of the GPU and land on its slice kernel, whose blocks need whole CUs
            // (rocprofv3, round 5: 529 us for the target's launch against 450 for the reference's) -- so that it runs beside that
            // assembly's stretch kernels instead, which leave the GPU idle
            if ((rc = ev_end()) != MXG_OK) return rc;
            const Tables *Tp = &T;
            DevBuf *oh = out.hash, *op = out.pos, *orc = out.rec, *of = out.fwd;
            const uint64_t on = out.n;
            const ChainIO ioc = *io;
            const uint32_t rk = b.rk, gc = gcap, np = n_place;
            uint32_t *const no = n_out;
            const uint32_t *cs = bp.cand_spread;
            pending_emit = [=](hipEvent_t behind) -> int {
                int rc2;
                if (behind) MXG_HIP(h, hipStreamWaitEvent(st, behind, 0));
                if ((rc2 = ev_begin(0, false, fine ? 4 : 1)) != MXG_OK) return rc2;
                const uint32_t gc0 = gcap, np0 = n_place;
                uint32_t *const no0 = n_out;
                gcap = gc; n_place = np; n_out = no;
                rc2 = emit(a->d_packed, *Tp, (uint32_t)n_ent, *oh, *op, *orc, *of, on, true, ctrl_host, &ioc, rk, (uint32_t)n_ent, cs);
                gcap = gc0; n_place = np0; n_out = no0;
                if (rc2 != MXG_OK) return rc2;
                return ev_end();
            };
            return MXG_OK;
        }
        if ((rc = ev_next(4)) != MXG_OK) return rc;
        if (io && io->wait) MXG_HIP(h, hipStreamWaitEvent(st, io->wait, 0));
        rc = emit(a->d_packed, T, (uint32_t)n_ent, *out.hash, *out.pos, *out.rec, *out.fwd, out.n, true, ctrl_host, io, b.rk,
                  (uint32_t)n_ent, bp.cand_spread);
        if (rc != MXG_OK) return rc;
        if (h->ev_sel_done[slot] && late) MXG_HIP(h, hipEventRecord(h->ev_sel_done[slot], st));
        return ev_end();
    }
    // the emit enqueue_sel held back, now: behind `behind` (an event of another stream) if given
    std::function<int(hipEvent_t)> pending_emit;
    bool defer_emit = false;
    int flush_emit(hipEvent_t behind)
    {
        if (!pending_emit) return MXG_OK;
        auto f = std::move(pending_emit);
        pending_emit = nullptr;
        return f(behind);
    }

    // Second half of a sparse batch, after the host has read the control block `ctrl` of a run that did not overflow:
    // accept the speculative emit, or (candidate-free stretches) emit to staging, run the dense fix-up, merge.
    // The candidate arrays of that run must still be intact in this driver's scratch.
    int complete_batch(Assembly *a, const Tables &T, const BatchGeom &g, OutArrays &out, const uint32_t *ctrl, uint32_t n_cap)
    {
        const size_t c0 = g.c0, c1 = g.c1;
        const uint64_t n_cand = ctrl[4];  // (layout: FinParams)
        uint32_t n_gaps = ctrl[1];
        const uint64_t total = ctrl[2];
        h->stat_candidates += n_cand;
        a->cand_hint = (uint32_t)std::min<uint64_t>(n_cand, 0xFFFFFFFFull);
        int rc;
        std::vector<uint4> gaps;
        if (n_cand == 0) {  // no candidate at all: every contig of the batch is one stretch
            for (size_t c = c0; c < c1; ++c) gaps.push_back(make_uint4((uint32_t)c, 0, (*T.ctg_nk)[c] - 1, 0));
            n_gaps = (uint32_t)gaps.size();
        } else if (n_gaps > GAP_CAP - SEL_IREQ_CAP) {  // (behind k_bs_select the array's upper part holds k_sel_stretch's requests)
            return set_err(h, MXG_ELIMIT, "more than %u candidate-free stretches in one batch; rerun with MXG_FLAG_DENSE_ONLY", GAP_CAP - SEL_IREQ_CAP);
        } else if (n_gaps) {
            gaps.resize(n_gaps);
            MXG_HIP(h, hipMemcpyAsync(gaps.data(), sc(SC_GAPS).p, (size_t)n_gaps * 16, hipMemcpyDeviceToHost, st));
            MXG_HIP(h, hipStreamSynchronize(st));
        }
        if (n_gaps == 0) {
            if (out.n + total > out.cap()) {  // the speculative emit did not fit: grow, emit again
                if ((rc = out_reserve(h, out, out.n + total, st)) != MXG_OK) return rc;
                if ((rc = emit(a->d_packed, T, n_cap, *out.hash, *out.pos, *out.rec, *out.fwd, out.n, true)) != MXG_OK) return rc;
            }
            out.n += total;
        } else {
            // main result to staging (before the candidate scratch is reused by the gap pass)
            MXG_HIP(h, sc(SC_ST_HASH).ensure(std::max<uint64_t>(total * 8, 16)));
            MXG_HIP(h, sc(SC_ST_POS).ensure(std::max<uint64_t>(total * 4, 16)));
            MXG_HIP(h, sc(SC_ST_REC).ensure(std::max<uint64_t>(total * 4, 16)));
            MXG_HIP(h, sc(SC_ST_FWD).ensure(std::max<uint64_t>(total, 16)));
            if ((rc = emit(a->d_packed, T, n_cap, sc(SC_ST_HASH), sc(SC_ST_POS), sc(SC_ST_REC), sc(SC_ST_FWD), 0, true)) != MXG_OK) return rc;
            uint64_t n_gap_mx = 0;
            if ((rc = process_gaps(a, T, gaps, &n_gap_mx)) != MXG_OK) return rc;
            if (total + n_gap_mx >= (1ull << 32)) return set_err(h, MXG_ELIMIT, "batch sketch too large to merge");
            if ((rc = out_reserve(h, out, out.n + total + n_gap_mx, st)) != MXG_OK) return rc;
            MergeParams mp;
            mp.a_hash = sc(SC_ST_HASH).as<uint64_t>(); mp.a_pos = sc(SC_ST_POS).as<uint32_t>();
            mp.a_rec = sc(SC_ST_REC).as<uint32_t>(); mp.a_fwd = sc(SC_ST_FWD).as<uint8_t>(); mp.nA = (uint32_t)total;
            mp.b_hash = sc(SC_G_HASH).as<uint64_t>(); mp.b_pos = sc(SC_G_POS).as<uint32_t>();
            mp.b_rec = sc(SC_G_REC).as<uint32_t>(); mp.b_fwd = sc(SC_G_FWD).as<uint8_t>(); mp.nB = (uint32_t)n_gap_mx;
            mp.o_hash = out.hash->as<uint64_t>() + out.n; mp.o_pos = out.pos->as<uint32_t>() + out.n;
            mp.o_rec = out.rec->as<uint32_t>() + out.n; mp.o_fwd = out.fwd->as<uint8_t>() + out.n;
            const uint32_t nt = mp.nA + mp.nB;
            if (nt) hipLaunchKernelGGL(k_merge, dim3((nt + 255) / 256), dim3(256), 0, st, mp);
            MXG_HIP(h, hipGetLastError());
            out.n += total + n_gap_mx;
        }
        return MXG_OK;
    }

    // stretches of any shape through k_stretch_tiles -> SC_G_* (hash, pos, rec) in (contig, position) order
    int sketch_stretches(Assembly *a, const Tables &T, std::vector<uint4> &gaps, uint64_t *n_gap_mx)
    {
        std::sort(gaps.begin(), gaps.end(), [](const uint4 &x, const uint4 &y) { return x.x != y.x ? x.x < y.x : x.y < y.y; });
        const uint32_t w = h->cfg.w;
        std::vector<StretchTile> tiles;
        for (const uint4 &g : gaps) {
            const uint32_t n = g.z - g.y + 1u;
            if (n < w) continue;  // (no window inside)
            const uint32_t n_win = n - w + 1u;
            // (a piece of a long stretch whose first window is the piece before's last: its tiles start at the second window, and the
            // first tile leaves out the arg-min of the window in front of it like every tile behind another)
            for (uint32_t t0 = (g.w & SEL_GAP_DROP) ? 1u : 0u; t0 < n_win; t0 += ST_WIN)
                tiles.push_back(StretchTile{g.x, g.y, t0, std::min(ST_WIN, n_win - t0)});
        }
        *n_gap_mx = 0;
        if (tiles.empty()) return MXG_OK;
        if (tiles.size() >= (1ull << 31)) return set_err(h, MXG_ELIMIT, "too many stretch tiles");
        int rc;
        const uint32_t nt = (uint32_t)tiles.size();
        if ((rc = upload(h, sc(SC_V_RUNS), tiles, st)) != MXG_OK) return rc;
        MXG_HIP(h, sc(SC_V_NK).ensure((size_t)nt * 4));
        MXG_HIP(h, sc(SC_V_REC).ensure((size_t)nt * 4));
        StretchParams sp;
        sp.tiles = sc(SC_V_RUNS).as<StretchTile>();
        sp.n_tiles = nt;
        sp.runs = T.d_runs;
        sp.ctg_run0 = T.d_ctg_run0;
        sp.ctg_rec = T.d_ctg_rec;
        sp.ctg_drop = T.d_ctg_drop;
        sp.packed = a->d_packed;
        sp.k = h->cfg.k;
        sp.w = w;
        sp.mult = 1ull ^ ((uint64_t)h->cfg.k * 0x90b45d39fb6da1faull);
        sp.tab = h->tab;
        sp.cnt = sc(SC_V_NK).as<uint32_t>();
        sp.off = nullptr;
        sp.o_hash = nullptr;
        sp.o_pos = sp.o_rec = nullptr;
        const bool v1 = h->cfg.variant == MXG_VARIANT_V1_MIN;
        if (v1) hipLaunchKernelGGL((k_stretch_tiles<MXG_VARIANT_V1_MIN, 0>), dim3(nt), dim3(256), 0, st, sp);
        else hipLaunchKernelGGL((k_stretch_tiles<MXG_VARIANT_V2_SUM, 0>), dim3(nt), dim3(256), 0, st, sp);
        MXG_HIP(h, hipGetLastError());
        std::vector<uint32_t> cnt(nt), off(nt);
        MXG_HIP(h, hipMemcpyAsync(cnt.data(), sp.cnt, (size_t)nt * 4, hipMemcpyDeviceToHost, st));
        MXG_HIP(h, hipStreamSynchronize(st));
        uint64_t total = 0;
        for (uint32_t t = 0; t < nt; ++t) {
            off[t] = (uint32_t)total;
            total += cnt[t];
        }
        if (total >= (1ull << 32)) return set_err(h, MXG_ELIMIT, "too many minimizers in candidate-free stretches");
        *n_gap_mx = total;
        if (!total) return MXG_OK;
        MXG_HIP(h, sc(SC_G_HASH).ensure(total * 8));
        MXG_HIP(h, sc(SC_G_POS).ensure(total * 4));
        MXG_HIP(h, sc(SC_G_REC).ensure(total * 4));
        MXG_HIP(h, sc(SC_G_FWD).ensure(total));
        MXG_HIP(h, hipMemsetAsync(sc(SC_G_FWD).p, 0, total, st));  // (strands are computed on demand: ensure_strand)
        MXG_HIP(h, hipMemcpyAsync(sc(SC_V_REC).p, off.data(), (size_t)nt * 4, hipMemcpyHostToDevice, st));
        sp.off = sc(SC_V_REC).as<uint32_t>();
        sp.o_hash = sc(SC_G_HASH).as<uint64_t>();
        sp.o_pos = sc(SC_G_POS).as<uint32_t>();
        sp.o_rec = sc(SC_G_REC).as<uint32_t>();
        if (v1) hipLaunchKernelGGL((k_stretch_tiles<MXG_VARIANT_V1_MIN, 1>), dim3(nt), dim3(256), 0, st, sp);
        else hipLaunchKernelGGL((k_stretch_tiles<MXG_VARIANT_V2_SUM, 1>), dim3(nt), dim3(256), 0, st, sp);
        MXG_HIP(h, hipGetLastError());
        MXG_HIP(h, hipStreamSynchronize(st));  // (off / tiles are read by the kernel until here)
        return MXG_OK;
    }

    // The stretches the device route left out (defer_stretch), for a whole assembly whose batches are all in place: sketched
    // as stand-alone contigs (k_stretch_tiles; beyond its window limit the dense pipeline), then merged into the assembly's n
    // minimizers (one pass over the sketch).
    int merge_deferred(Assembly *a, const Tables &T, std::vector<uint4> &gaps, uint64_t n, uint64_t *n_out)
    {
        int rc;
        uint64_t n_gap_mx = 0;
        *n_out = n;
        if (knob_set(h, "MXG_DEBUG_BATCH")) {  // (diagnostics: what the device route handed over, by length)
            uint32_t hist[33] = {0};
            for (const uint4 &g : gaps) {
                uint32_t len = g.z - g.y + 1u, b = 0;
                while ((2u << b) <= len) ++b;
                ++hist[b];
            }
            fprintf(stderr, "[mxg] %s: %zu stretches handed over, by length:", a->name.c_str(), gaps.size());
            for (uint32_t b = 0; b < 33; ++b)
                if (hist[b]) fprintf(stderr, " 2^%u: %u", b, hist[b]);
            for (size_t q = 0; q < gaps.size() && q < 6; ++q)
                fprintf(stderr, " | contig %u k-mers %u..%u", gaps[q].x, gaps[q].y, gaps[q].z);
            fprintf(stderr, "\n");
        }
        if (h->cfg.w <= ST_WMAX && !knob_set(h, "MXG_STRETCH_DENSE")) {
            if ((rc = sketch_stretches(a, T, gaps, &n_gap_mx)) != MXG_OK) return rc;
            for (const uint4 &g : gaps) h->stat_dense_kmers += g.z - g.y + 1u;
        } else if ((rc = process_gaps(a, T, gaps, &n_gap_mx)) != MXG_OK) {  // (dense_all counts its k-mers itself)
            return rc;
        }
        if (!n_gap_mx) return MXG_OK;
        if (n + n_gap_mx >= (1ull << 32)) return set_err(h, MXG_ELIMIT, "sketch too large to merge");
        OutArrays out{&a->d_hash, &a->d_pos, &a->d_rec, &a->d_fwd, n};
        if ((rc = out_reserve(h, out, n + n_gap_mx, st)) != MXG_OK) return rc;
        MXG_HIP(h, sc(SC_ST_HASH).ensure(std::max<uint64_t>(n * 8, 16)));
        MXG_HIP(h, sc(SC_ST_POS).ensure(std::max<uint64_t>(n * 4, 16)));
        MXG_HIP(h, sc(SC_ST_REC).ensure(std::max<uint64_t>(n * 4, 16)));
        MXG_HIP(h, sc(SC_ST_FWD).ensure(std::max<uint64_t>(n, 16)));
        if (n) {
            MXG_HIP(h, hipMemcpyAsync(sc(SC_ST_HASH).p, a->d_hash.p, n * 8, hipMemcpyDeviceToDevice, st));
            MXG_HIP(h, hipMemcpyAsync(sc(SC_ST_POS).p, a->d_pos.p, n * 4, hipMemcpyDeviceToDevice, st));
            MXG_HIP(h, hipMemcpyAsync(sc(SC_ST_REC).p, a->d_rec.p, n * 4, hipMemcpyDeviceToDevice, st));
            MXG_HIP(h, hipMemcpyAsync(sc(SC_ST_FWD).p, a->d_fwd.p, n, hipMemcpyDeviceToDevice, st));
        }
        MergeParams mp;
        mp.a_hash = sc(SC_ST_HASH).as<uint64_t>(); mp.a_pos = sc(SC_ST_POS).as<uint32_t>();
        mp.a_rec = sc(SC_ST_REC).as<uint32_t>(); mp.a_fwd = sc(SC_ST_FWD).as<uint8_t>(); mp.nA = (uint32_t)n;
        mp.b_hash = sc(SC_G_HASH).as<uint64_t>(); mp.b_pos = sc(SC_G_POS).as<uint32_t>();
        mp.b_rec = sc(SC_G_REC).as<uint32_t>(); mp.b_fwd = sc(SC_G_FWD).as<uint8_t>(); mp.nB = (uint32_t)n_gap_mx;
        mp.o_hash = a->d_hash.as<uint64_t>(); mp.o_pos = a->d_pos.as<uint32_t>();
        mp.o_rec = a->d_rec.as<uint32_t>(); mp.o_fwd = a->d_fwd.as<uint8_t>();
        hipLaunchKernelGGL(k_merge, dim3((mp.nA + mp.nB + 255) / 256), dim3(256), 0, st, mp);
        MXG_HIP(h, hipGetLastError());
        MXG_HIP(h, hipStreamSynchronize(st));
        *n_out = n + n_gap_mx;
        return MXG_OK;
    }

    // every contig of T, appended to `out` (synchronous: one sync per batch, retries and gap fix-ups inline)
    // (c_start / out.n: the redo of an assembly's batches from contig c_start on, behind the minimizers already in place)
    int sparse_all(Assembly *a, const Tables &T, OutArrays &out, uint32_t tau_hi, double cand_frac, size_t c_start = 0)
    {
        const size_t n_ctg = T.ctg_rec->size();
        const uint32_t S = a->S_sparse;
        size_t c0 = c_start;
        while (c0 < n_ctg) {
            BatchGeom g;
            batch_geom(T, c0, g);
            const size_t c1 = g.c1;
            uint64_t wave_cap = default_wave_cap(S, cand_frac);
            int rcp = ensure_pinned_ctrl(h);
            if (rcp != MXG_OK) return rcp;
            uint32_t *const ctrl = h->pinned_ctrl + 16 * (PINNED_SLOTS - 1);  // the slot of the synchronous path
            uint64_t n_cap64 = 0;
            for (int attempt = 0;; ++attempt) {
                uint32_t n_cap_now = 0;
                memset(ctrl, 0xFF, 64);
                int rc = enqueue_sparse(a, T, g, wave_cap, tau_hi, out, ctrl, &n_cap_now);
                if (rc != MXG_OK) return rc;
                n_cap64 = n_cap_now;
                MXG_HIP(h, stream_wait(st));
                if (ctrl[0] == 0) break;  // no wave overflowed its slice
                if (attempt >= 2) return set_err(h, MXG_EDEVICE, "internal error: candidate arena keeps overflowing");
                wave_cap = std::min<uint64_t>((uint64_t)ctrl[0] + 64, 64ull * S);  // exact need is known: redo the batch
                h->arena_cap_hint = wave_cap;
            }
            uint32_t ctrl_copy[16];
            memcpy(ctrl_copy, ctrl, 64);
            int rcb = complete_batch(a, T, g, out, ctrl_copy, (uint32_t)n_cap64);
            if (rcb != MXG_OK) return rcb;
            c0 = c1;
        }
        return MXG_OK;
    }
};

// strip -> run, for every strip of the sparse strip table: block r writes run r's strips (once per assembly; the batch kernels of
// the k = 32 route read it instead of searching run_strip0 per strip: 12-15 dependent loads per strip on a fragmented assembly)
__global__ __launch_bounds__(256) void k_strip_runs(const uint32_t *__restrict__ run_strip0, uint32_t n_runs, uint32_t *__restrict__ strip_run)
{
    const uint32_t r = blockIdx.x;
    if (r >= n_runs) return;
    const uint32_t s0 = run_strip0[r], s1 = run_strip0[r + 1];
    for (uint32_t s = s0 + threadIdx.x; s < s1; s += 256u) strip_run[s] = r;
}

static int prepare_tables(mxg_handle *h, Assembly *a)
{
    if (a->tables_ready) return MXG_OK;
    const size_t n_runs = a->runs.size();
    if (n_runs >= (1ull << 31)) return set_err(h, MXG_ELIMIT, "too many valid runs (%zu)", n_runs);
    bool ovf = false;
    build_strip_tables(a->runs, S_DENSE, a->strip0_dense, &ovf);
    a->S_sparse = choose_sparse_S(h, a->total_kmers);
    build_strip_tables(a->runs, (int)a->S_sparse, a->strip0_sparse, &ovf);
    if (ovf) return set_err(h, MXG_ELIMIT, "too many strips");
    a->g0.resize(n_runs + 1);
    uint64_t g = 0;
    for (size_t r = 0; r < n_runs; ++r) {
        a->g0[r] = g;
        g += a->runs[r].n_kmers;
    }
    a->g0[n_runs] = g;
    int rc;
    if ((rc = upload(h, a->d_runs, a->runs)) != MXG_OK) return rc;
    if ((rc = upload(h, a->d_strip0_dense, a->strip0_dense)) != MXG_OK) return rc;
    if ((rc = upload(h, a->d_strip0_sparse, a->strip0_sparse)) != MXG_OK) return rc;
    if ((rc = upload(h, a->d_g0, a->g0)) != MXG_OK) return rc;
    if ((rc = upload(h, a->d_ctg_nk, a->ctg_nk)) != MXG_OK) return rc;
    if ((rc = upload(h, a->d_ctg_rec, a->ctg_rec)) != MXG_OK) return rc;
    if ((rc = upload(h, a->d_ctg_run0, a->ctg_run0)) != MXG_OK) return rc;
    {
        std::vector<uint4> ci(a->ctg_rec.size());
        for (size_t c = 0; c < ci.size(); ++c) {
            const uint32_t r0 = a->ctg_run0[c], r1 = a->ctg_run0[c + 1];
            ci[c] = make_uint4(r0, r1 - r0, r1 > r0 ? a->runs[r0].pos0 : 0u, a->ctg_rec[c]);
        }
        if ((rc = upload(h, a->d_ctg_info, ci)) != MXG_OK) return rc;
    }
    if (a->any_drop && (rc = upload(h, a->d_ctg_drop, a->ctg_drop)) != MXG_OK) return rc;
    if (n_runs) {
        std::vector<RunX> rx(n_runs);
        for (size_t r = 0; r < n_runs; ++r) {
            const Run &q = a->runs[r];
            rx[r] = RunX{q.base_off, q.n_kmers, q.contig, q.kidx0, a->strip0_sparse[r] * a->S_sparse, a->ctg_nk[q.contig], 0u};
        }
        if ((rc = upload(h, a->d_runx, rx)) != MXG_OK) return rc;
        const uint32_t n_strips = a->strip0_sparse[n_runs];
        MXG_HIP(h, a->d_strip_run.ensure((size_t)n_strips * 4 + 16));
        hipLaunchKernelGGL(k_strip_runs, dim3((uint32_t)n_runs), dim3(256), 0, h->stream, a->d_strip0_sparse.as<uint32_t>(), (uint32_t)n_runs,
                           a->d_strip_run.as<uint32_t>());
        MXG_HIP(h, hipGetLastError());
    }
    MXG_HIP(h, hipStreamSynchronize(h->stream));
    a->tables_ready = true;
    return MXG_OK;
}

// uploads / tables / output capacity for one assembly; *empty = nothing eligible (the sketch is empty)
// The packed bases of an assembly that was parsed on the host go to HBM -- when the assembly is added (api.cpp: commit), so that
// a sketch starts from resident bases whatever route loaded them (the first sketch used to carry this copy: 10 ms per Gbp).
int upload_packed(mxg_handle *h, Assembly *a)
{
    if (a->d_packed || !a->has_bases || a->h_packed.empty()) return MXG_OK;
    MXG_HIP(h, hipSetDevice(h->device));
    MXG_HIP(h, a->d_packed_own.ensure(a->h_packed.size() * 4));
    MXG_HIP(h, hipMemcpyAsync(a->d_packed_own.p, a->h_packed.data(), a->h_packed.size() * 4, hipMemcpyHostToDevice, h->stream));
    MXG_HIP(h, hipStreamSynchronize(h->stream));
    a->d_packed = a->d_packed_own.as<uint32_t>();
    std::vector<uint32_t>().swap(a->h_packed);  // refetched from HBM on demand (mxg_write_tsv without text)
    return MXG_OK;
}

// (diagnostics, MXG_DEBUG_COLD=1: host milliseconds between marks -- where a fresh handle's first step goes)
static void cold_mark_g(const mxg_handle *h, const char *what)
{
    static thread_local std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    if (!knob_set(h, "MXG_DEBUG_COLD")) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[mxg]   . %s: %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
    last = now;
}

static int prepare_sketch(mxg_handle *h, Assembly *a, Tables &T, bool *empty)
{
    cold_mark_g(h, "(prepare_sketch begins)");
    if (!a->has_bases) return set_err(h, MXG_EINVAL, "assembly '%s' has no bases to sketch", a->name.c_str());
    MXG_HIP(h, hipSetDevice(h->device));
    const uint32_t w = h->cfg.w;
    a->has_sketch = false;
    a->host_valid = false;
    a->fwd_valid = false;
    a->foreign_sketch = false;
    a->flags_valid = false;
    h->graph.valid = false;
    a->n_mx_seen = std::max(a->n_mx_seen, a->n_mx);
    a->n_mx = 0;
    *empty = false;

    int rc0 = upload_packed(h, a);  // (bases to HBM: the loaders do it when the assembly is added; here for whatever did not)
    if (rc0 != MXG_OK) return rc0;
    if (a->runs.empty()) {  // nothing eligible: empty sketch
        a->has_sketch = true;
        *empty = true;
        return MXG_OK;
    }
    cold_mark_g(h, "upload_packed");
    int rc = prepare_tables(h, a);
    if (rc != MXG_OK) return rc;
    cold_mark_g(h, "prepare_tables");
    if (!h->d_init_tab.p) {
        std::vector<uint4> it;
        make_init_tab(h->cfg.k, it);
        if (it.empty()) it.push_back(make_uint4(0, 0, 0, 0));
        if ((rc = upload(h, h->d_init_tab, it)) != MXG_OK) return rc;
        MXG_HIP(h, hipStreamSynchronize(h->stream));
    }
    T.runs = &a->runs;
    T.ctg_nk = &a->ctg_nk;
    T.ctg_rec = &a->ctg_rec;
    T.ctg_run0 = &a->ctg_run0;
    T.strip0_dense = &a->strip0_dense;
    T.strip0_sparse = &a->strip0_sparse;
    T.g0 = &a->g0;
    T.d_runs = a->d_runs.as<Run>();
    T.d_strip0_dense = a->d_strip0_dense.as<uint32_t>();
    T.d_strip0_sparse = a->d_strip0_sparse.as<uint32_t>();
    T.d_strip_run = a->d_strip_run.as<uint32_t>();
    T.d_ctg_nk = a->d_ctg_nk.as<uint32_t>();
    T.d_ctg_rec = a->d_ctg_rec.as<uint32_t>();
    T.d_ctg_run0 = a->d_ctg_run0.as<uint32_t>();
    T.d_ctg_info = a->d_ctg_info.as<uint4>();
    T.d_g0 = a->d_g0.as<uint64_t>();
    T.d_ctg_drop = a->any_drop ? a->d_ctg_drop.as<uint8_t>() : nullptr;
    T.recs = &a->recs;
    // output capacity estimate: density 2/(w+1) per k-mer plus slack; grown on demand.  (Two and a half times the i.i.d. density:
    // low-complexity runs report every second or third k-mer -- bench.py's repeat-rich Gbp holds 3.8 M minimizers where i.i.d.
    // sequence holds 2.0 M, and a first sketch that outgrew arrays sized for 3.0 M was redone whole: 17 bytes per entry.)
    uint64_t cap = (uint64_t)(5.0 * (double)a->total_kmers / (double)(w + 1)) + 4096;
    MXG_HIP(h, a->d_hash.ensure(cap * 8));
    MXG_HIP(h, a->d_pos.ensure(cap * 4));
    MXG_HIP(h, a->d_rec.ensure(cap * 4));
    MXG_HIP(h, a->d_fwd.ensure(cap));
    cold_mark_g(h, "init table + output arrays");
    return MXG_OK;
}

// What a sketch needs of an assembly besides its bases, made when the assembly is ADDED (api.cpp: commit) instead of inside its
// first sketch: run / strip / contig tables in HBM (nine small uploads, the strip -> run table built on the stream), the output
// arrays, the filter's bitmap.  A fresh handle's first step spent 1.2 ms of host time on them at 3 Gbp + 3 Gbp before and between
// its enqueues (MXG_DEBUG_COLD=1), 45 % on top of the step.  Failures are left for the sketch to report.
int prewarm_assembly(mxg_handle *h, Assembly *a)
{
    if (!a->has_bases || a->runs.empty() || !a->d_packed || knob_set(h, "MXG_NO_PREWARM")) return MXG_OK;
    Tables T;
    bool empty = false;
    int rc = prepare_sketch(h, a, T, &empty);
    if (rc == MXG_OK && !empty && bs_possible(h, a) && knob_u64(h, "MXG_BS", 1) != 0) rc = bs_prepare(h, a);
    return rc;
}

// Sparse path: expected c candidates per window; it pays while candidates are a small fraction of k-mers.
// c trades the work behind the hash kernel (linear in c) against candidate-free stretches (a candidate is followed by
// one with probability e^-c).  With the stretches fixed up on the device (k_gap_fix) they are cheap, so large assemblies
// run with c = 10 (about 450 stretches per 10^9 k-mers at w = 1000); small ones keep c = 18, where a stretch (then handled by the
// host-driven route) turns up once per ~4 x 10^9 k-mers, and save the three extra launches per batch.
// mxg_config.cand_per_window fixes c; MXG_DEV_GAPS=0|1 forces the route (test / profiling knobs, read per call).
// Capacity of the device route's per-stretch arrays for a batch of an assembly: a power of two between GAP_DEV_MAX and
// GAP_DEV_CAP_MAX, thirty times what i.i.d. sequence of the assembly's size holds at ten candidates per window (repeat-rich
// genomes: satellite arrays, low-complexity runs -- 13 stretches per Mbp on bench.py's repeat-rich workload against 0.45), or two
// and a half times what earlier sketches of the assembly met.  MXG_GAP_DEV_CAP: test knob.
static uint32_t gap_capacity(const mxg_handle *h, const Assembly *a)
{
    const uint64_t forced = knob_u64(h, "MXG_GAP_DEV_CAP", 0);
    double want = forced ? (double)forced : std::max(32e-6, 2.5 * (a ? a->gap_rate_hint : 0.0)) * (double)(a ? a->total_kmers : 0);
    uint32_t cap = GAP_DEV_MAX;
    while (cap < GAP_DEV_CAP_MAX && (double)cap < want) cap *= 2;
    return forced ? (uint32_t)std::min<uint64_t>(std::max<uint64_t>(forced, 64), GAP_DEV_CAP_MAX) : cap;
}

struct SparsePlan {
    bool sparse;
    bool dev_gaps;
    uint32_t gcap;         // device route: stretches a batch's arrays hold (gap_capacity)
    double frac;
    uint32_t tau_hi;
    double gap_rate;       // device route: expected candidate-free stretches per k-mer (prior, or what earlier sketches met)
    uint64_t gap_kmers;    // device route: k-mers that hold ~GAP_DEV_MAX / 2 expected stretches (0: not the device route)
    uint64_t batch_kmers;  // device route: batches small enough for ~GAP_DEV_MAX / 2 expected stretches (0: the default size;
                           // a quarter until round 3: on repeat-rich sequence, whose batches this limit cuts, half as many batches
                           // are 20 % faster, and a batch that overflows all the same is re-sized and enqueued again)
};
static SparsePlan sparse_plan(const mxg_handle *h, const Assembly *a)
{
    SparsePlan sp;
    const uint64_t big = env_u64(h, "MXG_DEV_GAPS_MIN_KMERS", 256ull << 20);
    sp.dev_gaps = a && a->total_kmers >= big && h->cfg.w <= GAP_DEV_NMAX / 2;
    if (knob_set(h, "MXG_DEV_GAPS")) sp.dev_gaps = knob_u64(h, "MXG_DEV_GAPS", 0) != 0 && h->cfg.w <= GAP_DEV_NMAX / 2;
    // measured on MI355X (3 Gbp + 3 Gbp, w=1000, batches of 512 Mi k-mers): 941 / 985 / 945 / 897 Gbp/s at c = 8 / 10 / 12 / 14
    // (w = 500, configs[3], S = 192: 1256 / 1313 / 1342 / 1259 Gbp/s at c = 9 / 10 / 11 / 12 -- candidates are twice as dense as at
    // w = 1000, so are the stretches at a given c, and the stretch kernels' share grows: one candidate more per window pays)
    // round 6, k = 32 route: the stretches between candidates are sketched by k_sel_stretch right behind the slice kernel (one wave
    // per slice that has any), so a stretch costs a few us of one wave and c = 8 pays (tools/sweep_cS.sh: 2.74 / 2.62 / 2.60 / 3.05 ms
    // per step at c = 10 / 8 / 7 / 6 on 3 Gbp + 3 Gbp, w = 1000; configs[3], w = 500: 1421 / 1468 / 1540 / 1579 / 1578 Gbp/s at
    // c = 11 / 10 / 9 / 8 / 7)
    const bool inl_route = h->cfg.k == 32 && h->cfg.variant == MXG_VARIANT_V2_SUM && knob_u64(h, "MXG_BS", 1) != 0 &&
                           knob_u64(h, "MXG_BS_SELECT", 1) != 0 && knob_u64(h, "MXG_SEL_INLINE", 1) != 0;
    const uint32_t c = h->cfg.cand_per_window ? h->cfg.cand_per_window
                                              : (sp.dev_gaps ? (uint32_t)env_u64(h, "MXG_DEV_CAND", inl_route ? 8 : (h->cfg.w < 700 ? 11 : 10)) : 18u);
    sp.frac = (double)c / (double)h->cfg.w;
    // even: the threshold then falls on the top 31-bit ring of the hash, which is all the sparse kernel rolls
    sp.tau_hi = std::max(2u, (uint32_t)std::min<double>(4294967294.0, sp.frac * 4294967296.0) & ~1u);
    sp.sparse = !(h->cfg.flags & MXG_FLAG_DENSE_ONLY) && sp.frac <= 0.125;
    sp.batch_kmers = 0;
    sp.gap_kmers = 0;
    sp.gap_rate = 0;
    sp.gcap = gap_capacity(h, a);
    if (sp.dev_gaps) {  // a candidate is followed by a stretch with probability e^-c
        // (a->gap_rate_hint: what earlier sketches of this assembly met, 25 % on top)
        // (behind k_sel_stretch only what that kernel does not take reaches the stretch kernels: on plain sequence next to nothing)
        const bool inl = h->cfg.k == 32 && h->cfg.variant == MXG_VARIANT_V2_SUM && knob_u64(h, "MXG_BS", 1) != 0 &&
                         knob_u64(h, "MXG_BS_SELECT", 1) != 0 && knob_u64(h, "MXG_SEL_INLINE", 1) != 0;
        // (... and once a sketch of the assembly has reported, what it met counts, not the prior)
        const double per_kmer = inl && a->gap_rate_hint > 0 ? std::max(a->gap_rate_hint * 1.25, 1e-9)
                                                            : std::max(sp.frac * std::exp(-(double)c) * (inl ? 0.02 : 1.0), a->gap_rate_hint * 1.25);
        sp.gap_rate = per_kmer;
        const double lim = (double)env_u64(h, "MXG_GAP_BUDGET", sp.gcap / 2) / std::max(per_kmer, 1e-30);  // expected stretches per batch
        sp.gap_kmers = (uint64_t)std::min<double>(std::max<double>(lim, (double)(1u << 20)), 9e18);
        if (lim < (double)SPARSE_BATCH_KMERS) sp.batch_kmers = std::max<uint64_t>((uint64_t)lim, 1u << 20);
    }
    return sp;
}

static int run_sketch_sync(mxg_handle *h, Assembly *a, const Tables &T, Driver &drv)
{
    OutArrays out{&a->d_hash, &a->d_pos, &a->d_rec, &a->d_fwd, 0};
    const SparsePlan sp = sparse_plan(h, a);
    int rc = sp.sparse ? drv.sparse_all(a, T, out, sp.tau_hi, sp.frac) : drv.dense_all(a->d_packed, T, out, true);
    if (rc != MXG_OK) return rc;
    MXG_HIP(h, hipStreamSynchronize(drv.st));
    a->n_mx = out.n;
    a->has_sketch = true;
    return MXG_OK;
}

int sketch_assembly(mxg_handle *h, Assembly *a)
{
    Assembly *list[1] = {a};
    return sketch_assemblies(h, list, 1);
}

// xchg_pack's kernel with everything read on the device: the sketch may still be in flight on the stream.  The header
// says -1 (the caller exchanges sizes first) unless the batch ended the common way -- no arena overflow, no
// candidate-free stretch, at least one candidate -- and the sketch fits the slot and the output arrays.
__global__ __launch_bounds__(256) void k_pack_slot_dev(const uint64_t *__restrict__ hash, const uint32_t *__restrict__ pos,
                                                       const uint32_t *__restrict__ rec, const uint32_t *__restrict__ n_ptr,
                                                       const uint32_t *__restrict__ ctrl, uint64_t out_cap, uint64_t cap,
                                                       long long fixed, long long *header, unsigned char *__restrict__ region,
                                                       uint32_t dev_gaps, uint32_t place4)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (fixed != -2) {  // the host already knows: empty assembly (0) or not through the one-batch pipeline (-1)
        if (i == 0) *header = fixed;
        return;
    }
    const uint64_t n = *n_ptr;
    // the same predicate as sketch_finish's (batch_ended_well): no arena overflow, no flag from the stretch kernels (word 6) or from
    // the slice kernel (word 13: a slice gave up), stretches either absent or -- on the device route -- all placed (none deferred,
    // and no more of them than k_emit's launch had placing blocks for: k_emit then places none and tells the host through ITS
    // word 3 only, which this kernel does not see)
    const bool ok = ctrl[0] == 0 && ctrl[6] == 0 && ctrl[13] == 0 && (dev_gaps ? (ctrl[11] == 0 && ctrl[1] <= place4) : ctrl[1] == 0) && (ctrl[4] | ctrl[5]) != 0 &&
                    n <= cap && n <= out_cap;
    if (i == 0) *header = ok ? (long long)n : -1ll;
    if (!ok || i >= n) return;
    reinterpret_cast<uint64_t *>(region)[i] = hash[i];
    reinterpret_cast<uint32_t *>(region + 8 * cap)[i] = pos[i];
    reinterpret_cast<uint32_t *>(region + 12 * cap)[i] = rec[i];
}

// mxg_sketch_pack_parts' kernels.  A part carries 12 bytes per minimizer, not 16: hash | pos | the first entry of every RECORD.
// The entries are in (record, position) order, so the record column is a step function of the entry index: 4 bytes per record
// travel instead of 4 per minimizer, and the receiver finds an entry's record by bisection (k_unpack_part).  Same predicate as
// k_pack_slot_dev; header word 1 = the sender's number of records.
__global__ __launch_bounds__(256) void k_pack_part_dev(const uint64_t *__restrict__ hash, const uint32_t *__restrict__ pos,
                                                       const uint32_t *__restrict__ rec, const uint32_t *__restrict__ n_ptr,
                                                       const uint32_t *__restrict__ ctrl, uint64_t out_cap, uint64_t cap,
                                                       long long fixed, long long *header, unsigned char *__restrict__ region,
                                                       uint32_t n_rec, uint32_t rcap, uint32_t dev_gaps, uint32_t place4)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (fixed != -2) {
        if (i == 0) {
            header[0] = fixed;
            header[1] = 0;
        }
        return;
    }
    const uint64_t n = *n_ptr;
    const bool ok = ctrl[0] == 0 && ctrl[6] == 0 && ctrl[13] == 0 && (dev_gaps ? (ctrl[11] == 0 && ctrl[1] <= place4) : ctrl[1] == 0) && (ctrl[4] | ctrl[5]) != 0 &&
                    n <= cap && n <= out_cap && n_rec <= rcap;
    if (i == 0) {
        header[0] = ok ? (long long)n : -1ll;
        header[1] = n_rec;
    }
    if (!ok || i >= n) return;
    reinterpret_cast<uint64_t *>(region)[i] = hash[i];
    reinterpret_cast<uint32_t *>(region + 8 * cap)[i] = pos[i];
    const uint32_t r = rec[i];
    if ((i == 0 || rec[i - 1] != r) && r < n_rec) reinterpret_cast<uint32_t *>(region + 12 * cap)[r] = (uint32_t)i;
}

// starts[r] = min(starts[r .. n_rec), n): a record without an entry takes the next record's first entry (one block)
__global__ __launch_bounds__(1024) void k_part_starts(const long long *__restrict__ header, uint32_t *__restrict__ starts, uint32_t n_rec)
{
    const long long n = header[0];
    if (n < 0) return;
    __shared__ uint32_t smin[1024];
    const uint32_t per = (n_rec + 1023u) / 1024u, lo = min(threadIdx.x * per, n_rec), hi = min(lo + per, n_rec);
    uint32_t m = 0xFFFFFFFFu;
    for (uint32_t q = lo; q < hi; ++q) m = min(m, starts[q]);
    smin[threadIdx.x] = m;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {  // suffix minimum over the threads' chunks
        const uint32_t v = threadIdx.x + off < 1024u ? smin[threadIdx.x + off] : 0xFFFFFFFFu;
        __syncthreads();
        smin[threadIdx.x] = min(smin[threadIdx.x], v);
        __syncthreads();
    }
    uint32_t carry = min(threadIdx.x + 1u < 1024u ? smin[threadIdx.x + 1u] : 0xFFFFFFFFu, (uint32_t)n);
    for (uint32_t q = hi; q-- > lo;) {
        carry = min(carry, starts[q]);
        starts[q] = carry;
    }
}

// Every assembly is cut into batches of whole records (SPARSE_BATCH_KMERS) and EVERY batch of EVERY assembly is enqueued
// completely -- hash -> order -> resolve (-> stretches on the device) -> emit -- alternating between the handle's two
// streams with their own scratch, before the host waits once for both streams.  A batch starts in the assembly's sketch
// where the batches before it end: that sum travels through a device word (ChainIO), and a batch whose predecessor ran on
// the other stream waits for it just before its emit, so its hash / order / resolve kernels overlap the predecessor's
// latency-bound tail.  What did not end the common way (arena overflow, stretches the device route could not hold,
// output beyond its estimate) is redone afterwards: from the candidate arrays when they are still intact in the
// scratch, else through the synchronous path.
// fuse_graph: also run the graph stage, enqueued BEHIND the sketches with upper bounds for the sizes and the counts read
// on the device (one host sync for the whole step; needs one batch per assembly).  xp: mxg_sketch_pack.
int sketch_assemblies(mxg_handle *h, Assembly *const *list, size_t n, bool fuse_graph, const XchgPackReq *xp)
{
    if (xp) {
        bool same = n == h->asms.size() && n <= MXG_MAX_ASSEMBLIES;
        for (size_t i = 0; same && i < n; ++i) same = list[i] == h->asms[i];
        if (!same || fuse_graph) return set_err(h, MXG_EINVAL, "mxg_sketch_pack: every assembly of the handle must have bases");
    }
    MXG_HIP(h, hipSetDevice(h->device));
    int rc = ensure_pinned_ctrl(h);
    if (rc != MXG_OK) return rc;
    // batches of multi-batch assemblies rotate over the streams (MXG_STREAMS = 2..4), so that one batch's latency-bound tail
    // (emit, stretch fix-up) has another batch's hash kernel to run beside.  Two are enough: measured on MI355X at 3 Gbp +
    // 3 Gbp, 1072 / 1076 / 1046 Gbp/s with 2 / 3 / 4 streams -- the sum of the kernels' own times (9 ms per step under
    // rocprofv3) already overlaps into 5.6 ms of wall time
    uint32_t n_str = (uint32_t)std::min<uint64_t>(4, std::max<uint64_t>(2, env_u64(h, "MXG_STREAMS", 2)));
    for (uint32_t x = 0; x + 2 < n_str; ++x)
        if (!h->stream_x[x]) MXG_HIP(h, hipStreamCreateWithFlags(&h->stream_x[x], hipStreamNonBlocking));
    Driver drv0(h, 0), drv1(h, 1), drv2(h, n_str > 2 ? 2 : 1), drv3(h, n_str > 3 ? 3 : 1);
    // profiling (tools/pmc_r03.sh): every batch on the handle's main stream, so that no two kernels overlap.  Every batch then uses
    // slot 0 (one scratch set, in stream order); the calls that keep one batch per assembly in flight for the stage behind them
    // (mxg_sketch_graph, mxg_sketch_pack) need a scratch set per assembly and ignore the switch.
    const bool one_stream = knob_set(h, "MXG_ONE_STREAM") && !fuse_graph && !xp;
    Driver *drvs[4] = {&drv0, one_stream ? &drv0 : &drv1, one_stream ? &drv0 : &drv2, one_stream ? &drv0 : &drv3};
    if (one_stream) n_str = 1;
    struct Item {
        size_t asm_i;
        Driver::BatchGeom g;
        int slot;
        uint32_t n_cap;
        uint32_t *hc;  // pinned control block
        bool bs;       // went through the k = 32 route (no candidate arrays to finish from)
        uint32_t place4 = 0;  // stretches the batch's k_emit launch has placing blocks for (4 * n_place)
    };
    const bool bs_env = env_u64(h, "MXG_BS", 1) != 0;
    const bool bs_select = env_u64(h, "MXG_BS_SELECT", 1) != 0;  // k_bs_select instead of count -> reorder -> resolve
    std::vector<Tables> tabs(n);
    std::vector<int> state(n, 0);  // 0 = synchronous path, 1 = enqueued, 2 = done
    std::vector<SparsePlan> plans(n);
    std::vector<Item> items;
    std::vector<size_t> item0(n + 1, 0);
    size_t last_on_slot[4] = {(size_t)-1, (size_t)-1, (size_t)-1, (size_t)-1};
    size_t n_enq = 0, next_slot = 0;
    int last_sel_slot = -1;  // the stream slot the last slice kernel of this call went to
    const bool stagger = knob_u64(h, "MXG_STAGGER", 1) != 0;
    // (an assembly's k_emit behind the NEXT assembly's slice kernel, beside that assembly's stretch kernels: enqueued in its own
    // place it starts when the next filter lets go of the GPU and lands on the next slice kernel, whose blocks need whole CUs --
    // 529 us for the target's launch against 450 for the reference's under rocprofv3; MXG_DEFER_EMIT=0: in its own place)
    const bool defer_emits = knob_u64(h, "MXG_DEFER_EMIT", 1) != 0 && knob_u64(h, "MXG_STAGGER", 1) == 1;
    for (int q = 0; q < 4; ++q)
        if (!h->ev_sel_done[q]) MXG_HIP(h, hipEventCreateWithFlags(&h->ev_sel_done[q], hipEventDisableTiming));
    const bool chain_modes = fuse_graph || xp;  // (these two need one batch per assembly)
    // the batches of assembly i -> the streams (attempt 0; attempt 1: once more for an assembly whose batches did not all end
    // the common way, now sized by what the first attempt saw: stretch density, candidate counts, slice capacity)
    std::vector<size_t> q_lo(n, 0), q_hi(n, 0);
    auto enqueue_asm = [&](size_t i, int attempt) -> int {
        q_lo[i] = q_hi[i] = items.size();
        state[i] = 0;
        plans[i] = sparse_plan(h, list[i]);
        if (!plans[i].sparse || i >= MXG_MAX_ASSEMBLIES) return MXG_OK;
        // the k = 32 route: the bit-sliced filter over the whole assembly, then one k_bs_select per batch (sketch_bs.hip); a second
        // attempt (a batch did not end the common way: slices beyond their queues, stretches beyond the device route) and run
        // tables with short runs between invalid bases take count -> reorder -> resolve behind the same bitmap
        bool use_bs = bs_env && bs_possible(h, list[i]);
        bool sel_ok = use_bs && bs_select && (attempt == 0 || list[i]->sel_again);
        cold_mark_g(h, "(enqueue_asm begins)");
        if (use_bs) {
            if ((rc = bs_prepare(h, list[i])) != MXG_OK) return rc;
            cold_mark_g(h, "bs_prepare");
            use_bs = list[i]->bs_ready;
            Assembly *a = list[i];
            if (use_bs && sel_ok && (a->sel_H_S != a->S_sparse || a->sel_H_w != h->cfg.w)) {
                a->sel_H = bs_select_halo(a, a->S_sparse, h->cfg.w);
                a->sel_H_S = a->S_sparse;
                a->sel_H_w = h->cfg.w;
            }
            sel_ok = sel_ok && use_bs && a->sel_H != 0;
        }
        // k_bs_select has no candidate arrays to size: its batches are as large as the stretch budget and 32-bit k-mer counts allow
        // (an assembly of 3 Gbp: two batches, ten launches in all, where the other route cuts seven)
        std::vector<Driver::BatchGeom> gs;
        std::vector<BsSelGeom> bgs;
        const size_t n_ctg = tabs[i].ctg_rec->size();
        for (int pass = sel_ok ? 0 : 1; pass < 2; ++pass) {
            uint64_t budget = plans[i].batch_kmers;
            if (pass == 0) {
                const uint64_t big = env_u64(h, "MXG_SEL_BATCH_KMERS", 3600ull << 20);  // (k-mers of a batch are counted in 32 bits)
                budget = std::min<uint64_t>(plans[i].gap_kmers ? plans[i].gap_kmers : big, big);
                if (knob_set(h, "MXG_SPARSE_BATCH_KMERS")) budget = std::min<uint64_t>(budget, SPARSE_BATCH_KMERS);  // (test knob)
            }
            gs.clear();
            bgs.clear();
            for (size_t c0 = 0; c0 < n_ctg;) {
                Driver::BatchGeom g;
                drv0.batch_geom(tabs[i], c0, g, budget);
                gs.push_back(g);
                c0 = g.c1;
            }
            if (pass == 1) break;
            for (size_t b = 0; sel_ok && b < gs.size(); ++b) {
                bgs.push_back(drv0.sel_geom(list[i], gs[b], plans[i].frac));
                sel_ok = bgs.back().ok;  // (strips and selected entries are counted in 32 bits: bs_select_geom checks them)
            }
            if (sel_ok) break;
        }
        cold_mark_g(h, "halo + batch geometry");
        if (gs.empty() || (chain_modes && gs.size() > 1) || items.size() + gs.size() >= PINNED_SLOTS - 1) return MXG_OK;
        sel_ok = sel_ok && use_bs;
        hipEvent_t ev_hash = nullptr;
        hipStream_t st_hash = nullptr;
        OutArrays out{&list[i]->d_hash, &list[i]->d_pos, &list[i]->d_rec, &list[i]->d_fwd, 0};
        // chain words: [item] = where the NEXT batch starts
        MXG_HIP(h, h->d_chain.ensure((size_t)PINNED_SLOTS * 8));
        if (list[i]->cand_hints.size() != gs.size()) list[i]->cand_hints.assign(gs.size(), list[i]->full_grid_once ? 0xFFFFFFFEu : 0u);
        list[i]->full_grid_once = false;
        for (size_t b = 0; b < gs.size(); ++b) {
            // a single-batch assembly keeps the round-1 placement: the LAST assembly goes to driver 0 = the handle's main
            // stream (whatever follows the sketches on that stream then waits for the other stream's chain, which has
            // finished earlier); batches of a multi-batch assembly simply alternate
            size_t sl;
            if (one_stream) sl = 0;
            else if (gs.size() == 1) sl = h->own_stream ? ((n - 1 - i) & 1) : (i & 1);
            else sl = next_slot++ % n_str;
            Driver &drv = *drvs[sl];
            Item it;
            it.asm_i = i;
            it.g = gs[b];
            it.slot = (int)sl;
            it.n_cap = 0;
            it.bs = sel_ok;
            if (use_bs && attempt > 0) {
                // (the bitmap of the first attempt is still there, and every stream has been waited for)
            } else if (use_bs && b == 0) {  // the filter, once per assembly, on the first batch's stream
                st_hash = drv.st;
                // ... behind the slice kernel of the assembly before it when that runs on another stream: each of the two fills
                // the register file, side by side they only take turns; the tails behind the slice kernel (stretches, emit)
                // leave room.  Only for assemblies of 2^31 k-mers and more: there the free-running streams gain 1 % of the step
                // (3.20 against 3.24 ms at 3 Gbp + 3 Gbp, tools/stagger_try.sh) and a filter that shares the GPU with a slice
                // kernel takes 0.64 ms instead of 0.44 -- neither kernel's time says anything about the kernel any more; at
                // 1 Gbp + 1 Gbp, where the tails weigh more, running free is 6 % faster and stays.  (MXG_STAGGER=0: never)
                hipEvent_t behind = nullptr;
                // (mxg_sketch_pack_parts: always -- the assembly before this one must END first, its part travels beside this filter)
                if (stagger && !one_stream && last_sel_slot >= 0 && drvs[last_sel_slot] != &drv &&
                    (list[i]->total_kmers >= (1ull << 31) || (xp && (xp->d_parts || xp->dg))))
                    behind = h->ev_sel_done[last_sel_slot];
                if (sel_ok && (rc = drv.clear_sel_ctrl(bgs[b])) != MXG_OK) return rc;
                if ((rc = bs_edges(h, list[i], drv.st)) != MXG_OK) return rc;  // (the two blocks that copy the edge chunks need not wait)
                if (behind) MXG_HIP(h, hipStreamWaitEvent(drv.st, behind, 0));
                if ((rc = drv.ev_begin(list[i]->total_bases, true)) != MXG_OK) return rc;
                if ((rc = bs_hash(h, list[i], plans[i].tau_hi, drv.st)) != MXG_OK) return rc;
                h->stat_bs_bases += list[i]->total_bases;
                if ((rc = drv.ev_end()) != MXG_OK) return rc;
                if (gs.size() > 1) {
                    while (h->ev_bs.size() <= i) {
                        hipEvent_t e;
                        MXG_HIP(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                        h->ev_bs.push_back(e);
                    }
                    ev_hash = h->ev_bs[i];
                    MXG_HIP(h, hipEventRecord(ev_hash, drv.st));
                }
            } else if (use_bs && drv.st != st_hash) {
                MXG_HIP(h, hipStreamWaitEvent(drv.st, ev_hash, 0));
            }
            it.hc = h->pinned_ctrl + 16 * items.size();
            memset(it.hc, 0xFF, 64);
            Driver::ChainIO io;
            io.dev_gaps = plans[i].dev_gaps;
            {   // placing blocks for twice the stretches the plan expects of this batch (+ 256), at most for all the arrays hold; a
                // batch that meets more than its launch can place reports so and is enqueued again with the density it met
                const double expect = plans[i].gap_rate * (double)it.g.nk;
                const uint32_t place4 = (uint32_t)std::min<double>((double)plans[i].gcap, 2.0 * expect + 256.0);
                drv.set_gaps(plans[i].gcap, list[i]->gap_rate_hint > 0 ? (place4 + 3u) / 4u : plans[i].gcap / 4u,
                             list[i]->gap_rate_hint > 0 ? expect : -1.0);
                if (const uint64_t forced = knob_u64(h, "MXG_GAP_PLACE", 0))  // test knob: placing blocks for this many stretches
                    drv.set_gaps(plans[i].gcap, (uint32_t)std::min<uint64_t>((forced + 3u) / 4u, plans[i].gcap / 4u));
                it.place4 = 4u * drv.n_place;
            }
            // tiles of 32 slices in k_emit (0.18 against 0.21 ms per step at 3 Gbp + 3 Gbp) unless stretches are so dense that
            // most tiles of that size would hold one (the tile then searches the stretch keys per minimizer: repeat-rich
            // sequence is 2 % slower with 32, 6 % with 64; tools/sweep_emit_ecb.sh)
            io.ecb = plans[i].gap_rate * 32.0 * 64.0 * list[i]->S_sparse < 0.5 ? 32u : 16u;
            uint64_t *chain = h->d_chain.as<uint64_t>();
            io.base_in = b == 0 ? nullptr : chain + (items.size() - 1);
            io.base_out = gs.size() > 1 ? chain + items.size() : nullptr;
            if (b > 0 && drvs[items.back().slot] != &drv) {  // predecessor on the other stream: wait for its count before the emit
                while (h->ev_sync.size() <= items.size()) {
                    hipEvent_t e;
                    MXG_HIP(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    h->ev_sync.push_back(e);
                }
                hipEvent_t e = h->ev_sync[items.size()];
                MXG_HIP(h, hipEventRecord(e, drvs[items.back().slot]->st));
                io.wait = e;
            }
            drv.n_out = nullptr;
            if (fuse_graph || xp) {
                MXG_HIP(h, h->d_nmx.ensure(MXG_MAX_ASSEMBLIES * 4));
                drv.n_out = h->d_nmx.as<uint32_t>() + i;
            }
            if (sel_ok) {
                drv.defer_emit = defer_emits && !chain_modes && gs.size() == 1 && !one_stream;
                if ((rc = drv.enqueue_sel(list[i], tabs[i], it.g, bgs[b], plans[i].tau_hi, out, it.hc, &io)) != MXG_OK) return rc;
                // the emits other drivers hold back (the assembly before this one): behind this slice kernel
                for (Driver *od : drvs)
                    if (od != &drv && (rc = od->flush_emit(h->ev_sel_done[sl])) != MXG_OK) return rc;
                last_sel_slot = (int)sl;
            } else if ((rc = drv.enqueue_sparse(list[i], tabs[i], it.g, drv.default_wave_cap(list[i]->S_sparse, plans[i].frac),
                                                plans[i].tau_hi, out, it.hc, &it.n_cap, &io, list[i]->cand_hints[b],
                                                use_bs ? list[i]->d_bs_out.as<uint32_t>() + 4 : nullptr)) != MXG_OK)
                return rc;
            last_on_slot[sl] = items.size();
            items.push_back(it);
        }
        state[i] = 1;
        q_hi[i] = items.size();
        return MXG_OK;
    };
    auto pack_part = [&](size_t i, hipStream_t st, Driver *drv, uint32_t place4) -> int {
        Assembly *a = list[i];
        if (xp->dg) {  // the partitioned graph stage's item slots instead of an exchange part
            const uint64_t oc = std::min<uint64_t>({a->d_hash.bytes / 8, a->d_pos.bytes / 4, a->d_rec.bytes / 4, a->d_fwd.bytes});
            MXG_HIP(h, h->d_nmx.ensure(MXG_MAX_ASSEMBLIES * 4));
            const int rcd = dg_pack_slots_dev(h, a, (uint32_t)i, *xp->dg, st, state[i] == 1 ? 1u : (state[i] == 2 ? 0u : 2u),
                                              h->d_nmx.as<uint32_t>() + i, drv ? drv->sc(SC_CTRL).as<uint32_t>() : nullptr, oc,
                                              state[i] == 1 && plans[i].dev_gaps ? 1u : 0u, place4);
            if (rcd != MXG_OK) return rcd;
            if (!h->ev_part[i]) MXG_HIP(h, hipEventCreateWithFlags(&h->ev_part[i], hipEventDisableTiming));
            MXG_HIP(h, hipEventRecord(h->ev_part[i], st));
            return MXG_OK;
        }
        const uint64_t cap = xp->caps[i];
        const uint64_t out_cap = std::min<uint64_t>({a->d_hash.bytes / 8, a->d_pos.bytes / 4, a->d_rec.bytes / 4, a->d_fwd.bytes});
        const long long fixed = state[i] == 1 ? -2ll : (state[i] == 2 ? 0ll : -1ll);
        const uint32_t grid = state[i] == 1 ? (uint32_t)std::max<uint64_t>((cap + 255) / 256, 1) : 1u;
        unsigned char *part = static_cast<unsigned char *>(xp->d_parts[i]);
        const uint64_t rcap = xp->rcaps[i], n_rec = a->recs.size();
        uint32_t *starts = reinterpret_cast<uint32_t *>(part + XCHG_PART_HEAD + 12 * cap);
        MXG_HIP(h, h->d_nmx.ensure(MXG_MAX_ASSEMBLIES * 4));
        if (state[i] == 1 && n_rec <= rcap && n_rec) MXG_HIP(h, hipMemsetAsync(starts, 0xFF, 4 * n_rec, st));
        hipLaunchKernelGGL(k_pack_part_dev, dim3(grid), dim3(256), 0, st, a->d_hash.as<uint64_t>(), a->d_pos.as<uint32_t>(),
                           a->d_rec.as<uint32_t>(), h->d_nmx.as<uint32_t>() + i,
                           drv ? drv->sc(SC_CTRL).as<uint32_t>() : h->d_nmx.as<uint32_t>(), out_cap, cap, fixed,
                           reinterpret_cast<long long *>(part), part + XCHG_PART_HEAD, (uint32_t)std::min<uint64_t>(n_rec, 0xFFFFFFFFu),
                           (uint32_t)std::min<uint64_t>(rcap, 0xFFFFFFFFu), state[i] == 1 && plans[i].dev_gaps ? 1u : 0u, place4);
        if (state[i] == 1 && n_rec <= rcap && n_rec)
            hipLaunchKernelGGL(k_part_starts, dim3(1), dim3(1024), 0, st, reinterpret_cast<const long long *>(part), starts, (uint32_t)n_rec);
        MXG_HIP(h, hipGetLastError());
        if (!h->ev_part[i]) MXG_HIP(h, hipEventCreateWithFlags(&h->ev_part[i], hipEventDisableTiming));
        MXG_HIP(h, hipEventRecord(h->ev_part[i], st));
        return MXG_OK;
    };
    const bool dbg_cold = knob_set(h, "MXG_DEBUG_COLD");  // (diagnostics: where the host's time goes before and between the enqueues)
    auto t_cold = std::chrono::steady_clock::now();
    auto cold_mark = [&](const char *what, size_t i) {
        if (!dbg_cold) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[mxg] sketch_assemblies: %s %zu: %.3f ms\n", what, i, std::chrono::duration<double, std::milli>(now - t_cold).count());
        t_cold = now;
    };
    for (size_t i = 0; i < n; ++i) {
        item0[i] = items.size();
        bool empty = false;
        if ((rc = prepare_sketch(h, list[i], tabs[i], &empty)) != MXG_OK) return rc;
        cold_mark("prepare_sketch", i);
        if (empty) {
            state[i] = 2;
            continue;
        }
        if ((rc = enqueue_asm(i, 0)) != MXG_OK) return rc;
        cold_mark("enqueue_asm", i);
        if (state[i] == 1) ++n_enq;
        if (xp && (xp->d_parts || xp->dg) && state[i] == 1) {
            // mxg_sketch_pack_parts: this assembly's part right behind its k_emit, on the stream that ran it -- the caller's
            // all-gather of the part travels while the next assembly is sketched
            const Item &it = items[q_lo[i]];
            if ((rc = pack_part(i, drvs[it.slot]->st, drvs[it.slot], it.place4)) != MXG_OK) return rc;
        }
    }
    for (Driver *od : drvs)  // (the last assembly's emit, held back for an assembly that did not come)
        if ((rc = od->flush_emit(nullptr)) != MXG_OK) return rc;
    item0[n] = items.size();
    for (size_t i = n; i-- > 0;)
        if (state[i] != 1) item0[i] = item0[i + 1];  // (assemblies without items: empty range)
    std::vector<int> slot_of(n, 0);
    for (size_t i = 0; i < n; ++i)
        if (state[i] == 1) slot_of[i] = items[item0[i]].slot;
    if (xp) {
        // the second stream joins the first; the pack kernels follow the sketches on it and read the counts there.  No
        // host sync: mxg_sketch_finish completes the bookkeeping after the caller's next sync on this stream.
        if (!h->ev_join) MXG_HIP(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        MXG_HIP(h, hipEventRecord(h->ev_join, h->stream2));
        MXG_HIP(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
        MXG_HIP(h, h->d_nmx.ensure(MXG_MAX_ASSEMBLIES * 4));
        unsigned char *base = static_cast<unsigned char *>(xp->d_slot);
        uint64_t off = xp->head_bytes;
        for (size_t i = 0; (xp->d_parts || xp->dg) && i < n; ++i)  // (parts: what was enqueued is packed already; the others say 0 / -1)
            if (state[i] != 1 && (rc = pack_part(i, h->stream, nullptr, 0)) != MXG_OK) return rc;
        for (size_t i = 0; !xp->d_parts && !xp->dg && i < n; ++i) {
            Assembly *a = list[i];
            const uint64_t cap = xp->caps[i];
            const uint64_t out_cap = std::min<uint64_t>({a->d_hash.bytes / 8, a->d_pos.bytes / 4, a->d_rec.bytes / 4, a->d_fwd.bytes});
            const long long fixed = state[i] == 1 ? -2ll : (state[i] == 2 ? 0ll : -1ll);
            const uint32_t grid = state[i] == 1 ? (uint32_t)std::max<uint64_t>((cap + 255) / 256, 1) : 1u;
            hipLaunchKernelGGL(k_pack_slot_dev, dim3(grid), dim3(256), 0, h->stream, a->d_hash.as<uint64_t>(),
                               a->d_pos.as<uint32_t>(), a->d_rec.as<uint32_t>(), h->d_nmx.as<uint32_t>() + i,
                               drvs[slot_of[i]]->sc(SC_CTRL).as<uint32_t>(), out_cap, cap, fixed,
                               reinterpret_cast<long long *>(base) + i, base + off, plans[i].dev_gaps ? 1u : 0u,
                               state[i] == 1 ? items[item0[i]].place4 : 0u);
            off += 16 * cap;
        }
        MXG_HIP(h, hipGetLastError());
        h->pend_list.assign(list, list + n);
        h->pend_state = state;
        h->pend_dev.assign(n, 0);
        for (size_t i = 0; i < n; ++i) h->pend_dev[i] = plans[i].dev_gaps ? 1 : 0;
        return MXG_OK;
    }
    bool fused = false;
    GraphBounds gb;
    if (fuse_graph && n_enq == n && n == h->asms.size() && n <= MXG_MAX_ASSEMBLIES) {
        bool same = true;
        for (size_t i = 0; i < n; ++i) same = same && list[i] == h->asms[i];
        if (same) {
            // the second stream joins the first; the graph stage follows the sketches on it
            if (!h->ev_join) MXG_HIP(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
            MXG_HIP(h, hipEventRecord(h->ev_join, h->stream2));
            MXG_HIP(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
            for (size_t i = 0; i < n; ++i) {
                Assembly *a = list[i];
                const uint64_t cap = std::min<uint64_t>({a->d_hash.bytes / 8, a->d_pos.bytes / 4, a->d_rec.bytes / 4, a->d_fwd.bytes});
                // (2 per window on i.i.d. sequence; repeat-rich sequence reaches ~3.8: what an earlier sketch of the assembly ended
                // with, 10 % on top, so that the fused graph survives on exactly the inputs the output arrays were widened for)
                const uint64_t iid = (uint64_t)(2.3 * (double)a->total_kmers / (double)(h->cfg.w + 1)) + 2048;
                gb.n_bound[i] = std::min<uint64_t>(cap, std::max<uint64_t>(iid, a->n_mx_seen + a->n_mx_seen / 10 + 2048));
                gb.n_ptr[i] = h->d_nmx.as<uint32_t>() + i;
            }
            fused = build_graph(h, GRAPH_FULL, nullptr, 0, &gb) == MXG_OK;  // (its sync is this call's sync)
        }
    }
    MXG_HIP(h, stream_wait(h->stream));
    MXG_HIP(h, stream_wait(h->stream2));
    for (hipStream_t sx : h->stream_x)
        if (sx) MXG_HIP(h, stream_wait(sx));
    size_t n_fast = 0;  // assemblies whose every batch ended the common way
    // what became of assembly i's batches: state 2 = its sketch is complete, 3 = enqueue it once more, 0 = synchronous path
    auto evaluate = [&](size_t i, bool final) -> int {
        Assembly *a = list[i];
        const size_t q0 = q_lo[i], q1 = q_hi[i];
        const uint64_t cap = std::min<uint64_t>({a->d_hash.bytes / 8, a->d_pos.bytes / 4, a->d_rec.bytes / 4, a->d_fwd.bytes});
        bool good = true;
        uint64_t total = 0, n_cand = 0, gap_kmers = 0;
        for (size_t q = q0; q < q1; ++q) {
            const uint32_t *c = items[q].hc;
            const uint64_t t = (uint64_t)c[6] | ((uint64_t)c[7] << 32);
            // (the one-call modes have already used the counts on the device: a stretch handed to the host undoes them)
            const bool dev = plans[i].dev_gaps && !(chain_modes && c[11] != 0 && c[11] != 0xFFFFFFFFu);
            // (not the device route: any stretch sends the batch to the general route below)
            // (a batch without any candidate: k_bs_select reports its contigs as stretches; the other route leaves it to the host)
            good = good && c[0] == 0 && c[3] == 0 && c[12] == 0 && (c[4] != 0 || items[q].bs) && c[4] != 0xFFFFFFFFu && (dev || c[1] == 0);
            total += t;
            n_cand += c[4];
            gap_kmers += c[10] == 0xFFFFFFFFu ? 0 : c[10];
            if (items[q].bs && c[15] != 0xFFFFFFFFu) h->stat_slice_stretches += c[15];
        }
        if (!(good && total <= cap) && knob_set(h, "MXG_DEBUG_BATCH")) {  // (diagnostics: the reports of an assembly's batches)
            for (size_t q = q0; q < q1; ++q) {
                const uint32_t *c = items[q].hc;
                fprintf(stderr, "[mxg] asm %zu batch %zu: ovf %u gaps %u sel %u flag %u cand %u nB %u total %u obase %u gapk %u (cap %llu)\n", i,
                        q - q0, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[8], c[10], (unsigned long long)cap);
            }
        }
        // what the batches saw of candidate-free stretches sizes the next sketch's batches (sparse_plan): real genomes hold
        // far more of them than the i.i.d. estimate (satellite arrays, low-complexity runs)
        for (size_t q = q0; q < q1; ++q) {
            const uint32_t *c = items[q].hc;
            // (1e-12: "a sketch has reported" -- an assembly whose stretches all went through k_sel_stretch leaves none over)
            if (c[1] != 0xFFFFFFFFu && items[q].g.nk) a->gap_rate_hint = std::max({a->gap_rate_hint, (double)c[1] / (double)items[q].g.nk, 1e-12});
        }
        if (good && total <= cap) {
            // stretches the device route left to the host (too long, too many minimizers, invalid bases inside)
            std::vector<uint4> deferred;
            for (size_t q = q0; q < q1 && !chain_modes; ++q) {
                const uint32_t nd = items[q].hc[11] == 0xFFFFFFFFu ? 0u : std::min(items[q].hc[11], GAP_DEFER_MAX);
                const uint4 *src = reinterpret_cast<const uint4 *>(h->pinned_defer) + (size_t)((items[q].hc - h->pinned_ctrl) / 16) * GAP_DEFER_MAX;
                deferred.insert(deferred.end(), src, src + nd);
            }
            uint64_t n_final = total;
            if (!deferred.empty()) {
                if ((rc = drv0.merge_deferred(a, tabs[i], deferred, total, &n_final)) != MXG_OK) return rc;
                h->stat_deferred += deferred.size();
                fused = false;  // (the graph stage ran on a sketch without them)
            }
            a->n_mx = n_final;
            a->has_sketch = true;
            h->stat_candidates += n_cand;
            h->stat_dense_kmers += gap_kmers;
            for (size_t q = q0; q < q1; ++q) a->cand_hints[q - q0] = items[q].hc[4];
            a->cand_hint = a->cand_hints[0];
            state[i] = 2;
            if (fused && total > gb.n_bound[i]) fused = false;  // a sketch outgrew the bound the graph stage was sized for
            ++n_fast;
        } else if (q1 - q0 == 1 && !items[q0].bs && items[q0].hc[0] == 0 && items[q0].hc[4] != 0xFFFFFFFFu &&
                   last_on_slot[items[q0].slot] == q0) {
            // one batch, no arena overflow, and its candidate arrays are still intact in the driver's scratch: finish from
            // there the general way (staging emit, dense fix-up of the stretches, merge) instead of redoing the batch
            Driver &drv = *drvs[items[q0].slot];
            OutArrays out{&a->d_hash, &a->d_pos, &a->d_rec, &a->d_fwd, 0};
            uint32_t ctrl_copy[16];
            memcpy(ctrl_copy, items[q0].hc, 64);
            if ((rc = drv.complete_batch(a, tabs[i], items[q0].g, out, ctrl_copy, items[q0].n_cap)) != MXG_OK) return rc;
            MXG_HIP(h, hipStreamSynchronize(drv.st));
            a->n_mx = out.n;
            a->has_sketch = true;
            state[i] = 2;
        } else if ((q1 - q0 > 1 || (items[q0].bs && plans[i].dev_gaps)) && !chain_modes && !final) {
            // several batches, first attempt: once more through the streams (a few ms per Gbp; the synchronous route costs
            // ten times that), with batches sized for the stretch density just seen (gap_rate_hint, above), every grid sized by
            // the batch's own candidate count where it reported one, and slices as large as the largest wave asked for
            for (size_t q = q0; q < q1; ++q) {
                const uint32_t *c = items[q].hc;
                if (c[0] != 0 && c[0] != 0xFFFFFFFFu) h->arena_cap_hint = std::max<uint64_t>(h->arena_cap_hint, (uint64_t)c[0] + 64);
            }
            // (k_bs_select again unless the slice kernel itself gave up somewhere -- a slice beyond its queue with no region left,
            // more selected candidates than a slice's room: word 12 -- and not merely more stretches than a batch holds)
            a->sel_again = true;
            for (size_t q = q0; q < q1; ++q)
                if (items[q].bs && items[q].hc[12] != 0) a->sel_again = false;
            a->cand_hints.clear();  // (the batches will be cut differently)
            a->cand_hint = 0xFFFFFFFEu;
            a->full_grid_once = true;
            ++h->stat_retries;
            state[i] = 3;
        } else if (q1 - q0 > 1 && !chain_modes) {
            // several batches: keep the leading ones that ended the common way AND lie where they belong (a batch starts
            // where its predecessors end), redo the first bad one and everything behind it batch by batch through the
            // synchronous route.  That route finishes stretches from the host, so it takes the threshold of the host route
            // (18 candidates per window: a stretch per ~10^8 k-mers instead of one per ~2 x 10^6).
            const bool dev = plans[i].dev_gaps;
            uint64_t offset = 0, nc = 0, gk = 0;
            size_t j = q0;
            for (; j < q1; ++j) {
                const uint32_t *c = items[j].hc;
                const uint64_t t = (uint64_t)c[6] | ((uint64_t)c[7] << 32), ob = (uint64_t)c[8] | ((uint64_t)c[9] << 32);
                const bool ok = c[0] == 0 && c[3] == 0 && c[12] == 0 && (c[4] != 0 || items[j].bs) && c[4] != 0xFFFFFFFFu && (dev || c[1] == 0) && ob == offset &&
                                offset + t <= cap;
                if (!ok) break;
                offset += t;
                nc += c[4];
                gk += c[10] == 0xFFFFFFFFu ? 0 : c[10];
                a->cand_hints[j - q0] = c[4];
            }
            for (size_t q = j; q < q1; ++q)  // (next time: the whole grid for what did not report, the count for what did)
                a->cand_hints[q - q0] = items[q].hc[4] != 0xFFFFFFFFu && items[q].hc[4] != 0 ? items[q].hc[4] : 0xFFFFFFFEu;
            a->cand_hint = a->cand_hints[0];
            SparsePlan rp = plans[i];
            if (!h->cfg.cand_per_window && 18.0 / (double)h->cfg.w <= 0.125) {
                rp.frac = 18.0 / (double)h->cfg.w;
                rp.tau_hi = std::max(2u, (uint32_t)std::min<double>(4294967294.0, rp.frac * 4294967296.0) & ~1u);
            }
            OutArrays out{&a->d_hash, &a->d_pos, &a->d_rec, &a->d_fwd, offset};
            h->stat_candidates += nc;
            h->stat_dense_kmers += gk;
            h->stat_batches_redone += q1 - j;
            if (j == q0) ++h->stat_sync_assemblies;
            // (the batches that are kept may have left stretches to the host: their lists are read before anything reuses the slots)
            std::vector<uint4> deferred;
            for (size_t q = q0; q < j; ++q) {
                const uint32_t nd = items[q].hc[11] == 0xFFFFFFFFu ? 0u : std::min(items[q].hc[11], GAP_DEFER_MAX);
                const uint4 *src = reinterpret_cast<const uint4 *>(h->pinned_defer) + (size_t)((items[q].hc - h->pinned_ctrl) / 16) * GAP_DEFER_MAX;
                deferred.insert(deferred.end(), src, src + nd);
            }
            if ((rc = drv0.sparse_all(a, tabs[i], out, rp.tau_hi, rp.frac, items[j].g.c0)) != MXG_OK) return rc;
            MXG_HIP(h, hipStreamSynchronize(drv0.st));
            uint64_t n_final = out.n;
            if (!deferred.empty()) {
                if ((rc = drv0.merge_deferred(a, tabs[i], deferred, out.n, &n_final)) != MXG_OK) return rc;
                h->stat_deferred += deferred.size();
            }
            a->n_mx = n_final;
            a->has_sketch = true;
            state[i] = 2;
        } else {
            state[i] = 0;  // redo synchronously
        }
        return MXG_OK;
    };
    for (size_t i = 0; i < n; ++i)
        if (state[i] == 1 && (rc = evaluate(i, chain_modes)) != MXG_OK) return rc;
    {
        std::vector<size_t> again;
        for (size_t i = 0; i < n; ++i)
            if (state[i] == 3) again.push_back(i);
        for (size_t i : again)
            if ((rc = enqueue_asm(i, 1)) != MXG_OK) return rc;
        for (Driver *od : drvs)
            if ((rc = od->flush_emit(nullptr)) != MXG_OK) return rc;
        if (!again.empty()) {
            MXG_HIP(h, stream_wait(h->stream));
            MXG_HIP(h, stream_wait(h->stream2));
            for (hipStream_t sx : h->stream_x)
                if (sx) MXG_HIP(h, stream_wait(sx));
        }
        for (size_t i : again)
            if (state[i] == 1 && (rc = evaluate(i, true)) != MXG_OK) return rc;
    }
    if (n_fast != n) fused = false;  // not the common case everywhere: the graph stage ran on incomplete input
    for (size_t i = 0; i < n; ++i) {
        if (state[i] != 0) continue;
        if (item0[i + 1] > item0[i]) {  // (was enqueued: the common route did not finish it)
            ++h->stat_sync_assemblies;
            h->stat_batches_redone += item0[i + 1] - item0[i];
        }
        if ((rc = run_sketch_sync(h, list[i], tabs[i], drv0)) != MXG_OK) return rc;
    }
    if ((rc = drv0.collect()) != MXG_OK) return rc;
    if (fuse_graph && !fused) {
        h->graph.valid = false;
        return build_graph(h);
    }
    return MXG_OK;
}

// second half of mxg_sketch_pack: after the stream has drained (the caller's sync on it), accept the sketches that ended
// the common way and redo the others through the synchronous path
int sketch_finish(mxg_handle *h)
{
    if (h->pend_list.empty()) return MXG_OK;
    MXG_HIP(h, hipSetDevice(h->device));
    MXG_HIP(h, stream_wait(h->stream));
    MXG_HIP(h, stream_wait(h->stream2));
    std::vector<Assembly *> list;
    std::vector<int> state;
    std::vector<unsigned char> devg;
    list.swap(h->pend_list);
    state.swap(h->pend_state);
    devg.swap(h->pend_dev);
    int rc;
    size_t q = 0;  // (one item per enqueued assembly, in order)
    for (size_t i = 0; i < list.size(); ++i) {
        Assembly *a = list[i];
        if (state[i] == 1) {
            const uint32_t *c = h->pinned_ctrl + 16 * q++;
            const uint64_t total = (uint64_t)c[6] | ((uint64_t)c[7] << 32), n_cand = c[4];
            const uint64_t cap = std::min<uint64_t>({a->d_hash.bytes / 8, a->d_pos.bytes / 4, a->d_rec.bytes / 4, a->d_fwd.bytes});
            // evaluate()'s `good` for a one-call mode, and what k_pack_slot_dev checked on the device: word 3 = the stretch kernels'
            // flag, word 12 = the slice kernel gave up on a slice (its output is then truncated), stretches absent or all placed on
            // the device (word 11 = handed to the host: the counts the pack kernel used would be without them)
            const bool dev = i < devg.size() && devg[i] && c[11] == 0;
            if (c[0] == 0 && c[3] == 0 && c[12] == 0 && (dev || c[1] == 0) && n_cand > 0 && c[4] != 0xFFFFFFFFu && total <= cap) {
                a->n_mx = total;
                a->has_sketch = true;
                h->stat_candidates += n_cand;
                a->cand_hint = (uint32_t)n_cand;
                continue;
            }
        } else if (state[i] == 2) {
            continue;  // (empty: prepare_sketch left it complete)
        }
        Tables T;
        bool empty = false;
        if ((rc = prepare_sketch(h, a, T, &empty)) != MXG_OK) return rc;
        if (empty) continue;
        Driver drv(h);
        if ((rc = run_sketch_sync(h, a, T, drv)) != MXG_OK) return rc;
    }
    Driver drv(h);
    return drv.collect();
}

int flush_timers(mxg_handle *h)
{
    if (h->ev_spans.empty()) return MXG_OK;
    MXG_HIP(h, hipSetDevice(h->device));
    MXG_HIP(h, hipStreamSynchronize(h->stream));
    MXG_HIP(h, hipStreamSynchronize(h->stream2));
    for (hipStream_t sx : h->stream_x)
        if (sx) MXG_HIP(h, hipStreamSynchronize(sx));
    for (auto &e : h->ev_spans) {
        float ms = 0;
        MXG_HIP(h, hipEventElapsedTime(&ms, e.a, e.b));
        if (e.is_hash) {
            h->tm.ms_hash += ms;
            h->tm.launches_hash += 1;
            h->tm.hash_bases += e.bases;
        } else {
            h->tm.ms_resolve += ms;
            if (e.kind == 2) h->tm.ms_reorder += ms;
            else if (e.kind == 3) h->tm.ms_resolve_k += ms;
            else if (e.kind == 4) h->tm.ms_emit += ms;
        }
    }
    h->ev_spans.clear();
    h->ev_used = 0;
    return MXG_OK;
}

int ensure_strand(mxg_handle *h, Assembly *a)
{
    if (a->fwd_valid || !a->has_sketch) return MXG_OK;
    MXG_HIP(h, hipSetDevice(h->device));
    MXG_HIP(h, a->d_fwd.ensure(std::max<uint64_t>(a->n_mx, 16)));
    if (a->has_bases && a->d_packed && a->n_mx && !a->foreign_sketch) {
        if (!a->d_rec_base.p) {
            std::vector<uint64_t> rb(a->recs.size());
            for (size_t r = 0; r < rb.size(); ++r) rb[r] = a->recs[r].base_off;
            int rc = upload(h, a->d_rec_base, rb);
            if (rc != MXG_OK) return rc;
            MXG_HIP(h, hipStreamSynchronize(h->stream));
        }
        hipLaunchKernelGGL(k_strand, dim3((uint32_t)((a->n_mx + 255) / 256)), dim3(256), 0, h->stream, a->d_packed,
                           a->d_rec_base.as<uint64_t>(), a->d_pos.as<uint32_t>(), a->d_rec.as<uint32_t>(), a->n_mx,
                           h->cfg.k, h->tab, a->d_fwd.as<uint8_t>());
        MXG_HIP(h, hipGetLastError());
    } else if (a->n_mx) {  // a sketch imported without bases and without strands: reported as forward
        MXG_HIP(h, hipMemsetAsync(a->d_fwd.p, 1, a->n_mx, h->stream));
    }
    MXG_HIP(h, hipStreamSynchronize(h->stream));
    a->fwd_valid = true;
    return MXG_OK;
}

// ---- exchange step of the multi-GPU path: pack / unpack of the all-gather buffer ---------------------------------
__global__ __launch_bounds__(256) void k_pack(const uint64_t *__restrict__ hash, const uint32_t *__restrict__ pos,
                                              const uint32_t *__restrict__ rec, uint64_t n, uint64_t nmax,
                                              unsigned char *__restrict__ buf)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    reinterpret_cast<uint64_t *>(buf)[i] = hash[i];
    reinterpret_cast<uint32_t *>(buf + 8 * nmax)[i] = pos[i];
    reinterpret_cast<uint32_t *>(buf + 12 * nmax)[i] = rec[i];
}

struct UnpackParams {
    const unsigned char *all;
    uint64_t nmax;
    uint64_t stride;       // bytes between two ranks' buffers
    uint32_t world;
    uint64_t start[65];    // exclusive prefix of counts (world <= 64)
    uint32_t rec_off[64];
    uint64_t *hash;
    uint32_t *pos, *rec;
};

__global__ __launch_bounds__(256) void k_unpack(const UnpackParams p)
{
    const uint64_t o = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (o >= p.start[p.world]) return;
    uint32_t r = 0;
    while (o >= p.start[r + 1]) ++r;  // world is small
    const uint64_t i = o - p.start[r];
    const unsigned char *buf = p.all + (size_t)r * p.stride;
    p.hash[o] = reinterpret_cast<const uint64_t *>(buf)[i];
    p.pos[o] = reinterpret_cast<const uint32_t *>(buf + 8 * p.nmax)[i];
    p.rec[o] = reinterpret_cast<const uint32_t *>(buf + 12 * p.nmax)[i] + p.rec_off[r];
}

int pack_sketch(mxg_handle *h, Assembly *a, void *d_buf, uint64_t nmax)
{
    if (!a->has_sketch) return set_err(h, MXG_EINVAL, "assembly '%s' has no sketch yet", a->name.c_str());
    if (nmax < a->n_mx || (nmax & 7)) return set_err(h, MXG_EINVAL, "nmax must be >= n and a multiple of 8");
    MXG_HIP(h, hipSetDevice(h->device));
    if (a->n_mx)
        hipLaunchKernelGGL(k_pack, dim3((uint32_t)((a->n_mx + 255) / 256)), dim3(256), 0, h->stream,
                           a->d_hash.as<uint64_t>(), a->d_pos.as<uint32_t>(), a->d_rec.as<uint32_t>(), a->n_mx, nmax,
                           static_cast<unsigned char *>(d_buf));
    MXG_HIP(h, hipGetLastError());
    // a handle with its own stream syncs: the caller hands the buffer to a collective on some other stream.  A handle
    // created on the CALLER's stream (mxg_config.stream) leaves the ordering to that stream.
    if (h->own_stream) MXG_HIP(h, hipStreamSynchronize(h->stream));
    return MXG_OK;
}

int unpack_gathered(mxg_handle *h, Assembly *a, const void *d_allbuf, uint32_t world, uint64_t nmax,
                    const uint64_t *counts, const uint64_t *rec_offsets, uint64_t stride_bytes)
{
    if (world == 0 || world > 64) return set_err(h, MXG_ELIMIT, "world size must be 1..64");
    MXG_HIP(h, hipSetDevice(h->device));
    UnpackParams up;
    up.all = static_cast<const unsigned char *>(d_allbuf);
    up.nmax = nmax;
    up.stride = stride_bytes ? stride_bytes : 16 * nmax;
    up.world = world;
    uint64_t total = 0;
    for (uint32_t r = 0; r < world; ++r) {
        if (counts[r] > nmax) return set_err(h, MXG_EINVAL, "counts[%u] exceeds nmax", r);
        up.start[r] = total;
        up.rec_off[r] = (uint32_t)rec_offsets[r];
        total += counts[r];
    }
    up.start[world] = total;
    MXG_HIP(h, a->d_hash.ensure(std::max<uint64_t>(total * 8, 16)));
    MXG_HIP(h, a->d_pos.ensure(std::max<uint64_t>(total * 4, 16)));
    MXG_HIP(h, a->d_rec.ensure(std::max<uint64_t>(total * 4, 16)));
    up.hash = a->d_hash.as<uint64_t>();
    up.pos = a->d_pos.as<uint32_t>();
    up.rec = a->d_rec.as<uint32_t>();
    if (total) hipLaunchKernelGGL(k_unpack, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, h->stream, up);
    MXG_HIP(h, hipGetLastError());
    a->n_mx = total;
    a->has_sketch = true;
    a->fwd_valid = false;   // strands do not travel; a gathered sketch reports '+'
    a->foreign_sketch = true;
    a->host_valid = false;
    a->flags_valid = false;
    h->graph.valid = false;
    return MXG_OK;
}

// ---- steady-state exchange with device-side counts -------------------------------------------------------------------
// A rank's slot: [header: one int64 per assembly = its minimizer count, -1: does not fit | region of assembly 0 (caps[0]
// entries laid out as k_pack does: hash | pos | rec) | region of assembly 1 | ...].  After the all-gather every rank
// unpacks all slots with the counts read from the headers ON THE DEVICE and runs the graph stage behind it with upper
// bounds (GraphBounds): the whole exchange + graph step has ONE host sync (build_graph's).
__global__ __launch_bounds__(256) void k_pack_slot(const uint64_t *__restrict__ hash, const uint32_t *__restrict__ pos,
                                                   const uint32_t *__restrict__ rec, uint64_t n, uint64_t cap, long long count,
                                                   long long *header, unsigned char *__restrict__ region)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i == 0) *header = count;
    if (i >= n) return;
    reinterpret_cast<uint64_t *>(region)[i] = hash[i];
    reinterpret_cast<uint32_t *>(region + 8 * cap)[i] = pos[i];
    reinterpret_cast<uint32_t *>(region + 12 * cap)[i] = rec[i];
}

int xchg_pack(mxg_handle *h, void *d_slot, uint64_t head_bytes, const uint64_t *caps)
{
    MXG_HIP(h, hipSetDevice(h->device));
    unsigned char *base = static_cast<unsigned char *>(d_slot);
    uint64_t off = head_bytes;
    for (size_t ai = 0; ai < h->asms.size(); ++ai) {
        Assembly *a = h->asms[ai];
        if (!a->has_sketch) return set_err(h, MXG_EINVAL, "assembly '%s' has no sketch yet", a->name.c_str());
        if (caps[ai] & 7) return set_err(h, MXG_EINVAL, "slot capacities must be multiples of 8");
        const bool fits = a->n_mx <= caps[ai];
        const uint64_t n = fits ? a->n_mx : 0;
        hipLaunchKernelGGL(k_pack_slot, dim3((uint32_t)std::max<uint64_t>((n + 255) / 256, 1)), dim3(256), 0, h->stream,
                           a->d_hash.as<uint64_t>(), a->d_pos.as<uint32_t>(), a->d_rec.as<uint32_t>(), n, caps[ai],
                           fits ? (long long)a->n_mx : -1ll, reinterpret_cast<long long *>(base) + ai, base + off);
        off += 16 * caps[ai];
    }
    MXG_HIP(h, hipGetLastError());
    if (h->own_stream) MXG_HIP(h, hipStreamSynchronize(h->stream));
    return MXG_OK;
}

struct UnpackSlotParams {
    const unsigned char *all;  // world slots
    uint64_t slot_bytes, region_off, cap;
    uint32_t world, a;
    uint32_t rec_off[64];
    uint64_t *hash;
    uint32_t *pos, *rec;
    uint32_t *n_dev;        // device word that receives the total (GraphBounds::n_ptr)
    uint32_t *host_total;   // pinned: total, 0xFFFFFFFF if some rank's header says "does not fit"
};

__global__ __launch_bounds__(256) void k_unpack_slot(const UnpackSlotParams p)
{
    const uint32_t r = blockIdx.y;
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    uint64_t start = 0, total = 0;
    bool bad = false;
    long long mine = 0;
    for (uint32_t q = 0; q < p.world; ++q) {  // world is small; the headers sit in L2
        const long long c = reinterpret_cast<const long long *>(p.all + (size_t)q * p.slot_bytes)[p.a];
        bad = bad || c < 0 || (uint64_t)c > p.cap;
        if (q < r) start += (uint64_t)max(c, 0ll);
        if (q == r) mine = c;
        total += (uint64_t)max(c, 0ll);
    }
    if (r == 0 && i == 0) {
        *p.n_dev = bad ? 0u : (uint32_t)total;
        *p.host_total = bad ? 0xFFFFFFFFu : (uint32_t)total;
    }
    if (bad || (long long)i >= mine) return;
    const unsigned char *reg = p.all + (size_t)r * p.slot_bytes + p.region_off;
    const uint64_t o = start + i;
    p.hash[o] = reinterpret_cast<const uint64_t *>(reg)[i];
    p.pos[o] = reinterpret_cast<const uint32_t *>(reg + 8 * p.cap)[i];
    p.rec[o] = reinterpret_cast<const uint32_t *>(reg + 12 * p.cap)[i] + p.rec_off[r];
}

// the same out of parts (k_pack_part_dev): 12 bytes per entry, the record found by bisection in the sender's table of first entries
struct UnpackPartParams {
    const unsigned char *all;  // world parts of ONE assembly
    uint64_t part_bytes, cap, rcap;
    uint32_t world;
    uint32_t rec_off[64];
    uint64_t *hash;
    uint32_t *pos, *rec;
    uint32_t *n_dev, *host_total;
};

__global__ __launch_bounds__(256) void k_unpack_part(const UnpackPartParams p)
{
    const uint32_t r = blockIdx.y;
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    uint64_t start = 0, total = 0;
    bool bad = false;
    long long mine = 0, my_recs = 0;
    for (uint32_t q = 0; q < p.world; ++q) {
        const long long *hd = reinterpret_cast<const long long *>(p.all + (size_t)q * p.part_bytes);
        const long long c = hd[0];
        bad = bad || c < 0 || (uint64_t)c > p.cap || (c > 0 && (hd[1] <= 0 || (uint64_t)hd[1] > p.rcap));
        if (q < r) start += (uint64_t)max(c, 0ll);
        if (q == r) {
            mine = c;
            my_recs = hd[1];
        }
        total += (uint64_t)max(c, 0ll);
    }
    if (r == 0 && i == 0) {
        *p.n_dev = bad ? 0u : (uint32_t)total;
        *p.host_total = bad ? 0xFFFFFFFFu : (uint32_t)total;
    }
    if (bad || (long long)i >= mine) return;
    const unsigned char *reg = p.all + (size_t)r * p.part_bytes + XCHG_PART_HEAD;
    const uint32_t *starts = reinterpret_cast<const uint32_t *>(reg + 12 * p.cap);
    uint32_t lo = 0, hi = (uint32_t)my_recs;  // the last record whose first entry is <= i (records without entries share the next one's)
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (starts[mid] <= (uint32_t)i) lo = mid + 1;
        else hi = mid;
    }
    const uint64_t o = start + i;
    p.hash[o] = reinterpret_cast<const uint64_t *>(reg)[i];
    p.pos[o] = reinterpret_cast<const uint32_t *>(reg + 8 * p.cap)[i];
    p.rec[o] = (lo ? lo - 1u : 0u) + p.rec_off[r];
}

// returns 1 (nothing usable: some sketch did not fit its slot on some rank, the caller exchanges sizes first) or the
// result of build_graph
int xchg_unpack_graph(mxg_handle *h, const void *d_all, uint32_t world, uint64_t slot_bytes, uint64_t head_bytes,
                      const uint64_t *caps, const uint64_t *rec_offsets, const void *const *d_all_parts, const uint64_t *rcaps)
{
    if (world == 0 || world > 64) return set_err(h, MXG_ELIMIT, "world size must be 1..64");
    const size_t A = h->asms.size();
    if (A == 0 || A > MXG_MAX_ASSEMBLIES) return set_err(h, MXG_EINVAL, "mxg_xchg_unpack_graph: 1..%d assemblies", MXG_MAX_ASSEMBLIES);
    MXG_HIP(h, hipSetDevice(h->device));
    {
        const int rcp = ensure_pinned_ctrl(h);
        if (rcp != MXG_OK) return rcp;
    }
    MXG_HIP(h, h->d_nmx.ensure(MXG_MAX_ASSEMBLIES * 4));
    GraphBounds gb;
    uint64_t off = head_bytes;
    for (size_t ai = 0; ai < A; ++ai) {
        Assembly *a = h->asms[ai];
        const uint64_t bound = (uint64_t)world * caps[ai];
        if (bound >= (1ull << 32)) return set_err(h, MXG_ELIMIT, "exchange slots too large");
        MXG_HIP(h, a->d_hash.ensure(std::max<uint64_t>(bound * 8, 16)));
        MXG_HIP(h, a->d_pos.ensure(std::max<uint64_t>(bound * 4, 16)));
        MXG_HIP(h, a->d_rec.ensure(std::max<uint64_t>(bound * 4, 16)));
        h->pinned_ctrl[16 * ai] = 0xFFFFFFFFu;
        if (d_all_parts) {
            UnpackPartParams pp;
            pp.all = static_cast<const unsigned char *>(d_all_parts[ai]);
            pp.cap = caps[ai];
            pp.rcap = rcaps[ai];
            pp.part_bytes = XCHG_PART_HEAD + 12 * caps[ai] + 4 * rcaps[ai];
            pp.world = world;
            for (uint32_t r = 0; r < world; ++r) pp.rec_off[r] = (uint32_t)rec_offsets[ai * world + r];
            pp.hash = a->d_hash.as<uint64_t>();
            pp.pos = a->d_pos.as<uint32_t>();
            pp.rec = a->d_rec.as<uint32_t>();
            pp.n_dev = h->d_nmx.as<uint32_t>() + ai;
            pp.host_total = h->pinned_ctrl + 16 * ai;
            hipLaunchKernelGGL(k_unpack_part, dim3((uint32_t)std::max<uint64_t>((caps[ai] + 255) / 256, 1), world), dim3(256), 0,
                               h->stream, pp);
        }
        UnpackSlotParams up;
        up.all = static_cast<const unsigned char *>(d_all);
        up.slot_bytes = slot_bytes;
        up.region_off = off;
        up.cap = caps[ai];
        up.world = world;
        up.a = (uint32_t)ai;
        for (uint32_t r = 0; r < world; ++r) up.rec_off[r] = (uint32_t)rec_offsets[ai * world + r];
        up.hash = a->d_hash.as<uint64_t>();
        up.pos = a->d_pos.as<uint32_t>();
        up.rec = a->d_rec.as<uint32_t>();
        up.n_dev = h->d_nmx.as<uint32_t>() + ai;
        up.host_total = h->pinned_ctrl + 16 * ai;
        if (!d_all_parts)
            hipLaunchKernelGGL(k_unpack_slot, dim3((uint32_t)std::max<uint64_t>((caps[ai] + 255) / 256, 1), world), dim3(256), 0,
                               h->stream, up);
        off += 16 * caps[ai];
        gb.n_bound[ai] = bound;
        gb.n_ptr[ai] = h->d_nmx.as<uint32_t>() + ai;
        a->has_sketch = true;
        a->n_mx = 0;  // (known after the sync below)
        a->fwd_valid = false;
        a->foreign_sketch = true;
        a->host_valid = false;
        a->flags_valid = false;
    }
    MXG_HIP(h, hipGetLastError());
    h->graph.valid = false;
    const int rc = build_graph(h, GRAPH_FULL, nullptr, 0, &gb);  // its sync is the exchange's sync
    if (rc != MXG_OK) return rc;
    bool bad = false;
    for (size_t ai = 0; ai < A; ++ai) {
        const uint32_t t = h->pinned_ctrl[16 * ai];
        bad = bad || t == 0xFFFFFFFFu;
        h->asms[ai]->n_mx = t == 0xFFFFFFFFu ? 0 : t;
    }
    if (bad) {
        h->graph.valid = false;
        return 1;
    }
    return MXG_OK;
}

int sync_sketch_to_host(mxg_handle *h, Assembly *a)
{
    if (!a->has_sketch) return set_err(h, MXG_EINVAL, "assembly '%s' has no sketch yet (call mxg_sketch)", a->name.c_str());
    if (a->host_valid) return MXG_OK;
    int rcs = ensure_strand(h, a);
    if (rcs != MXG_OK) return rcs;
    MXG_HIP(h, hipSetDevice(h->device));
    a->h_hash.resize(a->n_mx);
    a->h_pos.resize(a->n_mx);
    a->h_rec.resize(a->n_mx);
    a->h_fwd.resize(a->n_mx);
    if (a->n_mx) {
        MXG_HIP(h, hipMemcpyAsync(a->h_hash.data(), a->d_hash.p, a->n_mx * 8, hipMemcpyDeviceToHost, h->stream));
        MXG_HIP(h, hipMemcpyAsync(a->h_pos.data(), a->d_pos.p, a->n_mx * 4, hipMemcpyDeviceToHost, h->stream));
        MXG_HIP(h, hipMemcpyAsync(a->h_rec.data(), a->d_rec.p, a->n_mx * 4, hipMemcpyDeviceToHost, h->stream));
        MXG_HIP(h, hipMemcpyAsync(a->h_fwd.data(), a->d_fwd.p, a->n_mx, hipMemcpyDeviceToHost, h->stream));
        MXG_HIP(h, hipStreamSynchronize(h->stream));
    }
    build_rec_first(a);
    a->host_valid = true;
    return MXG_OK;
}

}  // namespace mxg
