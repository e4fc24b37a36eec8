// sketch_bs.hip -- the k = 32 route of the sketch stage (replaces `indexlr`, reference ntJoin:204-205; semantics SURVEY.md App. A).
//
//   k_hash_bs       (bs_kernels.h)  the bit-sliced ring filter over the whole assembly, straight from the 2-bit packed bases (bit
//                                   planes are made in registers): one bit per base position, "the 32-mer starting here may have
//                                   canonical hash < tau" (a superset)
//   k_bs_resolve    (here)          one block per chunk of 65 536 positions: the filter's bits of the chunk and of a halo on
//                                   either side -> candidates in position order -> valid ones (run table) with their exact
//                                   64-bit hashes < tau -> the window decision of k_resolve (sketch.hip) on the block's own
//                                   candidates, with every neighbour it can need inside the halo -> the selected ones laid
//                                   out per block for k_emit, candidate-free stretches for k_gap_fix.
// The halo: w - 1 k-mers to the left (the window decision), the longest stretch the device route sketches (GAP_DEV_NMAX) plus
// w to the right (a stretch is reported by the candidate in front of it and must end inside the block's range; a longer one
// sends the batch to the general route, as it does in sketch.hip).  A block whose range the run table describes in more than
// BSR_RUNS runs, or that finds more candidates than its LDS holds, also sends the batch there (ctrl[6]).
#include <algorithm>

#include "bs_kernels.h"
#include "nthash_dev.h"
#include "scan_kernels.h"
#include "sketch_bs.h"

namespace mxg {

namespace {

__device__ __forceinline__ uint32_t lds_base2(const uint32_t *pk, uint32_t rel)  // 16 bases from LDS words, any alignment
{
    const uint32_t wi = rel >> 4, sh = (rel & 15u) * 2u;
    return __builtin_amdgcn_alignbit(pk[wi + 1], pk[wi], sh);
}

// canonical hash (fwd + rev) of the 32-mer at LDS base index rel: half position tables (init32_half, nthash_dev.h:
// half[j][v] = {srol^{4(3-j)} f4[v], srol^{4j} r4[v]}; bytes 0..3 and 4..7 of the k-mer through the same four tables, joined by
// two rotations by 16) -- 16 KB of LDS where the full position tables take 32 KB and a second block per CU
__device__ __forceinline__ uint64_t hash32_lds(const uint32_t *pk, uint32_t rel, const uint4 *half)
{
    const uint32_t wi = rel >> 4, sh = (rel & 15u) * 2u;
    const uint32_t w0 = pk[wi], w1 = pk[wi + 1], w2 = pk[wi + 2];
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
    return (((uint64_t)(a.y ^ c.y) << 32) | (a.x ^ c.x)) + (((uint64_t)(a.w ^ c.w) << 32) | (a.z ^ c.z));
}

// exclusive prefix of v over the block's threads (16 waves); total = the block's sum.  sh: 40 words; two barriers; a following
// call may start at once.  (block_exclusive of scan_kernels.h has every thread walk the 16 wave totals: ~120 instructions per
// wave where this takes ~40, and the kernel is bound by instruction issue.)
__device__ __forceinline__ uint32_t bsr_scan(uint32_t v, uint32_t *sh, uint32_t &total)
{
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t incl = wave_inclusive_u32(v, lane);
    if (lane == 63u) sh[wv] = incl;
    __syncthreads();
    if (wv == 0 && lane < 16u) {
        const uint32_t x = sh[lane];
        uint32_t inc = x;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 16);
            if (lane >= (uint32_t)o) inc += t;
        }
        sh[16u + lane] = inc - x;
        if (lane == 15u) sh[32] = inc;
    }
    __syncthreads();
    total = sh[32];
    return sh[16u + wv] + incl - v;
}
// the same for one flag per thread (rank inside the wave from the ballot)
__device__ __forceinline__ uint32_t bsr_scan_flag(bool f, uint32_t *sh, uint32_t &total)
{
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t m = __ballot(f);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (lane == 0u) sh[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    if (wv == 0 && lane < 16u) {
        const uint32_t x = sh[lane];
        uint32_t inc = x;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 16);
            if (lane >= (uint32_t)o) inc += t;
        }
        sh[16u + lane] = inc - x;
        if (lane == 15u) sh[32] = inc;
    }
    __syncthreads();
    total = sh[32];
    return sh[16u + wv] + rank;
}

constexpr uint32_t BSR_PAD = 4;  // sentinel entries on either side of the candidates (the scans look at four at a time)
struct BsLds {
    uint4 *btab;        // [1024] half position tables of the direct hash formula
    uint32_t *pk;       // packed words of the block's range (+ 3)
    uint32_t *nat;      // position-order bitmap of the range: word g = strip g (32 positions); lies over cand (dead before)
    uint32_t *posl;     // candidates: position relative to the range start
    uint4 *cand;        // candidates of the range in order: {k-mer index, contig, hash lo, hash hi}; [-PAD, n + PAD)
    uint32_t *cnk;      // ... the k-mer count of the candidate's contig
    Run *runs;          // [BSR_RUNS]
    uint32_t *rnk;      // [BSR_RUNS] k-mer count of the run's contig
    uint32_t *sh;       // [256 + 8] scan scratch
};

}  // namespace

// LDS of a block: half position tables | cand | posl | cnk | pk | runs | rnk | scan scratch; the position-order bitmap lies over cand
// (it is dead before that is written), which max_cand must be large enough for (the host sizes max_cand)
static __host__ __device__ inline uint32_t bsr_range_cap(const BsResolveParams &p) { return BS_CHUNK + (p.halo_l + p.halo_r) * BSR_HALO_LANE; }
size_t bs_resolve_lds(const BsResolveParams &p)
{
    const size_t rc = bsr_range_cap(p);
    return 1024 * 16 + ((size_t)p.max_cand + 2 * BSR_PAD) * 16 + (size_t)p.max_cand * 8 + (rc / 16 + 8) * 4 + BSR_RUNS * (sizeof(Run) + 4) +
           (256 + 8) * 4;
}

// (profiling builds: the block stops after phase n but still reports, so that the batch ends the common way)
#define BSR_STAMP(k)                                                    \
    if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 16u + (k)] = __builtin_readcyclecounter();
#define BSR_ABLATE(n)                                                   \
    if (p.ablate == (n)) {                                              \
        if (threadIdx.x == 0) {                                         \
            count_publish(p.cnt, p.sup, blockIdx.x, 0u);                \
            atomicAdd(&p.cand_spread[(blockIdx.x & 63u) * 32u], 1u);    \
        }                                                               \
        return;                                                         \
    }

__global__ __launch_bounds__(BSR_THREADS, 8) void k_bs_resolve(const BsResolveParams p)
{
    extern __shared__ uint4 lds_raw[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t c = p.chunk_lo + blockIdx.x;  // the chunk whose OUT words this block owns
    // positions: the chunk's OUT words cover strips -1 .. 2046 of the chunk
    const int64_t core_lo = (int64_t)c * BS_CHUNK - 32, core_hi = core_lo + BS_CHUNK;
    const int64_t n_pos = (int64_t)p.n_chunks * BS_CHUNK;
    const int64_t range_lo = std::max<int64_t>(0, core_lo - (int64_t)p.halo_l * BSR_HALO_LANE);
    const int64_t range_hi = std::min<int64_t>(n_pos, core_hi + (int64_t)p.halo_r * BSR_HALO_LANE);
    const uint32_t n_strips = (uint32_t)((range_hi - range_lo) >> 5);  // (range_lo, range_hi are multiples of 32)
    const uint32_t range_cap = bsr_range_cap(p);
    // ---- LDS carving (see bs_resolve_lds)
    BsLds L;
    unsigned char *bp = reinterpret_cast<unsigned char *>(lds_raw);
    L.btab = reinterpret_cast<uint4 *>(bp); bp += 1024 * 16;
    L.nat = reinterpret_cast<uint32_t *>(bp);
    L.cand = reinterpret_cast<uint4 *>(bp) + BSR_PAD; bp += ((size_t)p.max_cand + 2 * BSR_PAD) * 16;
    L.posl = reinterpret_cast<uint32_t *>(bp); bp += (size_t)p.max_cand * 4;
    L.cnk = reinterpret_cast<uint32_t *>(bp); bp += (size_t)p.max_cand * 4;
    L.pk = reinterpret_cast<uint32_t *>(bp); bp += (range_cap / 16 + 8) * 4;
    L.runs = reinterpret_cast<Run *>(bp); bp += BSR_RUNS * sizeof(Run);
    L.rnk = reinterpret_cast<uint32_t *>(bp); bp += BSR_RUNS * 4;
    L.sh = reinterpret_cast<uint32_t *>(bp);
    __shared__ uint32_t s_nwin, s_flag, s_nraw, s_ncand;
    __shared__ uint32_t s_klo_ctg, s_klo, s_khi_ctg, s_khi;  // the contig cut by the range's start / end and its k-mer there
    if (tid == 0) { s_flag = 0; s_nwin = 0; }
    BSR_STAMP(0)
    // ---- loads.  Everything the block reads from global memory is requested at once, by role: wave 0 = the chunk's own 64
    // lanes of OUT words, the first threads of wave 1 = the halo lanes (the last lanes of the chunks before, the first of the
    // chunks behind), every thread a share of the packed bases and of the position tables, the first BSR_RUNS + 32 a run.
    const uint32_t rc = (uint32_t)(range_lo / BS_CHUNK);
    const uint32_t run0_c = p.chunk_run0[rc];
    uint4 bt0;  // half tables (see init32_half): entry tid = table tid / 256, byte value tid % 256
    {
        const uint32_t j = tid >> 8, vb = tid & 255u;
        const uint4 f = p.init_tab[256u + (j + 4u) * 256u + vb], r = p.init_tab[256u + j * 256u + vb];
        bt0 = make_uint4(f.x, f.y, r.z, r.w);
    }
    const uint64_t w0 = (uint64_t)range_lo >> 4;            // (a multiple of 2: range_lo is a multiple of 32)
    const uint32_t nw = (uint32_t)((range_hi - range_lo) >> 4) + 3u;
    const int64_t strip0 = range_lo >> 5;
    // the filter's words of the range: a plain bitmap, word = strip (32 positions)
    constexpr uint32_t NWV3 = (3072u + BSR_THREADS - 1u) / BSR_THREADS;  // n_strips <= (65536 + 32 * 1024) / 32
    uint32_t natw[NWV3];
#pragma unroll
    for (uint32_t u = 0; u < NWV3; ++u) {
        const uint32_t g = tid + u * BSR_THREADS;
        natw[u] = g < n_strips ? p.out[strip0 + g] : 0u;
    }
    // the packed words of the range, four per load
    constexpr uint32_t PKV = (1536u + BSR_THREADS - 1u) / BSR_THREADS;
    uint4 pkv[PKV];
#pragma unroll
    for (uint32_t u = 0; u < PKV; ++u) {
        const uint32_t i = (tid + u * BSR_THREADS) * 4u;
        pkv[u] = make_uint4(0u, 0u, 0u, 0u);
        if (i < nw) {
            const uint64_t wq = w0 + i;
            if (wq + 4 <= p.n_words) {
                const uint2 lo = *reinterpret_cast<const uint2 *>(p.packed + wq), hi = *reinterpret_cast<const uint2 *>(p.packed + wq + 2);
                pkv[u] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            } else {
                if (wq < p.n_words) pkv[u].x = p.packed[wq];
                if (wq + 1 < p.n_words) pkv[u].y = p.packed[wq + 1];
                if (wq + 2 < p.n_words) pkv[u].z = p.packed[wq + 2];
            }
        }
    }
    // run window: the runs that overlap the range, of this batch only; every thread looks at one run
    uint32_t r_first = std::max(run0_c, p.run_lo);
    Run my_run{};
    const bool my_run_on = tid < BSR_RUNS + 32u && r_first + tid < p.run_hi;
    if (my_run_on) my_run = p.runs[r_first + tid];
    const uint32_t my_nk = my_run_on ? p.ctg_nk[my_run.contig] : 0u;
    uint64_t beyond_off = ~0ull;  // does the run behind the ones looked at still start inside the range?
    if (tid == 0 && r_first + BSR_RUNS + 32u < p.run_hi) beyond_off = p.runs[r_first + BSR_RUNS + 32u].base_off;
    BSR_STAMP(1)
    // ---- into LDS
    L.btab[tid] = bt0;
#pragma unroll
    for (uint32_t u = 0; u < PKV; ++u) {
        const uint32_t i = (tid + u * BSR_THREADS) * 4u;
        if (i < nw) {
            L.pk[i] = pkv[u].x; L.pk[i + 1] = pkv[u].y; L.pk[i + 2] = pkv[u].z; L.pk[i + 3] = pkv[u].w;
        }
    }
#pragma unroll
    for (uint32_t u = 0; u < NWV3; ++u) {
        const uint32_t g = tid + u * BSR_THREADS;
        if (g < n_strips) L.nat[g] = natw[u];
    }
    {
        // (chunk_run0 is exact for a chunk's first position; the range starts behind it: the runs that end before it are
        // skipped here)
        bool in = false;
        const Run &run = my_run;
        if (my_run_on) in = (int64_t)(run.base_off + run.n_kmers) > range_lo && (int64_t)run.base_off < range_hi;
        // runs are sorted by position: the overlapping ones are consecutive
        const uint64_t mb = __ballot(in);
        constexpr uint32_t NWV = BSR_THREADS / 64u;
        __shared__ uint32_t s_cnt[NWV], s_first[NWV];
        if (lane == 0) {
            s_cnt[tid >> 6] = (uint32_t)__popcll(mb);
            s_first[tid >> 6] = mb ? (uint32_t)__builtin_ctzll(mb) + (tid & ~63u) : 0xFFFFFFFFu;
        }
        __syncthreads();
        uint32_t first = 0xFFFFFFFFu, total = 0;
#pragma unroll
        for (uint32_t u = 0; u < 3u && u < NWV; ++u) {  // (the runs looked at sit in the first BSR_RUNS + 32 threads)
            first = std::min(first, s_first[u]);
            total += s_cnt[u];
        }
        if (in) {
            const uint32_t at = tid - first;
            if (at < BSR_RUNS) {
                L.runs[at] = run;
                L.rnk[at] = my_nk;
            }
        }
        if (tid == 0) {
            s_nwin = std::min(total, BSR_RUNS);
            // more runs than one pass of the block sees (or than the window holds): the general route takes the batch
            if (total > BSR_RUNS || (int64_t)std::min<uint64_t>(beyond_off, (uint64_t)INT64_MAX) < range_hi) s_flag = 1;
        }
    }
    __syncthreads();
    BSR_STAMP(2)
    BSR_ABLATE(1)
    const uint32_t n_win = s_nwin;
    BSR_STAMP(3)
    BSR_ABLATE(2)
    // ---- positions of the set bits, in order: thread t takes the strips [4 t, 4 t + 4)
    {
        uint32_t wd[4], cnt = 0;
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            const uint32_t g = 4u * tid + u;
            wd[u] = g < n_strips ? L.nat[g] : 0u;
            cnt += (uint32_t)__popc(wd[u]);
        }
        uint32_t total;
        uint32_t at = bsr_scan(cnt, L.sh, total);
        if (tid == 0) {
            s_nraw = std::min(total, p.max_cand);
            if (total > p.max_cand) s_flag = 1;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            uint32_t word = wd[u];
            while (word) {
                const uint32_t t = (uint32_t)__builtin_ctz(word);
                word &= word - 1u;
                if (at < p.max_cand) L.posl[at] = (4u * tid + u) * 32u + t;
                ++at;
            }
        }
    }
    __syncthreads();
    BSR_STAMP(4)
    BSR_ABLATE(3)
    const uint32_t n_raw = s_nraw;
    // ---- valid ones with their exact hashes; thread t takes the raw candidates [t * ipt, (t + 1) * ipt)
    constexpr uint32_t IPT_MAX = 3072u / BSR_THREADS;  // max_cand <= 3072
    const uint32_t ipt = (n_raw + BSR_THREADS - 1u) / BSR_THREADS;
    uint64_t vh[IPT_MAX];
    uint32_t vk[IPT_MAX], vc[IPT_MAX], vp[IPT_MAX], vn[IPT_MAX];
    uint32_t keep = 0;
#pragma unroll
    for (uint32_t u = 0; u < IPT_MAX; ++u) {
        const uint32_t i = tid * ipt + u;
        vh[u] = 0; vk[u] = 0; vc[u] = 0; vp[u] = 0; vn[u] = 0;
        if (u < ipt && i < n_raw) {
            const uint32_t rel = L.posl[i];
            vp[u] = rel;
            const uint64_t pos = (uint64_t)range_lo + rel;
            // the run holding the position: last run of the window with base_off <= pos
            uint32_t lo = 0, hi = n_win;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (L.runs[mid].base_off <= pos) lo = mid + 1; else hi = mid;
            }
            if (lo > 0) {
                const Run &run = L.runs[lo - 1];
                if (pos - run.base_off < run.n_kmers) {
                    const uint64_t h = hash32_lds(L.pk, rel, L.btab);
                    if (h < p.tau) {
                        vh[u] = h;
                        vk[u] = run.kidx0 + (uint32_t)(pos - run.base_off);
                        vc[u] = run.contig;
                        vn[u] = L.rnk[lo - 1];
                        keep |= 1u << u;
                    }
                }
            }
        }
    }
    {
        uint32_t total;
        uint32_t at = bsr_scan((uint32_t)__popc(keep), L.sh, total);
        if (tid == 0) s_ncand = total;
        // (behind the scan's barriers the bitmap and the raw positions are dead: cand / posl / cnk may be written)
#pragma unroll
        for (uint32_t u = 0; u < IPT_MAX; ++u)
            if ((keep >> u) & 1u) {
                L.cand[at] = make_uint4(vk[u], vc[u], (uint32_t)vh[u], (uint32_t)(vh[u] >> 32));
                L.posl[at] = vp[u];  // (position relative to the range, at the candidate's new index)
                L.cnk[at] = vn[u];
                ++at;
            }
    }
    // the k-mer index at which the range cuts a contig (first / last run of the window)
    if (tid == 0) {
        s_klo_ctg = 0xFFFFFFFFu; s_klo = 0; s_khi_ctg = 0xFFFFFFFFu; s_khi = 0;
        if (n_win) {
            const Run &a = L.runs[0];
            const uint32_t kl = (int64_t)a.base_off >= range_lo ? a.kidx0 : a.kidx0 + (uint32_t)(range_lo - (int64_t)a.base_off);
            if (kl > 0) { s_klo_ctg = a.contig; s_klo = kl; }
            const Run &b = L.runs[n_win - 1];
            const uint32_t kh = b.kidx0 + (uint32_t)std::min<int64_t>(b.n_kmers, range_hi - (int64_t)b.base_off) - 1u;
            if (kh + 1u < L.rnk[n_win - 1]) { s_khi_ctg = b.contig; s_khi = kh; }
        }
    }
    __syncthreads();
    BSR_STAMP(5)
    BSR_ABLATE(4)
    const uint32_t n_c = s_ncand;
    if (tid < BSR_PAD) {  // sentinels: a contig no candidate has
        L.cand[-1 - (int)tid] = make_uint4(0u, 0xFFFFFFFFu, 0u, 0u);
        L.cand[n_c + tid] = make_uint4(0u, 0xFFFFFFFFu, 0u, 0u);
    }
    __syncthreads();
    const uint32_t w = p.w, wm1 = w - 1u;
    // (chunk 0: positions -32..-1 do not exist)
    const uint32_t core_rel_lo = (uint32_t)(std::max(core_lo, range_lo) - range_lo);
    const uint32_t core_rel_hi = (uint32_t)(std::min(core_hi, range_hi) - range_lo);
    // ---- the window decision on the block's own candidates: k_resolve's scans (sketch.hip), everything in LDS.  Pass 1: every
    // candidate looks at its four neighbours on either side (one ds_read_b128 each), which decides four in five; pass 2: the
    // undecided ones, laid end to end, walk on four neighbours at a time with full waves.  The flags go to LDS (the raw
    // position list is dead: its first bytes are reused); the selected ones then leave in order.
    uint8_t *selb = reinterpret_cast<uint8_t *>(L.pk);            // [max_cand] 0 / 1 (the packed words are dead behind the hashes)
    uint32_t *undl = reinterpret_cast<uint32_t *>(L.pk) + (p.max_cand / 4u + 4u);  // undecided candidates: index | state
    uint32_t n_own = 0;
    auto scan_left = [&](uint32_t i, uint32_t kx, uint32_t cg, uint64_t h, uint32_t t0, uint32_t &Ld, bool &done, bool one) {
        for (uint32_t t = t0; !done; t += 4) {
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                const uint4 e = L.cand[(int)i - (int)t - (int)u];
                const uint32_t d = kx - e.x;
                const bool stop = e.y != cg || d > wm1;
                const bool hit = !stop && ((((uint64_t)e.w << 32) | e.z) < h);
                Ld = (!done && hit) ? d - 1u : Ld;
                done = done || stop || hit;
            }
            if (one) break;
        }
    };
    auto scan_right = [&](uint32_t i, uint32_t kx, uint32_t cg, uint64_t h, uint32_t need, uint32_t t0, bool &sel, bool &done, bool one) {
        for (uint32_t t = t0; !done; t += 4) {
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                const uint4 e = L.cand[i + t + u];
                const uint32_t d = e.x - kx;
                const bool stop = e.y != cg || d > need;
                const bool hit = !stop && ((((uint64_t)e.w << 32) | e.z) <= h);
                sel = (!done && hit) ? false : sel;
                done = done || stop || hit;
            }
            if (one) break;
        }
    };
    for (uint32_t i0 = 0; i0 < n_c; i0 += BSR_THREADS) {  // (one round unless max_cand > 1024)
        const uint32_t i = i0 + tid;
        bool undecided = false;
        uint32_t st_word = 0;
        if (i < n_c) {
            bool sel = false;
            const uint32_t rel = L.posl[i];
            if (rel >= core_rel_lo && rel < core_rel_hi) {  // (a halo candidate: its own block decides it)
                ++n_own;
                const uint4 me = L.cand[i];
                const uint64_t h = ((uint64_t)me.w << 32) | me.z;
                const uint32_t kx = me.x, cg = me.y, nk = L.cnk[i];
                // is everything this candidate can need inside the range?
                const uint32_t need_lo = kx > wm1 ? kx - wm1 : 0u, need_hi = std::min(nk - 1u, kx + wm1);
                if ((cg == s_klo_ctg && need_lo < s_klo) || (cg == s_khi_ctg && need_hi > s_khi)) {
                    s_flag = 1;
                } else {
                    uint32_t Ld = std::min(kx, wm1);
                    bool ldone = false;
                    scan_left(i, kx, cg, h, 1u, Ld, ldone, true);
                    if (!ldone) {
                        undecided = true;  // (left scan unfinished)
                    } else {
                        const uint32_t Rd = std::min(nk - 1u - kx, wm1);
                        sel = Ld + Rd + 1u >= w;
                        const uint32_t need = wm1 - std::min(Ld, wm1);
                        bool rdone = !(sel && need > 0);
                        scan_right(i, kx, cg, h, need, 1u, sel, rdone, true);
                        if (!rdone) {
                            undecided = true;
                            st_word = 0x80000000u | (Ld << 16);  // (left part known: Ld < 2^15 as w <= 2048)
                        } else {
                            if (p.ctg_drop && sel && kx <= wm1 && Ld == kx && p.ctg_drop[cg]) sel = false;
                            if (h == 0xFFFFFFFFFFFFFFFFull) sel = false;
                        }
                    }
                    // candidate-free stretches behind this candidate (the candidate in front of a stretch reports it)
                    const uint4 nxt = L.cand[i + 1u];
                    if (nxt.y == cg) {
                        if (nxt.x - kx - 1u >= w) {
                            const uint32_t idx = atomicAdd(&p.ctrl[1], 1u);
                            if (idx < p.gap_cap) p.gaps[idx] = make_uint4(cg, kx + 1u, nxt.x - 1u, blockIdx.x * p.rk);
                        }
                    } else if (nk - 1u - kx >= w) {
                        // no further candidate of the contig in the range: the stretch runs to the contig's end -- unless the
                        // contig goes on behind the range, where its next candidate hides: then the stretch is longer than the
                        // device route takes
                        if (cg == s_khi_ctg) s_flag = 1;
                        else {
                            const uint32_t idx = atomicAdd(&p.ctrl[1], 1u);
                            if (idx < p.gap_cap) p.gaps[idx] = make_uint4(cg, kx + 1u, nk - 1u, blockIdx.x * p.rk);
                        }
                    }
                }
            }
            selb[i] = sel ? 1 : 0;
        }
        uint32_t n_und;
        const uint32_t ua = bsr_scan_flag(undecided, L.sh, n_und);
        if (undecided) undl[ua] = i | st_word;  // (i < 2^15)
        __syncthreads();
        for (uint32_t q = tid; q < n_und; q += BSR_THREADS) {  // pass 2
            const uint32_t wd = undl[q], ii = wd & 0x7FFFu;
            const uint4 me = L.cand[ii];
            const uint64_t h = ((uint64_t)me.w << 32) | me.z;
            const uint32_t kx = me.x, cg = me.y, nk = L.cnk[ii];
            uint32_t Ld;
            if (wd & 0x80000000u) {
                Ld = (wd >> 16) & 0x7FFFu;
            } else {
                Ld = std::min(kx, wm1);
                bool ldone = false;
                scan_left(ii, kx, cg, h, 5u, Ld, ldone, false);
            }
            const uint32_t Rd = std::min(nk - 1u - kx, wm1);
            bool sel = Ld + Rd + 1u >= w;
            const uint32_t need = wm1 - std::min(Ld, wm1);
            bool rdone = !(sel && need > 0);
            scan_right(ii, kx, cg, h, need, (wd & 0x80000000u) ? 5u : 1u, sel, rdone, false);
            if (p.ctg_drop && sel && kx <= wm1 && Ld == kx && p.ctg_drop[cg]) sel = false;
            if (h == 0xFFFFFFFFFFFFFFFFull) sel = false;
            selb[ii] = sel ? 1 : 0;
        }
        __syncthreads();
    }
    // the selected ones, in order
    uint32_t out_at = 0;
    for (uint32_t i0 = 0; i0 < n_c; i0 += BSR_THREADS) {
        const uint32_t i = i0 + tid;
        const bool sel = i < n_c && selb[i];
        uint32_t tot;
        const uint32_t at = out_at + bsr_scan_flag(sel, L.sh, tot);
        out_at += tot;
        if (sel && at < p.rk) {
            const uint4 me = L.cand[i];
            const size_t dst = (size_t)blockIdx.x * p.rk + at;
            p.cs_h[dst] = ((uint64_t)me.w << 32) | me.z;
            p.cs_k[dst] = me.x;
            p.cs_c[dst] = me.y;
        }
    }
    BSR_STAMP(6)
    BSR_ABLATE(5)
    // ---- stretches in front of a contig's first candidate, and contigs without any: reported by the block whose own
    // positions hold the contig's first k-mer
    for (uint32_t r = tid; r < n_win; r += BSR_THREADS) {
        const Run &run = L.runs[r];
        if (run.kidx0 != 0) continue;
        const int64_t rel64 = (int64_t)run.base_off - range_lo;
        if (rel64 < (int64_t)core_rel_lo || rel64 >= (int64_t)core_rel_hi) continue;
        const uint32_t cg = run.contig, nk = L.rnk[r];
        // first candidate at or behind the contig's first position
        uint32_t lo = 0, hi = n_c;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((int64_t)L.posl[mid] < rel64) lo = mid + 1; else hi = mid;
        }
        uint32_t khi = 0;
        bool push = false;
        const uint4 e = L.cand[lo];  // (index n_c: a sentinel)
        if (e.y == cg) {
            if (e.x >= w) { push = true; khi = e.x - 1u; }
        } else if (cg == s_khi_ctg) {
            s_flag = 1;  // the contig leaves the range without a candidate
        } else {
            push = true; khi = nk - 1u;  // no candidate at all (eligible: nk >= w)
        }
        if (push) {
            const uint32_t idx = atomicAdd(&p.ctrl[1], 1u);
            if (idx < p.gap_cap) p.gaps[idx] = make_uint4(cg, 0u, khi, blockIdx.x * p.rk);
        }
    }
    BSR_STAMP(7)
    // ---- counts: selected (two-level, for k_emit), own candidates (statistics: 64 counters on their own lines, summed by
    // k_emit's reporting tile -- one counter for all blocks cost 70 us per launch in same-address atomics)
    {
        const uint32_t own_w = wave_sum_u32(n_own);
        __shared__ uint32_t s_own[BSR_THREADS / 64];
        if (lane == 0) s_own[tid >> 6] = own_w;
        __syncthreads();
        if (tid == 0) {
            if (out_at > p.rk) s_flag = 1;
            count_publish(p.cnt, p.sup, blockIdx.x, std::min(out_at, p.rk));
            uint32_t own = 0;
            for (uint32_t u = 0; u < BSR_THREADS / 64; ++u) own += s_own[u];
            if (own) atomicAdd(&p.cand_spread[(blockIdx.x & 63u) * 32u], own);
            if (s_flag) p.ctrl[6] = 1;
        }
    }
    BSR_STAMP(8)
}

void launch_bs_resolve(const BsResolveParams &p, uint32_t n_blocks, hipStream_t st)
{
    hipLaunchKernelGGL(k_bs_resolve, dim3(n_blocks), dim3(BSR_THREADS), bs_resolve_lds(p), st, p);
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
    // padded copies of the first and of the last chunk's words (the filter reads whole chunks and, per lane, the two words in
    // front of its 64: zeros in front of the assembly, zeros = base A behind it): [head copy | tail copy], BS_EDGE_WORDS each
    MXG_HIP(h, a->d_bs_tail.ensure((size_t)2 * BS_EDGE_WORDS * 4));
    MXG_HIP(h, hipMemsetAsync(a->d_bs_tail.p, 0, (size_t)2 * BS_EDGE_WORDS * 4, h->stream));
    {
        uint32_t *edge = a->d_bs_tail.as<uint32_t>();
        const uint64_t n_head = std::min<uint64_t>(a->packed_words, BS_CHUNK_WORDS);
        MXG_HIP(h, hipMemcpyAsync(edge + 2, a->d_packed, (size_t)n_head * 4, hipMemcpyDeviceToDevice, h->stream));
        const uint64_t tail_lo = (uint64_t)(n_chunks - 1) * BS_CHUNK_WORDS;
        if (n_chunks > 1) {  // (one chunk: it is the first one, and the head copy is padded behind as well)
            const uint64_t from = tail_lo - 2;  // with the two words in front
            if (a->packed_words > from)
                MXG_HIP(h, hipMemcpyAsync(edge + BS_EDGE_WORDS, a->d_packed + from, (size_t)std::min<uint64_t>(a->packed_words - from, BS_CHUNK_WORDS + 2) * 4,
                                          hipMemcpyDeviceToDevice, h->stream));
        }
    }
    // chunk -> first run whose k-mers end behind the chunk's first position (runs are sorted)
    std::vector<uint32_t> run0((size_t)n_chunks + 1);
    size_t r = 0;
    for (uint32_t c = 0; c <= n_chunks; ++c) {
        const uint64_t p0 = (uint64_t)c * BS_CHUNK;
        while (r < a->runs.size() && a->runs[r].base_off + a->runs[r].n_kmers <= p0) ++r;
        run0[c] = (uint32_t)r;
    }
    MXG_HIP(h, a->d_bs_run0.ensure(run0.size() * 4));
    MXG_HIP(h, hipMemcpyAsync(a->d_bs_run0.p, run0.data(), run0.size() * 4, hipMemcpyHostToDevice, h->stream));
    MXG_HIP(h, hipStreamSynchronize(h->stream));
    a->bs_ready = true;
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
    const char *eb = getenv("MXG_BS_BLOCKS");
    const unsigned env_blocks = eb && atoi(eb) > 0 ? (unsigned)atoi(eb) : 512u;
    const uint32_t blocks = std::min<uint32_t>((uint32_t)env_blocks, (a->bs_chunks + 3u) / 4u);  // two waves per SIMD are resident (see k_hash_bs)
    const uint32_t *head = a->d_bs_tail.as<uint32_t>() + 2, *tail = a->bs_chunks > 1 ? head + BS_EDGE_WORDS : head;
    hipLaunchKernelGGL(k_hash_bs, dim3(blocks), dim3(256), 0, st, a->d_packed, head, tail, a->d_bs_out.as<uint32_t>() + BS_OUT_PAD, 0u,
                       a->bs_chunks, tt, a->bs_chunks - 1u);
    MXG_HIP(h, hipGetLastError());
    return MXG_OK;
}

}  // namespace mxg
