// sketch.hip -- the minimizer-sketch stage on gfx950 (replaces `indexlr`, reference ntJoin:204-205).
//
// Semantics: for every record, over its VALID k-mers only, the rightmost
// arg-min of canonical ntHash in every window of w consecutive valid k-mers; distinct arg-mins in
// position order; printed hash = second ntHash value ext(min_hash).
//
// Stateless, data-parallel formulation used here (equivalent to the stateful ring-buffer loop, proved
// in DESIGN.md and checked against the oracle): k-mer p (contig-local valid-k-mer index) is a
// minimizer iff  L(p) + R(p) + 1 >= w  where
//     L(p) = number of consecutive k-mers left of p with hash >= h(p)   (capped by p and w-1)
//     R(p) = number of consecutive k-mers right of p with hash >  h(p)  (capped by n-1-p and w-1)
// i.e. there is room for a window of w k-mers around p in which p is the rightmost minimum.
//
// Kernels:
//   k_hash<S,...>   one lane per strip of S consecutive k-mers inside one valid run: k warm-up steps, then
//                   rolling forward / reverse-complement ntHash (split-rotate done on 32-bit halves),
//                   table-driven (LDS, one ds_read_b128 per base).  Emits CANDIDATES (min_hash, contig-local
//                   k-mer index, contig|strand).  Dense mode: every k-mer is a candidate.
//   k_resolve       one lane per candidate: scan neighbouring candidates left/right for a blocker.
//   k_count/k_emit  ordered stream compaction of the selected candidates into the sketch arrays
//                   (out_hash = ext(min_hash), pos from the run table, record index, strand).
#include <algorithm>

#include "mxg_internal.h"
#include "scan_kernels.h"

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
__device__ __forceinline__ uint64_t canonical(const H2 &h, bool &forward)
{
    uint64_t f = ((uint64_t)h.fhi << 32) | h.flo, r = ((uint64_t)h.rhi << 32) | h.rlo;
    forward = f <= r;
    if (VARIANT == MXG_VARIANT_V1_MIN) return forward ? f : r;
    return f + r;
}

__device__ __forceinline__ uint64_t ext_hash(uint64_t h0, uint64_t mult)
{
    uint64_t t = h0 * mult;  // mult = 1 ^ (k * MULTISEED)
    return t ^ (t >> 27);
}

struct HashParams {
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

// Dense mode: every valid k-mer becomes a candidate at arena slot (its global k-mer index - g_base).
template <int S, int VARIANT>
__global__ __launch_bounds__(256) void k_hash_dense(const HashParams p)
{
    __shared__ uint4 tab[20];
    if (threadIdx.x < 20) tab[threadIdx.x] = p.tab.e[threadIdx.x];
    __syncthreads();
    const uint32_t s = p.strip_lo + blockIdx.x * 256u + threadIdx.x;
    if (s >= p.strip_hi) return;
    uint32_t lo = p.run_lo, hi = p.run_hi;  // run_strip0[lo] <= s < run_strip0[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (p.run_strip0[mid] <= s) lo = mid; else hi = mid;
    }
    const Run run = p.runs[lo];
    const uint32_t j0 = (s - p.run_strip0[lo]) * (uint32_t)S;
    const uint32_t len = min((uint32_t)S, run.n_kmers - j0);
    const uint64_t b = run.base_off + j0;
    const uint64_t gi = p.run_g0[lo] + j0 - p.g_base;
    const uint32_t kidx = run.kidx0 + j0;
    const uint32_t k = p.k;

    H2 h = {0u, 0u, 0u, 0u};
    for (uint32_t t = 0; t < k; t += 16) {  // warm-up: k steps with no outgoing base
        uint32_t chunk = fetch16(p.packed, b + t);
        uint32_t n = min(16u, k - t);
        for (uint32_t u = 0; u < n; ++u) {
            nt_step(h, tab[16 + (chunk & 3u)]);
            chunk >>= 2;
        }
    }
    {
        bool fw;
        uint64_t h0 = canonical<VARIANT>(h, fw);
        p.cand_h[gi] = h0;
        p.cand_k[gi] = kidx;
        p.cand_c[gi] = run.contig | (fw ? 0u : 0x80000000u);
    }
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
                bool fw;
                uint64_t h0 = canonical<VARIANT>(h, fw);
                p.cand_h[gi + j] = h0;
                p.cand_k[gi + j] = kidx + j;
                p.cand_c[gi + j] = run.contig | (fw ? 0u : 0x80000000u);
            }
        }
    }
}

// One lane per candidate.  sel[i] = 1 iff candidate i is a minimizer (see file header).
__global__ __launch_bounds__(256) void k_resolve(const uint64_t *__restrict__ ch, const uint32_t *__restrict__ ck,
                                                 const uint32_t *__restrict__ cc, uint32_t n,
                                                 const uint32_t *__restrict__ ctg_nk, uint32_t w,
                                                 uint8_t *__restrict__ sel)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t h = ch[i];
    const uint32_t kx = ck[i];
    const uint32_t c = cc[i] & 0x7FFFFFFFu;
    const uint32_t nk = ctg_nk[c];
    const uint32_t wm1 = w - 1;
    uint32_t L = min(kx, wm1);
    for (uint32_t j = i; j-- > 0;) {
        if ((cc[j] & 0x7FFFFFFFu) != c) break;
        uint32_t d = kx - ck[j];
        if (d > wm1) break;
        if (ch[j] < h) {  // strictly smaller on the left blocks (ties: rightmost wins)
            L = d - 1;
            break;
        }
    }
    uint32_t R = min(nk - 1 - kx, wm1);
    bool s = (L + R + 1 >= w);  // enough room if nothing blocks on the right
    if (s && L < wm1) {
        // need R >= w-1-L : look at candidates up to that distance only
        const uint32_t need = wm1 - L;
        for (uint32_t j = i + 1; j < n; ++j) {
            if ((cc[j] & 0x7FFFFFFFu) != c) break;
            uint32_t d = ck[j] - kx;
            if (d > need) break;
            if (ch[j] <= h) {  // smaller-or-equal on the right blocks
                s = false;
                break;
            }
        }
    }
    sel[i] = (s && h != 0xFFFFFFFFFFFFFFFFull) ? 1 : 0;  // btllib never reports min_hash == 2^64-1
}

struct EmitParams {
    const uint8_t *sel;
    const uint64_t *ch;
    const uint32_t *ck, *cc;
    uint32_t n;
    const uint32_t *bsum;  // exclusive block offsets
    const Run *runs;
    const uint32_t *ctg_run0, *ctg_rec;
    uint64_t mult;         // 1 ^ (k * MULTISEED)
    uint64_t out_base;     // where this batch starts in the output arrays
    uint64_t *o_hash;
    uint32_t *o_pos, *o_rec;
    uint8_t *o_fwd;
};

__global__ __launch_bounds__(256) void k_emit(const EmitParams p)
{
    __shared__ uint32_t sh[256];
    uint32_t base = blockIdx.x * TILE + threadIdx.x * TILE_PER_THREAD;
    uint32_t c = 0;
    for (int u = 0; u < TILE_PER_THREAD; ++u)
        if (base + u < p.n) c += p.sel[base + u];
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t t = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint64_t o = p.out_base + p.bsum[blockIdx.x] + sh[threadIdx.x] - c;
    if (c == 0) return;
    for (int u = 0; u < TILE_PER_THREAD; ++u) {
        uint32_t i = base + u;
        if (i < p.n && p.sel[i]) {
            uint32_t cs = p.cc[i], ctg = cs & 0x7FFFFFFFu, kx = p.ck[i];
            // contig-local valid-k-mer index -> base position, through the contig's run table
            uint32_t lo = p.ctg_run0[ctg], hi = p.ctg_run0[ctg + 1];
            while (hi - lo > 1) {
                uint32_t mid = (lo + hi) >> 1;
                if (p.runs[mid].kidx0 <= kx) lo = mid; else hi = mid;
            }
            p.o_hash[o] = ext_hash(p.ch[i], p.mult);
            p.o_pos[o] = p.runs[lo].pos0 + (kx - p.runs[lo].kidx0);
            p.o_rec[o] = p.ctg_rec[ctg];
            p.o_fwd[o] = (cs >> 31) ? 0 : 1;
            ++o;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------------------------
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
static int upload(mxg_handle *h, DevBuf &b, const std::vector<T> &v)
{
    MXG_HIP(h, b.ensure(std::max<size_t>(v.size() * sizeof(T), 16)));
    if (!v.empty())
        MXG_HIP(h, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    return MXG_OK;
}

constexpr int S_DENSE = 128;
constexpr uint64_t DENSE_BATCH_KMERS = 96ull << 20;  // arena = 16 B per k-mer

template <int S>
static void launch_hash_dense(mxg_handle *h, const HashParams &p, uint32_t n_strips)
{
    dim3 grid((n_strips + 255) / 256), block(256);
    if (h->cfg.variant == MXG_VARIANT_V1_MIN)
        hipLaunchKernelGGL((k_hash_dense<S, MXG_VARIANT_V1_MIN>), grid, block, 0, h->stream, p);
    else
        hipLaunchKernelGGL((k_hash_dense<S, MXG_VARIANT_V2_SUM>), grid, block, 0, h->stream, p);
}

int sketch_assembly(mxg_handle *h, Assembly *a)
{
    if (!a->has_bases) return set_err(h, MXG_EINVAL, "assembly '%s' has no bases to sketch", a->name.c_str());
    MXG_HIP(h, hipSetDevice(h->device));
    const uint32_t k = h->cfg.k, w = h->cfg.w;
    a->has_sketch = false;
    a->host_valid = false;
    a->flags_valid = false;
    h->graph.valid = false;
    a->n_mx = 0;

    // bases to HBM
    if (!a->d_packed) {
        MXG_HIP(h, a->d_packed_own.ensure(a->h_packed.size() * 4));
        MXG_HIP(h, hipMemcpyAsync(a->d_packed_own.p, a->h_packed.data(), a->h_packed.size() * 4,
                                  hipMemcpyHostToDevice, h->stream));
        MXG_HIP(h, hipStreamSynchronize(h->stream));
        a->d_packed = a->d_packed_own.as<uint32_t>();
        if (a->has_text || !(h->cfg.flags & MXG_FLAG_DROP_SEQ)) {
            std::vector<uint32_t>().swap(a->h_packed);  // text (if any) serves --seq; otherwise refetched on demand
        }
    }
    const size_t n_runs = a->runs.size();
    if (n_runs == 0) {  // nothing eligible: empty sketch
        a->has_sketch = true;
        return MXG_OK;
    }
    if (n_runs >= (1ull << 31)) return set_err(h, MXG_ELIMIT, "too many valid runs (%zu)", n_runs);

    const int S = S_DENSE;
    std::vector<uint32_t> strip0(n_runs + 1);
    std::vector<uint64_t> g0(n_runs + 1);
    {
        uint64_t s = 0, g = 0;
        for (size_t r = 0; r < n_runs; ++r) {
            strip0[r] = (uint32_t)s;
            g0[r] = g;
            s += (a->runs[r].n_kmers + S - 1) / S;
            g += a->runs[r].n_kmers;
            if (s >= (1ull << 32)) return set_err(h, MXG_ELIMIT, "too many strips");
        }
        strip0[n_runs] = (uint32_t)s;
        g0[n_runs] = g;
    }
    int rc;
    if ((rc = upload(h, h->s_runs, a->runs)) != MXG_OK) return rc;
    if ((rc = upload(h, h->s_strip0, strip0)) != MXG_OK) return rc;
    if ((rc = upload(h, h->s_g0, g0)) != MXG_OK) return rc;
    if ((rc = upload(h, h->s_ctg_nk, a->ctg_nk)) != MXG_OK) return rc;
    if ((rc = upload(h, h->s_ctg_rec, a->ctg_rec)) != MXG_OK) return rc;
    if ((rc = upload(h, h->s_ctg_run0, a->ctg_run0)) != MXG_OK) return rc;
    MXG_HIP(h, h->s_total.ensure(64));

    // output capacity estimate: density 2/(w+1) per k-mer, generous slack; grown on demand
    uint64_t cap = (uint64_t)(3.0 * (double)a->total_kmers / (double)(w + 1)) + 4096;
    MXG_HIP(h, a->d_hash.ensure(cap * 8));
    MXG_HIP(h, a->d_pos.ensure(cap * 4));
    MXG_HIP(h, a->d_rec.ensure(cap * 4));
    MXG_HIP(h, a->d_fwd.ensure(cap));
    cap = std::min<uint64_t>({a->d_hash.bytes / 8, a->d_pos.bytes / 4, a->d_rec.bytes / 4, a->d_fwd.bytes});

    const uint64_t mult = 1ull ^ ((uint64_t)k * 0x90b45d39fb6da1faull);
    const size_t n_ctg = a->ctg_rec.size();
    uint64_t n_out = 0;
    size_t c0 = 0;
    while (c0 < n_ctg) {
        // batch = whole contigs [c0, c1)
        size_t c1 = c0;
        uint64_t nk = 0;
        while (c1 < n_ctg && (c1 == c0 || nk + a->ctg_nk[c1] <= DENSE_BATCH_KMERS)) nk += a->ctg_nk[c1++];
        if (nk >= (1ull << 31))
            return set_err(h, MXG_ELIMIT, "record '%s' has %llu valid k-mers; the dense path handles < 2^31 per record",
                           a->recs[a->ctg_rec[c0]].id.c_str(), (unsigned long long)nk);
        const uint32_t r_lo = a->ctg_run0[c0], r_hi = a->ctg_run0[c1];
        MXG_HIP(h, h->s_cand_h.ensure(nk * 8));
        MXG_HIP(h, h->s_cand_k.ensure(nk * 4));
        MXG_HIP(h, h->s_cand_c.ensure(nk * 4));
        MXG_HIP(h, h->s_sel.ensure(nk));
        const uint32_t n_cand = (uint32_t)nk;
        const uint32_t n_tiles = (n_cand + TILE - 1) / TILE;
        MXG_HIP(h, h->s_bsum.ensure((size_t)n_tiles * 4 + 16));

        HashParams hp;
        hp.packed = a->d_packed;
        hp.runs = h->s_runs.as<Run>();
        hp.run_strip0 = h->s_strip0.as<uint32_t>();
        hp.run_g0 = h->s_g0.as<uint64_t>();
        hp.run_lo = r_lo;
        hp.run_hi = r_hi;
        hp.strip_lo = strip0[r_lo];
        hp.strip_hi = strip0[r_hi];
        hp.g_base = g0[r_lo];
        hp.k = k;
        hp.cand_h = h->s_cand_h.as<uint64_t>();
        hp.cand_k = h->s_cand_k.as<uint32_t>();
        hp.cand_c = h->s_cand_c.as<uint32_t>();
        hp.tab = h->tab;
        const bool timing = (h->cfg.flags & MXG_FLAG_TIMING) != 0;
        if (timing) MXG_HIP(h, hipEventRecord(h->ev0, h->stream));
        launch_hash_dense<S_DENSE>(h, hp, hp.strip_hi - hp.strip_lo);
        if (timing) MXG_HIP(h, hipEventRecord(h->ev1, h->stream));
        MXG_HIP(h, hipGetLastError());

        hipLaunchKernelGGL(k_resolve, dim3((n_cand + 255) / 256), dim3(256), 0, h->stream, hp.cand_h, hp.cand_k,
                           hp.cand_c, n_cand, h->s_ctg_nk.as<uint32_t>(), w, h->s_sel.as<uint8_t>());
        hipLaunchKernelGGL(k_count, dim3(n_tiles), dim3(256), 0, h->stream, h->s_sel.as<uint8_t>(), n_cand,
                           h->s_bsum.as<uint32_t>());
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, h->stream, h->s_bsum.as<uint32_t>(), n_tiles,
                           h->s_total.as<uint64_t>());
        MXG_HIP(h, hipGetLastError());
        uint64_t total = 0;
        MXG_HIP(h, hipMemcpyAsync(&total, h->s_total.p, 8, hipMemcpyDeviceToHost, h->stream));
        MXG_HIP(h, hipStreamSynchronize(h->stream));
        if (timing) {
            float ms = 0;
            MXG_HIP(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
            h->tm.ms_hash += ms;
            h->tm.launches_hash += 1;
            uint64_t bases = 0;
            for (size_t c = c0; c < c1; ++c) bases += a->recs[a->ctg_rec[c]].len;
            h->tm.hash_bases += bases;
        }
        if (n_out + total > cap) {
            uint64_t need = n_out + total;
            MXG_HIP(h, grow_preserve(a->d_hash, n_out * 8, need * 8, h->stream));
            MXG_HIP(h, grow_preserve(a->d_pos, n_out * 4, need * 4, h->stream));
            MXG_HIP(h, grow_preserve(a->d_rec, n_out * 4, need * 4, h->stream));
            MXG_HIP(h, grow_preserve(a->d_fwd, n_out, need, h->stream));
            cap = std::min<uint64_t>({a->d_hash.bytes / 8, a->d_pos.bytes / 4, a->d_rec.bytes / 4, a->d_fwd.bytes});
        }
        EmitParams ep;
        ep.sel = h->s_sel.as<uint8_t>();
        ep.ch = hp.cand_h;
        ep.ck = hp.cand_k;
        ep.cc = hp.cand_c;
        ep.n = n_cand;
        ep.bsum = h->s_bsum.as<uint32_t>();
        ep.runs = hp.runs;
        ep.ctg_run0 = h->s_ctg_run0.as<uint32_t>();
        ep.ctg_rec = h->s_ctg_rec.as<uint32_t>();
        ep.mult = mult;
        ep.out_base = n_out;
        ep.o_hash = a->d_hash.as<uint64_t>();
        ep.o_pos = a->d_pos.as<uint32_t>();
        ep.o_rec = a->d_rec.as<uint32_t>();
        ep.o_fwd = a->d_fwd.as<uint8_t>();
        hipLaunchKernelGGL(k_emit, dim3(n_tiles), dim3(256), 0, h->stream, ep);
        MXG_HIP(h, hipGetLastError());
        n_out += total;
        h->stat_dense_kmers += nk;
        c0 = c1;
    }
    MXG_HIP(h, hipStreamSynchronize(h->stream));
    a->n_mx = n_out;
    a->has_sketch = true;
    return MXG_OK;
}

int sync_sketch_to_host(mxg_handle *h, Assembly *a)
{
    if (!a->has_sketch) return set_err(h, MXG_EINVAL, "assembly '%s' has no sketch yet (call mxg_sketch)", a->name.c_str());
    if (a->host_valid) return MXG_OK;
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
