// dgraph.hip -- the graph stage distributed over ranks by hash range, device side.
//
// The union approach (every rank all-gathers every sketch and builds the whole graph) does N times the work on every
// rank.  Here every minimizer is sent to the rank that OWNS its hash; the owner decides uniqueness and intersection for
// its hashes with the ordinary kernels of graph.hip and numbers its vertices; the verdict (flags + global vertex id)
// travels back; adjacency stays where the records are (a record is never split over ranks) and reaches the owners of the
// two end points as messages; every owner emits the edges whose first supporter's source vertex it owns.
// The collectives themselves (all-to-all with uneven splits) are issued by the caller (ntjoin_amd/dist.py over
// torch.distributed = RCCL); this file packs, unpacks and counts.
//
//   sender                                   owner
//   k_dg_owner_count / k_dg_pack_items  -->  k_dg_items_to_soa, build_graph(GRAPH_DG_VERTICES), k_dg_item_result
//   k_dg_shared_cnt / k_dg_compact      <--  (flags | global vertex id << 8 per item, in the sender's bucket order)
//   k_dg_msg_count / k_dg_pack_msgs     -->  build_graph(GRAPH_DG_EDGES): k_apply_msgs, k_edge_flags, k_edges
#include <algorithm>
#include <cstring>

#include "mxg_internal.h"
#include "scan_kernels.h"

namespace mxg {

static constexpr uint32_t DG_NONE = 0xFFFFFFFFu;

// rank that owns a hash: multiplicative mix, then the top bits scaled to [0, world)
__device__ __forceinline__ uint32_t dg_owner(uint64_t hash, uint32_t world)
{
    const uint32_t m = (uint32_t)((hash * 0x9E3779B97F4A7C15ull) >> 32);
    return (uint32_t)(((uint64_t)m * world) >> 32);
}

// Bucketing by destination without hot atomics.  A block owns DG_IPB consecutive items: it builds the histogram of
// their destinations in LDS (one LDS atomic per wave and distinct key), then touches global memory once per destination
// -- to add its counts (counting kernels) or to reserve its range of every bucket (packing kernels); inside a reserved
// range the slots come from LDS cursors.  (One global atomic per wave and key: 62 k same-address atomics for 4 M items
// on a single rank, ~10 ns each.)
constexpr uint32_t DG_IPB = 1024;  // items per block (4 rounds of 256 threads; with 4096 the packing kernels of a 2 x 10^5-item step ran on 49 blocks: 23 us)

__device__ __forceinline__ void lds_hist(uint32_t *lh, uint32_t key, bool active)
{
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t todo = __ballot(active); todo;) {
        const int leader = __builtin_ctzll(todo);
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
        const uint64_t same = __ballot(active && key == k0);
        if ((int)lane == leader) atomicAdd(&lh[k0], (uint32_t)__popcll(same));
        todo &= ~same;
    }
}
// a slot inside the block's range of bucket `key` (lcur[key] advances by the wave's lanes of that key)
__device__ __forceinline__ uint32_t lds_slot(uint32_t *lcur, uint32_t key, bool active)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t slot = 0;
    for (uint64_t todo = __ballot(active); todo;) {
        const int leader = __builtin_ctzll(todo);
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
        const uint64_t same = __ballot(active && key == k0);
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(&lcur[k0], (uint32_t)__popcll(same));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        if (active && key == k0) slot = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return slot;
}

__global__ __launch_bounds__(256) void k_dg_owner_count(const uint64_t *__restrict__ hash, uint64_t n, uint32_t world,
                                                        unsigned long long *cnt)
{
    __shared__ uint32_t lh[64];
    if (threadIdx.x < 64) lh[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t i0 = (uint64_t)blockIdx.x * DG_IPB;
    for (uint32_t it = 0; it < DG_IPB / 256; ++it) {
        const uint64_t i = i0 + it * 256u + threadIdx.x;
        const bool in = i < n;
        lds_hist(lh, in ? dg_owner(hash[i], world) : 0u, in);
    }
    __syncthreads();
    if (threadIdx.x < world && lh[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

// item = {hash, pos, global record}; perm[i] = where minimizer i went in the send buffer
__global__ __launch_bounds__(256) void k_dg_pack_items(const uint64_t *__restrict__ hash, const uint32_t *__restrict__ pos,
                                                       const uint32_t *__restrict__ rec, uint64_t n, uint32_t world,
                                                       uint32_t rec_off, unsigned long long *cursor, uint4 *items, uint32_t *perm)
{
    __shared__ uint32_t lh[64], lcur[64];
    __shared__ unsigned long long lbase[64];
    if (threadIdx.x < 64) lh[threadIdx.x] = lcur[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t i0 = (uint64_t)blockIdx.x * DG_IPB;
    for (uint32_t it = 0; it < DG_IPB / 256; ++it) {
        const uint64_t i = i0 + it * 256u + threadIdx.x;
        const bool in = i < n;
        lds_hist(lh, in ? dg_owner(hash[i], world) : 0u, in);
    }
    __syncthreads();
    if (threadIdx.x < world) lbase[threadIdx.x] = lh[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], (unsigned long long)lh[threadIdx.x]) : 0ull;
    __syncthreads();
    for (uint32_t it = 0; it < DG_IPB / 256; ++it) {
        const uint64_t i = i0 + it * 256u + threadIdx.x;
        const bool in = i < n;
        const uint64_t h = in ? hash[i] : 0;
        const uint32_t d = in ? dg_owner(h, world) : 0u;
        const uint32_t sl = lds_slot(lcur, d, in);
        if (in) {
            const uint64_t s = lbase[d] + sl;
            items[s] = make_uint4((uint32_t)h, (uint32_t)(h >> 32), pos[i], rec[i] + rec_off);
            perm[i] = (uint32_t)s;
        }
    }
}

// The receive buffer holds, per source rank, one section per assembly; assembly a's items are its `world` sections in
// source order.  start[s] = first item of source s's section in the buffer, first[s] = exclusive prefix of the counts.
struct DgSections {
    uint32_t world;
    uint64_t start[64];
    uint64_t first[65];
};
__device__ __forceinline__ uint64_t dg_locate(const DgSections &sc, uint64_t o)  // item o of the assembly -> buffer index
{
    uint32_t s = 0;
    while (o >= sc.first[s + 1]) ++s;  // world is small
    return sc.start[s] + (o - sc.first[s]);
}

__global__ __launch_bounds__(256) void k_dg_items_to_soa(const uint4 *__restrict__ items, const DgSections sc, uint64_t *hash,
                                                         uint32_t *pos, uint32_t *rec)
{
    const uint64_t o = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (o >= sc.first[sc.world]) return;
    const uint4 it = items[dg_locate(sc, o)];
    hash[o] = ((uint64_t)it.y << 32) | it.x;
    pos[o] = it.z;
    rec[o] = it.w;
}

// what the owner tells the sender about an item: flags | (global vertex id or NONE) << 8, at the item's place in the
// receive layout (the return trip uses the same splits backwards)
__global__ __launch_bounds__(256) void k_dg_item_result(const uint8_t *__restrict__ flags, const uint32_t *__restrict__ ivid,
                                                        const DgSections sc, const uint32_t *__restrict__ gbase_ptr,
                                                        unsigned long long *out)
{
    const uint64_t o = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (o >= sc.first[sc.world]) return;
    const uint32_t gbase = *gbase_ptr;  // this owner's first global vertex id, still on the device
    const uint32_t v = ivid[o];
    out[dg_locate(sc, o)] = ((unsigned long long)(v == DG_NONE ? DG_NONE : v + gbase) << 8) | flags[o];
}

// ---- sender, second half: adjacency of ITS records as messages to the owners of the two end points ---------------
struct DgAdjParams {
    const unsigned long long *ret;  // the owners' verdicts in send-buffer order
    const uint32_t *perm;           // minimizer i -> its place in that order
    const uint32_t *rec;
    uint32_t n;
    uint32_t *cnt, *sup;            // two-level counts of shared minimizers per 256 (scan_kernels.h)
    uint32_t *fg, *frec;            // shared minimizers in order: global vertex id, record
    uint32_t *n_shared;             // device scalar
    uint8_t *flags_out;             // the assembly's flags array (mxg_get_mx_flags)
};

__global__ __launch_bounds__(256) void k_dg_shared_cnt(const DgAdjParams p)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    bool sh = false;
    if (i < p.n) {
        const unsigned long long v = p.ret[p.perm[i]];
        p.flags_out[i] = (uint8_t)v;
        sh = ((uint8_t)v & MXG_MX_SHARED) != 0;
    }
    const uint32_t c = (uint32_t)__syncthreads_count(sh ? 1 : 0);
    if (threadIdx.x == 0) count_publish(p.cnt, p.sup, blockIdx.x, c);
}

__global__ __launch_bounds__(256) void k_dg_compact(const DgAdjParams p)
{
    __shared__ uint32_t sh_scan[256];
    __shared__ uint32_t sh_before;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    unsigned long long v = 0;
    bool sh = false;
    if (i < p.n) {
        v = p.ret[p.perm[i]];
        sh = ((uint8_t)v & MXG_MX_SHARED) != 0;
    }
    const uint32_t rank = block_exclusive_256(sh ? 1u : 0u, sh_scan);
    if (threadIdx.x < 64) {
        const uint32_t bef = count_prefix(p.cnt, p.sup, blockIdx.x);
        if (threadIdx.x == 0) sh_before = bef;
        if (blockIdx.x + 1 == gridDim.x) {
            const uint32_t all = count_prefix(p.cnt, p.sup, gridDim.x);
            if (threadIdx.x == 0) *p.n_shared = all;
        }
    }
    __syncthreads();
    if (sh) {
        const uint32_t r = sh_before + rank;
        p.fg[r] = (uint32_t)(v >> 8);
        p.frec[r] = p.rec[i];
    }
}

// owner of a global vertex id: bases[r] <= g < bases[r + 1]
__device__ __forceinline__ uint32_t dg_vertex_owner(const uint32_t *__restrict__ bases, uint32_t world, uint32_t g)
{
    uint32_t lo = 0, hi = world;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (bases[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

// Records cut between ranks (sub-record shards): the adjacency between the last shared minimizer of one rank's piece and
// the first shared minimizer of the next rank's piece is seen by neither.  `ghost` = {record, global vertex id} of the
// nearest shared minimizer BEFORE this rank's first one (from the ranks before it, mxg_dg_set_ghosts; record 2^32-1: none):
// pair number ns - 1 (otherwise unused: pairs are (r, r + 1) for r + 1 < ns) stands for (ghost, shared minimizer 0).
struct DgPair {
    bool in;
    uint32_t u, v;
};
__device__ __forceinline__ DgPair dg_pair(const uint32_t *__restrict__ fg, const uint32_t *__restrict__ frec, uint32_t ns, uint32_t r,
                                          const uint32_t *__restrict__ ghost)
{
    DgPair p{false, 0u, 0u};
    if (r + 1 < ns) {
        if (frec[r] == frec[r + 1]) p = DgPair{true, fg[r], fg[r + 1]};
    } else if (r + 1 == ns && ghost && ghost[0] != 0xFFFFFFFFu && ghost[0] == frec[0]) {
        p = DgPair{true, ghost[1], fg[0]};
    }
    return p;
}

// pair r = (shared minimizer r, r + 1) of the same record: one message to the owner of each end
__global__ __launch_bounds__(256) void k_dg_msg_count(const uint32_t *__restrict__ fg, const uint32_t *__restrict__ frec,
                                                      const uint32_t *__restrict__ n_shared, const uint32_t *__restrict__ bases,
                                                      uint32_t world, unsigned long long *cnt, const uint32_t *__restrict__ ghost)
{
    __shared__ uint32_t lh[64];
    if (threadIdx.x < 64) lh[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t ns = *n_shared;
    const uint32_t r0 = blockIdx.x * DG_IPB;
    if (r0 < ns) {
        for (uint32_t it = 0; it < DG_IPB / 256; ++it) {
            const uint32_t r = r0 + it * 256u + threadIdx.x;
            const DgPair pr = dg_pair(fg, frec, ns, r, ghost);
            lds_hist(lh, pr.in ? dg_vertex_owner(bases, world, pr.u) : 0u, pr.in);
            lds_hist(lh, pr.in ? dg_vertex_owner(bases, world, pr.v) : 0u, pr.in);
        }
    }
    __syncthreads();
    if (threadIdx.x < world && lh[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

__global__ __launch_bounds__(256) void k_dg_pack_msgs(const uint32_t *__restrict__ fg, const uint32_t *__restrict__ frec,
                                                      const uint32_t *__restrict__ n_shared, const uint32_t *__restrict__ bases,
                                                      uint32_t world, uint32_t assembly, unsigned long long *cursor, uint4 *msgs,
                                                      const uint32_t *__restrict__ ghost)
{
    __shared__ uint32_t lh[64], lcur[64];
    __shared__ unsigned long long lbase[64];
    if (threadIdx.x < 64) lh[threadIdx.x] = lcur[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t ns = *n_shared;
    const uint32_t r0 = blockIdx.x * DG_IPB;
    if (r0 >= ns) return;  // block-uniform
    for (uint32_t it = 0; it < DG_IPB / 256; ++it) {
        const uint32_t r = r0 + it * 256u + threadIdx.x;
        const DgPair pr = dg_pair(fg, frec, ns, r, ghost);
        lds_hist(lh, pr.in ? dg_vertex_owner(bases, world, pr.u) : 0u, pr.in);
        lds_hist(lh, pr.in ? dg_vertex_owner(bases, world, pr.v) : 0u, pr.in);
    }
    __syncthreads();
    if (threadIdx.x < world) lbase[threadIdx.x] = lh[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], (unsigned long long)lh[threadIdx.x]) : 0ull;
    __syncthreads();
    for (uint32_t it = 0; it < DG_IPB / 256; ++it) {
        const uint32_t r = r0 + it * 256u + threadIdx.x;
        const DgPair pr = dg_pair(fg, frec, ns, r, ghost);
        const bool in = pr.in;
        const uint32_t u = pr.u, v = pr.v;
        const uint32_t ou = in ? dg_vertex_owner(bases, world, u) : 0u, ov = in ? dg_vertex_owner(bases, world, v) : 0u;
        const uint32_t su = lds_slot(lcur, ou, in);
        const uint32_t sv = lds_slot(lcur, ov, in);
        if (in) {
            msgs[lbase[ou] + su] = make_uint4(assembly, u - bases[ou], v, 0u);         // nxt[a][u] = v at the owner of u
            msgs[lbase[ov] + sv] = make_uint4(assembly | 256u, v - bases[ov], u, 0u);  // prv[a][v] = u at the owner of v
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// per-destination counters -> host through pinned memory, polling wait
static int dg_counts_to_host(mxg_handle *h, size_t n_words, uint64_t *out)
{
    if (!h->pinned_dg) MXG_HIP(h, hipHostMalloc((void **)&h->pinned_dg, MXG_MAX_ASSEMBLIES * 64 * 8));
    MXG_HIP(h, hipMemcpyAsync(h->pinned_dg, h->dg_cnt.p, n_words * 8, hipMemcpyDeviceToHost, h->stream));
    MXG_HIP(h, stream_wait(h->stream));
    memcpy(out, h->pinned_dg, n_words * 8);
    return MXG_OK;
}

static int dg_cursor(mxg_handle *h, uint32_t world, const uint64_t *starts)
{
    MXG_HIP(h, h->dg_cursor.ensure(64 * 8));
    MXG_HIP(h, hipMemcpyAsync(h->dg_cursor.p, starts, (size_t)world * 8, hipMemcpyHostToDevice, h->stream));
    return MXG_OK;
}

// sender: how many minimizers of every assembly go to every rank; counts[a * world + r]
int dg_owner_counts(mxg_handle *h, uint32_t world, uint64_t *counts)
{
    if (world == 0 || world > 64) return set_err(h, MXG_ELIMIT, "world size must be 1..64");
    MXG_HIP(h, hipSetDevice(h->device));
    const size_t A = h->asms.size();
    MXG_HIP(h, h->dg_cnt.ensure(A * 64 * 8 + 64));
    MXG_HIP(h, hipMemsetAsync(h->dg_cnt.p, 0, A * 64 * 8, h->stream));
    for (size_t a = 0; a < A; ++a) {
        Assembly *as = h->asms[a];
        if (!as->has_sketch) return set_err(h, MXG_EINVAL, "assembly '%s' has no sketch", as->name.c_str());
        if (as->n_mx)
            hipLaunchKernelGGL(k_dg_owner_count, dim3((uint32_t)((as->n_mx + DG_IPB - 1) / DG_IPB)), dim3(256), 0, h->stream,
                               as->d_hash.as<uint64_t>(), as->n_mx, world, h->dg_cnt.as<unsigned long long>() + a * 64);
    }
    MXG_HIP(h, hipGetLastError());
    std::vector<uint64_t> tmp(A * 64);
    int rc = dg_counts_to_host(h, A * 64, tmp.data());
    if (rc != MXG_OK) return rc;
    for (size_t a = 0; a < A; ++a)
        for (uint32_t r = 0; r < world; ++r) counts[a * world + r] = tmp[a * 64 + r];
    return MXG_OK;
}

// sender: assembly a's minimizers as 16-byte items bucketed by owner; bucket r starts at item starts[r] of d_send
int dg_pack_items(mxg_handle *h, Assembly *a, uint32_t world, uint32_t rec_offset, const uint64_t *starts, void *d_send)
{
    MXG_HIP(h, hipSetDevice(h->device));
    int rc = dg_cursor(h, world, starts);
    if (rc != MXG_OK) return rc;
    MXG_HIP(h, a->d_perm.ensure(std::max<uint64_t>(a->n_mx * 4, 16)));
    if (a->n_mx)
        hipLaunchKernelGGL(k_dg_pack_items, dim3((uint32_t)((a->n_mx + DG_IPB - 1) / DG_IPB)), dim3(256), 0, h->stream,
                           a->d_hash.as<uint64_t>(), a->d_pos.as<uint32_t>(), a->d_rec.as<uint32_t>(), a->n_mx, world, rec_offset,
                           h->dg_cursor.as<unsigned long long>(), static_cast<uint4 *>(d_send), a->d_perm.as<uint32_t>());
    MXG_HIP(h, hipGetLastError());
    if (h->own_stream) MXG_HIP(h, hipStreamSynchronize(h->stream));  // (see pack_sketch)
    return MXG_OK;
}

static int dg_sections(mxg_handle *h, uint32_t world, const uint64_t *sec_start, const uint64_t *sec_count, DgSections *sc)
{
    if (world == 0 || world > 64) return set_err(h, MXG_ELIMIT, "world size must be 1..64");
    sc->world = world;
    uint64_t tot = 0;
    for (uint32_t s = 0; s < world; ++s) {
        sc->start[s] = sec_start[s];
        sc->first[s] = tot;
        tot += sec_count[s];
    }
    for (uint32_t s = world; s <= 64; ++s) sc->first[s] = tot;
    return MXG_OK;
}

// owner: the items received for assembly a (one section per source rank) become its (unordered) sketch
int dg_set_items(mxg_handle *h, Assembly *a, const void *d_items, uint32_t world, const uint64_t *sec_start,
                 const uint64_t *sec_count)
{
    MXG_HIP(h, hipSetDevice(h->device));
    DgSections sc;
    int rc = dg_sections(h, world, sec_start, sec_count, &sc);
    if (rc != MXG_OK) return rc;
    const uint64_t n = sc.first[world];
    if (n >= (1ull << 31)) return set_err(h, MXG_ELIMIT, "too many items for one owner");
    MXG_HIP(h, a->d_hash.ensure(std::max<uint64_t>(n * 8, 16)));
    MXG_HIP(h, a->d_pos.ensure(std::max<uint64_t>(n * 4, 16)));
    MXG_HIP(h, a->d_rec.ensure(std::max<uint64_t>(n * 4, 16)));
    if (n)
        hipLaunchKernelGGL(k_dg_items_to_soa, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, h->stream,
                           static_cast<const uint4 *>(d_items), sc, a->d_hash.as<uint64_t>(), a->d_pos.as<uint32_t>(),
                           a->d_rec.as<uint32_t>());
    MXG_HIP(h, hipGetLastError());
    a->n_mx = n;
    a->has_sketch = true;
    a->fwd_valid = false;
    a->foreign_sketch = true;
    a->host_valid = false;
    a->flags_valid = false;
    h->graph.valid = false;
    return MXG_OK;
}

// owner: verdict per item of assembly a (after build_graph(GRAPH_DG_VERTICES)), written into the return buffer at the
// places the items came from
int dg_item_results(mxg_handle *h, Assembly *a, const void *d_gbase, uint32_t world, const uint64_t *sec_start,
                    const uint64_t *sec_count, void *d_out)
{
    MXG_HIP(h, hipSetDevice(h->device));
    DgSections sc;
    int rc = dg_sections(h, world, sec_start, sec_count, &sc);
    if (rc != MXG_OK) return rc;
    if (sc.first[world] != a->n_mx) return set_err(h, MXG_EINVAL, "sections do not add up to the assembly's items");
    if (a->n_mx) {
        if (h->graph.nv_stride == 0) {  // some assembly received nothing: no vertex, and nobody made d_ivid
            MXG_HIP(h, a->d_ivid.ensure(a->n_mx * 4 + 16));
            MXG_HIP(h, hipMemsetAsync(a->d_ivid.p, 0xFF, a->n_mx * 4, h->stream));
        }
        hipLaunchKernelGGL(k_dg_item_result, dim3((uint32_t)((a->n_mx + 255) / 256)), dim3(256), 0, h->stream,
                           a->d_flags.as<uint8_t>(), a->d_ivid.as<uint32_t>(), sc, static_cast<const uint32_t *>(d_gbase),
                           static_cast<unsigned long long *>(d_out));
    }
    MXG_HIP(h, hipGetLastError());
    if (h->own_stream) MXG_HIP(h, hipStreamSynchronize(h->stream));
    return MXG_OK;
}

// ---- records cut between ranks: the last shared minimizer of every rank travels to the ranks after it -----------------------
__global__ void k_dg_last(const uint32_t *__restrict__ fg, const uint32_t *__restrict__ frec, const uint32_t *__restrict__ n_shared,
                          uint32_t *out)
{
    const uint32_t ns = *n_shared;
    out[0] = ns ? frec[ns - 1] : 0xFFFFFFFFu;
    out[1] = ns ? fg[ns - 1] : 0xFFFFFFFFu;
}
__global__ void k_dg_pick_ghost(const uint32_t *__restrict__ all, uint32_t n_asm, uint32_t rank, uint32_t *ghost)
{
    const uint32_t a = threadIdx.x;
    if (a >= n_asm) return;
    uint32_t rec = 0xFFFFFFFFu, g = 0xFFFFFFFFu;
    for (uint32_t q = rank; q-- > 0;) {  // the nearest rank before this one that has a shared minimizer of assembly a
        const uint32_t *e = all + ((size_t)q * n_asm + a) * 2;
        if (e[0] != 0xFFFFFFFFu) {
            rec = e[0];
            g = e[1];
            break;
        }
    }
    ghost[2 * a] = rec;
    ghost[2 * a + 1] = g;
}

// sender, after the verdicts came back: {record, global vertex id} of this rank's LAST shared minimizer of every assembly
// (2^32-1, 2^32-1: none) into d_out (u32[A][2], device).  No host sync.
int dg_last_shared(mxg_handle *h, const void *d_ret, void *d_out)
{
    MXG_HIP(h, hipSetDevice(h->device));
    const size_t A = h->asms.size();
    for (size_t ai = 0; ai < A; ++ai) {
        Assembly *a = h->asms[ai];
        const uint32_t n = (uint32_t)a->n_mx;
        const uint32_t blocks = (n + 255) / 256;
        MXG_HIP(h, a->d_flags.ensure(std::max<uint32_t>(n, 16)));
        MXG_HIP(h, a->d_dgtmp.ensure(((size_t)sup_words(blocks) + blocks + 16) * 4));
        MXG_HIP(h, a->d_fg.ensure((size_t)n * 4 + 16));
        MXG_HIP(h, a->d_frec.ensure((size_t)n * 4 + 16));
        uint32_t *sup = a->d_dgtmp.as<uint32_t>(), *cnt = sup + sup_words(blocks), *n_shared = cnt + blocks;
        MXG_HIP(h, hipMemsetAsync(sup, 0, (size_t)sup_words(blocks) * 4, h->stream));
        MXG_HIP(h, hipMemsetAsync(n_shared, 0, 4, h->stream));
        if (n) {
            DgAdjParams p;
            p.ret = static_cast<const unsigned long long *>(d_ret);
            p.perm = a->d_perm.as<uint32_t>();
            p.rec = a->d_rec.as<uint32_t>();
            p.n = n;
            p.cnt = cnt;
            p.sup = sup;
            p.fg = a->d_fg.as<uint32_t>();
            p.frec = a->d_frec.as<uint32_t>();
            p.n_shared = n_shared;
            p.flags_out = a->d_flags.as<uint8_t>();
            hipLaunchKernelGGL(k_dg_shared_cnt, dim3(blocks), dim3(256), 0, h->stream, p);
            hipLaunchKernelGGL(k_dg_compact, dim3(blocks), dim3(256), 0, h->stream, p);
        }
        hipLaunchKernelGGL(k_dg_last, dim3(1), dim3(1), 0, h->stream, a->d_fg.as<uint32_t>(), a->d_frec.as<uint32_t>(), n_shared,
                           static_cast<uint32_t *>(d_out) + 2 * ai);
    }
    MXG_HIP(h, hipGetLastError());
    if (h->own_stream) MXG_HIP(h, hipStreamSynchronize(h->stream));
    return MXG_OK;
}

// sender: d_all = the all-gather of every rank's dg_last_shared (u32[world][A][2]); from here on the message kernels also
// make the adjacency across the cut before this rank's first shared minimizer.  d_all = NULL: off.
int dg_set_ghosts(mxg_handle *h, const void *d_all, uint32_t world, uint32_t rank)
{
    MXG_HIP(h, hipSetDevice(h->device));
    h->dg_ghost_on = false;
    if (!d_all) return MXG_OK;
    if (rank >= world) return set_err(h, MXG_EINVAL, "rank out of range");
    const size_t A = h->asms.size();
    MXG_HIP(h, h->dg_ghost.ensure(MXG_MAX_ASSEMBLIES * 8));
    hipLaunchKernelGGL(k_dg_pick_ghost, dim3(1), dim3(MXG_MAX_ASSEMBLIES), 0, h->stream, static_cast<const uint32_t *>(d_all), (uint32_t)A,
                       rank, h->dg_ghost.as<uint32_t>());
    MXG_HIP(h, hipGetLastError());
    h->dg_ghost_on = true;
    return MXG_OK;
}

// sender: the verdicts are back (d_ret, send-buffer order).  Flags of every assembly, its shared minimizers in order, and
// how many adjacency messages go to every rank: counts[a * world + r]; d_bases = [world + 1] first global vertex id of
// every rank (device).  One host sync for all assemblies.
int dg_msg_counts(mxg_handle *h, uint32_t world, const void *d_ret, const void *d_bases, uint64_t *counts)
{
    MXG_HIP(h, hipSetDevice(h->device));
    const size_t A = h->asms.size();
    MXG_HIP(h, h->dg_cnt.ensure(A * 64 * 8 + 64));
    MXG_HIP(h, hipMemsetAsync(h->dg_cnt.p, 0, A * 64 * 8, h->stream));
    for (size_t ai = 0; ai < A; ++ai) {
        Assembly *a = h->asms[ai];
        const uint32_t n = (uint32_t)a->n_mx;
        const uint32_t blocks = (n + 255) / 256;
        MXG_HIP(h, a->d_flags.ensure(std::max<uint32_t>(n, 16)));
        MXG_HIP(h, a->d_dgtmp.ensure(((size_t)sup_words(blocks) + blocks + 16) * 4));
        MXG_HIP(h, a->d_fg.ensure((size_t)n * 4 + 16));
        MXG_HIP(h, a->d_frec.ensure((size_t)n * 4 + 16));
        uint32_t *sup = a->d_dgtmp.as<uint32_t>(), *cnt = sup + sup_words(blocks), *n_shared = cnt + blocks;
        MXG_HIP(h, hipMemsetAsync(sup, 0, (size_t)sup_words(blocks) * 4, h->stream));
        MXG_HIP(h, hipMemsetAsync(n_shared, 0, 4, h->stream));
        if (n) {
            DgAdjParams p;
            p.ret = static_cast<const unsigned long long *>(d_ret);
            p.perm = a->d_perm.as<uint32_t>();
            p.rec = a->d_rec.as<uint32_t>();
            p.n = n;
            p.cnt = cnt;
            p.sup = sup;
            p.fg = a->d_fg.as<uint32_t>();
            p.frec = a->d_frec.as<uint32_t>();
            p.n_shared = n_shared;
            p.flags_out = a->d_flags.as<uint8_t>();
            hipLaunchKernelGGL(k_dg_shared_cnt, dim3(blocks), dim3(256), 0, h->stream, p);
            hipLaunchKernelGGL(k_dg_compact, dim3(blocks), dim3(256), 0, h->stream, p);
            hipLaunchKernelGGL(k_dg_msg_count, dim3((n + DG_IPB - 1) / DG_IPB), dim3(256), 0, h->stream, p.fg, p.frec, n_shared,
                               static_cast<const uint32_t *>(d_bases), world, h->dg_cnt.as<unsigned long long>() + ai * 64,
                               h->dg_ghost_on ? h->dg_ghost.as<uint32_t>() + 2 * ai : nullptr);
            MXG_HIP(h, hipGetLastError());
        }
        a->flags_valid = true;
        a->flags_on_host = false;
    }
    std::vector<uint64_t> tmp(A * 64);
    int rc = dg_counts_to_host(h, A * 64, tmp.data());
    if (rc != MXG_OK) return rc;
    for (size_t ai = 0; ai < A; ++ai)
        for (uint32_t r = 0; r < world; ++r) counts[ai * world + r] = tmp[ai * 64 + r];
    return MXG_OK;
}

// sender: the messages of assembly a (after dg_msg_counts), bucket r at message starts[r] of d_send
int dg_pack_msgs(mxg_handle *h, Assembly *a, uint32_t assembly, uint32_t world, const void *d_bases, const uint64_t *starts,
                 void *d_send)
{
    MXG_HIP(h, hipSetDevice(h->device));
    const uint32_t n = (uint32_t)a->n_mx;
    if (n == 0) return MXG_OK;
    int rc = dg_cursor(h, world, starts);
    if (rc != MXG_OK) return rc;
    const uint32_t blocks = (n + 255) / 256;
    uint32_t *n_shared = a->d_dgtmp.as<uint32_t>() + sup_words(blocks) + blocks;
    hipLaunchKernelGGL(k_dg_pack_msgs, dim3((n + DG_IPB - 1) / DG_IPB), dim3(256), 0, h->stream, a->d_fg.as<uint32_t>(), a->d_frec.as<uint32_t>(),
                       n_shared, static_cast<const uint32_t *>(d_bases), world, assembly, h->dg_cursor.as<unsigned long long>(),
                       static_cast<uint4 *>(d_send), h->dg_ghost_on ? h->dg_ghost.as<uint32_t>() + 2 * assembly : nullptr);
    MXG_HIP(h, hipGetLastError());
    if (h->own_stream) MXG_HIP(h, hipStreamSynchronize(h->stream));
    return MXG_OK;
}

// ------------------------------------------------------------------------------------------------------
// Steady state: fixed-capacity slots.  Once the sizes of one exact exchange are known, every (source, destination) pair
// gets a slot of fixed capacity PER ASSEMBLY (64-byte header: the count + cap[a] items); the all-to-alls then have equal
// splits (no size exchange), and the receivers read the counts from the headers ON THE DEVICE (no host sync until the
// stage's last kernel).  A count above its capacity raises an overflow word and the caller repeats the step the exact way.
// The buffer is ASSEMBLY-MAJOR (round 6): assembly a's slots for all `world` destinations lie side by side, so ONE all-to-all
// per assembly carries them -- and can leave as soon as that assembly is sketched, while the next one still is (dg_pack_slots_dev).
// ------------------------------------------------------------------------------------------------------
struct DgSlots {
    uint32_t world, n_asm;
    uint32_t cap[8];      // items per assembly and slot
    uint32_t off[8];      // first verdict of assembly a inside a destination's verdict area (the verdict buffers are [world][items])
    uint32_t items;       // items per destination, all assemblies
    uint64_t abase[8];    // byte offset of assembly a's `world` slots
    uint64_t astride[8];  // bytes per slot of assembly a: 64 (header: word 0 = count) + 16 * cap[a]
    uint64_t total;       // bytes of the whole buffer
};

static int dg_layout(mxg_handle *h, uint32_t world, uint32_t n_asm, const uint32_t *cap, DgSlots *L)
{
    if (world == 0 || world > 64 || n_asm == 0 || n_asm > 8) return set_err(h, MXG_ELIMIT, "slots: 1..64 ranks, 1..8 assemblies");
    L->world = world;
    L->n_asm = n_asm;
    uint32_t off = 0;
    for (uint32_t a = 0; a < 8; ++a) {
        L->cap[a] = a < n_asm ? cap[a] : 0;
        L->off[a] = off;
        off += L->cap[a];
    }
    L->items = off;
    uint64_t at = 0;
    for (uint32_t a = 0; a < 8; ++a) {
        L->abase[a] = at;
        L->astride[a] = 64 + (uint64_t)L->cap[a] * 16;
        if (a < n_asm) at += (uint64_t)world * L->astride[a];
    }
    L->total = at;
    return MXG_OK;
}

// the slot of assembly a for destination / from source s (its first word = the count, its items 64 bytes further)
__device__ __forceinline__ const unsigned char *slot_at(const unsigned char *buf, const DgSlots &L, uint32_t a, uint32_t s)
{
    return buf + L.abase[a] + (size_t)s * L.astride[a];
}
__device__ __forceinline__ unsigned long long slot_count(const unsigned char *buf, const DgSlots &L, uint32_t a, uint32_t s)
{
    return *reinterpret_cast<const unsigned long long *>(slot_at(buf, L, a, s));
}
// a sender whose sketch was not usable (dg_pack_slots_dev) says so with a count no slot can hold: overflow like any count above
// the capacity, but NOTHING in the slot is an item then (a count that merely exceeds the capacity leaves `cap` good items)
constexpr unsigned long long DG_SLOT_INVALID = 1ull << 40;
__device__ __forceinline__ uint32_t slot_items(unsigned long long raw, uint32_t cap)
{
    return raw >= DG_SLOT_INVALID ? 0u : (uint32_t)min(raw, (unsigned long long)cap);
}

// mxg_sketch_dg_pack_slots: the sketch may still be in flight on the stream -- its length is read on the device and the batch's
// control words say whether it ended the common way (the predicate of k_pack_slot_dev, sketch.hip).  mode 0: n is the host's.
struct DgDevN {
    uint32_t mode;  // 0: host count; 1: count + predicate on the device; 2: the host knows the sketch is not usable
    const uint32_t *n_ptr, *ctrl;
    uint64_t out_cap;
    uint32_t dev_gaps, place4;
};

__global__ __launch_bounds__(256) void k_dg_pack_slots(const uint64_t *__restrict__ hash, const uint32_t *__restrict__ pos,
                                                       const uint32_t *__restrict__ rec, uint64_t n, uint32_t a, uint32_t rec_off,
                                                       const DgSlots L, unsigned char *send, uint32_t *perm, const DgDevN dv)
{
    __shared__ uint32_t lh[64], lcur[64];
    __shared__ unsigned long long lbase[64];
    if (dv.mode) {  // (block-uniform)
        bool ok = dv.mode == 1;
        if (ok) {
            const uint32_t *ctrl = dv.ctrl;
            n = *dv.n_ptr;
            ok = ctrl[0] == 0 && ctrl[6] == 0 && ctrl[13] == 0 && (dv.dev_gaps ? (ctrl[11] == 0 && ctrl[1] <= dv.place4) : ctrl[1] == 0) &&
                 (ctrl[4] | ctrl[5]) != 0 && n <= dv.out_cap;
        }
        if (!ok) {  // every destination sees a count far above any capacity: all ranks repeat the step the exact way
            if (blockIdx.x == 0 && threadIdx.x < L.world)
                atomicAdd(reinterpret_cast<unsigned long long *>(send + L.abase[a] + (size_t)threadIdx.x * L.astride[a]), DG_SLOT_INVALID);
            return;
        }
        if ((uint64_t)blockIdx.x * DG_IPB >= n) return;
    }
    if (threadIdx.x < 64) lh[threadIdx.x] = lcur[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t i0 = (uint64_t)blockIdx.x * DG_IPB;
    for (uint32_t it = 0; it < DG_IPB / 256; ++it) {
        const uint64_t i = i0 + it * 256u + threadIdx.x;
        const bool in = i < n;
        lds_hist(lh, in ? dg_owner(hash[i], L.world) : 0u, in);
    }
    __syncthreads();
    if (threadIdx.x < L.world && lh[threadIdx.x])  // the header word of the destination's slot is the bucket's cursor
        lbase[threadIdx.x] = atomicAdd(reinterpret_cast<unsigned long long *>(send + L.abase[a] + (size_t)threadIdx.x * L.astride[a]),
                                       (unsigned long long)lh[threadIdx.x]);
    __syncthreads();
    for (uint32_t it = 0; it < DG_IPB / 256; ++it) {
        const uint64_t i = i0 + it * 256u + threadIdx.x;
        const bool in = i < n;
        const uint64_t h = in ? hash[i] : 0;
        const uint32_t d = in ? dg_owner(h, L.world) : 0u;
        const uint32_t sl = lds_slot(lcur, d, in);
        if (in) {
            const uint64_t idx = lbase[d] + sl;
            if (idx < L.cap[a]) {
                reinterpret_cast<uint4 *>(send + L.abase[a] + (size_t)d * L.astride[a] + 64)[idx] =
                    make_uint4((uint32_t)h, (uint32_t)(h >> 32), pos[i], rec[i] + rec_off);
                perm[i] = d * L.items + L.off[a] + (uint32_t)idx;
            } else {
                perm[i] = 0;  // slot overflow: the receiver raises the overflow word, the step is repeated
            }
        }
    }
}

// assembly a's items of all sources, in source order, as SoA; n_out[a] = their number; *ovf |= some count > capacity
__global__ __launch_bounds__(256) void k_dg_slots_to_soa(const unsigned char *__restrict__ recv, const DgSlots L, uint32_t a,
                                                         uint64_t *hash, uint32_t *pos, uint32_t *rec, uint32_t *n_out, uint32_t *ovf)
{
    const uint32_t cap = L.cap[a];
    const uint64_t o = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const bool any = o < (uint64_t)L.world * cap;
    if (!any && o != 0) return;  // (thread 0 reads the headers whatever the capacity: a slot without room can still say "overflow")
    const uint32_t src = any ? (uint32_t)(o / cap) : 0u, idx = any ? (uint32_t)(o % cap) : 0u;
    uint32_t before = 0, mine = 0, total = 0;
    for (uint32_t s = 0; s < L.world; ++s) {
        const unsigned long long raw = slot_count(recv, L, a, s);
        if (raw > cap && o == 0) *ovf = 1;
        const uint32_t c = slot_items(raw, cap);
        if (s < src) before += c;
        if (s == src) mine = c;
        total += c;
    }
    if (o == 0) n_out[a] = total;
    if (!any || idx >= mine) return;
    const uint4 it = reinterpret_cast<const uint4 *>(slot_at(recv, L, a, src) + 64)[idx];
    const uint32_t t = before + idx;
    hash[t] = ((uint64_t)it.y << 32) | it.x;
    pos[t] = it.z;
    rec[t] = it.w;
}

// the verdicts of assembly a's items, written where the items sat: out[src][off[a] + idx]
__global__ __launch_bounds__(256) void k_dg_slot_results(const uint8_t *__restrict__ flags, const uint32_t *__restrict__ ivid,
                                                         const unsigned char *__restrict__ recv, const DgSlots L, uint32_t a,
                                                         const uint32_t *__restrict__ gbase_ptr, unsigned long long *out,
                                                         const uint64_t *__restrict__ pj_fail, uint32_t *ovf)
{
    const uint32_t cap = L.cap[a];
    const uint64_t o = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    // the owner's LDS join gave up (a key of huge multiplicity): flags and vertex ids mean nothing -- every item leaves as "no
    // vertex", so that no message is built on them, and the step is reported as overflowed (all ranks repeat it the exact way)
    const bool failed = *pj_fail != 0;
    if (failed && o == 0) *ovf = 1;
    if (o >= (uint64_t)L.world * cap) return;
    const uint32_t src = (uint32_t)(o / cap), idx = (uint32_t)(o % cap);
    uint32_t before = 0, mine = 0;
    for (uint32_t s = 0; s <= src; ++s) {
        const uint32_t c = slot_items(slot_count(recv, L, a, s), cap);
        if (s < src) before += c; else mine = c;
    }
    if (idx >= mine) return;
    const uint32_t t = before + idx, v = failed ? DG_NONE : ivid[t];
    out[(size_t)src * L.items + L.off[a] + idx] = ((unsigned long long)(v == DG_NONE ? DG_NONE : v + *gbase_ptr) << 8) | (failed ? 0u : flags[t]);
}

// adjacency messages into slots of M messages per destination (header word 0 = count)
__global__ __launch_bounds__(256) void k_dg_pack_msg_slots(const uint32_t *__restrict__ fg, const uint32_t *__restrict__ frec,
                                                           const uint32_t *__restrict__ n_shared, const uint32_t *__restrict__ bases,
                                                           uint32_t world, uint32_t assembly, uint32_t M, unsigned char *send,
                                                           const uint32_t *__restrict__ ghost)
{
    __shared__ uint32_t lh[64], lcur[64];
    __shared__ unsigned long long lbase[64];
    if (threadIdx.x < 64) lh[threadIdx.x] = lcur[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t ns = *n_shared;
    const uint32_t r0 = blockIdx.x * DG_IPB;
    if (r0 >= ns) return;  // block-uniform
    const uint64_t stride = 64 + (uint64_t)M * 16;
    for (uint32_t it = 0; it < DG_IPB / 256; ++it) {
        const uint32_t r = r0 + it * 256u + threadIdx.x;
        const DgPair pr = dg_pair(fg, frec, ns, r, ghost);
        lds_hist(lh, pr.in ? dg_vertex_owner(bases, world, pr.u) : 0u, pr.in);
        lds_hist(lh, pr.in ? dg_vertex_owner(bases, world, pr.v) : 0u, pr.in);
    }
    __syncthreads();
    if (threadIdx.x < world && lh[threadIdx.x])
        lbase[threadIdx.x] = atomicAdd(reinterpret_cast<unsigned long long *>(send + (size_t)threadIdx.x * stride),
                                       (unsigned long long)lh[threadIdx.x]);
    __syncthreads();
    for (uint32_t it = 0; it < DG_IPB / 256; ++it) {
        const uint32_t r = r0 + it * 256u + threadIdx.x;
        const DgPair pr = dg_pair(fg, frec, ns, r, ghost);
        const bool in = pr.in;
        const uint32_t u = pr.u, v = pr.v;
        const uint32_t ou = in ? dg_vertex_owner(bases, world, u) : 0u, ov = in ? dg_vertex_owner(bases, world, v) : 0u;
        const uint32_t su = lds_slot(lcur, ou, in);
        const uint32_t sv = lds_slot(lcur, ov, in);
        if (in) {
            const uint64_t iu = lbase[ou] + su, iv = lbase[ov] + sv;
            if (iu < M) reinterpret_cast<uint4 *>(send + (size_t)ou * stride + 64)[iu] = make_uint4(assembly, u - bases[ou], v, 0u);
            if (iv < M) reinterpret_cast<uint4 *>(send + (size_t)ov * stride + 64)[iv] = make_uint4(assembly | 256u, v - bases[ov], u, 0u);
        }
    }
}

__global__ __launch_bounds__(256) void k_apply_msg_slots(const unsigned char *__restrict__ recv, uint32_t world, uint32_t M,
                                                         uint32_t nvs, uint32_t *adj, uint32_t *ovf)
{
    const uint64_t o = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (o >= (uint64_t)world * M) return;
    const uint32_t src = (uint32_t)(o / M), idx = (uint32_t)(o % M);
    const uint64_t stride = 64 + (uint64_t)M * 16;
    const unsigned long long raw = *reinterpret_cast<const unsigned long long *>(recv + (size_t)src * stride);
    if (raw > M && idx == 0) *ovf = 1;
    if (idx >= min(raw, (unsigned long long)M)) return;
    const uint4 m = reinterpret_cast<const uint4 *>(recv + (size_t)src * stride + 64)[idx];
    adj[((size_t)(m.x & 255u) * nvs + m.y) * 2u + ((m.x >> 8) ? 1u : 0u)] = m.z;  // (graph.hip: adj[a][u] = {successor, predecessor})
}

// sender: assembly a's minimizers into the slots of d_send (headers zeroed by the caller)
int dg_pack_slots(mxg_handle *h, Assembly *a, uint32_t ai, uint32_t rec_offset, uint32_t world, uint32_t n_asm, const uint32_t *cap,
                  void *d_send)
{
    MXG_HIP(h, hipSetDevice(h->device));
    DgSlots L;
    int rc = dg_layout(h, world, n_asm, cap, &L);
    if (rc != MXG_OK) return rc;
    if (ai >= n_asm) return MXG_EINVAL;
    MXG_HIP(h, a->d_perm.ensure(std::max<uint64_t>(a->n_mx * 4, 16)));
    // the slot headers are the buckets' cursors: this assembly's are cleared here, on the stream the packing kernel runs on
    MXG_HIP(h, hipMemset2DAsync(static_cast<unsigned char *>(d_send) + L.abase[ai], L.astride[ai], 0, 64, world, h->stream));
    if (a->n_mx)
        hipLaunchKernelGGL(k_dg_pack_slots, dim3((uint32_t)((a->n_mx + DG_IPB - 1) / DG_IPB)), dim3(256), 0, h->stream,
                           a->d_hash.as<uint64_t>(), a->d_pos.as<uint32_t>(), a->d_rec.as<uint32_t>(), a->n_mx, ai, rec_offset, L,
                           static_cast<unsigned char *>(d_send), a->d_perm.as<uint32_t>(), DgDevN{0, nullptr, nullptr, 0, 0, 0});
    MXG_HIP(h, hipGetLastError());
    if (h->own_stream) MXG_HIP(h, hipStreamSynchronize(h->stream));
    return MXG_OK;
}

// the same behind a sketch that is still in flight on `st` (sketch_assemblies, mxg_sketch_dg_pack_slots): the count is read on the
// device (n_ptr; ctrl = the batch's control words), the grid covers what the output arrays hold.  mode 2: the host already knows
// that the sketch did not go through the one-batch pipeline; mode 0 with n = 0: an assembly without k-mers.
int dg_pack_slots_dev(mxg_handle *h, Assembly *a, uint32_t ai, const DgPackReq &rq, hipStream_t st, uint32_t mode, const uint32_t *n_ptr,
                      const uint32_t *ctrl, uint64_t out_cap, uint32_t dev_gaps, uint32_t place4)
{
    DgSlots L;
    int rc = dg_layout(h, rq.world, rq.n_asm, rq.cap, &L);
    if (rc != MXG_OK) return rc;
    if (ai >= rq.n_asm) return MXG_EINVAL;
    unsigned char *send = static_cast<unsigned char *>(rq.d_send);
    MXG_HIP(h, hipMemset2DAsync(send + L.abase[ai], L.astride[ai], 0, 64, rq.world, st));
    if (mode == 1) MXG_HIP(h, a->d_perm.ensure(std::max<uint64_t>(out_cap * 4, 16)));
    const uint64_t bound = mode == 1 ? out_cap : 0;
    const uint32_t grid = (uint32_t)std::max<uint64_t>((bound + DG_IPB - 1) / DG_IPB, 1);
    if (mode != 0)
        hipLaunchKernelGGL(k_dg_pack_slots, dim3(grid), dim3(256), 0, st, a->d_hash.as<uint64_t>(), a->d_pos.as<uint32_t>(),
                           a->d_rec.as<uint32_t>(), (uint64_t)0, ai, rq.rec_off[ai], L, send, a->d_perm.as<uint32_t>(),
                           DgDevN{mode, n_ptr, ctrl, out_cap, dev_gaps, place4});
    MXG_HIP(h, hipGetLastError());
    return MXG_OK;
}

// owner: everything between the item exchange and the verdict exchange, no host sync: items of every assembly out of
// the slots, uniqueness / intersection / local vertex ids with the counts read on the device, vertex count -> d_nv
int dg_owner_slots(mxg_handle *h, uint32_t world, uint32_t n_asm, const uint32_t *cap, const void *d_recv, void *d_nv)
{
    MXG_HIP(h, hipSetDevice(h->device));
    DgSlots L;
    int rc = dg_layout(h, world, n_asm, cap, &L);
    if (rc != MXG_OK) return rc;
    if (n_asm != h->asms.size()) return set_err(h, MXG_EINVAL, "slots: the owner handle has %zu assemblies", h->asms.size());
    MXG_HIP(h, h->d_nmx.ensure(MXG_MAX_ASSEMBLIES * 4 + 16));
    uint32_t *ovf = h->d_nmx.as<uint32_t>() + MXG_MAX_ASSEMBLIES;  // the word after the counts
    MXG_HIP(h, hipMemsetAsync(ovf, 0, 16, h->stream));              // (... and the join's "failed" word behind it)
    GraphBounds gb;
    for (uint32_t ai = 0; ai < n_asm; ++ai) {
        Assembly *a = h->asms[ai];
        const uint64_t bound = (uint64_t)world * L.cap[ai];
        MXG_HIP(h, a->d_hash.ensure(std::max<uint64_t>(bound * 8, 16)));
        MXG_HIP(h, a->d_pos.ensure(std::max<uint64_t>(bound * 4, 16)));
        MXG_HIP(h, a->d_rec.ensure(std::max<uint64_t>(bound * 4, 16)));
        hipLaunchKernelGGL(k_dg_slots_to_soa, dim3((uint32_t)std::max<uint64_t>((bound + 255) / 256, 1)), dim3(256), 0, h->stream,
                               static_cast<const unsigned char *>(d_recv), L, ai, a->d_hash.as<uint64_t>(), a->d_pos.as<uint32_t>(),
                               a->d_rec.as<uint32_t>(), h->d_nmx.as<uint32_t>(), ovf);
        a->n_mx = bound;  // an upper bound until mxg_dg_edges_slots reads the counts back
        a->has_sketch = true;
        a->fwd_valid = false;
        a->foreign_sketch = true;
        a->host_valid = false;
        gb.n_bound[ai] = bound;
        gb.n_ptr[ai] = h->d_nmx.as<uint32_t>() + ai;
    }
    MXG_HIP(h, hipGetLastError());
    rc = build_graph(h, GRAPH_DG_VERTICES, d_nv, 0, &gb);
    if (rc == MXG_OK && h->own_stream) MXG_HIP(h, hipStreamSynchronize(h->stream));  // d_nv feeds a collective elsewhere
    return rc;
}

int dg_slot_results(mxg_handle *h, uint32_t world, uint32_t n_asm, const uint32_t *cap, const void *d_recv, const void *d_gbase,
                    void *d_out)
{
    MXG_HIP(h, hipSetDevice(h->device));
    DgSlots L;
    int rc = dg_layout(h, world, n_asm, cap, &L);
    if (rc != MXG_OK) return rc;
    for (uint32_t ai = 0; ai < n_asm; ++ai) {
        Assembly *a = h->asms[ai];
        const uint64_t bound = (uint64_t)world * L.cap[ai];
        if (!bound) continue;
        if (h->graph.nv_stride == 0) {  // no vertex at this owner: nobody made d_ivid
            MXG_HIP(h, a->d_ivid.ensure(bound * 4 + 16));
            MXG_HIP(h, hipMemsetAsync(a->d_ivid.p, 0xFF, bound * 4, h->stream));
        }
        hipLaunchKernelGGL(k_dg_slot_results, dim3((uint32_t)((bound + 255) / 256)), dim3(256), 0, h->stream,
                           a->d_flags.as<uint8_t>(), a->d_ivid.as<uint32_t>(), static_cast<const unsigned char *>(d_recv), L, ai,
                           static_cast<const uint32_t *>(d_gbase), static_cast<unsigned long long *>(d_out), dg_pj_fail_word(h),
                           h->d_nmx.as<uint32_t>() + MXG_MAX_ASSEMBLIES);
    }
    MXG_HIP(h, hipGetLastError());
    if (h->own_stream) MXG_HIP(h, hipStreamSynchronize(h->stream));
    return MXG_OK;
}

// sender: verdicts -> flags, shared minimizers in order, adjacency messages of every assembly into the message slots
int dg_pack_msg_slots(mxg_handle *h, uint32_t world, uint32_t M, const void *d_ret, const void *d_bases, void *d_send)
{
    MXG_HIP(h, hipSetDevice(h->device));
    const size_t A = h->asms.size();
    MXG_HIP(h, hipMemset2DAsync(d_send, 64 + (size_t)M * 16, 0, 64, world, h->stream));  // headers = cursors
    for (size_t ai = 0; ai < A; ++ai) {
        Assembly *a = h->asms[ai];
        const uint32_t n = (uint32_t)a->n_mx;
        const uint32_t blocks = (n + 255) / 256;
        MXG_HIP(h, a->d_flags.ensure(std::max<uint32_t>(n, 16)));
        MXG_HIP(h, a->d_dgtmp.ensure(((size_t)sup_words(blocks) + blocks + 16) * 4));
        MXG_HIP(h, a->d_fg.ensure((size_t)n * 4 + 16));
        MXG_HIP(h, a->d_frec.ensure((size_t)n * 4 + 16));
        uint32_t *sup = a->d_dgtmp.as<uint32_t>(), *cnt = sup + sup_words(blocks), *n_shared = cnt + blocks;
        MXG_HIP(h, hipMemsetAsync(sup, 0, (size_t)sup_words(blocks) * 4, h->stream));
        MXG_HIP(h, hipMemsetAsync(n_shared, 0, 4, h->stream));
        if (n) {
            DgAdjParams p;
            p.ret = static_cast<const unsigned long long *>(d_ret);
            p.perm = a->d_perm.as<uint32_t>();
            p.rec = a->d_rec.as<uint32_t>();
            p.n = n;
            p.cnt = cnt;
            p.sup = sup;
            p.fg = a->d_fg.as<uint32_t>();
            p.frec = a->d_frec.as<uint32_t>();
            p.n_shared = n_shared;
            p.flags_out = a->d_flags.as<uint8_t>();
            hipLaunchKernelGGL(k_dg_shared_cnt, dim3(blocks), dim3(256), 0, h->stream, p);
            hipLaunchKernelGGL(k_dg_compact, dim3(blocks), dim3(256), 0, h->stream, p);
            hipLaunchKernelGGL(k_dg_pack_msg_slots, dim3((n + DG_IPB - 1) / DG_IPB), dim3(256), 0, h->stream, p.fg, p.frec, n_shared,
                               static_cast<const uint32_t *>(d_bases), world, (uint32_t)ai, M, static_cast<unsigned char *>(d_send),
                               h->dg_ghost_on ? h->dg_ghost.as<uint32_t>() + 2 * ai : nullptr);
        }
        a->flags_valid = true;
        a->flags_on_host = false;
    }
    MXG_HIP(h, hipGetLastError());
    if (h->own_stream) MXG_HIP(h, hipStreamSynchronize(h->stream));
    return MXG_OK;
}

// owner: adjacency out of the message slots, edges, the stage's one host sync; *overflow != 0: some slot was too small
// somewhere on this rank's receiving side, the results are incomplete
int dg_edges_slots(mxg_handle *h, const void *d_recv, uint32_t world, uint32_t M, uint64_t *n_vertices, uint64_t *n_edges,
                   uint32_t *overflow)
{
    MXG_HIP(h, hipSetDevice(h->device));
    Graph &g = h->graph;
    const uint32_t A = (uint32_t)h->asms.size();
    const uint64_t nvs = g.nv_stride;
    uint32_t *ovf = h->d_nmx.as<uint32_t>() + MXG_MAX_ASSEMBLIES;
    if (nvs) {
        const size_t anv = (size_t)A * nvs;
        MXG_HIP(h, h->g_nxt.ensure(2 * anv * 4));
        MXG_HIP(h, hipMemsetAsync(h->g_nxt.p, 0xFF, 2 * anv * 4, h->stream));
        const uint64_t tot = (uint64_t)world * M;
        if (tot)
            hipLaunchKernelGGL(k_apply_msg_slots, dim3((uint32_t)((tot + 255) / 256)), dim3(256), 0, h->stream,
                               static_cast<const unsigned char *>(d_recv), world, M, (uint32_t)nvs, h->g_nxt.as<uint32_t>(),
                               ovf);
        MXG_HIP(h, hipGetLastError());
    }
    int rc = build_graph(h, GRAPH_DG_EDGES_APPLIED);
    if (rc != MXG_OK) return rc;
    uint32_t back[MXG_MAX_ASSEMBLIES + 4];
    MXG_HIP(h, hipMemcpy(back, h->d_nmx.p, sizeof back, hipMemcpyDeviceToHost));
    for (uint32_t ai = 0; ai < A; ++ai) h->asms[ai]->n_mx = back[ai];  // the real item counts
    if (back[MXG_MAX_ASSEMBLIES + 2] | back[MXG_MAX_ASSEMBLIES + 3]) h->dg_pj_off = true;  // (the LDS join failed: reported as overflow above)
    if (n_vertices) *n_vertices = g.nv;
    if (n_edges) *n_edges = g.ne;
    if (overflow) *overflow = back[MXG_MAX_ASSEMBLIES];
    return MXG_OK;
}

}  // namespace mxg
