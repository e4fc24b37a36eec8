// paths.hip -- first "next" row: on the minimizer graph already resident in HBM, the global edge filter,
// the per-component branch filtering and the extraction of linear paths, i.e. what the reference does with igraph at
//   filter_graph_global   bin/ntjoin.py:80-89        filter_graph            bin/ntjoin.py:69-77
//   is_graph_linear       bin/ntjoin.py:105-111      check_circularity       bin/ntjoin.py:113-135
//   determine_source_vertex bin/ntjoin.py:91-103     find_paths(_process)    bin/ntjoin.py:137-176
//
// Data-parallel formulation (vertices = shared minimizers, a few million; degree <= 2 per assembly):
//   components            hook (CAS on roots, larger index under smaller, path halving) + compress, until nothing changes
//   branch filtering      for t = n, n+1, ... : degree (atomics), per-component "has a branch node" flag, kill the edges
//                         incident to a branch node of such a component whose weight < t
//   cycles                per all-degree-2 component: min-position vertex (atomicMin of pos<<32|v), drop the edge to
//                         its highest-position neighbour
//   ordering              every chain edge becomes two arcs; succ(u->v) = the arc leaving v that does not return to u;
//                         pointer jumping gives each arc the end of its list and the number of arcs up to it, so the
//                         arcs of the source->target list know their vertex's index in the path
#include <algorithm>

#include "mxg_internal.h"
#include "scan_kernels.h"

namespace mxg {

static constexpr uint32_t NIL = 0xFFFFFFFFu;
// scratch buffers of this file (mxg_handle::pbuf); PV / PF / PC keep the paths on the device after find_paths
enum { ALIVE, COMP, SUB, DEG, NONLIN, FLAG, CNTV, CNTE, CNTD1, MAXDEG, FILL, NB, NBE, KEY0, KEY1, KEY2, SUCC0, CNT0, END0,
       SUCC1, CNT1, END1, SIZE, ISSRC, FIRST, RANK, BSUM, TOTAL, PV, PF, PC, SG_FLAG, SG_EXCL, SG_FIRST, SG_REC, SG_PATH,
       SG_STAT, XT_MIN, XT_MAX, PBUF_COUNT };
static_assert(PBUF_COUNT <= 48, "mxg_handle::pbuf too small");

__device__ __forceinline__ uint32_t find_root(uint32_t *parent, uint32_t v)
{
    uint32_t p = parent[v];
    while (p != v) {
        uint32_t g = parent[p];
        if (g != p) parent[v] = g;  // path halving (benign race: only ever points closer to the root)
        v = p;
        p = g;
    }
    return v;
}

__global__ __launch_bounds__(256) void kp_iota(uint32_t *a, uint32_t n)
{
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) a[i] = i;
}

// global filter: edges lighter than n are dropped unless n <= every assembly's weight (bin/ntjoin.py:83)
__global__ __launch_bounds__(256) void kp_alive(const double *__restrict__ ew, uint32_t ne, double n_min, int do_filter,
                                                uint8_t *__restrict__ alive)
{
    uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e < ne) alive[e] = (do_filter && ew[e] < n_min) ? 0 : 1;
}

__global__ __launch_bounds__(256) void kp_hook(const uint32_t *__restrict__ eu, const uint32_t *__restrict__ ev,
                                               const uint8_t *__restrict__ alive, uint32_t ne, uint32_t *parent,
                                               uint32_t *changed)
{
    uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= ne || !alive[e]) return;
    const uint32_t u = eu[e], v = ev[e];
    while (true) {
        const uint32_t ru = find_root(parent, u), rv = find_root(parent, v);
        if (ru == rv) break;
        const uint32_t hi = max(ru, rv), lo = min(ru, rv);
        // hook only while hi is still a root (a CAS, not a min: a min would also re-point a hi that has been hooked
        // meanwhile and so cut it off the tree it had just joined)
        if (atomicCAS(&parent[hi], hi, lo) == hi) {
            *changed = 1;
            break;
        }
    }
}

__global__ __launch_bounds__(256) void kp_compress(uint32_t *parent, uint32_t n)
{
    // read-only walk: with path halving here, another thread's late `parent[v] = grandparent` could land after this
    // thread's `parent[v] = root` and leave v one level short of its root
    uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= n) return;
    uint32_t r = v, p = parent[r];
    while (p != r) {
        r = p;
        p = parent[r];
    }
    parent[v] = r;
}

// number of components = number of roots (block-reduced, one atomic per block)
__global__ __launch_bounds__(256) void kp_count_roots(const uint32_t *__restrict__ parent, uint32_t n, uint32_t *count)
{
    __shared__ uint32_t sh[256];
    uint32_t v = blockIdx.x * 256u + threadIdx.x;
    sh[threadIdx.x] = (v < n && parent[v] == v) ? 1u : 0u;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0 && sh[0]) atomicAdd(count, sh[0]);
}

__global__ __launch_bounds__(256) void kp_degree(const uint32_t *__restrict__ eu, const uint32_t *__restrict__ ev,
                                                 const uint8_t *__restrict__ alive, uint32_t ne, uint32_t *deg)
{
    uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= ne || !alive[e]) return;
    atomicAdd(&deg[eu[e]], 1u);
    atomicAdd(&deg[ev[e]], 1u);
}

// nonlin[root of the globally filtered component] = 1 if the component has a vertex of degree > 2
__global__ __launch_bounds__(256) void kp_mark_branch(const uint32_t *__restrict__ deg, const uint32_t *__restrict__ comp,
                                                      uint32_t nv, uint8_t *nonlin, uint32_t *any)
{
    uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v < nv && deg[v] > 2) {
        nonlin[comp[v]] = 1;
        *any = 1;
    }
}

// filter_graph on every component that is not linear yet: drop edges incident to a branch node whose weight < t
__global__ __launch_bounds__(256) void kp_branch_filter(const uint32_t *__restrict__ eu, const uint32_t *__restrict__ ev,
                                                        const double *__restrict__ ew, uint32_t ne,
                                                        const uint32_t *__restrict__ deg, const uint32_t *__restrict__ comp,
                                                        const uint8_t *__restrict__ nonlin, double t, uint8_t *alive)
{
    uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= ne || !alive[e]) return;
    uint32_t u = eu[e], v = ev[e];
    if (nonlin[comp[u]] && (deg[u] > 2 || deg[v] > 2) && ew[e] < t) alive[e] = 0;
}

struct SubStats {  // per root of a sub-component (after filtering)
    uint32_t *cnt_v, *cnt_e, *cnt_d1, *max_deg;
};

// Per-root tallies with one atomic per (wave, root): neighbouring vertex ids are neighbours in the first assembly, so a
// wave usually sees one or two roots, and a 6 M-vertex chain would otherwise queue 6 M same-address atomics (~10 ns
// each on MI355X: kp_vertex_stats and kp_edge_stats took 70 ms each).  `pred` lanes contribute 1 to arr[key].
__device__ __forceinline__ void wave_count_by_key(uint32_t *arr, uint32_t key, bool active, bool pred)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t pm = __ballot(active && pred);
    for (uint64_t todo = __ballot(active); todo;) {
        const int leader = __builtin_ctzll(todo);
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
        const uint64_t same = __ballot(active && key == k0);
        const uint32_t c = (uint32_t)__popcll(same & pm);
        if ((int)lane == leader && c) atomicAdd(&arr[k0], c);
        todo &= ~same;
    }
}

__global__ __launch_bounds__(256) void kp_vertex_stats(const uint32_t *__restrict__ deg, const uint32_t *__restrict__ sub,
                                                       uint32_t nv, SubStats s)
{
    uint32_t v = blockIdx.x * 256u + threadIdx.x;
    const bool in = v < nv;
    const uint32_t r = in ? sub[v] : 0u, d = in ? deg[v] : 0u;
    wave_count_by_key(s.cnt_v, r, in, true);
    wave_count_by_key(s.cnt_d1, r, in, d == 1);
    if (in && d > 2) s.max_deg[r] = 3;  // only "some vertex has degree > 2" is ever asked (sub_kind)
}

// edge counts + neighbour slots (only meaningful where max degree <= 2)
__global__ __launch_bounds__(256) void kp_edge_stats(const uint32_t *__restrict__ eu, const uint32_t *__restrict__ ev,
                                                     const uint8_t *__restrict__ alive, uint32_t ne,
                                                     const uint32_t *__restrict__ sub, uint32_t *cnt_e, uint32_t *fill,
                                                     uint32_t *nb, uint32_t *nbe)
{
    uint32_t e = blockIdx.x * 256u + threadIdx.x;
    const bool in = e < ne && alive[e];
    wave_count_by_key(cnt_e, in ? sub[eu[e]] : 0u, in, true);
    if (!in) return;
    uint32_t u = eu[e], v = ev[e];
    uint32_t su = atomicAdd(&fill[u], 1u), sv = atomicAdd(&fill[v], 1u);
    if (su < 2) { nb[2 * u + su] = v; nbe[2 * u + su] = e; }
    if (sv < 2) { nb[2 * v + sv] = u; nbe[2 * v + sv] = e; }
}

// classification of a sub-component by its root r: 1 = simple chain (>= 2 vertices), 2 = simple cycle, 0 = neither
__device__ __forceinline__ int sub_kind(const SubStats &s, uint32_t r)
{
    if (s.max_deg[r] > 2) return 0;
    if (s.cnt_v[r] >= 2 && s.cnt_e[r] == s.cnt_v[r] - 1 && s.cnt_d1[r] == 2) return 1;
    if (s.cnt_d1[r] == 0 && s.cnt_e[r] == s.cnt_v[r] && s.cnt_v[r] >= 3) return 2;
    return 0;
}

// cycles: the vertex with the smallest position in the first highest-weight assembly (bin/ntjoin.py:116-124)
__global__ __launch_bounds__(256) void kp_cycle_min(const uint32_t *__restrict__ sub, uint32_t nv, SubStats s,
                                                    const uint32_t *__restrict__ pos_first, unsigned long long *key)
{
    uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= nv) return;
    uint32_t r = sub[v];
    if (sub_kind(s, r) == 2) atomicMin(&key[r], ((unsigned long long)pos_first[v] << 32) | v);
}

// ... and the edge to its highest-position neighbour is dropped (bin/ntjoin.py:125-133); the two become the endpoints
__global__ __launch_bounds__(256) void kp_cycle_break(uint32_t nv, const uint32_t *__restrict__ sub, SubStats s,
                                                      const uint32_t *__restrict__ pos_first,
                                                      const unsigned long long *__restrict__ key, uint32_t *nb, uint32_t *nbe,
                                                      uint8_t *alive, uint32_t *deg)
{
    uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= nv || sub[r] != r || sub_kind(s, r) != 2) return;
    uint32_t mv = (uint32_t)key[r];
    uint32_t a = nb[2 * mv], b = nb[2 * mv + 1];
    uint32_t slot = pos_first[b] > pos_first[a] ? 1u : 0u;  // ties: first neighbour (stable sort, reverse=True)
    uint32_t hn = nb[2 * mv + slot], e = nbe[2 * mv + slot];
    alive[e] = 0;
    nb[2 * mv + slot] = NIL;
    uint32_t hs = nb[2 * hn] == mv && nbe[2 * hn] == e ? 0u : 1u;
    nb[2 * hn + hs] = NIL;
    deg[mv] = 1;
    deg[hn] = 1;
}

// endpoints of chains (incl. broken cycles): source = smallest position in the LAST highest-weight assembly, target =
// largest (bin/ntjoin.py:95-103; ties resolved towards the later vertex, as `.pop()` does)
__global__ __launch_bounds__(256) void kp_endpoints(const uint32_t *__restrict__ deg, const uint32_t *__restrict__ sub,
                                                    uint32_t nv, SubStats s, const uint32_t *__restrict__ pos_last,
                                                    unsigned long long *src_key, unsigned long long *tgt_key)
{
    uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= nv || deg[v] != 1) return;
    uint32_t r = sub[v];
    if (sub_kind(s, r) == 0) return;
    atomicMin(&src_key[r], ((unsigned long long)pos_last[v] << 32) | (0xFFFFFFFFu - v));
    atomicMax(&tgt_key[r], ((unsigned long long)pos_last[v] << 32) | v);
}

// arcs: arc 2e = eu[e]->ev[e], arc 2e+1 = ev[e]->eu[e]
__global__ __launch_bounds__(256) void kp_arc_init(const uint32_t *__restrict__ eu, const uint32_t *__restrict__ ev,
                                                   const uint8_t *__restrict__ alive, uint32_t ne,
                                                   const uint32_t *__restrict__ sub, SubStats s,
                                                   const uint32_t *__restrict__ nb, const uint32_t *__restrict__ nbe,
                                                   uint32_t *succ, uint32_t *cnt, uint32_t *endv)
{
    uint32_t a = blockIdx.x * 256u + threadIdx.x;
    if (a >= 2 * ne) return;
    uint32_t e = a >> 1;
    succ[a] = NIL;
    cnt[a] = 0;
    endv[a] = NIL;
    if (!alive[e] || sub_kind(s, sub[eu[e]]) == 0) return;
    uint32_t from = (a & 1) ? ev[e] : eu[e], to = (a & 1) ? eu[e] : ev[e];
    cnt[a] = 1;
    endv[a] = to;
    for (int sl = 0; sl < 2; ++sl) {  // the arc leaving `to` that does not go back over edge e
        uint32_t x = nb[2 * to + sl];
        if (x != NIL && nbe[2 * to + sl] != e) {
            uint32_t e2 = nbe[2 * to + sl];
            succ[a] = 2 * e2 + (eu[e2] == to ? 0u : 1u);
        }
    }
    (void)from;
}

__global__ __launch_bounds__(256) void kp_arc_jump(const uint32_t *__restrict__ succ_in, const uint32_t *__restrict__ cnt_in,
                                                   const uint32_t *__restrict__ end_in, uint32_t *succ_out, uint32_t *cnt_out,
                                                   uint32_t *end_out, uint32_t n_arcs)
{
    uint32_t a = blockIdx.x * 256u + threadIdx.x;
    if (a >= n_arcs) return;
    uint32_t s = succ_in[a];
    if (s == NIL) {
        succ_out[a] = NIL;
        cnt_out[a] = cnt_in[a];
        end_out[a] = end_in[a];
    } else {
        succ_out[a] = succ_in[s];
        cnt_out[a] = cnt_in[a] + cnt_in[s];
        end_out[a] = end_in[s];
    }
}

__global__ __launch_bounds__(256) void kp_widen(const uint8_t *__restrict__ in, uint32_t n, uint32_t *__restrict__ out)
{
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// accepted sub-components, flagged at their source vertex: size[v] = number of vertices of the path starting at v
__global__ __launch_bounds__(256) void kp_path_sizes(uint32_t nv, const uint32_t *__restrict__ sub, SubStats s,
                                                     const unsigned long long *__restrict__ src_key,
                                                     const unsigned long long *__restrict__ tgt_key, uint32_t *size,
                                                     uint8_t *is_src)
{
    uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= nv || sub[r] != r || sub_kind(s, r) == 0) return;
    uint32_t src = 0xFFFFFFFFu - (uint32_t)src_key[r], tgt = (uint32_t)tgt_key[r];
    if (src == tgt) return;  // equal positions: the reference's shortest path has one vertex and is rejected
    size[src] = s.cnt_v[r];
    is_src[src] = 1;
}

struct WriteParams {
    const uint32_t *eu, *ev;
    const uint8_t *alive;
    uint32_t ne;
    const uint32_t *sub;
    const unsigned long long *src_key, *tgt_key;
    const uint32_t *cnt, *endv;
    const uint32_t *cnt_v;
    const uint8_t *is_src;
    const uint32_t *first;   // exclusive scan of size[] over vertices: start of the path whose source is v
    const uint32_t *rank;    // exclusive scan of is_src: path index
    uint32_t *path_vertex;
};

__global__ __launch_bounds__(256) void kp_write_arcs(const WriteParams p)
{
    uint32_t a = blockIdx.x * 256u + threadIdx.x;
    if (a >= 2 * p.ne || p.cnt[a] == 0) return;
    uint32_t e = a >> 1;
    uint32_t to = (a & 1) ? p.eu[e] : p.ev[e];
    uint32_t r = p.sub[to];
    uint32_t src = 0xFFFFFFFFu - (uint32_t)p.src_key[r], tgt = (uint32_t)p.tgt_key[r];
    if (!p.is_src[src] || p.endv[a] != tgt) return;  // not an arc of the source->target list
    // a is the i-th arc of that list with cnt = V - i arcs up to the end; it enters path index i
    p.path_vertex[p.first[src] + (p.cnt_v[r] - p.cnt[a])] = to;
}

__global__ __launch_bounds__(256) void kp_write_sources(uint32_t nv, const uint8_t *__restrict__ is_src,
                                                        const uint32_t *__restrict__ first, const uint32_t *__restrict__ rank,
                                                        const uint32_t *__restrict__ comp, const uint32_t *__restrict__ size,
                                                        uint32_t *path_vertex, uint64_t *path_first, uint32_t *path_comp)
{
    uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= nv || !is_src[v]) return;
    path_vertex[first[v]] = v;
    path_first[rank[v]] = first[v];
    path_comp[rank[v]] = comp[v];
    (void)size;
}

// ------------------------------------------------------------------------------------------------------
static int exclusive_scan_u32(mxg_handle *h, const uint32_t *in, uint32_t n, DevBuf &bsum, uint32_t *out, uint64_t *d_total)
{
    const uint32_t tiles = (n + TILE - 1) / TILE;
    MXG_HIP(h, bsum.ensure((size_t)tiles * 4 + 16));
    hipLaunchKernelGGL(k_tile_sum_u32, dim3(tiles), dim3(256), 0, h->stream, in, n, bsum.as<uint32_t>());
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, h->stream, bsum.as<uint32_t>(), tiles, d_total);
    hipLaunchKernelGGL(k_tile_excl_u32, dim3(tiles), dim3(256), 0, h->stream, in, n, bsum.as<uint32_t>(), out);
    MXG_HIP(h, hipGetLastError());
    return MXG_OK;
}

static int components(mxg_handle *h, const uint32_t *eu, const uint32_t *ev, const uint8_t *alive, uint32_t ne, uint32_t nv,
                      uint32_t *parent, uint32_t *d_flag)
{
    const dim3 gv((nv + 255) / 256), ge((ne + 255) / 256), b(256);
    hipLaunchKernelGGL(kp_iota, gv, b, 0, h->stream, parent, nv);
    for (int it = 0; it < 64; ++it) {
        MXG_HIP(h, hipMemsetAsync(d_flag, 0, 4, h->stream));
        if (ne) hipLaunchKernelGGL(kp_hook, ge, b, 0, h->stream, eu, ev, alive, ne, parent, d_flag);
        hipLaunchKernelGGL(kp_compress, gv, b, 0, h->stream, parent, nv);
        uint32_t changed = 0;
        MXG_HIP(h, hipMemcpyAsync(&changed, d_flag, 4, hipMemcpyDeviceToHost, h->stream));
        MXG_HIP(h, hipStreamSynchronize(h->stream));
        if (!changed) return MXG_OK;
    }
    return set_err(h, MXG_EDEVICE, "internal error: component labelling did not converge");
}

int find_paths(mxg_handle *h, int64_t n_min)
{
    Graph &g = h->graph;
    if (!g.valid) return set_err(h, MXG_EINVAL, "mxg_find_paths: call mxg_build_graph first");
    MXG_HIP(h, hipSetDevice(h->device));
    Paths &P = h->paths;
    P = Paths();
    const uint32_t nv = (uint32_t)g.nv, ne = (uint32_t)g.ne, A = g.n_asm;
    if (nv == 0) {
        P.first.assign(1, 0);
        P.valid = true;
        return MXG_OK;
    }
    // weights: global filter rule, iteration bound, the two "highest weight" assemblies the reference consults
    double wmin = h->asms[0]->weight, wmax = h->asms[0]->weight, wsum = 0;
    for (auto *a : h->asms) {
        wmin = std::min(wmin, a->weight);
        wmax = std::max(wmax, a->weight);
        wsum += a->weight;
    }
    uint32_t first_max = 0, last_max = 0;
    for (uint32_t a = 0; a < A; ++a)
        if (h->asms[a]->weight == wmax) {
            last_max = a;
        }
    for (uint32_t a = 0; a < A; ++a)
        if (h->asms[a]->weight == wmax) {
            first_max = a;
            break;
        }
    const uint32_t *eu = h->g_eu.as<uint32_t>(), *ev = h->g_ev.as<uint32_t>();
    const double *ew = h->g_ew.as<double>();
    const uint32_t *pos_first = h->g_vpos.as<uint32_t>() + (size_t)first_max * g.nv_stride;
    const uint32_t *pos_last = h->g_vpos.as<uint32_t>() + (size_t)last_max * g.nv_stride;

    DevBuf *B = h->pbuf;  // scratch of this stage
    const size_t ne1 = std::max<uint32_t>(ne, 1);
    MXG_HIP(h, B[ALIVE].ensure(ne1));
    MXG_HIP(h, B[COMP].ensure((size_t)nv * 4));
    MXG_HIP(h, B[SUB].ensure((size_t)nv * 4));
    MXG_HIP(h, B[DEG].ensure((size_t)nv * 4));
    MXG_HIP(h, B[NONLIN].ensure(nv));
    MXG_HIP(h, B[FLAG].ensure(64));
    const dim3 gv((nv + 255) / 256), ge((ne1 + 255) / 256), b(256);
    uint8_t *alive = B[ALIVE].as<uint8_t>();
    uint32_t *comp = B[COMP].as<uint32_t>(), *sub = B[SUB].as<uint32_t>(), *deg = B[DEG].as<uint32_t>();
    uint32_t *d_flag = B[FLAG].as<uint32_t>();

    const int do_filter = !((double)n_min <= wmin);
    if (ne) hipLaunchKernelGGL(kp_alive, ge, b, 0, h->stream, ew, ne, (double)n_min, do_filter, alive);
    int rc = components(h, eu, ev, alive, ne, nv, comp, d_flag);
    if (rc != MXG_OK) return rc;
    MXG_HIP(h, hipMemsetAsync(d_flag + 1, 0, 4, h->stream));
    hipLaunchKernelGGL(kp_count_roots, gv, b, 0, h->stream, comp, nv, d_flag + 1);
    uint32_t n_comp = 0;
    MXG_HIP(h, hipMemcpyAsync(&n_comp, d_flag + 1, 4, hipMemcpyDeviceToHost, h->stream));  // read at the next sync

    // branch filtering, all components at once: thresholds n, n+1, ... while some component still has a branch node
    for (double t = (double)n_min; t <= wsum; t += 1.0) {
        MXG_HIP(h, hipMemsetAsync(deg, 0, (size_t)nv * 4, h->stream));
        MXG_HIP(h, hipMemsetAsync(B[NONLIN].p, 0, nv, h->stream));
        MXG_HIP(h, hipMemsetAsync(d_flag, 0, 4, h->stream));
        if (ne) hipLaunchKernelGGL(kp_degree, ge, b, 0, h->stream, eu, ev, alive, ne, deg);
        hipLaunchKernelGGL(kp_mark_branch, gv, b, 0, h->stream, deg, comp, nv, B[NONLIN].as<uint8_t>(), d_flag);
        uint32_t any = 0;
        MXG_HIP(h, hipMemcpyAsync(&any, d_flag, 4, hipMemcpyDeviceToHost, h->stream));
        MXG_HIP(h, hipStreamSynchronize(h->stream));
        if (!any) break;
        hipLaunchKernelGGL(kp_branch_filter, ge, b, 0, h->stream, eu, ev, ew, ne, deg, comp, B[NONLIN].as<uint8_t>(), t, alive);
    }
    // final degrees and sub-components
    MXG_HIP(h, hipMemsetAsync(deg, 0, (size_t)nv * 4, h->stream));
    if (ne) hipLaunchKernelGGL(kp_degree, ge, b, 0, h->stream, eu, ev, alive, ne, deg);
    if ((rc = components(h, eu, ev, alive, ne, nv, sub, d_flag)) != MXG_OK) return rc;

    for (int i : {CNTV, CNTE, CNTD1, MAXDEG, FILL, SIZE, FIRST, RANK}) MXG_HIP(h, B[i].ensure((size_t)nv * 4 + 16));
    MXG_HIP(h, B[ISSRC].ensure(nv + 16));
    MXG_HIP(h, B[NB].ensure((size_t)nv * 8));
    MXG_HIP(h, B[NBE].ensure((size_t)nv * 8));
    for (int i : {KEY0, KEY1, KEY2}) MXG_HIP(h, B[i].ensure((size_t)nv * 8));
    for (int i : {CNTV, CNTE, CNTD1, MAXDEG, FILL, SIZE}) MXG_HIP(h, hipMemsetAsync(B[i].p, 0, (size_t)nv * 4, h->stream));
    MXG_HIP(h, hipMemsetAsync(B[ISSRC].p, 0, nv, h->stream));
    MXG_HIP(h, hipMemsetAsync(B[NB].p, 0xFF, (size_t)nv * 8, h->stream));
    MXG_HIP(h, hipMemsetAsync(B[NBE].p, 0xFF, (size_t)nv * 8, h->stream));
    MXG_HIP(h, hipMemsetAsync(B[KEY0].p, 0xFF, (size_t)nv * 8, h->stream));  // cycle min key
    MXG_HIP(h, hipMemsetAsync(B[KEY1].p, 0xFF, (size_t)nv * 8, h->stream));  // source key (min)
    MXG_HIP(h, hipMemsetAsync(B[KEY2].p, 0, (size_t)nv * 8, h->stream));     // target key (max)
    SubStats st{B[CNTV].as<uint32_t>(), B[CNTE].as<uint32_t>(), B[CNTD1].as<uint32_t>(), B[MAXDEG].as<uint32_t>()};
    uint32_t *nb = B[NB].as<uint32_t>(), *nbe = B[NBE].as<uint32_t>();
    hipLaunchKernelGGL(kp_vertex_stats, gv, b, 0, h->stream, deg, sub, nv, st);
    if (ne) hipLaunchKernelGGL(kp_edge_stats, ge, b, 0, h->stream, eu, ev, alive, ne, sub, st.cnt_e, B[FILL].as<uint32_t>(), nb, nbe);
    hipLaunchKernelGGL(kp_cycle_min, gv, b, 0, h->stream, sub, nv, st, pos_first, B[KEY0].as<unsigned long long>());
    hipLaunchKernelGGL(kp_cycle_break, gv, b, 0, h->stream, nv, sub, st, pos_first, B[KEY0].as<unsigned long long>(), nb, nbe,
                       alive, deg);
    hipLaunchKernelGGL(kp_endpoints, gv, b, 0, h->stream, deg, sub, nv, st, pos_last, B[KEY1].as<unsigned long long>(),
                       B[KEY2].as<unsigned long long>());
    MXG_HIP(h, hipGetLastError());

    // arcs + pointer jumping
    const uint32_t n_arcs = 2 * ne;
    const size_t na1 = std::max<uint32_t>(n_arcs, 1);
    for (int i : {SUCC0, CNT0, END0, SUCC1, CNT1, END1}) MXG_HIP(h, B[i].ensure(na1 * 4));
    const dim3 ga((uint32_t)((na1 + 255) / 256));
    int cur = 0;
    if (n_arcs) {
        hipLaunchKernelGGL(kp_arc_init, ga, b, 0, h->stream, eu, ev, alive, ne, sub, st, nb, nbe, B[SUCC0].as<uint32_t>(),
                           B[CNT0].as<uint32_t>(), B[END0].as<uint32_t>());
        int rounds = 1;
        while ((1u << rounds) < nv) ++rounds;
        for (int r = 0; r < rounds; ++r) {
            const int s0 = cur ? SUCC1 : SUCC0, c0 = cur ? CNT1 : CNT0, e0 = cur ? END1 : END0;
            const int s1 = cur ? SUCC0 : SUCC1, c1 = cur ? CNT0 : CNT1, e1 = cur ? END0 : END1;
            hipLaunchKernelGGL(kp_arc_jump, ga, b, 0, h->stream, B[s0].as<uint32_t>(), B[c0].as<uint32_t>(), B[e0].as<uint32_t>(),
                               B[s1].as<uint32_t>(), B[c1].as<uint32_t>(), B[e1].as<uint32_t>(), n_arcs);
            cur ^= 1;
        }
    }
    MXG_HIP(h, hipGetLastError());
    // accepted paths: sizes flagged at the source vertex, ordered by source vertex index
    hipLaunchKernelGGL(kp_path_sizes, gv, b, 0, h->stream, nv, sub, st, B[KEY1].as<unsigned long long>(),
                       B[KEY2].as<unsigned long long>(), B[SIZE].as<uint32_t>(), B[ISSRC].as<uint8_t>());
    MXG_HIP(h, B[TOTAL].ensure(64));
    uint64_t *d_tot = B[TOTAL].as<uint64_t>();
    if ((rc = exclusive_scan_u32(h, B[SIZE].as<uint32_t>(), nv, B[BSUM], B[FIRST].as<uint32_t>(), d_tot)) != MXG_OK) return rc;
    // path index = exclusive scan of is_src (as u32): reuse SIZE? keep it simple: widen is_src into FILL
    MXG_HIP(h, hipMemsetAsync(B[FILL].p, 0, (size_t)nv * 4, h->stream));
    hipLaunchKernelGGL(kp_widen, gv, b, 0, h->stream, B[ISSRC].as<uint8_t>(), nv, B[FILL].as<uint32_t>());
    if ((rc = exclusive_scan_u32(h, B[FILL].as<uint32_t>(), nv, B[BSUM], B[RANK].as<uint32_t>(), d_tot + 1)) != MXG_OK) return rc;
    uint64_t tot[2] = {0, 0};
    MXG_HIP(h, hipMemcpyAsync(tot, d_tot, 16, hipMemcpyDeviceToHost, h->stream));
    MXG_HIP(h, hipStreamSynchronize(h->stream));
    const uint64_t n_pv = tot[0], n_paths = tot[1];
    MXG_HIP(h, B[PV].ensure(std::max<uint64_t>(n_pv * 4, 16)));
    MXG_HIP(h, B[PF].ensure((n_paths + 1) * 8));
    MXG_HIP(h, B[PC].ensure(std::max<uint64_t>(n_paths * 4, 16)));
    if (n_paths) {
        WriteParams wp;
        wp.eu = eu; wp.ev = ev; wp.alive = alive; wp.ne = ne; wp.sub = sub;
        wp.src_key = B[KEY1].as<unsigned long long>(); wp.tgt_key = B[KEY2].as<unsigned long long>();
        wp.cnt = B[cur ? CNT1 : CNT0].as<uint32_t>(); wp.endv = B[cur ? END1 : END0].as<uint32_t>();
        wp.cnt_v = st.cnt_v; wp.is_src = B[ISSRC].as<uint8_t>();
        wp.first = B[FIRST].as<uint32_t>(); wp.rank = B[RANK].as<uint32_t>();
        wp.path_vertex = B[PV].as<uint32_t>();
        if (n_arcs) hipLaunchKernelGGL(kp_write_arcs, ga, b, 0, h->stream, wp);
        hipLaunchKernelGGL(kp_write_sources, gv, b, 0, h->stream, nv, wp.is_src, wp.first, wp.rank, comp, B[SIZE].as<uint32_t>(),
                           wp.path_vertex, B[PF].as<uint64_t>(), B[PC].as<uint32_t>());
        MXG_HIP(h, hipGetLastError());
    }
    P.vertex.resize(n_pv);
    P.first.resize(n_paths + 1);
    P.component.resize(n_paths);
    if (n_pv) MXG_HIP(h, hipMemcpyAsync(P.vertex.data(), B[PV].p, n_pv * 4, hipMemcpyDeviceToHost, h->stream));
    if (n_paths) {
        MXG_HIP(h, hipMemcpyAsync(P.first.data(), B[PF].p, n_paths * 8, hipMemcpyDeviceToHost, h->stream));
        MXG_HIP(h, hipMemcpyAsync(P.component.data(), B[PC].p, n_paths * 4, hipMemcpyDeviceToHost, h->stream));
    }
    MXG_HIP(h, hipStreamSynchronize(h->stream));
    P.first[n_paths] = n_pv;
    P.n_components = n_comp;
    P.valid = true;
    return MXG_OK;
}

// ------------------------------------------------------------------------------------------------------
// row f4: what the scaffolder derives from the paths for one assembly
//   find_mx_min_max   reference bin/ntjoin_assemble.py:688-702   per contig: min / max position over its graph vertices
//   format_path       reference bin/ntjoin_assemble.py:175-218   runs of path vertices on the same contig ("segments")
//                     with what determine_orientation (:30-50) and calc_start/end_coord (:52-65) need of each run:
//                     number of vertices, min / max position, number of increasing / decreasing consecutive pairs
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kx_extremes(const uint32_t *__restrict__ vrec, const uint32_t *__restrict__ vpos,
                                                   uint32_t nv, uint32_t *mn, uint32_t *mx)
{
    uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= nv) return;
    atomicMin(&mn[vrec[v]], vpos[v]);
    atomicMax(&mx[vrec[v]], vpos[v]);
}

__global__ __launch_bounds__(256) void ks_mark_starts(const uint64_t *__restrict__ path_first, uint32_t n_paths, uint32_t *flag)
{
    uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p < n_paths) flag[path_first[p]] = 1;
}

// flag[i] = 1 where a segment starts: a path starts there, or the contig differs from the previous path vertex's
__global__ __launch_bounds__(256) void ks_flags(const uint32_t *__restrict__ pv, uint32_t n, const uint32_t *__restrict__ vrec,
                                                uint32_t *flag)
{
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (i == 0 || vrec[pv[i]] != vrec[pv[i - 1]]) flag[i] = 1;
}

__global__ __launch_bounds__(256) void ks_heads(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ excl, uint32_t n,
                                                const uint32_t *__restrict__ pv, const uint32_t *__restrict__ vrec,
                                                const uint64_t *__restrict__ path_first, uint32_t n_paths,
                                                uint32_t *seg_first, uint32_t *seg_rec, uint32_t *seg_path)
{
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const uint32_t s = excl[i];
    seg_first[s] = i;
    seg_rec[s] = vrec[pv[i]];
    uint32_t lo = 0, hi = n_paths;  // last path with path_first <= i
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (path_first[mid] <= i) lo = mid; else hi = mid;
    }
    seg_path[s] = lo;
}

// one wave per segment: {n, min pos, max pos, increasing pairs, decreasing pairs}
__global__ __launch_bounds__(256) void ks_stats(const uint32_t *__restrict__ seg_first, uint32_t n_seg, uint32_t n,
                                                const uint32_t *__restrict__ pv, const uint32_t *__restrict__ vpos,
                                                uint32_t *stat)
{
    const uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (s >= n_seg) return;
    const uint32_t lo = seg_first[s], hi = s + 1 < n_seg ? seg_first[s + 1] : n;
    uint32_t mn = 0xFFFFFFFFu, mx = 0, inc = 0, dec = 0;
    for (uint32_t i = lo + lane; i < hi; i += 64) {
        const uint32_t p = vpos[pv[i]];
        mn = min(mn, p);
        mx = max(mx, p);
        if (i + 1 < hi) {
            const uint32_t q = vpos[pv[i + 1]];
            inc += p < q;
            dec += p > q;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
        inc += (uint32_t)__shfl_xor((int)inc, o, 64);
        dec += (uint32_t)__shfl_xor((int)dec, o, 64);
    }
    if (lane == 0) {
        uint32_t *o = stat + (size_t)s * 5;
        o[0] = hi - lo; o[1] = mn; o[2] = mx; o[3] = inc; o[4] = dec;
    }
}

int mx_extremes(mxg_handle *h, uint32_t a)
{
    Graph &g = h->graph;
    if (!g.valid) return set_err(h, MXG_EINVAL, "mxg_mx_extremes: call mxg_build_graph first");
    if (a >= g.n_asm) return set_err(h, MXG_EINVAL, "assembly index %u out of range", a);
    MXG_HIP(h, hipSetDevice(h->device));
    const size_t nr = h->asms[a]->recs.size();
    Segments &S = h->segs;
    S.ext_min.assign(nr, 0xFFFFFFFFu);
    S.ext_max.assign(nr, 0u);
    if (nr == 0 || g.nv == 0) return MXG_OK;
    DevBuf *B = h->pbuf;
    MXG_HIP(h, B[XT_MIN].ensure(nr * 4));
    MXG_HIP(h, B[XT_MAX].ensure(nr * 4));
    MXG_HIP(h, hipMemsetAsync(B[XT_MIN].p, 0xFF, nr * 4, h->stream));
    MXG_HIP(h, hipMemsetAsync(B[XT_MAX].p, 0, nr * 4, h->stream));
    const uint32_t nv = (uint32_t)g.nv;
    hipLaunchKernelGGL(kx_extremes, dim3((nv + 255) / 256), dim3(256), 0, h->stream,
                       h->g_vrec.as<uint32_t>() + (size_t)a * g.nv_stride, h->g_vpos.as<uint32_t>() + (size_t)a * g.nv_stride, nv,
                       B[XT_MIN].as<uint32_t>(), B[XT_MAX].as<uint32_t>());
    MXG_HIP(h, hipGetLastError());
    MXG_HIP(h, hipMemcpyAsync(S.ext_min.data(), B[XT_MIN].p, nr * 4, hipMemcpyDeviceToHost, h->stream));
    MXG_HIP(h, hipMemcpyAsync(S.ext_max.data(), B[XT_MAX].p, nr * 4, hipMemcpyDeviceToHost, h->stream));
    MXG_HIP(h, hipStreamSynchronize(h->stream));
    return MXG_OK;
}

int path_segments(mxg_handle *h, uint32_t a)
{
    Graph &g = h->graph;
    const Paths &P = h->paths;
    if (!g.valid || !P.valid) return set_err(h, MXG_EINVAL, "mxg_path_segments: call mxg_find_paths first");
    if (a >= g.n_asm) return set_err(h, MXG_EINVAL, "assembly index %u out of range", a);
    MXG_HIP(h, hipSetDevice(h->device));
    Segments &S = h->segs;
    S.path.clear(); S.record.clear(); S.first.clear(); S.stat.clear();
    const uint32_t n = (uint32_t)P.vertex.size(), n_paths = (uint32_t)P.component.size();
    if (n == 0) return MXG_OK;
    DevBuf *B = h->pbuf;
    const uint32_t *pv = B[PV].as<uint32_t>();
    const uint64_t *pf = B[PF].as<uint64_t>();
    const uint32_t *vrec = h->g_vrec.as<uint32_t>() + (size_t)a * g.nv_stride;
    const uint32_t *vpos = h->g_vpos.as<uint32_t>() + (size_t)a * g.nv_stride;
    MXG_HIP(h, B[SG_FLAG].ensure((size_t)n * 4 + 16));
    MXG_HIP(h, B[SG_EXCL].ensure((size_t)n * 4 + 16));
    MXG_HIP(h, B[TOTAL].ensure(64));
    MXG_HIP(h, hipMemsetAsync(B[SG_FLAG].p, 0, (size_t)n * 4, h->stream));
    const dim3 gn((n + 255) / 256), b(256);
    hipLaunchKernelGGL(ks_mark_starts, dim3((n_paths + 255) / 256), b, 0, h->stream, pf, n_paths, B[SG_FLAG].as<uint32_t>());
    hipLaunchKernelGGL(ks_flags, gn, b, 0, h->stream, pv, n, vrec, B[SG_FLAG].as<uint32_t>());
    int rc = exclusive_scan_u32(h, B[SG_FLAG].as<uint32_t>(), n, B[BSUM], B[SG_EXCL].as<uint32_t>(), B[TOTAL].as<uint64_t>());
    if (rc != MXG_OK) return rc;
    uint64_t n_seg64 = 0;
    MXG_HIP(h, hipMemcpyAsync(&n_seg64, B[TOTAL].p, 8, hipMemcpyDeviceToHost, h->stream));
    MXG_HIP(h, hipStreamSynchronize(h->stream));
    const uint32_t n_seg = (uint32_t)n_seg64;
    MXG_HIP(h, B[SG_FIRST].ensure((size_t)n_seg * 4 + 16));
    MXG_HIP(h, B[SG_REC].ensure((size_t)n_seg * 4 + 16));
    MXG_HIP(h, B[SG_PATH].ensure((size_t)n_seg * 4 + 16));
    MXG_HIP(h, B[SG_STAT].ensure((size_t)n_seg * 20 + 16));
    hipLaunchKernelGGL(ks_heads, gn, b, 0, h->stream, B[SG_FLAG].as<uint32_t>(), B[SG_EXCL].as<uint32_t>(), n, pv, vrec, pf,
                       n_paths, B[SG_FIRST].as<uint32_t>(), B[SG_REC].as<uint32_t>(), B[SG_PATH].as<uint32_t>());
    hipLaunchKernelGGL(ks_stats, dim3((n_seg + 3) / 4), b, 0, h->stream, B[SG_FIRST].as<uint32_t>(), n_seg, n, pv, vpos,
                       B[SG_STAT].as<uint32_t>());
    MXG_HIP(h, hipGetLastError());
    S.path.resize(n_seg); S.record.resize(n_seg); S.first.resize(n_seg); S.stat.resize((size_t)n_seg * 5);
    MXG_HIP(h, hipMemcpyAsync(S.path.data(), B[SG_PATH].p, (size_t)n_seg * 4, hipMemcpyDeviceToHost, h->stream));
    MXG_HIP(h, hipMemcpyAsync(S.record.data(), B[SG_REC].p, (size_t)n_seg * 4, hipMemcpyDeviceToHost, h->stream));
    MXG_HIP(h, hipMemcpyAsync(S.first.data(), B[SG_FIRST].p, (size_t)n_seg * 4, hipMemcpyDeviceToHost, h->stream));
    MXG_HIP(h, hipMemcpyAsync(S.stat.data(), B[SG_STAT].p, (size_t)n_seg * 20, hipMemcpyDeviceToHost, h->stream));
    MXG_HIP(h, hipStreamSynchronize(h->stream));
    return MXG_OK;
}

}  // namespace mxg
