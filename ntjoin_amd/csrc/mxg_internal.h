// mxg_internal.h -- internal structures of libntjoin_mx.so (not part of the C-ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "ntjoin_mx.h"

namespace mxg {

// ---- device-side tables ------------------------------------------------------------------------
// One maximal run of valid (ACGTU) bases that holds at least one k-mer, inside an ELIGIBLE contig
// (a record with >= w valid k-mers; shorter records have no window and yield no minimizer,
// SURVEY.md A.3).  Windows are taken over the concatenation of a contig's runs (valid k-mers only).
struct Run {
    uint64_t base_off;  // global base index (into the packed array) of the run's first base
    uint32_t n_kmers;   // run_len - k + 1
    uint32_t contig;    // eligible-contig index
    uint32_t kidx0;     // contig-local valid-k-mer index of the run's first k-mer
    uint32_t pos0;      // contig-local base position of the run's first base
};

// a run as k_bs_select (sketch_bs.hip) reads it: one aligned 32-byte entry where Run + the strip prefix + the contig's k-mer
// count are three dependent table reads (strip0 belongs to the assembly's sparse strip length)
struct alignas(32) RunX {
    uint64_t base_off;
    uint32_t n_kmers, contig, kidx0;
    uint32_t strip0S;   // (strips of the sparse strip table in front of the run) x (k-mers per strip), mod 2^32
    uint32_t nk;        // valid k-mers of the run's contig
    uint32_t pad;
};

struct HashTab {  // ntHash step table, entry (out*4+in), out==4: warm-up step with no outgoing base
    // x,y = low/high word of  srol^k(SEED[out]) ^ SEED[in]            (forward update term)
    // z,w = low/high word of  SEED[comp(out)] ^ srol^k(SEED[comp(in)]) (reverse update term, before sror)
    uint4 e[20];
};

// ---- host-side helpers ---------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    // grow-only; contents are NOT preserved
    hipError_t ensure(size_t need)
    {
        if (need <= bytes) return hipSuccess;
        static const bool dbg = getenv("MXG_DEBUG_ALLOC") != nullptr;  // (diagnostics: what a handle's first step allocates)
        const auto t0 = std::chrono::steady_clock::now();
        const size_t had = bytes;
        release();
        size_t want = need + need / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return e;
        }
        bytes = want;
        if (dbg)
            fprintf(stderr, "[mxg] alloc %.1f MB (had %.1f MB): %.3f ms\n", want / 1048576.0, had / 1048576.0,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        return hipSuccess;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct Record {
    std::string id;
    uint64_t len = 0;       // bases in the record
    uint64_t base_off = 0;  // global base offset in the packed array (multiple of 16)
    uint64_t text_off = 0;  // offset of the record's text in Assembly::text (if kept)
};

struct Assembly {
    std::string name;
    double weight = 0.0;
    std::vector<Record> recs;
    uint64_t total_bases = 0;
    uint64_t shard_lo = 0, shard_hi = ~0ull;  // records this handle holds bases for (fasta_shard); others: id + length only
    // bases
    bool has_bases = false;
    std::string text;                // concatenated record text as read (kept unless MXG_FLAG_DROP_SEQ)
    bool has_text = false;
    std::vector<uint32_t> h_packed;  // host 2-bit packing (dropped after upload)
    // device ingest (ingest.hip): the raw FASTA text in HBM + the tile index that maps a base to its byte
    bool text_on_device = false;
    uint64_t text_bytes = 0;
    DevBuf d_text, d_ing_items, d_ing_cnt, d_ing_sub, d_ing_pbase, d_ing_item0;
    std::vector<uint64_t> ing_item0;  // [n_records + 1] first tile of every record
    DevBuf d_packed_own;
    const uint32_t *d_packed = nullptr;  // owned (above) or borrowed
    uint64_t packed_words = 0;
    // eligibility tables (host), uploaded by the sketch driver
    std::vector<Run> runs;
    std::vector<uint32_t> ctg_rec;   // eligible contig -> record index
    std::vector<uint32_t> ctg_nk;    // eligible contig -> valid k-mer count
    std::vector<uint32_t> ctg_run0;  // eligible contig -> first run (size n+1)
    std::vector<uint8_t> ctg_drop;   // split load only: the contig is a piece whose first minimizer belongs to the shard before
    bool any_drop = false;
    bool split_first_cont = false;   // split load: this handle's first record continues a record begun on the shard before
    uint64_t total_kmers = 0;        // over eligible contigs
    // device copies of the tables above + strip / k-mer prefix tables (built once, by the first sketch)
    bool tables_ready = false;
    uint32_t S_sparse = 256;  // strip length chosen for the sparse hash kernel
    uint32_t cand_hint = 0;   // candidates of the last sparse run (k_resolve: which blocks may load before the count arrives)
    bool full_grid_once = false;       // the next enqueue sizes every grid for the candidate capacity, not the estimate
    double gap_rate_hint = 0;          // candidate-free stretches per k-mer met by earlier sketches (sizes the batches)
    std::vector<uint32_t> cand_hints;  // ... per batch of the pipelined multi-batch driver
    std::vector<uint32_t> strip0_dense, strip0_sparse;  // [n_runs+1] exclusive prefix of strips per run
    std::vector<uint64_t> g0;                           // [n_runs+1] exclusive prefix of k-mers per run
    DevBuf d_runs, d_strip0_dense, d_strip0_sparse, d_g0, d_ctg_nk, d_ctg_rec, d_ctg_run0, d_ctg_drop;
    DevBuf d_ctg_info;  // uint4 per contig {first run, runs, first run's pos0, record}: what k_emit needs of a contig in one read
    DevBuf d_strip_run;  // strip of the sparse strip table -> its run (k_strip_runs)
    DevBuf d_runx;       // RunX per run
    // k = 32 route (sketch_bs.hip): the bases transposed for the bit-sliced ring filter, its result, chunk -> first run
    bool bs_ready = false, bs_impossible = false;
    uint32_t bs_chunks = 0;
    DevBuf d_bs_tail, d_bs_out;  // k = 32 route: padded copies of the first / last chunk's words, the filter's bitmap
    bool sel_again = false;  // the second attempt of an assembly's batches may take k_bs_select again (only the stretch budget failed)
    uint32_t sel_H = 0, sel_H_S = 0, sel_H_w = 0;  // k_bs_select: halo strips (0: the route does not take this run table) for (S, w)
    // sketch (device, ordered by (record,pos)) + lazily filled host mirror
    bool has_sketch = false;
    uint64_t n_mx = 0;
    uint64_t n_mx_seen = 0;  // the largest sketch an earlier mxg_sketch* of this assembly ended with (sizes the fused graph stage)
    DevBuf d_hash, d_pos, d_rec, d_fwd, d_rec_base;
    bool fwd_valid = false;  // d_fwd filled (lazily: k_strand)
    bool foreign_sketch = false;  // sketch holds minimizers of records this handle has no bases for (gathered / imported)
    bool host_valid = false;
    std::vector<uint64_t> h_hash;
    std::vector<uint32_t> h_pos, h_rec;
    std::vector<uint8_t> h_fwd;
    std::vector<uint64_t> rec_first;
    // graph stage
    DevBuf d_flags, d_slot, d_shared, d_ivid;
    DevBuf d_perm, d_fg, d_frec, d_dgtmp;  // distributed graph stage (dgraph.hip), sender side
    std::vector<uint8_t> h_flags;
    bool flags_valid = false;   // device flags computed
    bool flags_on_host = false; // h_flags mirrors d_flags
};

struct Graph {
    bool valid = false;       // device results computed
    bool host_valid = false;  // host mirrors below filled (lazily, by graph_to_host)
    uint32_t n_asm = 0;
    uint64_t nv = 0, ne = 0, nv_stride = 0;
    std::vector<uint64_t> vhash;
    std::vector<uint32_t> vpos, vrec;  // [a*nv+v]
    std::vector<uint32_t> eu, ev, esup;
    std::vector<double> ew;
};

struct Paths {  // result of mxg_find_paths (host copies)
    bool valid = false;
    uint64_t n_components = 0;        // components of the globally filtered graph (singletons included)
    std::vector<uint32_t> vertex;     // concatenated paths, vertex indices of the graph, source -> target
    std::vector<uint64_t> first;      // [n_paths+1]
    std::vector<uint32_t> component;  // [n_paths] component (root vertex index) of the globally filtered graph
};

struct Segments {  // results of mxg_path_segments / mxg_mx_extremes (host copies)
    std::vector<uint32_t> path, record, first;  // per segment
    std::vector<uint32_t> stat;                 // per segment: n, min pos, max pos, increasing pairs, decreasing pairs
    std::vector<uint32_t> ext_min, ext_max;     // per record of the assembly last asked for
};

struct Timers {
    double ms_hash = 0, ms_resolve = 0, ms_graph = 0;
    // MXG_FLAG_TIMING_FINE: one span per kernel (ms_resolve stays the sum of the three behind the hash kernel)
    double ms_reorder = 0, ms_resolve_k = 0, ms_emit = 0, ms_join = 0, ms_vertices = 0, ms_edges = 0;
    uint64_t launches_hash = 0, hash_bases = 0;
};

// MXG_FLAG_TIMING: event pairs recorded around the hash kernel / the rest of a batch.  Events come from a pool and are
// read back lazily (flush_timers: mxg_get_stats, mxg_reset_timers), so timing adds two hipEventRecord per pair to the
// hot path and nothing else.
struct TimedSpan {
    hipEvent_t a, b;
    uint64_t bases;
    bool is_hash;
    int kind;  // 0 hash, 1 everything behind it (coarse timing), 2 reorder, 3 resolve, 4 emit (fine timing)
};

}  // namespace mxg

struct mxg_handle {
    mxg_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // second stream of the pipelined multi-assembly sketch (always library-owned)
    uint64_t timing_batches = 0;    // batches enqueued with MXG_FLAG_TIMING (MXG_TIMING_SAMPLE picks one in n)
    hipStream_t stream_x[2] = {nullptr, nullptr};  // third / fourth stream: batches of multi-batch assemblies (created on first use)
    bool own_stream = false;
    std::string err;
    std::vector<mxg::Assembly *> asms;
    mxg::Graph graph;
    std::vector<hipEvent_t> ev_pool;       // every event ever created for timing; [0, ev_used) are in flight
    size_t ev_used = 0;
    std::vector<mxg::TimedSpan> ev_spans;  // not yet folded into tm
    mxg::DevBuf d_nmx;              // fused sketch+graph call: sketch sizes on the device
    mxg::DevBuf d_chain;            // pipelined batches: where the next batch of an assembly starts in its sketch (u64 per batch)
    std::vector<hipEvent_t> ev_sync;  // ... and the events by which a batch waits for its predecessor on the other stream
    hipEvent_t ev_join = nullptr;   // ... and the event that joins the second stream into the first
    hipEvent_t ev_part[MXG_MAX_ASSEMBLIES] = {};  // mxg_sketch_pack_parts: assembly a's part is packed
    mxg::DevBuf dbg_buf;            // (profiling: MXG_BSR_DBG)
    uint32_t dbg_blocks = 0;
    std::vector<hipEvent_t> ev_bs;  // k = 32 route: "the assembly's filter has run" (batches on other streams wait for it)
    mxg::DevBuf dg_cnt, dg_cursor;  // dgraph.hip: per-destination counts / cursors
    mxg::DevBuf dg_ghost;           // ... {record, global vertex id} of the shared minimizer before this rank's first, per assembly
    bool dg_ghost_on = false;
    mxg::Paths paths;
    mxg::DevBuf pbuf[48];  // scratch of paths.hip
    mxg::Segments segs;
    mxg::Timers tm;
    mxg::HashTab tab{};
    mxg::DevBuf d_init_tab;  // byte table of the direct hash formula (256 x 16 B), built by the first sketch
    uint64_t stat_candidates = 0, stat_dense_kmers = 0, stat_unique = 0;
    uint64_t stat_bs_bases = 0;  // bases the bit-sliced filter (k = 32 route) has covered
    uint64_t stat_sel_slices = 0;  // slices enqueued through k_bs_select
    uint64_t stat_slice_stretches = 0;  // candidate-free stretches handed to k_sel_stretch
    hipEvent_t ev_sel_done[4] = {nullptr, nullptr, nullptr, nullptr};  // recorded behind every slice kernel, per stream slot
    uint64_t stat_graph_join = 0; // mxg_stats::graph_join
    uint32_t pj_cap1_P1 = 0;     // two-level join: coarse partitions and the records one of them must hold, as an earlier call's
    uint64_t pj_cap1_need = 0;   // cursors reported them (a key of large multiplicity skews the partitions)
    uint64_t pj_learnt_sig = 0;  // the sketches (count and sizes) the three fields around this one were learnt on
    bool pj_overflowed = false;  // graph stage: the partitioned join overflowed once (build_graph then starts with the global table)
    bool dg_pj_off = false;      // owner of a partitioned graph stage: the LDS join failed once over the slots (global table from then on)
    uint64_t stat_retries = 0;   // assemblies enqueued a second time (their batches did not all end the common way)
    uint64_t stat_deferred = 0;  // candidate-free stretches the device route handed to the host
    uint64_t stat_batches_redone = 0, stat_sync_assemblies = 0;  // batches that did not end the common way / assemblies redone whole
    // scratch reused across calls
    mxg::DevBuf scratch[4][40];  // indexed by mxg::Scratch (sketch.hip): one set per in-flight sketch driver (= stream)
    std::vector<mxg::Assembly *> pend_list;  // mxg_sketch_pack in flight: assemblies and how each was enqueued
    std::vector<int> pend_state;
    std::vector<unsigned char> pend_dev;     // ... and whether its stretches went the device route (sketch_finish accepts those)
    mxg::DevBuf g_part;     // partitioned join (graph.hip): partition offsets of every bucketing block
    mxg::DevBuf g_recs1;    // two-level join: the coarse partitions' records
    mxg::DevBuf g_keys, g_cnt, g_vid, g_ctl, g_vhash, g_vpos, g_vrec, g_fv, g_frec, g_nxt, g_prv, g_eflag,
        g_ebs, g_eu, g_ev, g_esup, g_ew;  // indexed by mxg::Scratch (sketch.hip) / graph.hip's own enum
    uint64_t arena_cap_hint = 0;
    // environment knobs (README: tuning / test / profiling switches), each parsed ONCE per handle, at its first use: what a
    // handle ran with is what mxg_knobs reports, whatever the environment says later
    struct Knob {
        bool set = false;
        uint64_t value = 0;
        std::string raw;
    };
    std::map<std::string, Knob> knobs;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_g[2] = {nullptr, nullptr};  // MXG_FLAG_TIMING_FINE: after the join, after vertices + adjacency
    uint64_t *pinned_dg = nullptr;    // dgraph.hip: pinned copy of the per-destination counters
    uint64_t *pinned_gctl = nullptr;  // pinned host copy of the graph stage's control block
    uint32_t *pinned_ctrl = nullptr;  // pinned host copies of per-assembly control blocks (pipelined sketch)
    std::vector<char> dot_part[2];    // a formatted part of the .mx.dot (mxg_dot_part_format -> mxg_dot_part_write)
    void *pinned_defer = nullptr;     // per control block: the stretches its batch left to the host (sketch.hip defer_stretch)
    // 128 MB of pinned host memory shared by the file routes (FASTA text on its way in: 4 x 32 MB, TSV text on its way out: 2 x 64 MB)
    // and the TSV writer's device windows: allocated on first use, kept until mxg_destroy -- a pinned allocation of this size
    // costs 25-55 ms, and the one-process route (mxgraph) would pay it four times
    void *pin_pool = nullptr;
    // ... and pinning is what costs: the pool is ordinary memory that a (detached) thread of the handle registers with HIP in four pieces of
    // 32 MB (pin_pool_start), so the first file's upload starts when the first staging buffer is pinned (~7 ms), not the last (~29 ms)
    std::atomic<uint32_t> pin_ready{0};   // pieces registered so far
    std::atomic<int> pin_state{0};        // 0 no pool, 1 the thread is at work, 2 done, 3 failed (pin_err)
    hipError_t pin_err = hipSuccess;
    bool pin_registered = false;          // the pool is mmap'ed memory registered piece by piece (else: hipHostMalloc)
    std::vector<std::pair<void *, size_t>> kept_maps;  // MXG_FLAG_ONE_SHOT: file mappings left for mxg_destroy / the process's end
    mxg::DevBuf tsv_win[2];
};

namespace mxg {

int set_err(mxg_handle *h, int code, const char *fmt, ...);
// environment knob `name` as the handle first saw it (host_io.cpp): its value, or dflt when it is unset / empty
uint64_t knob_u64(const mxg_handle *h, const char *name, uint64_t dflt);
bool knob_set(const mxg_handle *h, const char *name);
const char *knob_raw(const mxg_handle *h, const char *name);  // the text it was set to, or nullptr
// a helper thread of the library points this at a string of its own: set_err then leaves the handle's message alone (host_io.cpp)
extern thread_local std::string *tl_err_sink;
constexpr size_t PIN_POOL_BYTES = 128ull << 20;
constexpr size_t PIN_PIECE_BYTES = 32ull << 20;
constexpr uint32_t PIN_PIECES = (uint32_t)(PIN_POOL_BYTES / PIN_PIECE_BYTES);
// host_io.cpp: starts the pool (no-op when there is one); waits until `pieces` pieces from the pool's start are pinned
hipError_t pin_pool_start(mxg_handle *h);
hipError_t pin_pool_wait(mxg_handle *h, uint32_t pieces);
void pin_pool_release(mxg_handle *h);
inline hipError_t pin_pool_get(mxg_handle *h, unsigned char **p)  // the whole pool
{
    hipError_t e = pin_pool_start(h);
    if (e == hipSuccess) e = pin_pool_wait(h, PIN_PIECES);
    if (e != hipSuccess && (e = pin_pool_start(h)) == hipSuccess) e = pin_pool_wait(h, PIN_PIECES);  // (once more, in one allocation)
    if (e != hipSuccess) return e;
    *p = static_cast<unsigned char *>(h->pin_pool);
    return hipSuccess;
}
// Wait for a stream by polling: hipStreamSynchronize parks the thread and its wake-up costs tens of microseconds, which
// is a tenth of a whole sketch+graph step at 2 x 100 Mbp.  The two syncs on the hot path (end of the sketch stage, end
// of the graph stage) poll for a bounded time, then fall back to the blocking wait.
inline hipError_t stream_wait(hipStream_t s)
{
    for (int spin = 0; spin < 200000; ++spin) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
    }
    return hipStreamSynchronize(s);
}

#define MXG_HIP(h, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return mxg::set_err((h), e_ == hipErrorOutOfMemory ? MXG_ENOMEM : MXG_EDEVICE,      \
                                "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                                __LINE__);                                                       \
    } while (0)

// host_io.cpp
int load_fasta(mxg_handle *h, Assembly *a, const char *path, uint32_t shard = 0, uint32_t n_shards = 1, bool split = false);
void shard_range(const uint64_t *lengths, uint64_t n, uint32_t shard, uint32_t n_shards, uint64_t *lo, uint64_t *hi);
int load_buffers(mxg_handle *h, Assembly *a, const uint8_t *ascii, const uint64_t *offsets,
                 const char *const *ids, uint64_t n_records);
int load_tsv(mxg_handle *h, Assembly *a, const char *path, std::vector<uint64_t> &hash,
             std::vector<uint32_t> &pos, std::vector<uint32_t> &rec);
void build_runs_from_lengths(mxg_handle *h, Assembly *a);  // N-free packed-device input
int write_tsv(mxg_handle *h, Assembly *a, const char *path, int with_pos, int with_strand, int with_seq);
// ingest.hip: FASTA -> packed bases + run table on the device (1: not for this route, use load_fasta); TSV text on the device
int load_fasta_device(mxg_handle *h, Assembly *a, const char *path, uint32_t n_threads);
int write_tsv_device(mxg_handle *h, Assembly *a, const char *path, int with_pos, int with_strand, int with_seq);
uint32_t host_threads(const mxg_handle *h);
// n_parts byte ranges written to fd at consecutive offsets from `off` on, by that many threads (host_io.cpp)
bool put_parallel(int fd, uint64_t off, const char *const *data, const size_t *len, uint32_t n_parts);
int write_dot(mxg_handle *h, const char *path);
int dot_part_format(mxg_handle *h, uint32_t part, uint32_t n_parts, uint64_t bytes[2]);
int dot_part_write(mxg_handle *h, const char *path, uint64_t v_off, uint64_t e_off, int first, int last);
void build_rec_first(Assembly *a);
std::string py_repr_str(const std::string &s);
std::string py_repr_float(double v);
void make_hash_tab(uint32_t k, HashTab *t);
void make_init_tab(uint32_t k, std::vector<uint4> &out);

// sketch.hip
int sketch_assembly(mxg_handle *h, Assembly *a);
struct XchgPackReq {  // mxg_sketch_pack: where the sketches go once they exist (see xchg_pack for the slot layout)
    void *d_slot;
    uint64_t head_bytes;
    const uint64_t *caps;
    // mxg_sketch_pack_parts: one buffer per assembly instead, [64 bytes: int64 count, int64 records | caps[a] hashes | caps[a]
    // positions | rcaps[a] first entries of the records], packed right behind that
    // assembly's own k_emit on the stream it ran on, with an event the caller's communication stream can wait for (ev_part)
    void *const *d_parts = nullptr;
    const uint64_t *rcaps = nullptr;  // ... and room for this many records' first entries behind the 12 bytes per entry
    // mxg_sketch_dg_pack_slots: instead, every assembly's minimizers into the item slots of the partitioned graph stage (dgraph.hip),
    // right behind that assembly's k_emit, with the same per-assembly event
    const struct DgPackReq *dg = nullptr;
};
struct DgPackReq {
    uint32_t world, n_asm;
    const uint32_t *cap;      // items per (destination, assembly) slot
    const uint32_t *rec_off;  // record index shift per assembly
    void *d_send;
};
constexpr uint64_t XCHG_PART_HEAD = 64;
int sketch_assemblies(mxg_handle *h, Assembly *const *list, size_t n, bool fuse_graph = false, const XchgPackReq *xp = nullptr);
int sketch_finish(mxg_handle *h);
int upload_packed(mxg_handle *h, Assembly *a);
int prewarm_assembly(mxg_handle *h, Assembly *a);  // tables, output arrays, the filter's bitmap: when the assembly is added
int sync_sketch_to_host(mxg_handle *h, Assembly *a);
int write_sketch_bin(mxg_handle *h, Assembly *a, const char *path);  // host_io.cpp
int load_sketch_bin(mxg_handle *h, Assembly *a, const char *path, std::vector<uint64_t> &hash, std::vector<uint32_t> &pos,
                    std::vector<uint32_t> &rec);
int ensure_strand(mxg_handle *h, Assembly *a);
int pack_sketch(mxg_handle *h, Assembly *a, void *d_buf, uint64_t nmax);
int unpack_gathered(mxg_handle *h, Assembly *a, const void *d_allbuf, uint32_t world, uint64_t nmax,
                    const uint64_t *counts, const uint64_t *rec_offsets, uint64_t stride_bytes = 0);
// graph.hip
enum { GRAPH_FULL = 0, GRAPH_DG_VERTICES = 1, GRAPH_DG_EDGES = 2, GRAPH_DG_EDGES_APPLIED = 3 /* nxt/prv already filled */ };
struct GraphBounds {  // fused sketch+graph call: per assembly an upper bound of its sketch and where the count will be
    uint64_t n_bound[MXG_MAX_ASSEMBLIES];
    const uint32_t *n_ptr[MXG_MAX_ASSEMBLIES];
};
int build_graph(mxg_handle *h, int mode = GRAPH_FULL, const void *d_msgs = nullptr, uint64_t n_msgs = 0,
                const GraphBounds *gb = nullptr);
int xchg_pack(mxg_handle *h, void *d_slot, uint64_t head_bytes, const uint64_t *caps);
int xchg_unpack_graph(mxg_handle *h, const void *d_all, uint32_t world, uint64_t slot_bytes, uint64_t head_bytes,
                      const uint64_t *caps, const uint64_t *rec_offsets, const void *const *d_all_parts = nullptr,
                      const uint64_t *rcaps = nullptr);
int graph_to_host(mxg_handle *h);
int find_paths(mxg_handle *h, int64_t n_min);  // paths.hip
// dgraph.hip
int dg_owner_counts(mxg_handle *h, uint32_t world, uint64_t *counts);
int dg_pack_items(mxg_handle *h, Assembly *a, uint32_t world, uint32_t rec_offset, const uint64_t *starts, void *d_send);
int dg_set_items(mxg_handle *h, Assembly *a, const void *d_items, uint32_t world, const uint64_t *sec_start,
                 const uint64_t *sec_count);
int dg_item_results(mxg_handle *h, Assembly *a, const void *d_gbase, uint32_t world, const uint64_t *sec_start,
                    const uint64_t *sec_count, void *d_out);
int dg_msg_counts(mxg_handle *h, uint32_t world, const void *d_ret, const void *d_bases, uint64_t *counts);
int dg_last_shared(mxg_handle *h, const void *d_ret, void *d_out);
int dg_set_ghosts(mxg_handle *h, const void *d_all, uint32_t world, uint32_t rank);
int dg_pack_msgs(mxg_handle *h, Assembly *a, uint32_t assembly, uint32_t world, const void *d_bases, const uint64_t *starts,
                 void *d_send);
int dg_pack_slots(mxg_handle *h, Assembly *a, uint32_t ai, uint32_t rec_offset, uint32_t world, uint32_t n_asm, const uint32_t *cap,
                  void *d_send);
int dg_pack_slots_dev(mxg_handle *h, Assembly *a, uint32_t ai, const DgPackReq &rq, hipStream_t st, uint32_t mode, const uint32_t *n_ptr,
                      const uint32_t *ctrl, uint64_t out_cap, uint32_t dev_gaps, uint32_t place4);
int dg_owner_slots(mxg_handle *h, uint32_t world, uint32_t n_asm, const uint32_t *cap, const void *d_recv, void *d_nv);
// the words behind the per-assembly counts of d_nmx: [MXG_MAX_ASSEMBLIES] = a slot overflowed, then (8-byte aligned) "the owner's LDS join failed"
inline uint64_t *dg_pj_fail_word(mxg_handle *h) { return reinterpret_cast<uint64_t *>(h->d_nmx.as<unsigned char>() + 4 * MXG_MAX_ASSEMBLIES + 8); }
int dg_slot_results(mxg_handle *h, uint32_t world, uint32_t n_asm, const uint32_t *cap, const void *d_recv, const void *d_gbase,
                    void *d_out);
int dg_pack_msg_slots(mxg_handle *h, uint32_t world, uint32_t M, const void *d_ret, const void *d_bases, void *d_send);
int dg_edges_slots(mxg_handle *h, const void *d_recv, uint32_t world, uint32_t M, uint64_t *n_vertices, uint64_t *n_edges,
                   uint32_t *overflow);
int path_segments(mxg_handle *h, uint32_t assembly);
int mx_extremes(mxg_handle *h, uint32_t assembly);
int flush_timers(mxg_handle *h);                // sketch.hip: fold the recorded event pairs into h->tm
int flags_to_host(mxg_handle *h, Assembly *a);

}  // namespace mxg
