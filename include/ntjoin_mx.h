/*
 * ntjoin_mx.h -- C-ABI of libntjoin_mx.so: the MI355X-native minimizer-sketch + minimizer-graph engine
 * that drops in for ntJoin's hot path (sketch -> uniqueness -> intersection -> adjacency edges).
 *
 * Plain C linkage, plain pointers and sizes; no C++/torch types cross this boundary.  Every function
 * returns 0 on success or a negative MXG_E* code; mxg_last_error(h) then holds a message.  The library
 * owns every buffer it returns (valid until the next call that recomputes it, or mxg_destroy);
 * a handle is not thread-safe, distinct handles are.  All device work runs on the handle's HIP stream
 * and each call blocks until its results are usable.  There is NO CPU fallback: without a usable HIP
 * device every compute entry point fails with MXG_EDEVICE.
 *
 * Reference interfaces replaced (file:line under the reference tree):
 *   mxg_add_assembly_fasta + mxg_sketch + mxg_write_tsv
 *        = `indexlr --seq --long --pos -k K -w W -t T X.fa > X.fa.kK.wW.tsv`          ntJoin:204-205
 *          (and its twin run_indexlr()                                                  bin/ntjoin_utils.py:195-202)
 *   mxg_add_assembly_tsv            = the parse half of read_minimizers()               bin/ntjoin_utils.py:167-185
 *   mxg_build_graph (uniqueness)    = the dup_mxs logic of read_minimizers()            bin/ntjoin_utils.py:182-193
 *   mxg_build_graph (intersection)  = filter_minimizers()                               bin/ntjoin_utils.py:152-165
 *   mxg_build_graph (edges/weights) = build_graph(), calc_total_weight()                bin/ntjoin_utils.py:83-141,54-56
 *   assembly order = refs (CLI order) then target, weights as double                    bin/ntjoin.py:178-186,
 *                                                                                       bin/ntjoin_assemble.py:788-807
 *   mxg_write_dot                   = Ntjoin.print_graph()                              bin/ntjoin.py:25-62
 */
#ifndef NTJOIN_MX_H
#define NTJOIN_MX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MXG_ABI_VERSION 2

/* error codes */
#define MXG_OK 0
#define MXG_EINVAL (-1)   /* bad argument / bad state                                   */
#define MXG_EIO (-2)      /* file could not be read / written / parsed                  */
#define MXG_ENOMEM (-3)   /* host or device allocation failed                           */
#define MXG_EDEVICE (-4)  /* no usable HIP device, or a HIP call / kernel failed        */
#define MXG_ELIMIT (-5)   /* an engine limit was exceeded (see mxg_last_error)          */

/* canonical-hash variant */
#define MXG_VARIANT_V2_SUM 0 /* min_hash = fwd+rev: current btllib; default                */
#define MXG_VARIANT_V1_MIN 1 /* min_hash = min(fwd,rev): what the reference's stale golden TSVs hold */

/* mxg_config.flags */
#define MXG_FLAG_DENSE_ONLY 0x1u  /* disable the sparse-candidate fast path (every k-mer is a candidate) */
#define MXG_FLAG_DROP_SEQ 0x2u    /* do not keep FASTA text on the host (mxg_write_tsv then decodes k-mers from the packed bases: upper-case) */
#define MXG_FLAG_TIMING 0x4u      /* bracket each kernel family with HIP events (read back through mxg_stats) */
#define MXG_FLAG_TIMING_FINE 0x8u /* ... one event pair per kernel (profiling; fills the per-kernel fields of mxg_stats) */
#define MXG_FLAG_ONE_SHOT 0x10u   /* the handle sketches once and writes its outputs once (the CLIs): mxg_write_outputs returns the bases, the
                                     text and the scratch buffers to the driver as soon as the TSVs are written, beside the .mx.dot writer;
                                     mxg_add_assembly_fasta leaves the file's mapping in place until mxg_destroy (or the process's end:
                                     taking 3 GB of page table apart costs ~25 ms per file) */

#define MXG_MAX_ASSEMBLIES 32

typedef struct mxg_handle mxg_handle;

typedef struct mxg_config {
    uint32_t struct_size;  /* = sizeof(mxg_config); lets the struct grow compatibly          */
    uint32_t k;            /* k-mer size, 1..1024  (ntJoin: k=32, ntJoin:36)                 */
    uint32_t w;            /* window size in k-mers, >= 1 (ntJoin: w=1000, ntJoin:33)        */
    uint32_t variant;      /* MXG_VARIANT_*                                                   */
    int32_t device;        /* HIP device ordinal, -1 = current device                         */
    uint32_t flags;        /* MXG_FLAG_*                                                      */
    void *stream;          /* hipStream_t to launch on; NULL = the library creates its own    */
    uint32_t cand_per_window; /* sparse path: expected candidates per window (0 = chosen by assembly size: 18 / 10) */
    uint32_t host_threads;    /* worker threads for file ingest and output (`indexlr -t`); 0 = min(16, cores) */
    uint32_t reserved[4];
} mxg_config;

/* host view of one assembly's ordered sketch: minimizer i is (out_hash[i], pos[i], record[i], strand[i]);
   entries are sorted by (record, pos); record indexes the assembly's FASTA records in input order. */
typedef struct mxg_sketch_view {
    uint64_t n;
    const uint64_t *out_hash;
    const uint32_t *pos;
    const uint32_t *record;
    const uint8_t *forward;      /* 1 = forward hash <= reverse hash ('+' under --strand) */
    uint64_t n_records;
    const uint64_t *record_first; /* n_records+1 offsets into the arrays above (CSR by record) */
} mxg_sketch_view;

/* device view (HBM pointers) of the same arrays, for collectives run by the caller (RCCL all-gather) */
typedef struct mxg_sketch_dview {
    uint64_t n;
    const void *out_hash; /* uint64[n] */
    const void *pos;      /* uint32[n] */
    const void *record;   /* uint32[n] */
    const void *forward;  /* uint8[n]  */
} mxg_sketch_dview;

/* per-minimizer classification after mxg_build_graph (parallel to mxg_sketch_view arrays) */
#define MXG_MX_UNIQUE 0x1u /* hash occurs exactly once in its assembly  -> key of mx_info (ntjoin_utils.py:187)  */
#define MXG_MX_SHARED 0x2u /* ... and exactly once in EVERY assembly    -> vertex of the minimizer graph          */
#define MXG_MX_INALL 0x4u  /* hash occurs (any number of times) in EVERY assembly: filter_minimizers' set rule (:152-165) */

typedef struct mxg_graph_view {
    uint32_t n_assemblies;
    uint64_t n_vertices;          /* |intersection|                                                     */
    const uint64_t *vertex_hash;  /* [n_vertices] out_hash = igraph vertex `name` (decimal string there) */
    /* [a * n_vertices + v]: where vertex v lies in assembly a  (= list_mx_info[a][name], ntjoin.py:183)  */
    const uint32_t *vertex_pos;
    const uint32_t *vertex_record;
    uint64_t n_edges;
    const uint32_t *edge_u;       /* [n_edges] vertex index, first-seen orientation (ntjoin_utils.py:101-108) */
    const uint32_t *edge_v;
    const uint32_t *edge_support; /* bit a set = assembly a (order of mxg_add_*) supports the edge      */
    const double *edge_weight;    /* sum of weights over support, in assembly order (ntjoin_utils.py:54-56) */
} mxg_graph_view;

typedef struct mxg_stats {
    uint32_t struct_size;
    uint32_t n_assemblies;
    uint64_t bases;            /* sum of record lengths over all assemblies                     */
    uint64_t kmers;            /* valid k-mers hashed (records with >= w valid k-mers)          */
    uint64_t minimizers;       /* sum of sketch sizes                                           */
    uint64_t candidates;       /* sparse path: candidates that reached the resolve kernel       */
    uint64_t dense_kmers;      /* k-mers re-processed by the dense (gap / fallback) path        */
    uint64_t unique;           /* minimizers flagged MXG_MX_UNIQUE                              */
    uint64_t vertices, edges;
    /* HIP-event times in ms, accumulated since mxg_create / mxg_reset_timers (MXG_FLAG_TIMING)  */
    double ms_hash;            /* hash + candidate kernels (the dominant kernel family)         */
    double ms_resolve;         /* window arg-min resolve + compaction + emit                    */
    double ms_graph;           /* uniqueness, intersection, vertex ids, edges                   */
    uint64_t launches_hash;    /* number of hash-kernel launches in ms_hash                     */
    uint64_t hash_kernel_bases;/* bases covered by those launches                               */
    /* MXG_FLAG_TIMING_FINE: ms_resolve split by kernel, ms_graph split into its three parts     */
    double ms_reorder;         /* arena entry -> exact hash -> ordered candidate slot            */
    double ms_resolve_kernel;  /* window arg-min decision per candidate                          */
    double ms_emit;            /* ordered compaction into the sketch arrays                      */
    double ms_join;            /* uniqueness + intersection (hash join, flags)                   */
    double ms_vertices;        /* vertex ids + adjacency arrays                                  */
    double ms_edges;           /* edge flags + edge compaction                                   */
    uint64_t bs_filter_bases;  /* bases covered by the bit-sliced filter (k = 32 route; 0: the rolling-hash kernel ran) */
    uint64_t graph_join;       /* the last mxg_build_graph's join: 1 = LDS tables per hash partition, 2 = the same behind coarse
                                  partitions (two levels), 3 = one global table; | 0x100: coarse partitions re-sized and the join
                                  redone, | 0x200: the LDS join gave up and the global table ran (a slot that was reserved)     */
    /* what the common route (every batch enqueued once, one host sync) could not finish: candidate floods beyond the estimate,
       stretches the device route cannot hold, output beyond its bound.  Such batches are redone one by one behind the good ones */
    uint64_t batches_redone;   /* batches redone through the synchronous route                               */
    uint64_t sync_assemblies;  /* assemblies none of whose batches could be kept                             */
    uint64_t retried_assemblies; /* assemblies whose batches went through the streams a second time, re-sized    */
    uint64_t deferred_stretches; /* candidate-free stretches sketched apart and merged in (satellites, low complexity) */
    uint64_t select_slices;    /* slices of 64 strips that went through k_bs_select (k = 32 route: bitmap -> selected minimizers
                                  in one kernel; 0: count -> reorder -> resolve ran)                                           */
    uint64_t slice_stretches;  /* candidate-free stretches (>= w k-mers without a candidate) that k_sel_stretch was handed: sketched
                                  one wave per slice right behind k_bs_select, their minimizers put into the slice's row            */
} mxg_stats;

/* ---- lifecycle ------------------------------------------------------------------------------- */
int mxg_abi_version(void);
int mxg_create(const mxg_config *cfg, mxg_handle **out);
void mxg_destroy(mxg_handle *h);
const char *mxg_last_error(const mxg_handle *h); /* h may be NULL: error of the last failed mxg_create */

/* ---- assemblies: call in the reference's order, references first (CLI order), target last -----
   every mxg_add_assembly_* returns the new assembly's index (>= 0) on success, a negative code on failure */
/* FASTA file (plain text; `>id comment`, multi-line, any case; non-ACGTU bytes invalidate k-mers). */
int mxg_add_assembly_fasta(mxg_handle *h, const char *name, double weight, const char *fasta_path);
/* Contig sharding for one-process-per-GPU runs: the same FASTA is opened by every rank; ALL records are registered
   (ids, lengths: record indices are global) but only the records of shard `shard` of `n_shards` are packed and
   sketched.  Shards are contiguous record ranges balanced by base count (mxg_shard_range), so concatenating the ranks'
   sketches in rank order yields the globally (record,pos)-sorted sketch. */
int mxg_add_assembly_fasta_shard(mxg_handle *h, const char *name, double weight, const char *fasta_path,
                                 uint32_t shard, uint32_t n_shards);
/* Sub-record sharding (chunks with a halo): as ..._fasta_shard, but shard s owns the BASE range
   [total*s/n, total*(s+1)/n) of the concatenated records, whatever the record boundaries: a record cut by the range is
   sketched in pieces.  Every window belongs to the shard its last k-mer starts in; a piece that does not start its
   record also loads the w valid k-mers before its first own one and withholds its first minimizer (the last minimizer
   of the piece before it), so the rank-ordered concatenation of the shards' sketches is exactly the sketch of the
   whole file: positions and record indices are those of the whole records.  mxg_assembly_shard gives the records the
   handle holds pieces of; mxg_assembly_continues says whether the first of them began on the shard before (its TSV
   line is then the continuation of that shard's last line).  Replaces a single `indexlr` process per assembly,
   reference ntJoin:204-205, for inputs with few very long records. */
int mxg_add_assembly_fasta_split(mxg_handle *h, const char *name, double weight, const char *fasta_path,
                                 uint32_t shard, uint32_t n_shards);
int mxg_assembly_continues(const mxg_handle *h, int assembly);  /* 1 / 0, negative: error */
/* records [*lo,*hi) of shard `shard`: cut points at multiples of total/n_shards of the cumulative base count
   (a record belongs to the shard its midpoint falls in).  Host only; usable without a device. */
int mxg_shard_range(const uint64_t *lengths, uint64_t n_records, uint32_t shard, uint32_t n_shards, uint64_t *lo,
                    uint64_t *hi);
/* the record range this handle sketches for an assembly ([0,n_records) unless added with ..._fasta_shard) */
int mxg_assembly_shard(const mxg_handle *h, int assembly, uint64_t *lo, uint64_t *hi);
/* Records already in host memory: record r is ascii[offsets[r] .. offsets[r+1]) with id ids[r]. */
int mxg_add_assembly_buffers(mxg_handle *h, const char *name, double weight, const uint8_t *ascii,
                             const uint64_t *offsets, const char *const *ids, uint64_t n_records);
/* Bases already resident in HBM, 2-bit packed (A=0,C=1,G=2,T=3; base i in bits 2*(i%16).. of 32-bit word
   i/16), no invalid bases.  record r starts at base rec_start[r] (a multiple of 16) and has rec_len[r]
   bases; the buffer must be readable 1 KiB past the last base.  d_packed is borrowed, not copied.
   ids may be NULL (records are then named "0","1",...). */
int mxg_add_assembly_packed_device(mxg_handle *h, const char *name, double weight, const void *d_packed,
                                   const uint64_t *rec_start, const uint64_t *rec_len,
                                   const char *const *ids, uint64_t n_records);
/* Sub-record sharding of bases already in HBM (the packed counterpart of mxg_add_assembly_fasta_split, for inputs that never
   were text: generated or produced on the device).  Every record is registered (ids, full lengths: record indices are global);
   the handle holds the bases [piece_lo[r], piece_hi[r]) of record r where piece_hi > piece_lo: base b of the record sits at
   packed index rec_start[r] + (b - (piece_lo[r] & ~15)) (rec_start a multiple of 16); piece_drop[r] bit 0 = the piece begins
   with the halo of the shard before it and withholds its first minimizer, bit 1 = the record began on an earlier shard (what
   mxg_assembly_continues reports for the handle's first record; set even when the halo reaches back to the record's base 0).  mxg_plan_split computes the pieces of shard `shard` of
   `n_shards` for N-free records (equal base ranges, a halo of w k-mers: the rule of mxg_add_assembly_fasta_split), so that the
   rank-ordered concatenation of the shards' sketches is the sketch of the whole assembly.  No counterpart in the reference
   (one process per assembly, ntJoin:204-205). */
int mxg_plan_split(const uint64_t *lengths, uint64_t n_records, uint32_t shard, uint32_t n_shards, uint32_t k, uint32_t w,
                   uint64_t *piece_lo, uint64_t *piece_hi, uint8_t *piece_drop);
int mxg_add_assembly_packed_device_pieces(mxg_handle *h, const char *name, double weight, const void *d_packed,
                                          const uint64_t *rec_start, const uint64_t *rec_len, const uint64_t *piece_lo,
                                          const uint64_t *piece_hi, const uint8_t *piece_drop, const char *const *ids,
                                          uint64_t n_records);
/* A sketch computed elsewhere: an indexlr TSV (`id \t hash:pos[:seq] ...`), parsed as read_minimizers does. */
int mxg_add_assembly_tsv(mxg_handle *h, const char *name, double weight, const char *tsv_path);
/* ... or the binary side-car mxg_write_sketch_bin left next to the TSV (the reference re-parses ~65
   bytes of text per minimizer in read_minimizers, bin/ntjoin_utils.py:167-193; the side-car is 12 bytes of raw arrays
   per minimizer).  Fails with MXG_EINVAL when the file was written with another k or hash variant. */
int mxg_add_assembly_bin(mxg_handle *h, const char *name, double weight, const char *bin_path);
/* ... or arrays (sorted by record, then pos); record_ids has n_records entries. */
int mxg_add_assembly_minimizers(mxg_handle *h, const char *name, double weight, const uint64_t *out_hash,
                                const uint32_t *pos, const uint32_t *record, uint64_t n,
                                const char *const *record_ids, uint64_t n_records);
int mxg_num_assemblies(const mxg_handle *h);
const char *mxg_assembly_name(const mxg_handle *h, int assembly);
/* id of record r of an assembly (first whitespace-delimited token of the FASTA header) */
const char *mxg_record_id(const mxg_handle *h, int assembly, uint64_t record);
uint64_t mxg_record_length(const mxg_handle *h, int assembly, uint64_t record);
uint64_t mxg_num_records(const mxg_handle *h, int assembly);
double mxg_assembly_weight(const mxg_handle *h, int assembly);

/* ---- sketch stage (replaces indexlr) --------------------------------------------------------- */
#define MXG_SKETCH_PENDING (-1) /* every assembly that has bases and no sketch yet                         */
#define MXG_SKETCH_ALL (-2)     /* every assembly that has bases (re-sketch); pipelined: one host sync for all */
int mxg_sketch(mxg_handle *h, int assembly /* index, MXG_SKETCH_PENDING or MXG_SKETCH_ALL */);
/* mxg_sketch(h, MXG_SKETCH_ALL) followed by mxg_build_graph(h) in ONE call and, in the common case, with ONE host sync: the
   graph kernels are enqueued behind the sketch kernels with upper bounds for the sizes and read the sketch sizes on the
   device.  Same results as the two calls (what ntJoin's `%.tsv` rule plus make_minimizer_graph produce, ntJoin:204-205,
   bin/ntjoin.py:189-204). */
int mxg_sketch_graph(mxg_handle *h);
int mxg_get_sketch(mxg_handle *h, int assembly, mxg_sketch_view *out);
/* (forward is NULL until the strands have been computed: mxg_get_sketch, mxg_write_tsv or mxg_compute_strands) */
int mxg_get_sketch_device(mxg_handle *h, int assembly, mxg_sketch_dview *out);
int mxg_compute_strands(mxg_handle *h, int assembly);
/* Replace an assembly's sketch by device arrays (e.g. the concatenation an all-gather produced);
   entries must be sorted by (record, pos).  The arrays are copied. */
int mxg_set_sketch_device(mxg_handle *h, int assembly, const void *d_out_hash, const void *d_pos,
                          const void *d_record, const void *d_forward, uint64_t n);
/* Exchange step of the multi-GPU path (one all-gather per assembly).  Every rank packs its sketch into
   one byte buffer of 16*nmax bytes laid out [out_hash u64 x nmax | pos u32 x nmax | record u32 x nmax] (nmax >= n,
   a multiple of 8), the caller all-gathers the buffers (RCCL), and mxg_set_sketch_gathered unpacks the `world`
   buffers in rank order -- counts[r] minimizers from rank r, record indices shifted by rec_offsets[r] -- straight into
   the assembly's sketch.  Strands do not travel (they are recomputed from bases on demand, or reported as '+'). */
int mxg_pack_sketch_device(mxg_handle *h, int assembly, void *d_buf, uint64_t nmax);
int mxg_set_sketch_gathered(mxg_handle *h, int assembly, const void *d_allbuf, uint32_t world, uint64_t nmax,
                            const uint64_t *counts, const uint64_t *rec_offsets);
/* the same when rank r's packed buffer starts at d_allbuf + r * stride_bytes (several assemblies in ONE all-gather) */
int mxg_set_sketch_gathered_strided(mxg_handle *h, int assembly, const void *d_allbuf, uint32_t world,
                                    uint64_t stride_bytes, uint64_t nmax, const uint64_t *counts,
                                    const uint64_t *rec_offsets);
/* Steady-state exchange with the counts on the device (the union path of ntjoin_amd/dist.py after its first step): a
   rank's slot = [int64 count per assembly (-1: does not fit) padded to head_bytes | caps[0] entries of assembly 0 as
   mxg_pack_sketch_device lays them out | caps[1] entries of assembly 1 | ...].  mxg_xchg_pack fills this handle's slot
   (every assembly, the counts written by the packing kernels); after ONE all-gather of the slots,
   mxg_xchg_unpack_graph on the union's handle unpacks all `world` slots with the counts read from the headers on the
   device (rec_offsets[a * world + r] = record index shift of rank r) and runs the graph stage behind it: one host sync
   for exchange + graph.  Returns 1 when some rank's header says "does not fit" (nothing usable: exchange sizes first,
   mxg_set_sketch_gathered), else as mxg_build_graph.  No counterpart in the reference (single process). */
int mxg_xchg_pack(mxg_handle *h, void *d_slot, uint64_t head_bytes, const uint64_t *caps);
/* mxg_sketch(h, MXG_SKETCH_ALL) + mxg_xchg_pack without the host sync in between: the sketches are enqueued, the packing
   kernels follow them on the stream and read the counts on the device (a sketch that did not end the common way -- arena
   overflow, candidate-free stretch, several batches -- or does not fit its slot travels as -1, so every rank falls back
   together).  The handle must have been created on the caller's stream (mxg_config.stream).  After the caller's next
   sync on that stream (mxg_xchg_unpack_graph on the union's handle does it), mxg_sketch_finish completes this handle's
   sketches (those that travelled as -1 are redone the ordinary way); until then the handle has no sketches. */
int mxg_sketch_pack(mxg_handle *h, void *d_slot, uint64_t head_bytes, const uint64_t *caps);
int mxg_sketch_finish(mxg_handle *h);
int mxg_xchg_unpack_graph(mxg_handle *h, const void *d_all, uint32_t world, uint64_t slot_bytes, uint64_t head_bytes,
                          const uint64_t *caps, const uint64_t *rec_offsets);
/* The same exchange with ONE BUFFER PER ASSEMBLY, so that an assembly's sketch can travel while the next assembly is still
   being sketched -- and with 12 bytes per minimizer instead of 16: part a = [64 bytes: int64 count (-1: does not fit), int64
   records | caps[a] hashes (u64) | caps[a] positions (u32) | rcaps[a] x u32: the first entry of every record of the sender
   (entries are in (record, position) order: the record column is a step function of the entry index, the receiver finds an
   entry's record by bisection)]; caps[a] a multiple of 8, rcaps[a] even and at least the sender's number of records;
   64 + 12 caps[a] + 4 rcaps[a] bytes.  mxg_sketch_pack_parts enqueues every assembly's sketch and packs each part right behind that assembly's own last
   kernel; mxg_part_packed_wait(h, a, stream) makes `stream` (a hipStream_t: the caller's communication stream) wait for part
   a -- the caller then issues the all-gather of part a on it, one collective per assembly.  mxg_xchg_unpack_graph_parts on
   the union's handle takes the gathered parts (d_all_parts[a] = world parts of assembly a, rank after rank) once the
   handle's stream has been made to wait for the collectives; returns as mxg_xchg_unpack_graph.  mxg_sketch_finish as
   above.  No counterpart in the reference (single process). */
int mxg_sketch_pack_parts(mxg_handle *h, void *const *d_parts, const uint64_t *caps, const uint64_t *rcaps);
int mxg_part_packed_wait(mxg_handle *h, int assembly, void *stream);
int mxg_xchg_unpack_graph_parts(mxg_handle *h, const void *const *d_all_parts, uint32_t world, const uint64_t *caps,
                                const uint64_t *rcaps, const uint64_t *rec_offsets);
/* indexlr TSV: `id \t out_hash[:pos][:+|-][:kmer] ( out_hash...)* \n`, one line per record, input order.
   path "-" = stdout. */
int mxg_write_tsv(mxg_handle *h, int assembly, const char *path, int with_pos, int with_strand,
                  int with_seq);
/* the same sketch as raw arrays (record ids, lengths, out_hash, pos): read back by mxg_add_assembly_bin */
int mxg_write_sketch_bin(mxg_handle *h, int assembly, const char *path);

/* ---- graph stage (replaces read_minimizers' uniqueness, filter_minimizers, build_graph) -------- */
int mxg_build_graph(mxg_handle *h);
/* flags[i] (MXG_MX_*) for minimizer i of the assembly's sketch */
int mxg_get_mx_flags(mxg_handle *h, int assembly, const uint8_t **flags, uint64_t *n);
int mxg_get_graph(mxg_handle *h, mxg_graph_view *out);
/* `.mx.dot` as Ntjoin.print_graph writes it (HEAD syntax); vertex/edge line order: vertices in first-assembly
   order, edges in first-seen order (the reference's own order is unspecified: python set order). */
int mxg_write_dot(mxg_handle *h, const char *path);
/* The .mx.dot written by several processes that each hold the WHOLE graph (one process per GPU, union route): part `part` of
   `n_parts` = the vertex lines [nv part / n, nv (part + 1) / n) and the edge lines likewise.  mxg_dot_part_format formats the
   part's two segments into memory and reports their sizes (bytes[0] vertices, bytes[1] edges); the caller exchanges the sizes
   (one small all-gather) and every process writes its segments at
       v_off = 10 + sum of bytes[0] of the parts before it,   e_off = 10 + sum of ALL bytes[0] + sum of bytes[1] of the parts before it
   (10 = strlen("graph G {\n"), written by the part with first != 0; the part with last != 0 appends "}\n").  The file must not
   be truncated by anybody after the first write.  Together the parts are byte for byte what mxg_write_dot writes.
   No counterpart in the reference (bin/ntjoin.py:25-67 is one process). */
int mxg_dot_part_format(mxg_handle *h, uint32_t part, uint32_t n_parts, uint64_t bytes[2]);
int mxg_dot_part_write(mxg_handle *h, const char *path, uint64_t v_off, uint64_t e_off, int first, int last);
/* The text outputs of a whole run in one call: <prefix>.mx.dot (mxg_write_dot) formatted by the host workers WHILE the TSVs of
   the assemblies (mxg_write_tsv with these flags; tsv_paths[a] == NULL: none for assembly a) are formatted on the device and
   written out -- the two use different resources (host threads / GPU + one writer), so a run's text output takes the longer
   of the two instead of their sum.  Same bytes as the separate calls.  Replaces the tail of reference ntJoin:204-205 (the
   `> $@` of every indexlr recipe) and bin/ntjoin.py:25-67 (print_graph) when one process does both. */
int mxg_write_outputs(mxg_handle *h, const char *dot_path, const char *const *tsv_paths, int with_pos, int with_strand,
                      int with_seq);

/* ---- next row: linear paths through the minimizer graph -------------------------------------------
   What the reference computes with igraph right after the graph (bin/ntjoin_assemble.py:759,779): filter_graph_global
   with minimum edge weight n (bin/ntjoin.py:80-89), then per component the branch filtering with rising thresholds
   (filter_graph :69-77, find_paths_process :137-161), circular components opened (check_circularity :113-135), source
   and target chosen by position in the highest-weight assembly (determine_source_vertex :91-103); a sub-component yields
   a path iff it is a simple chain.  Vertices are indices into mxg_graph_view; paths are ordered by source vertex. */
typedef struct mxg_paths_view {
    uint64_t n_components;           /* components of the globally filtered graph (bin/ntjoin.py:166-167)         */
    uint64_t n_paths;
    const uint64_t *path_first;      /* [n_paths+1] offsets into path_vertex                                     */
    const uint32_t *path_vertex;     /* vertex indices, source -> target                                          */
    const uint32_t *path_component;  /* [n_paths] id of the component of the globally filtered graph it came from */
} mxg_paths_view;
int mxg_find_paths(mxg_handle *h, int64_t min_edge_weight /* ntJoin's n */, mxg_paths_view *out);

/* ---- next row: what the scaffolder derives from the paths for one assembly (the target) ------------
   mxg_mx_extremes   = find_mx_min_max (bin/ntjoin_assemble.py:688-702): per record of the assembly, the smallest and
                       largest position among its minimizers that are graph vertices (no vertex: min = 2^32-1, max = 0).
   mxg_path_segments = the grouping loop of format_path (bin/ntjoin_assemble.py:175-218): every path of the last
                       mxg_find_paths is cut into runs of consecutive vertices lying on the same record of `assembly`;
                       per run what determine_orientation (:30-50) and calc_start/end_coord (:52-65) need. */
typedef struct mxg_segments_view {
    uint64_t n_segments;             /* in path order, then position in the path                                  */
    const uint32_t *seg_path;        /* index into mxg_paths_view                                                 */
    const uint32_t *seg_record;      /* record (contig) of the assembly                                           */
    const uint32_t *seg_first;       /* offset of the run's first vertex in mxg_paths_view.path_vertex            */
    const uint32_t *seg_stat;        /* 5 per run: vertices, min pos, max pos, increasing pairs, decreasing pairs */
} mxg_segments_view;
int mxg_path_segments(mxg_handle *h, int assembly, mxg_segments_view *out);
int mxg_mx_extremes(mxg_handle *h, int assembly, const uint32_t **min_pos, const uint32_t **max_pos, uint64_t *n_records);

/* ---- graph stage distributed over ranks by hash range (one process per GPU) --------------------------
   No counterpart in the reference (it is one process).  Uniqueness and intersection need every occurrence of a hash in
   one place: every minimizer travels to the rank that owns its hash, the owner runs the ordinary graph kernels on what it
   received and numbers its vertices, the verdict travels back, and adjacency (which stays with the records) reaches the
   owners of both end points as messages.  These entry points pack, unpack and count on the device; the all-to-all
   exchanges between them are the caller's (ntjoin_amd/dist.py: torch.distributed, backend "nccl" = RCCL).
   Sender side (the handle that holds the sketches):
     mxg_dg_owner_counts  counts[a * world + r] = minimizers of assembly a owned by rank r
     mxg_dg_pack_items    assembly a's minimizers as 16-byte items {hash, pos, record + rec_offset}, bucket r starting at
                          item starts[r] of d_send
     mxg_dg_msg_counts    after the verdicts came back (d_ret: 8 bytes per item, same layout as d_send): flags of every
                          assembly, and counts[a * world + r] = adjacency messages of assembly a for rank r; d_bases =
                          device array u32[world + 1], first global vertex id of every rank
     mxg_dg_pack_msgs     those messages (16 bytes each), bucket r starting at message starts[r] of d_send
   Owner side (a second handle with the same assemblies registered, no sketches):
     mxg_dg_set_items     assembly a's received items: source s's section starts at item sec_start[s] of d_items and holds
                          sec_count[s] items
     mxg_dg_vertices      uniqueness, intersection, local vertex ids; the owner's vertex count is written to the DEVICE
                          word d_n_vertices (u64) -- no host sync
     mxg_dg_item_results  the verdict of every item (flags | global vertex id << 8, or 2^32-1 << 8) written to d_out at the
                          item's place in the receive layout; d_gbase = device u32 holding this owner's first global id
     mxg_dg_edges         adjacency from the received messages, then the edges whose first supporter's source vertex is
                          local (the stage's sync); afterwards mxg_get_graph gives this owner's vertices and edges
                          (edge_u: local vertex index, edge_v: global vertex id) */
int mxg_dg_owner_counts(mxg_handle *h, uint32_t world, uint64_t *counts);
int mxg_dg_pack_items(mxg_handle *h, int assembly, uint32_t world, uint32_t rec_offset, const uint64_t *starts, void *d_send);
int mxg_dg_set_items(mxg_handle *h, int assembly, const void *d_items, uint32_t world, const uint64_t *sec_start,
                     const uint64_t *sec_count);
int mxg_dg_vertices(mxg_handle *h, void *d_n_vertices);
int mxg_dg_item_results(mxg_handle *h, int assembly, const void *d_gbase, uint32_t world, const uint64_t *sec_start,
                        const uint64_t *sec_count, void *d_out);
int mxg_dg_msg_counts(mxg_handle *h, uint32_t world, const void *d_ret, const void *d_bases, uint64_t *counts);
int mxg_dg_pack_msgs(mxg_handle *h, int assembly, uint32_t world, const void *d_bases, const uint64_t *starts, void *d_send);
/* Records cut between ranks (mxg_add_assembly_fasta_split / ..._packed_device_pieces): the adjacency between the last shared
   minimizer of one rank's piece and the first shared minimizer of the next rank's piece lies with neither rank.  After the
   verdicts came back, mxg_dg_last_shared writes {record, global vertex id} of this rank's LAST shared minimizer of every
   assembly (2^32-1: none) to d_out (device u32[A][2]); the caller all-gathers these (u32[world][A][2]) and hands them to
   mxg_dg_set_ghosts, after which mxg_dg_msg_counts / mxg_dg_pack_msgs / mxg_dg_pack_msg_slots also emit the pair (nearest
   shared minimizer of that record on an earlier rank, this rank's first).  d_all = NULL switches it off.  No host sync. */
int mxg_dg_last_shared(mxg_handle *h, const void *d_ret, void *d_out);
int mxg_dg_set_ghosts(mxg_handle *h, const void *d_all, uint32_t world, uint32_t rank);
int mxg_dg_edges(mxg_handle *h, const void *d_msgs, uint64_t n_msgs, uint64_t *n_vertices, uint64_t *n_edges);
/* Steady state of the same exchange with FIXED-CAPACITY SLOTS: once one exact step has shown the sizes, every (source,
   destination, assembly) triple gets a slot = 64-byte header (word 0 = the count) + cap[a] 16-byte items (<= 8 assemblies);
   the all-to-alls then have equal splits and the receivers read the counts on the device: no size exchange, no host sync
   before mxg_dg_edges_slots.  A count above its capacity sets *overflow there: repeat the step the exact way.  The item
   buffer is ASSEMBLY-MAJOR: assembly a's `world` slots lie side by side at byte offset sum over a' < a of
   world * (64 + 16 cap[a']), so that one all-to-all per assembly carries them (the verdict buffers stay [world][sum cap]).
   Message slots: 64-byte header (word 0 = count) + max_msgs 16-byte messages.  mxg_dg_pack_slots clears its assembly's
   headers, mxg_dg_pack_msg_slots all of them at its start.
   mxg_sketch_dg_pack_slots = mxg_sketch(h, MXG_SKETCH_ALL) + mxg_dg_pack_slots for every assembly WITHOUT the host sync in
   between: the sketches are enqueued, every assembly's items are packed right behind its own last kernel with the count
   read on the device (rec_offsets[a] = that assembly's record index shift), and mxg_part_packed_wait(h, a, stream) lets the
   caller's communication stream send assembly a's slots while the next assembly is still being sketched.  A sketch that did
   not end the common way reaches every destination as a count far above any capacity (overflow: all ranks repeat the step the
   exact way).  After the caller's sync on the handle's stream: mxg_sketch_finish.  No counterpart in the reference. */
int mxg_sketch_dg_pack_slots(mxg_handle *h, uint32_t world, uint32_t n_asm, const uint32_t *cap, const uint32_t *rec_offsets,
                             void *d_send);
int mxg_dg_pack_slots(mxg_handle *h, int assembly, uint32_t rec_offset, uint32_t world, uint32_t n_asm, const uint32_t *cap,
                      void *d_send);
int mxg_dg_owner_slots(mxg_handle *h, uint32_t world, uint32_t n_asm, const uint32_t *cap, const void *d_recv, void *d_n_vertices);
int mxg_dg_slot_results(mxg_handle *h, uint32_t world, uint32_t n_asm, const uint32_t *cap, const void *d_recv,
                        const void *d_gbase, void *d_out /* [world][sum cap] u64 */);
int mxg_dg_pack_msg_slots(mxg_handle *h, uint32_t world, uint32_t max_msgs, const void *d_ret, const void *d_bases, void *d_send);
int mxg_dg_edges_slots(mxg_handle *h, const void *d_recv, uint32_t world, uint32_t max_msgs, uint64_t *n_vertices,
                       uint64_t *n_edges, uint32_t *overflow);

/* ---- text helpers used by the writers (host only; usable without a device) --------------------- */
/* python repr() of a float / of a str, as Ntjoin.print_graph's f-strings produce them.  Returns the length
   written (excluding the NUL), or the length needed if it exceeds cap. */
size_t mxg_py_repr_double(double v, char *buf, size_t cap);
size_t mxg_py_repr_str(const char *s, char *buf, size_t cap);


/* ---- synthetic inputs (bench / test support; no counterpart in the reference, whose tests hold four small FASTA files)
   SURVEY.md 8(d) configs 2-5: assemblies of 0.1-20 Gbp "generated on device from a counter-based RNG mirrored on the
   CPU".  The genome is a pure function of (seed, coordinate): 32 bases per splitmix64 output; an assembly is a list of
   segments, output bases [dst_base, dst_base+len) = genome coordinates src..src+len-1, reverse-complemented when rc,
   each base substituted with probability sub_per_65536/65536 (decided by a second splitmix64 stream indexed by the
   coordinate).  Segments must be sorted by dst_base (multiples of 16, non-overlapping); words outside are zero.
   Formulas: ntjoin_amd/csrc/synth.hip, mirrored in numpy by ntjoin_amd/synth.py. */
typedef struct mxg_synth_seg {
    uint64_t dst_base; /* first output base (multiple of 16)          */
    uint64_t src;      /* genome coordinate of the segment's first base (its LAST output base when rc) */
    uint64_t len;      /* bases                                       */
    uint32_t rc;       /* 1: output is the reverse complement         */
    uint32_t reserved;
} mxg_synth_seg;
/* fills d_out[0..n_words) (HBM, 2-bit packed as mxg_add_assembly_packed_device expects); device < 0 = current device */
int mxg_synth_fill_packed_device(void *d_out, uint64_t n_words, const mxg_synth_seg *segs, uint64_t n_segs, uint64_t seed,
                                 uint64_t sub_seed, uint32_t sub_per_65536, int device);
/* the same words computed on the host (n_threads workers); usable without a device */
int mxg_synth_fill_packed_host(uint32_t *out, uint64_t n_words, const mxg_synth_seg *segs, uint64_t n_segs, uint64_t seed,
                               uint64_t sub_seed, uint32_t sub_per_65536, uint32_t n_threads);
/* FASTA text of packed host records (ids "<id_prefix><r>", `line` bases per line): materialises a synthetic assembly
   as a file for the end-to-end (FASTA -> .tsv + .mx.dot) measurement.  Host only. */
int mxg_synth_write_fasta(const char *path, const uint32_t *packed, const uint64_t *rec_start, const uint64_t *rec_len,
                          uint64_t n_records, const char *id_prefix, uint32_t line, uint32_t n_threads);

/* ---- introspection ----------------------------------------------------------------------------- */
int mxg_get_stats(mxg_handle *h, mxg_stats *out);
int mxg_reset_timers(mxg_handle *h);
/* The environment knobs (README: MXG_* tuning / test / profiling switches) this handle has read and found SET, as
   "NAME=value NAME=value ..." sorted by name.  A knob is parsed once per handle, at its first use, so this is what the
   handle really ran with.  Returns the text's length; writes at most cap - 1 bytes + NUL (buf may be NULL).  No reference
   counterpart (bench.py records it next to every number). */
size_t mxg_knobs(mxg_handle *h, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* NTJOIN_MX_H */
