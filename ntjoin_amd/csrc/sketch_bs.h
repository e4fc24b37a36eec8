// sketch_bs.h -- the k = 32 route of the sketch stage: bit-sliced ring filter (bs_kernels.h) + one kernel that turns its
// candidate bitmap into minimizers (exact hashes, window decision, candidate-free stretches).  Launchers for sketch.hip.
#pragma once
#include "mxg_internal.h"

namespace mxg {

#ifndef MXG_BS_CHUNK_DEFINED
#define MXG_BS_CHUNK_DEFINED
constexpr uint32_t BS_CHUNK = 65536;      // base positions per chunk of the bit-sliced filter (bs_kernels.h)
#endif
constexpr uint32_t BSR_THREADS = 1024;   // 16 waves per chunk: the block is a chain of short dependent phases, parallel slack hides them
constexpr uint32_t BSR_RUNS = 96;        // runs of the run table a block keeps in LDS (more overlap its range: host redo)
constexpr uint32_t BSR_HALO_LANE = 1024; // base positions per halo unit (one lane of a chunk)

struct BsResolveParams {
    // the filter's result for the whole assembly, the assembly's bases and tables
    const uint32_t *out;       // the filter's bitmap: bit p = position p (bs_kernels.h)
    uint32_t n_chunks;
    const uint32_t *packed;
    uint64_t n_words;
    const Run *runs;
    const uint32_t *chunk_run0;  // [n_chunks + 1] first run whose k-mers end behind the chunk's first position
    const uint32_t *ctg_nk;
    const uint8_t *ctg_drop;     // see ResolveParams::ctg_drop (sketch.hip); may be null
    const uint4 *init_tab;       // byte table of the direct hash formula (make_init_tab)
    HashTab tab;
    // the batch: contigs [ctg_lo, ctg_hi) = runs [run_lo, run_hi); one block per chunk from chunk_lo on
    uint32_t run_lo, run_hi, ctg_lo, ctg_hi;
    uint32_t chunk_lo;
    uint32_t k, w;
    uint64_t tau;
    uint32_t halo_l, halo_r;     // halo in units of BSR_HALO_LANE positions
    uint32_t max_cand;           // candidates a block can hold (dynamic LDS is sized for it)
    // results: the selected candidates of block b from entry b * rk on + two-level counts (k_emit), stretches (k_gap_fix)
    uint32_t rk;
    uint64_t *cs_h;
    uint32_t *cs_k, *cs_c;
    uint32_t *cnt, *sup;
    uint4 *gaps;
    uint32_t gap_cap;
    uint32_t ablate;             // (profiling builds: stop after phase n; 0 = run)
    unsigned long long *dbg;     // (profiling: 16 cycle stamps per block, or null)
    uint32_t *cand_spread;       // 64 counters, 32 words apart: the blocks' own candidates (k_emit adds them up for the report)
    uint32_t *ctrl;              // [1] stretches, [6] "the host must redo this batch"
};

size_t bs_resolve_lds(const BsResolveParams &p);
void launch_bs_resolve(const BsResolveParams &p, uint32_t n_blocks, hipStream_t st);

// layout + filter (per assembly)
bool bs_possible(const mxg_handle *h, const Assembly *a);
int bs_prepare(mxg_handle *h, Assembly *a);                                  // T / Q / chunk_run0, once per assembly
int bs_hash(mxg_handle *h, Assembly *a, uint32_t tau_hi, hipStream_t st);    // the filter over the whole assembly -> a->d_bs_out

}  // namespace mxg
