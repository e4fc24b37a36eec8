// sketch_bs.h -- the k = 32 route of the sketch stage: bit-sliced ring filter (bs_kernels.h) + one kernel that turns its
// candidate bitmap into selected minimizers (exact hashes, window decision, candidate-free stretches).  Launchers for sketch.hip.
#pragma once
#include "mxg_internal.h"

namespace mxg {

#ifndef MXG_BS_CHUNK_DEFINED
#define MXG_BS_CHUNK_DEFINED
constexpr uint32_t BS_CHUNK = 65536;      // base positions per chunk of the bit-sliced filter (bs_kernels.h)
#endif

// k_bs_select: one WAVE per slice of 64 consecutive strips of the assembly's strip table (lane = strip): H halo strips, T own
// strips, H halo strips (T + 2 H = 64).  The slice's candidates never leave the wave's LDS: bits -> queue -> exact hashes ->
// window decision -> the selected ones, laid out per slice for k_emit (the layout k_resolve writes per block of 256 candidates).
constexpr uint32_t SEL_PAD = 8;          // sentinel entries on either side of a wave's candidate list (the scans look at eight at a time)
constexpr uint32_t SEL_REQ = 8;          // stretches per slice whose end lies behind the slice's strips (found by walking on)
constexpr uint32_t SEL_GAP_DROP = 0x80000000u;  // in a reported stretch's fourth word (the reporting slice's first entry): "leave the first window's arg-min out"
constexpr uint32_t SEL_MAX_H = 12;       // largest halo (strips) the route takes: 40 own strips per slice
// k_sel_stretch (round 6): the candidate-free stretches that lie between two candidates of one slice are sketched by a kernel of
// their own right behind the slice kernel, one wave per slice that has any, and their minimizers put into the slice's row
constexpr uint32_t SEL_INL_R = 4;        // windows per lane in one piece of a stretch: a piece has at most 64 R windows
constexpr uint32_t SEL_INL_PIECES = 6;   // pieces per stretch; longer stretches go to k_gap_fix
constexpr uint32_t SEL_INL_TMP = 128;    // minimizers of one stretch (more: k_gap_fix)
constexpr uint32_t SEL_IREQ_CAP = 1u << 18;  // requests per batch (the upper part of the stretch array)

struct BsSelParams {
    const uint32_t *bm;          // the filter's bitmap: bit p = base position p of the packed array (bs_kernels.h)
    const uint32_t *packed;
    const RunX *runx;            // [n_runs] the run table as this kernel reads it
    const uint32_t *strip_run;   // [n_strips_asm] the run of every strip (k_strip_runs)
    const uint8_t *ctg_drop;     // see ResolveParams::ctg_drop (sketch.hip); may be null
    const uint4 *ptab;           // position tables of the direct hash formula (make_init_tab, entries 256..: 2048 x 16 B)
    uint32_t n_strips_asm;       // strips of the whole assembly (halo strips may lie outside the batch)
    uint32_t strip_lo, strip_hi; // the batch's strips: every one of them is an own strip of exactly one slice
    uint32_t S, H, T;            // k-mers per strip, halo strips on either side, own strips per slice
    uint32_t n_slices;
    uint32_t w;
    uint64_t tau;
    uint32_t qcap;               // raw candidates a wave's LDS queue holds
    // a slice with more raw candidates than qcap works in one of n_ovf global-memory regions of ovf_cap entries (a wave's
    // worst case: 64 S) instead; no region left: the host redoes the batch (ctrl[6])
    uint64_t *ovf_h;
    uint32_t *ovf_e;
    uint32_t ovf_cap, n_ovf;
    uint32_t *ovf_next;          // ticket counter of the regions (zeroed with the control block)
    // results: the selected candidates of slice s from entry s * rk on + two-level counts (k_emit), stretches (k_gap_fix)
    uint32_t rk;
    uint4 *cs;                   // {hash lo, hash hi, k-mer index, contig}
    uint32_t *cnt, *sup;
    uint4 *gaps;
    uint32_t gap_cap;
    uint32_t gap_nmax;           // k-mers of the longest stretch reported in one piece (0: any); see sel_push_gap
    uint32_t *cand_spread;       // 64 counters, 32 words apart: the slices' own candidates (k_emit adds them up for the report)
    uint32_t *ctrl;              // [1] stretches, [6] "the host must redo this batch"
    uint32_t ablate;             // (profiling builds: every slice stops after phase n; 0 = run)
    // inl_amax != 0: a slice's stretches become requests for k_sel_stretch, {contig, first, last k-mer, slice | number in the
    // slice << 24 | stretches of the slice << 27}, the slice's requests next to one another; ctrl[15] counts them (0: every
    // stretch goes straight to k_gap_fix)
    uint32_t inl_amax;
    uint4 *ireq;
    uint32_t ireq_cap;
};

struct SelStretchParams {
    const uint32_t *packed;
    const Run *runs;
    const uint32_t *ctg_run0;
    const uint8_t *ctg_drop;     // see ResolveParams::ctg_drop (sketch.hip); may be null
    const uint4 *byte_tab;       // make_init_tab's first 256 entries (init_direct)
    HashTab tab;                 // the terms of an ntHash step
    uint32_t w, amax, rk;
    uint4 *cs;                   // the slices' rows (k_bs_select) and their two-level counts
    uint32_t *cnt, *sup;
    const uint4 *ireq;
    uint32_t ireq_cap;
    uint32_t *ctrl;              // [15] requests, [1] stretches (what does not fit here goes on to k_gap_fix)
    uint32_t *tickets;           // 64 counters, 32 words apart, zero: request 64 t + c is handed out by ticket t of counter c
    uint4 *gaps;
    uint32_t gap_cap, gap_nmax;
    uint32_t ablate;             // (profiling only, MXG_SST_ABLATE: 1 no rolls, 2 no first hash, 4 no window scans, 8 nothing per request, 32 no row update)
};
// the longest piece the window w allows (sketch_bs.hip: stretch_sketch); 0: no stretch is taken this way
uint32_t bs_select_inline_amax(uint32_t w);
int launch_sel_stretch(mxg_handle *h, const SelStretchParams &p, hipStream_t st);

struct BsSelGeom {
    uint32_t H, T, n_slices, rk, qcap, waves, ovf_cap, n_ovf;
    size_t lds;
    bool ok;
};
// halo (strips on either side of a slice's own strips) that gives every own candidate w k-mers of its contig on either side --
// or the contig's end -- whatever the run table looks like; 0: more than SEL_MAX_H strips (short runs between invalid bases)
uint32_t bs_select_halo(const Assembly *a, uint32_t S, uint32_t w);
BsSelGeom bs_select_geom(uint32_t S, uint32_t H, uint32_t w, double frac, uint32_t n_strips, uint32_t qcap_force, uint32_t rk_force = 0);
int launch_bs_select(mxg_handle *h, const BsSelParams &p, const BsSelGeom &g, hipStream_t st);

// layout + filter (per assembly)
bool bs_possible(const mxg_handle *h, const Assembly *a);
int bs_prepare(mxg_handle *h, Assembly *a);                                  // padded edge chunks, once per assembly
int bs_edges(mxg_handle *h, Assembly *a, hipStream_t st);                    // the padded copies of the first and last chunk's words (before every filter launch)
int bs_hash(mxg_handle *h, Assembly *a, uint32_t tau_hi, hipStream_t st);    // the filter over the whole assembly -> a->d_bs_out

}  // namespace mxg
