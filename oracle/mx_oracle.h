/*
 * mx_oracle.h -- CPU oracle for the ntJoin minimizer-sketch hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load or execute it, and only as the checker / reported CPU baseline.
 *
 * What it restates: the behaviour of btllib's `indexlr` as ntJoin invokes it
 * (reference ntJoin:204-205: `indexlr --seq --long --pos -k K -w W -t T`).
 * btllib is a third-party dependency that is NOT vendored under
 * /root/reference and is required un-pinned (reference requirements.txt:4),
 * so the arithmetic below restates the published ntHash / indexlr algorithm
 *  and is pinned by the reference's own golden files:
 *   - tests/expected_outputs/{ref.fa,scaf.f-f.fa}.k32.w1000.tsv  (variant V1, hashes+positions, bit-exact)
 *   - tests/ntjoin_test.py:133,141,148,157,202,207,220            (variant V2, positions)
 * Hash VALUES under the default variant V2 are pinned by no reference file
 * ("parity unpinned" for V2 hash values; the ext() formula is pinned via V1).
 */
#ifndef MX_ORACLE_H
#define MX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* canonical-hash variants */
#define MXO_VARIANT_V2_SUM 0 /* min_hash = fwd + rev (mod 2^64): current btllib, pinned by HEAD tests  */
#define MXO_VARIANT_V1_MIN 1 /* min_hash = min(fwd, rev): older btllib, pinned by expected_outputs tsv files */

typedef struct {
    uint64_t out_hash; /* what indexlr prints: second ntHash value, ext(min_hash)     */
    uint64_t min_hash; /* ordering key: canonical hash of the k-mer                    */
    uint32_t pos;      /* 0-based offset of the k-mer's first base within the record   */
    uint8_t forward;   /* fwd_hash <= rev_hash  (printed as +/- only with --strand)    */
} mxo_minimizer;

/* split-rotate primitives */
uint64_t mxo_srol(uint64_t x);
uint64_t mxo_sror(uint64_t x);
uint64_t mxo_srol_n(uint64_t x, unsigned n);
/* seed of a base; 0 for any byte that is not A/C/G/T/U (either case) */
uint64_t mxo_seed(unsigned char c);
uint64_t mxo_seed_comp(unsigned char c);

/* direct (non-rolling) forward / reverse-complement hashes of seq[0..k) ; returns 0 if any byte invalid */
int mxo_nthash_direct(const char *seq, unsigned k, uint64_t *fwd, uint64_t *rev);
/* canonical + extension */
uint64_t mxo_canonical(uint64_t fwd, uint64_t rev, int variant);
uint64_t mxo_ext_hash(uint64_t min_hash, unsigned k);

/*
 * Per-k-mer hashes of a record.  For i in [0, len-k]: valid[i] != 0 iff the k-mer has only ACGTU
 * bases; min_hash[i], out_hash[i], forward[i] are then set.  Rolling formulas are used where
 * the previous k-mer was valid, the direct formula otherwise.  Any output pointer may be NULL.
 * Returns the number of valid k-mers.
 */
size_t mxo_kmer_hashes(const char *seq, size_t len, unsigned k, int variant, uint64_t *min_hash,
                       uint64_t *out_hash, uint8_t *forward, uint8_t *valid);

/*
 * Minimizer sketch of one record, btllib-style STATEFUL loop (ring buffer of w+1 hashed k-mers,
 * rescan with `<=` when the current minimum leaves the window).  *out is malloc'd; caller frees.
 */
size_t mxo_sketch_stateful(const char *seq, size_t len, unsigned k, unsigned w, int variant,
                           mxo_minimizer **out);
/* the same loop with its per-k-mer arrays in a caller-owned workspace (zero-initialise; free with mxo_workspace_free) */
typedef struct {
    uint64_t *mh, *oh;
    uint8_t *fw, *ok;
    size_t cap;
    mxo_minimizer *ring;
    size_t ring_cap;
} mxo_workspace;
size_t mxo_sketch_stateful_ws(const char *seq, size_t len, unsigned k, unsigned w, int variant,
                              mxo_minimizer **out, mxo_workspace *ws);
void mxo_workspace_free(mxo_workspace *ws);
/*
 * Same result from the STATELESS definition (rightmost arg-min of every window of w consecutive
 * valid k-mers, distinct arg-mins in order), computed with a monotone deque.  Independent code path
 * used to cross-check the stateful loop and as the definition the GPU kernels implement.
 */
size_t mxo_sketch_stateless(const char *seq, size_t len, unsigned k, unsigned w, int variant,
                            mxo_minimizer **out);

void mxo_free(void *p);

/*
 * Whole-file driver: read FASTA `path`, sketch each record (stateful loop), write the indexlr TSV
 * (`id \t hash[:pos][:strand][:seq] ...\n`) to `out_path` ("-" = stdout).  Returns 0 on success.
 * stats (may be NULL): [0]=records, [1]=bases, [2]=minimizers.
 */
int mxo_sketch_fasta_to_tsv(const char *path, const char *out_path, unsigned k, unsigned w,
                            int variant, int with_pos, int with_strand, int with_seq,
                            uint64_t *stats);

/*
 * CPU baseline legs (mx_oracle_mt.c).  mxo_sketch_packed_mt: `indexlr -t T` on 2-bit packed N-free records (record r =
 * bases [rec_start[r], +rec_len[r]) of `packed`, 16 bases per word): n_threads workers over chunks of chunk_kmers
 * k-mers (0 = 256 Ki) with a w-1 halo, each running the stateful loop above; outputs are malloc'd arrays sorted by
 * (record, pos) (free with mxo_free).  Returns the number of minimizers.
 */
size_t mxo_sketch_packed_mt(const uint32_t *packed, const uint64_t *rec_start, const uint64_t *rec_len, size_t n_rec, unsigned k,
                            unsigned w, int variant, unsigned n_threads, uint64_t chunk_kmers, uint64_t **out_hash,
                            uint32_t **out_pos, uint32_t **out_rec);
/*
 * Graph stage on arrays: uniqueness per assembly, intersection, adjacency edges with support masks and weights
 * (reference bin/ntjoin_utils.py:182-193,152-165,83-141), single-threaded.  counts = {unique minimizers, vertices, edges}.
 * Edge arrays (optional, malloc'd): source hash, target hash (first-seen orientation), support mask, weight.
 */
int mxo_graph(unsigned A, const uint64_t *const *hash, const uint32_t *const *rec, const uint64_t *n, const double *weights,
              uint64_t counts[3], uint64_t **eu, uint64_t **ev, uint32_t **esup, double **ew);

#ifdef __cplusplus
}
#endif
#endif
