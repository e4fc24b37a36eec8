/*
 * mx_oracle_mt.c -- the CPU baseline legs of the oracle.  TEST INFRASTRUCTURE ONLY (see mx_oracle.h): loaded by tests/
 * and by bench.py's cpu_baseline, never by the product.
 *
 * (1) mxo_sketch_packed_mt: `indexlr -t T` as ntJoin runs it (reference ntJoin:47-48,204-205: record-level worker
 *     threads; SURVEY.md 8(d) CPU baseline (2): "rolling hash + ring buffer, one record per worker, long records
 *     chunked, -t $(nproc)").  Work items are chunks of CHUNK k-mers of one record plus a halo of w-1 k-mers on the
 *     left; every item runs the SAME stateful loop as mxo_sketch_stateful (mx_oracle.c, pinned by the reference's golden
 *     files) and reports the arg-mins of the windows that END inside its chunk; neighbouring chunks can only repeat one
 *     minimizer at their seam (arg-mins are non-decreasing in the window), which is dropped when the
 *     pieces are concatenated.  Input is 2-bit packed (what the benchmark holds); every worker expands its chunk to
 *     ASCII first, so the hashing runs through the byte tables of the pinned single-record code path.
 * (2) mxo_graph: read_minimizers' uniqueness, filter_minimizers and build_graph (reference bin/ntjoin_utils.py:182-193,
 *     152-165, 83-141) restated on arrays in plain C, single-threaded like the reference; checked against
 *     oracle/graph_oracle.py (itself pinned by fixtures generated from the imported reference) in tests/.
 */
#include <malloc.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mx_oracle.h"

typedef struct {
    uint32_t rec;
    uint64_t k0, k1; /* windows ending at k-mers [k0, k1) of the record are this item's */
    mxo_minimizer *mx;
    size_t n;
} item_t;

typedef struct {
    const uint32_t *packed;
    const uint64_t *rec_start, *rec_len;
    unsigned k, w;
    int variant;
    item_t *items;
    size_t n_items;
    volatile size_t next;
    pthread_mutex_t mu;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    char *buf = NULL;
    size_t cap = 0;
    mxo_workspace ws;
    memset(&ws, 0, sizeof ws);
    for (;;) {
        pthread_mutex_lock(&j->mu);
        size_t i = j->next++;
        pthread_mutex_unlock(&j->mu);
        if (i >= j->n_items) break;
        item_t *it = &j->items[i];
        /* the piece starts w-1 k-mers before the chunk (k0 >= w-1 always), so its first window ends at k-mer k0 */
        const uint64_t b0 = it->k0 - (uint64_t)(j->w - 1); /* first base (= first k-mer) of the piece */
        const uint64_t nb = (it->k1 - b0) + j->k - 1;     /* bases of the piece                      */
        if (nb + 1 > cap) {
            cap = nb + 1;
            buf = (char *)realloc(buf, cap);
            if (!buf) { fprintf(stderr, "mx_oracle: out of memory\n"); abort(); }
        }
        const uint64_t g0 = j->rec_start[it->rec] + b0;
        for (uint64_t q = 0; q < nb; ++q) {
            const uint64_t g = g0 + q;
            buf[q] = "ACGT"[(j->packed[g >> 4] >> (2 * (g & 15))) & 3u];
        }
        buf[nb] = 0;
        it->n = mxo_sketch_stateful_ws(buf, (size_t)nb, j->k, j->w, j->variant, &it->mx, &ws);
        for (size_t m = 0; m < it->n; ++m) it->mx[m].pos += (uint32_t)b0;
    }
    free(buf);
    mxo_workspace_free(&ws);
    return NULL;
}

size_t mxo_sketch_packed_mt(const uint32_t *packed, const uint64_t *rec_start, const uint64_t *rec_len, size_t n_rec, unsigned k,
                            unsigned w, int variant, unsigned n_threads, uint64_t chunk_kmers, uint64_t **out_hash,
                            uint32_t **out_pos, uint32_t **out_rec)
{
    *out_hash = NULL;
    *out_pos = *out_rec = NULL;
    if (!chunk_kmers) chunk_kmers = 1u << 18;
    if (!n_threads) n_threads = 1;
    /* every work item allocates ~17 B per k-mer in the pinned single-record routine: keep those blocks in the threads'
       arenas instead of mmap/munmap-ing them per item (with 256 workers the page faults serialise in the kernel) */
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    size_t n_items = 0;
    for (size_t r = 0; r < n_rec; ++r) {
        if (rec_len[r] < k) continue;
        const uint64_t nk = rec_len[r] - k + 1;
        if (nk < w) continue;
        n_items += (size_t)((nk - (w - 1) + chunk_kmers - 1) / chunk_kmers);
    }
    item_t *items = (item_t *)calloc(n_items ? n_items : 1, sizeof(item_t));
    size_t t = 0;
    for (size_t r = 0; r < n_rec; ++r) {
        if (rec_len[r] < k) continue;
        const uint64_t nk = rec_len[r] - k + 1;
        if (nk < w) continue;
        /* the first window ends at k-mer w-1: chunk c owns window ends [w-1 + c*C, w-1 + (c+1)*C) */
        for (uint64_t e0 = w - 1; e0 < nk; e0 += chunk_kmers) {
            items[t].rec = (uint32_t)r;
            items[t].k0 = e0;
            items[t].k1 = e0 + chunk_kmers < nk ? e0 + chunk_kmers : nk;
            ++t;
        }
    }
    job_t job;
    memset(&job, 0, sizeof job);
    job.packed = packed; job.rec_start = rec_start; job.rec_len = rec_len;
    job.k = k; job.w = w; job.variant = variant;
    job.items = items; job.n_items = n_items; job.next = 0;
    pthread_mutex_init(&job.mu, NULL);
    pthread_t *th = (pthread_t *)calloc(n_threads, sizeof(pthread_t));
    for (unsigned i = 1; i < n_threads; ++i) pthread_create(&th[i], NULL, worker, &job);
    worker(&job);
    for (unsigned i = 1; i < n_threads; ++i) pthread_join(th[i], NULL);
    free(th);
    pthread_mutex_destroy(&job.mu);
    size_t total = 0;
    for (size_t i = 0; i < n_items; ++i) total += items[i].n;
    uint64_t *oh = (uint64_t *)malloc((total ? total : 1) * 8);
    uint32_t *op = (uint32_t *)malloc((total ? total : 1) * 4), *orc = (uint32_t *)malloc((total ? total : 1) * 4);
    size_t n = 0;
    for (size_t i = 0; i < n_items; ++i) {
        for (size_t m = 0; m < items[i].n; ++m) {
            const mxo_minimizer *x = &items[i].mx[m];
            if (n && orc[n - 1] == items[i].rec && op[n - 1] >= x->pos) continue; /* the seam's repeat */
            oh[n] = x->out_hash; op[n] = x->pos; orc[n] = items[i].rec;
            ++n;
        }
        free(items[i].mx);
    }
    free(items);
    *out_hash = oh; *out_pos = op; *out_rec = orc;
    return n;
}

/* ---- graph stage --------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t key;
    uint32_t seen, dup, vid, used;
} gslot;

static inline size_t gfind(gslot *tab, size_t mask, uint64_t key)
{
    size_t s = (size_t)((key * 0x9E3779B97F4A7C15ULL) >> 20) & mask;
    while (tab[s].used && tab[s].key != key) s = (s + 1) & mask;
    return s;
}

typedef struct {
    uint64_t key; /* min(u,v) << 32 | max(u,v) */
    uint32_t idx, used;
} eslot;

/*
 * in:  A assemblies in the reference's order (refs, then target); hash[a][i], rec[a][i] for i < n[a], sorted by
 *      (record, position); weights[a].
 * out: counts[0] = minimizers unique in their assembly (summed), counts[1] = vertices, counts[2] = edges; if the edge
 *      pointers are non-NULL they receive malloc'd arrays (source hash, target hash, support mask, weight) in the order
 *      the reference's dictionary would first see them (bin/ntjoin_utils.py:101-108).
 */
int mxo_graph(unsigned A, const uint64_t *const *hash, const uint32_t *const *rec, const uint64_t *n, const double *weights,
              uint64_t counts[3], uint64_t **eu, uint64_t **ev, uint32_t **esup, double **ew)
{
    if (A == 0 || A > 32) return -1;
    uint64_t N = 0;
    for (unsigned a = 0; a < A; ++a) N += n[a];
    size_t cap = 1024;
    while (cap < 2 * N) cap <<= 1;
    gslot *tab = (gslot *)calloc(cap, sizeof(gslot));
    if (!tab) return -2;
    const size_t mask = cap - 1;
    /* read_minimizers: a hash seen twice in an assembly is dropped from it (bin/ntjoin_utils.py:182-192) */
    for (unsigned a = 0; a < A; ++a)
        for (uint64_t i = 0; i < n[a]; ++i) {
            gslot *s = &tab[gfind(tab, mask, hash[a][i])];
            if (!s->used) { s->used = 1; s->key = hash[a][i]; s->vid = 0xFFFFFFFFu; }
            if (s->seen & (1u << a)) s->dup |= 1u << a;
            s->seen |= 1u << a;
        }
    const uint32_t full = A == 32 ? 0xFFFFFFFFu : ((1u << A) - 1u);
    uint64_t n_unique = 0, nv = 0;
    for (unsigned a = 0; a < A; ++a)
        for (uint64_t i = 0; i < n[a]; ++i) {
            const gslot *s = &tab[gfind(tab, mask, hash[a][i])];
            if (!(s->dup & (1u << a))) ++n_unique;
        }
    /* filter_minimizers (:152-165): kept = in every assembly's de-duplicated list.  build_graph (:83-141): consecutive
       kept minimizers of a contig make an edge; the first (s,t) orientation seen is the key */
    size_t ecap = 1024;
    while (ecap < 2 * N) ecap <<= 1;
    eslot *et = (eslot *)calloc(ecap, sizeof(eslot));
    uint64_t ne = 0, ealloc = N ? N : 1;
    uint64_t *U = (uint64_t *)malloc(ealloc * 8), *V = (uint64_t *)malloc(ealloc * 8);
    uint32_t *S = (uint32_t *)malloc(ealloc * 4);
    if (!et || !U || !V || !S) return -2;
    for (unsigned a = 0; a < A; ++a) {
        int have_prev = 0;
        uint32_t prev_rec = 0, prev_vid = 0;
        uint64_t prev_hash = 0;
        for (uint64_t i = 0; i < n[a]; ++i) {
            gslot *s = &tab[gfind(tab, mask, hash[a][i])];
            if (s->seen != full || s->dup) continue;
            if (s->vid == 0xFFFFFFFFu) s->vid = (uint32_t)nv++;
            if (have_prev && prev_rec == rec[a][i]) {
                const uint32_t u = prev_vid, v = s->vid;
                const uint64_t key = u < v ? ((uint64_t)u << 32) | v : ((uint64_t)v << 32) | u;
                size_t q = (size_t)((key * 0x9E3779B97F4A7C15ULL) >> 20) & (ecap - 1);
                while (et[q].used && et[q].key != key) q = (q + 1) & (ecap - 1);
                if (!et[q].used) {
                    et[q].used = 1; et[q].key = key; et[q].idx = (uint32_t)ne;
                    U[ne] = prev_hash; V[ne] = hash[a][i]; S[ne] = 0;
                    ++ne;
                }
                S[et[q].idx] |= 1u << a;
            }
            have_prev = 1; prev_rec = rec[a][i]; prev_vid = s->vid; prev_hash = hash[a][i];
        }
    }
    counts[0] = n_unique; counts[1] = nv; counts[2] = ne;
    if (eu && ev && esup && ew) {
        double *W = (double *)malloc((ne ? ne : 1) * 8);
        for (uint64_t e = 0; e < ne; ++e) {
            double s = 0.0; /* python: sum(weights[f] for f in support), support in assembly order */
            for (unsigned a = 0; a < A; ++a)
                if (S[e] & (1u << a)) s = s + weights[a];
            W[e] = s;
        }
        *eu = U; *ev = V; *esup = S; *ew = W;
    } else {
        free(U); free(V); free(S);
    }
    free(et);
    free(tab);
    return 0;
}
