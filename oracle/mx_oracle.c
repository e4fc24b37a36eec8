/*
 * mx_oracle.c -- CPU oracle (plain C) for the minimizer-sketch stage.  TEST INFRASTRUCTURE ONLY
 * (see mx_oracle.h).  Restates btllib `indexlr` as invoked at reference ntJoin:204-205; semantics
 * per SURVEY.md Appendix A; pinned by tests/golden (reference tests/expected_outputs tsv files and the
 * coordinates asserted in reference tests/ntjoin_test.py:133,148,202).
 */
#include "mx_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- ntHash constants (validated bit-exactly by the V1 golden TSVs) ---- */
static const uint64_t SEED_A = 0x3c8bfbb395c60474ULL;
static const uint64_t SEED_C = 0x3193c18562a02b4cULL;
static const uint64_t SEED_G = 0x20323ed082572324ULL;
static const uint64_t SEED_T = 0x295549f54be24456ULL;
static const uint64_t MULTISEED = 0x90b45d39fb6da1faULL;
static const unsigned MULTISHIFT = 27;

uint64_t mxo_seed(unsigned char c)
{
    switch (c) {
    case 'A': case 'a': return SEED_A;
    case 'C': case 'c': return SEED_C;
    case 'G': case 'g': return SEED_G;
    case 'T': case 't': case 'U': case 'u': return SEED_T;
    default: return 0; /* invalid base: k-mer is skipped (A.3) */
    }
}

uint64_t mxo_seed_comp(unsigned char c)
{
    switch (c) {
    case 'A': case 'a': return SEED_T;
    case 'C': case 'c': return SEED_G;
    case 'G': case 'g': return SEED_C;
    case 'T': case 't': case 'U': case 'u': return SEED_A;
    default: return 0;
    }
}

/* split rotate: low 33 bits and high 31 bits rotate left by one, each within itself (A.1) */
uint64_t mxo_srol(uint64_t x)
{
    uint64_t m = ((x & 0x8000000000000000ULL) >> 30) | ((x & 0x100000000ULL) >> 32);
    return ((x << 1) & 0xFFFFFFFDFFFFFFFFULL) | m;
}

uint64_t mxo_sror(uint64_t x)
{
    uint64_t m = ((x & 0x200000000ULL) << 30) | ((x & 1ULL) << 32);
    return ((x >> 1) & 0xFFFFFFFEFFFFFFFFULL) | m;
}

uint64_t mxo_srol_n(uint64_t x, unsigned n)
{
    /* independent formulation: rotate the 33-bit and 31-bit halves by n mod 33 / n mod 31 */
    const uint64_t LO_MASK = 0x1FFFFFFFFULL; /* bits 0..32 */
    uint64_t lo = x & LO_MASK;
    uint64_t hi = x >> 33; /* 31 bits */
    unsigned a = n % 33, b = n % 31;
    if (a) lo = ((lo << a) | (lo >> (33 - a))) & LO_MASK;
    if (b) hi = ((hi << b) | (hi >> (31 - b))) & 0x7FFFFFFFULL;
    return (hi << 33) | lo;
}

int mxo_nthash_direct(const char *seq, unsigned k, uint64_t *fwd, uint64_t *rev)
{
    uint64_t f = 0, r = 0;
    for (unsigned j = 0; j < k; ++j) {
        uint64_t s = mxo_seed((unsigned char)seq[j]);
        if (!s) return 0;
        f ^= mxo_srol_n(s, k - 1 - j);
        r ^= mxo_srol_n(mxo_seed_comp((unsigned char)seq[j]), j);
    }
    *fwd = f;
    *rev = r;
    return 1;
}

uint64_t mxo_canonical(uint64_t fwd, uint64_t rev, int variant)
{
    if (variant == MXO_VARIANT_V1_MIN) return fwd < rev ? fwd : rev;
    return fwd + rev; /* V2: wraps mod 2^64 */
}

uint64_t mxo_ext_hash(uint64_t min_hash, unsigned k)
{
    /* second hash of ntHash's multi-hash extension: index i = 1 (A.2) */
    uint64_t t = min_hash * (1ULL ^ ((uint64_t)k * MULTISEED));
    t ^= t >> MULTISHIFT;
    return t;
}

size_t mxo_kmer_hashes(const char *seq, size_t len, unsigned k, int variant, uint64_t *min_hash,
                       uint64_t *out_hash, uint8_t *forward, uint8_t *valid)
{
    if (k == 0 || len < k) return 0;
    size_t n = len - k + 1, nvalid = 0;
    size_t run = 0; /* consecutive valid bases ending at the base just consumed */
    uint64_t f = 0, r = 0;
    int have_prev = 0; /* k-mer at i-1 was valid -> f,r hold its hashes */
    /* per-byte tables of the four terms of the rolling update (A.1), built once per call */
    uint64_t t_in_f[256], t_out_f[256], t_in_r[256], t_out_r[256];
    for (unsigned c = 0; c < 256; ++c) {
        uint64_t s = mxo_seed((unsigned char)c), sc = mxo_seed_comp((unsigned char)c);
        t_in_f[c] = s;                   /* SEED[in]                 */
        t_out_f[c] = mxo_srol_n(s, k);   /* srol^k(SEED[out])        */
        t_out_r[c] = sc;                 /* SEED[comp(out)]          */
        t_in_r[c] = mxo_srol_n(sc, k);   /* srol^k(SEED[comp(in)])   */
    }
    for (size_t j = 0; j + 1 < k; ++j) run = t_in_f[(unsigned char)seq[j]] ? run + 1 : 0;
    for (size_t i = 0; i < n; ++i) {
        unsigned char cin = (unsigned char)seq[i + k - 1];
        run = t_in_f[cin] ? run + 1 : 0;
        int ok = run >= k;
        if (ok) {
            if (have_prev) {
                unsigned char cout = (unsigned char)seq[i - 1];
                /* rolling update (A.1) */
                f = mxo_srol(f) ^ t_out_f[cout] ^ t_in_f[cin];
                r = mxo_sror(r ^ t_out_r[cout] ^ t_in_r[cin]);
            } else {
                mxo_nthash_direct(seq + i, k, &f, &r);
            }
            uint64_t h0 = mxo_canonical(f, r, variant);
            if (min_hash) min_hash[i] = h0;
            if (out_hash) out_hash[i] = mxo_ext_hash(h0, k);
            if (forward) forward[i] = (uint8_t)(f <= r);
            ++nvalid;
        }
        if (valid) valid[i] = (uint8_t)ok;
        have_prev = ok;
    }
    return nvalid;
}

/* growable output */
typedef struct {
    mxo_minimizer *v;
    size_t n, cap;
} mvec;

static void mvec_push(mvec *m, mxo_minimizer x)
{
    if (m->n == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 64;
        m->v = (mxo_minimizer *)realloc(m->v, m->cap * sizeof(mxo_minimizer));
        if (!m->v) { fprintf(stderr, "mx_oracle: out of memory\n"); abort(); }
    }
    m->v[m->n++] = x;
}

/*
 * Stateful loop, following btllib's Indexlr::minimize / calc_minimizer:
 * a ring of w+1 hashed k-mers indexed by the count of VALID k-mers seen so far (`idx`); from
 * idx+1 >= w on, window = the last w valid k-mers; if there is no current minimum or it lies left
 * of the window, rescan the window left->right taking `<=` (rightmost minimum); otherwise the
 * newest k-mer replaces the current one iff its min_hash is `<=`; the current minimum is emitted
 * iff its position is greater than the last emitted position and its min_hash != 2^64-1.
 */
size_t mxo_sketch_stateful(const char *seq, size_t len, unsigned k, unsigned w, int variant,
                           mxo_minimizer **out)
{
    mxo_workspace ws;
    memset(&ws, 0, sizeof ws);
    size_t n = mxo_sketch_stateful_ws(seq, len, k, w, variant, out, &ws);
    mxo_workspace_free(&ws);
    return n;
}

void mxo_workspace_free(mxo_workspace *ws)
{
    free(ws->mh); free(ws->oh); free(ws->fw); free(ws->ok); free(ws->ring);
    memset(ws, 0, sizeof *ws);
}

/* the loop itself; the per-k-mer arrays live in a caller-owned workspace that grows on demand, so that a worker thread of
   the threaded driver (mx_oracle_mt.c) allocates once, not once per chunk */
size_t mxo_sketch_stateful_ws(const char *seq, size_t len, unsigned k, unsigned w, int variant,
                              mxo_minimizer **out, mxo_workspace *ws)
{
    *out = NULL;
    if (k == 0 || w == 0 || k > len || (size_t)w > len - k + 1) return 0;
    size_t n = len - k + 1;
    if (n > ws->cap) {
        free(ws->mh); free(ws->oh); free(ws->fw); free(ws->ok);
        ws->cap = n + n / 8;
        ws->mh = (uint64_t *)malloc(ws->cap * sizeof(uint64_t));
        ws->oh = (uint64_t *)malloc(ws->cap * sizeof(uint64_t));
        ws->fw = (uint8_t *)malloc(ws->cap);
        ws->ok = (uint8_t *)malloc(ws->cap);
    }
    size_t ring_n = (size_t)w + 1;
    if (ring_n > ws->ring_cap) {
        free(ws->ring);
        ws->ring_cap = ring_n;
        ws->ring = (mxo_minimizer *)malloc(ring_n * sizeof(mxo_minimizer));
    }
    uint64_t *mh = ws->mh, *oh = ws->oh;
    uint8_t *fw = ws->fw, *ok = ws->ok;
    mxo_minimizer *ring = ws->ring;
    if (!mh || !oh || !fw || !ok || !ring) { fprintf(stderr, "mx_oracle: out of memory\n"); abort(); }
    mxo_kmer_hashes(seq, len, k, variant, mh, oh, fw, ok);

    mvec res = {0, 0, 0};
    long long min_pos_prev = -1;
    const mxo_minimizer *cur = NULL;
    size_t idx = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!ok[i]) continue; /* invalid k-mers are skipped and occupy no window slot */
        mxo_minimizer *hk = &ring[idx % ring_n];
        hk->min_hash = mh[i];
        hk->out_hash = oh[i];
        hk->pos = (uint32_t)i;
        hk->forward = fw[i];
        if (idx + 1 >= w) {
            size_t left = idx + 1 - w, right = idx + 1;
            const mxo_minimizer *mleft = &ring[left % ring_n];
            const mxo_minimizer *mright = &ring[(right - 1) % ring_n];
            if (cur == NULL || cur->pos < mleft->pos) {
                cur = mleft;
                for (size_t j = left; j < right; ++j) {
                    const mxo_minimizer *mj = &ring[j % ring_n];
                    if (mj->min_hash <= cur->min_hash) cur = mj;
                }
            } else if (mright->min_hash <= cur->min_hash) {
                cur = mright;
            }
            if ((long long)cur->pos > min_pos_prev && cur->min_hash != UINT64_MAX) {
                min_pos_prev = (long long)cur->pos;
                mvec_push(&res, *cur);
            }
        }
        ++idx;
    }
    *out = res.v;
    return res.n;
}

/* Stateless definition with a monotone deque (independent of the loop above). */
size_t mxo_sketch_stateless(const char *seq, size_t len, unsigned k, unsigned w, int variant,
                            mxo_minimizer **out)
{
    *out = NULL;
    if (k == 0 || w == 0 || k > len || (size_t)w > len - k + 1) return 0;
    size_t n = len - k + 1;
    uint64_t *mh = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint64_t *oh = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint8_t *fw = (uint8_t *)malloc(n);
    uint8_t *ok = (uint8_t *)malloc(n);
    size_t nv = mxo_kmer_hashes(seq, len, k, variant, mh, oh, fw, ok);
    /* compact valid k-mer positions: v_0 < v_1 < ... */
    size_t *vpos = (size_t *)malloc((nv ? nv : 1) * sizeof(size_t));
    size_t t = 0;
    for (size_t i = 0; i < n; ++i) if (ok[i]) vpos[t++] = i;
    /* deque of indices into vpos; hashes strictly increasing front->back, so among equal hashes
       only the newest (rightmost) survives and the front is the rightmost arg-min of the window */
    size_t *dq = (size_t *)malloc((nv ? nv : 1) * sizeof(size_t));
    size_t head = 0, tail = 0;
    mvec res = {0, 0, 0};
    long long last = -1;
    for (size_t e = 0; e < nv; ++e) {
        uint64_t h = mh[vpos[e]];
        while (tail > head && mh[vpos[dq[tail - 1]]] >= h) --tail;
        dq[tail++] = e;
        if (e + 1 >= w) {
            size_t lo = e + 1 - w;
            while (dq[head] < lo) ++head;
            size_t a = vpos[dq[head]];
            if ((long long)a > last && mh[a] != UINT64_MAX) {
                mxo_minimizer m;
                m.min_hash = mh[a]; m.out_hash = oh[a]; m.pos = (uint32_t)a; m.forward = fw[a];
                mvec_push(&res, m);
                last = (long long)a;
            }
        }
    }
    free(dq); free(vpos);
    free(mh); free(oh); free(fw); free(ok);
    *out = res.v;
    return res.n;
}

void mxo_free(void *p) { free(p); }

/* ---- FASTA -> TSV driver (record id = first whitespace-delimited token after '>') ---- */
typedef struct {
    char *buf;
    size_t n, cap;
} sbuf;

static void sbuf_add(sbuf *s, const char *p, size_t n)
{
    if (s->n + n + 1 > s->cap) {
        while (s->n + n + 1 > s->cap) s->cap = s->cap ? s->cap * 2 : 4096;
        s->buf = (char *)realloc(s->buf, s->cap);
        if (!s->buf) { fprintf(stderr, "mx_oracle: out of memory\n"); abort(); }
    }
    memcpy(s->buf + s->n, p, n);
    s->n += n;
    s->buf[s->n] = 0;
}

static void emit_record(FILE *fo, const char *id, const sbuf *seq, unsigned k, unsigned w,
                        int variant, int with_pos, int with_strand, int with_seq, uint64_t *stats)
{
    mxo_minimizer *m = NULL;
    size_t nm = mxo_sketch_stateful(seq->buf ? seq->buf : "", seq->n, k, w, variant, &m);
    fputs(id, fo);
    fputc('\t', fo);
    for (size_t i = 0; i < nm; ++i) {
        if (i) fputc(' ', fo);
        fprintf(fo, "%llu", (unsigned long long)m[i].out_hash);
        if (with_pos) fprintf(fo, ":%u", m[i].pos);
        if (with_strand) fprintf(fo, ":%c", m[i].forward ? '+' : '-');
        if (with_seq) { fputc(':', fo); fwrite(seq->buf + m[i].pos, 1, k, fo); }
    }
    fputc('\n', fo);
    if (stats) { stats[0] += 1; stats[1] += seq->n; stats[2] += nm; }
    free(m);
}

int mxo_sketch_fasta_to_tsv(const char *path, const char *out_path, unsigned k, unsigned w,
                            int variant, int with_pos, int with_strand, int with_seq,
                            uint64_t *stats)
{
    FILE *fi = fopen(path, "rb");
    if (!fi) { fprintf(stderr, "mx_oracle: cannot open %s\n", path); return -1; }
    FILE *fo = (strcmp(out_path, "-") == 0) ? stdout : fopen(out_path, "wb");
    if (!fo) { fclose(fi); fprintf(stderr, "mx_oracle: cannot open %s\n", out_path); return -1; }
    if (stats) stats[0] = stats[1] = stats[2] = 0;
    char *line = NULL;
    size_t lcap = 0;
    ssize_t ll;
    sbuf seq = {0, 0, 0};
    char *id = NULL;
    while ((ll = getline(&line, &lcap, fi)) >= 0) {
        while (ll > 0 && (line[ll - 1] == '\n' || line[ll - 1] == '\r')) line[--ll] = 0;
        if (ll > 0 && line[0] == '>') {
            if (id) emit_record(fo, id, &seq, k, w, variant, with_pos, with_strand, with_seq, stats);
            free(id);
            size_t e = 1;
            while (line[e] && line[e] != ' ' && line[e] != '\t') ++e;
            id = (char *)malloc(e);
            memcpy(id, line + 1, e - 1);
            id[e - 1] = 0;
            seq.n = 0;
            if (seq.buf) seq.buf[0] = 0;
        } else if (id && ll > 0) {
            sbuf_add(&seq, line, (size_t)ll);
        }
    }
    if (id) emit_record(fo, id, &seq, k, w, variant, with_pos, with_strand, with_seq, stats);
    free(id);
    free(seq.buf);
    free(line);
    fclose(fi);
    if (fo != stdout) fclose(fo); else fflush(fo);
    return 0;
}

#ifdef MXO_MAIN
/* mx_oracle -k K -w W [--variant v1|v2] [--pos] [--strand] [--seq] [-o out.tsv] in.fa */
int main(int argc, char **argv)
{
    unsigned k = 32, w = 1000;
    int variant = MXO_VARIANT_V2_SUM, with_pos = 0, with_strand = 0, with_seq = 0;
    const char *in = NULL, *outp = "-";
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-k") && i + 1 < argc) k = (unsigned)atoi(argv[++i]);
        else if (!strcmp(argv[i], "-w") && i + 1 < argc) w = (unsigned)atoi(argv[++i]);
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) outp = argv[++i];
        else if (!strcmp(argv[i], "--variant") && i + 1 < argc) {
            ++i;
            variant = (!strcmp(argv[i], "v1") || !strcmp(argv[i], "min")) ? MXO_VARIANT_V1_MIN
                                                                           : MXO_VARIANT_V2_SUM;
        } else if (!strcmp(argv[i], "--pos")) with_pos = 1;
        else if (!strcmp(argv[i], "--strand")) with_strand = 1;
        else if (!strcmp(argv[i], "--seq")) with_seq = 1;
        else if (!strcmp(argv[i], "--long") || !strcmp(argv[i], "--id")) { /* accepted, no-op */ }
        else if (!strcmp(argv[i], "-t") && i + 1 < argc) ++i;
        else if (argv[i][0] != '-') in = argv[i];
        else { fprintf(stderr, "mx_oracle: unknown option %s\n", argv[i]); return 2; }
    }
    if (!in) { fprintf(stderr, "usage: mx_oracle -k K -w W [--variant v1|v2] [--pos] [--strand] [--seq] [-o out] in.fa\n"); return 2; }
    return mxo_sketch_fasta_to_tsv(in, outp, k, w, variant, with_pos, with_strand, with_seq, NULL) ? 1 : 0;
}
#endif
