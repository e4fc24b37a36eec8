// api.cpp -- extern "C" entry points of libntjoin_mx.so (declared in include/ntjoin_mx.h).
#include <algorithm>
#include <chrono>
#include <new>
#include <thread>

#include <sys/mman.h>

#include "mxg_internal.h"

using namespace mxg;

static thread_local std::string g_create_err;

extern "C" {

int mxg_abi_version(void) { return MXG_ABI_VERSION; }

const char *mxg_last_error(const mxg_handle *h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int mxg_create(const mxg_config *cfg, mxg_handle **out)
{
    if (out) *out = nullptr;
    if (!cfg || !out) {
        g_create_err = "mxg_create: null argument";
        return MXG_EINVAL;
    }
    if (cfg->struct_size != sizeof(mxg_config)) {
        g_create_err = "mxg_create: mxg_config.struct_size does not match this library (ABI mismatch)";
        return MXG_EINVAL;
    }
    if (cfg->k < 1 || cfg->k > 1024 || cfg->w < 1 || cfg->w > (1u << 24) || cfg->variant > MXG_VARIANT_V1_MIN) {
        g_create_err = "mxg_create: need 1 <= k <= 1024, 1 <= w <= 2^24, variant in {0,1}";
        return MXG_EINVAL;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_err = std::string("mxg_create: no usable HIP device (") + hipGetErrorString(e) +
                       "); this engine has no CPU fallback";
        return MXG_EDEVICE;
    }
    int dev = cfg->device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= ndev) {
        g_create_err = "mxg_create: device ordinal out of range";
        return MXG_EINVAL;
    }
    mxg_handle *h = new (std::nothrow) mxg_handle();
    if (!h) {
        g_create_err = "mxg_create: out of memory";
        return MXG_ENOMEM;
    }
    h->cfg = *cfg;
    h->device = dev;
    auto fail = [&](hipError_t er, const char *what) {
        g_create_err = std::string("mxg_create: ") + what + ": " + hipGetErrorString(er);
        mxg_destroy(h);
        return MXG_EDEVICE;
    };
    if ((e = hipSetDevice(dev)) != hipSuccess) return fail(e, "hipSetDevice");
    if (cfg->stream) {
        h->stream = (hipStream_t)cfg->stream;
    } else {
        if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess)
            return fail(e, "hipStreamCreate");
        h->own_stream = true;
    }
    if (cfg->stream) {
        // HIP multiplexes its streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and the caller's
        // stream and ours may land on the same one, which serialises the two assemblies' kernels (measured: 0.41 ms
        // instead of 0.35 ms per step).  Streams of another priority class get queues of their own.
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        if ((e = hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, prio_hi)) != hipSuccess) return fail(e, "hipStreamCreate");
    } else if ((e = hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking)) != hipSuccess) {
        return fail(e, "hipStreamCreate");
    }
    if ((e = hipEventCreate(&h->ev0)) != hipSuccess) return fail(e, "hipEventCreate");
    if ((e = hipEventCreate(&h->ev1)) != hipSuccess) return fail(e, "hipEventCreate");
    make_hash_tab(cfg->k, &h->tab);
    *out = h;
    return MXG_OK;
}

void mxg_destroy(mxg_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->stream2) {
        (void)hipStreamSynchronize(h->stream2);
        (void)hipStreamDestroy(h->stream2);
    }
    for (hipStream_t s : h->stream_x)
        if (s) {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
    for (auto *a : h->asms) delete a;
    h->asms.clear();
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    for (hipEvent_t e : h->ev_part)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_sync) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_bs) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_sel_done)
        if (e) (void)hipEventDestroy(e);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (hipEvent_t e : h->ev_g)
        if (e) (void)hipEventDestroy(e);
    if (h->pinned_ctrl) (void)hipHostFree(h->pinned_ctrl);
    if (h->pinned_defer) (void)hipHostFree(h->pinned_defer);
    pin_pool_release(h);
    for (auto &m : h->kept_maps) munmap(m.first, m.second);
    h->kept_maps.clear();
    if (h->pinned_gctl) (void)hipHostFree(h->pinned_gctl);
    if (h->pinned_dg) (void)hipHostFree(h->pinned_dg);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

static int new_assembly(mxg_handle *h, const char *name, double weight, Assembly **out)
{
    if (!h) return MXG_EINVAL;
    if (!name) return set_err(h, MXG_EINVAL, "assembly name is NULL");
    if (h->asms.size() >= MXG_MAX_ASSEMBLIES) return set_err(h, MXG_ELIMIT, "at most %d assemblies", MXG_MAX_ASSEMBLIES);
    for (auto *a : h->asms)
        if (a->name == name)
            return set_err(h, MXG_EINVAL, "assembly name '%s' already used (names key the graph's support lists)", name);
    Assembly *a = new (std::nothrow) Assembly();
    if (!a) return set_err(h, MXG_ENOMEM, "out of memory");
    a->name = name;
    a->weight = weight;
    *out = a;
    return MXG_OK;
}

static int commit(mxg_handle *h, Assembly *a, int rc)
{
    if (rc == MXG_OK) rc = upload_packed(h, a);  // (bases parsed on the host: resident in HBM from here on, like the other routes')
    if (rc != MXG_OK) {
        delete a;
        return rc;
    }
    h->asms.push_back(a);
    if (prewarm_assembly(h, a) != MXG_OK) h->err.clear();  // (what cannot be prepared now is reported by the sketch that needs it)
    h->graph.valid = false;
    h->pj_overflowed = false;
    h->pj_cap1_P1 = 0;
    h->pj_cap1_need = 0;
    return (int)h->asms.size() - 1;
}

int mxg_add_assembly_fasta(mxg_handle *h, const char *name, double weight, const char *fasta_path)
{
    Assembly *a;
    int rc = new_assembly(h, name, weight, &a);
    if (rc != MXG_OK) return rc;
    if (!fasta_path) return commit(h, a, set_err(h, MXG_EINVAL, "fasta_path is NULL"));
    try {
        // regular files: raw text to HBM, classified and packed there (ingest.hip); MXG_HOST_INGEST=1 keeps the host parser
        rc = knob_set(h, "MXG_HOST_INGEST") ? 1 : load_fasta_device(h, a, fasta_path, host_threads(h));
        if (rc == 1) {
            Assembly *fresh;
            delete a;
            if ((rc = new_assembly(h, name, weight, &fresh)) != MXG_OK) return rc;
            a = fresh;
            rc = load_fasta(h, a, fasta_path);
        }
    } catch (const std::bad_alloc &) {
        rc = set_err(h, MXG_ENOMEM, "out of host memory reading '%s'", fasta_path);
    }
    return commit(h, a, rc);
}

int mxg_add_assembly_fasta_shard(mxg_handle *h, const char *name, double weight, const char *fasta_path,
                                 uint32_t shard, uint32_t n_shards)
{
    Assembly *a;
    int rc = new_assembly(h, name, weight, &a);
    if (rc != MXG_OK) return rc;
    if (!fasta_path || n_shards == 0 || shard >= n_shards)
        return commit(h, a, set_err(h, MXG_EINVAL, "need fasta_path and shard < n_shards"));
    try {
        rc = load_fasta(h, a, fasta_path, shard, n_shards);
    } catch (const std::bad_alloc &) {
        rc = set_err(h, MXG_ENOMEM, "out of host memory reading '%s'", fasta_path);
    }
    return commit(h, a, rc);
}

int mxg_add_assembly_fasta_split(mxg_handle *h, const char *name, double weight, const char *fasta_path,
                                 uint32_t shard, uint32_t n_shards)
{
    Assembly *a;
    int rc = new_assembly(h, name, weight, &a);
    if (rc != MXG_OK) return rc;
    if (!fasta_path || n_shards == 0 || shard >= n_shards)
        return commit(h, a, set_err(h, MXG_EINVAL, "need fasta_path and shard < n_shards"));
    try {
        rc = load_fasta(h, a, fasta_path, shard, n_shards, true);
    } catch (const std::bad_alloc &) {
        rc = set_err(h, MXG_ENOMEM, "out of host memory reading '%s'", fasta_path);
    }
    return commit(h, a, rc);
}

int mxg_assembly_continues(const mxg_handle *h, int assembly)
{
    if (!h || assembly < 0 || (size_t)assembly >= h->asms.size()) return MXG_EINVAL;
    return h->asms[assembly]->split_first_cont ? 1 : 0;
}

int mxg_shard_range(const uint64_t *lengths, uint64_t n_records, uint32_t shard, uint32_t n_shards, uint64_t *lo,
                    uint64_t *hi)
{
    if ((!lengths && n_records) || !lo || !hi || n_shards == 0 || shard >= n_shards) return MXG_EINVAL;
    shard_range(lengths, n_records, shard, n_shards, lo, hi);
    return MXG_OK;
}

int mxg_assembly_shard(const mxg_handle *h, int assembly, uint64_t *lo, uint64_t *hi)
{
    if (!h || !lo || !hi || assembly < 0 || (size_t)assembly >= h->asms.size()) return MXG_EINVAL;
    const Assembly *a = h->asms[assembly];
    *lo = std::min<uint64_t>(a->shard_lo, a->recs.size());
    *hi = std::min<uint64_t>(a->shard_hi, a->recs.size());
    return MXG_OK;
}

int mxg_add_assembly_buffers(mxg_handle *h, const char *name, double weight, const uint8_t *ascii,
                             const uint64_t *offsets, const char *const *ids, uint64_t n_records)
{
    Assembly *a;
    int rc = new_assembly(h, name, weight, &a);
    if (rc != MXG_OK) return rc;
    if ((!ascii || !offsets) && n_records) return commit(h, a, set_err(h, MXG_EINVAL, "ascii/offsets is NULL"));
    try {
        rc = load_buffers(h, a, ascii, offsets, ids, n_records);
    } catch (const std::bad_alloc &) {
        rc = set_err(h, MXG_ENOMEM, "out of host memory");
    }
    return commit(h, a, rc);
}

int mxg_add_assembly_packed_device(mxg_handle *h, const char *name, double weight, const void *d_packed,
                                   const uint64_t *rec_start, const uint64_t *rec_len,
                                   const char *const *ids, uint64_t n_records)
{
    Assembly *a;
    int rc = new_assembly(h, name, weight, &a);
    if (rc != MXG_OK) return rc;
    if (!d_packed || !rec_start || !rec_len) return commit(h, a, set_err(h, MXG_EINVAL, "null argument"));
    uint64_t end = 0;
    for (uint64_t r = 0; r < n_records && rc == MXG_OK; ++r) {
        if (rec_start[r] & 15) rc = set_err(h, MXG_EINVAL, "rec_start[%llu] is not a multiple of 16", (unsigned long long)r);
        if (rec_len[r] >= (1ull << 32)) rc = set_err(h, MXG_ELIMIT, "record %llu too long", (unsigned long long)r);
        Record rec;
        rec.id = (ids && ids[r]) ? std::string(ids[r]) : std::to_string(r);
        rec.len = rec_len[r];
        rec.base_off = rec_start[r];
        a->recs.push_back(rec);
        end = std::max(end, rec_start[r] + rec_len[r]);
    }
    if (rc == MXG_OK) {
        a->d_packed = static_cast<const uint32_t *>(d_packed);
        a->packed_words = (end + 15) / 16 + 1;
        build_runs_from_lengths(h, a);
    }
    return commit(h, a, rc);
}

int mxg_plan_split(const uint64_t *lengths, uint64_t n_records, uint32_t shard, uint32_t n_shards, uint32_t k, uint32_t w,
                   uint64_t *piece_lo, uint64_t *piece_hi, uint8_t *piece_drop)
{
    if ((!lengths && n_records) || !piece_lo || !piece_hi || !piece_drop || n_shards == 0 || shard >= n_shards || k == 0 || w == 0)
        return MXG_EINVAL;
    unsigned __int128 total = 0;
    for (uint64_t r = 0; r < n_records; ++r) total += lengths[r];
    const uint64_t cut_lo = (uint64_t)(total * shard / n_shards), cut_hi = (uint64_t)(total * (shard + 1) / n_shards);
    uint64_t cum = 0;
    for (uint64_t r = 0; r < n_records; ++r) {
        const uint64_t len = lengths[r], r0 = cum, r1 = cum + len;
        cum = r1;
        piece_lo[r] = piece_hi[r] = 0;
        piece_drop[r] = 0;
        if (len == 0 || r1 <= cut_lo || r0 >= cut_hi) continue;
        // own k-mer starts [P_lo, P_hi): the same rule as the FASTA route's plan_pieces (host_io.cpp), every k-mer valid
        const uint64_t P_lo = std::max(cut_lo, r0) - r0, P_hi = std::min(cut_hi, r1) - r0;
        piece_hi[r] = P_hi == len ? len : std::min<uint64_t>(len, P_hi + k - 1);
        if (P_lo > 0) {
            piece_drop[r] = 2;  // the record began on an earlier shard (even when the halo reaches back to its first base)
            const uint64_t nk = len >= k ? len - k + 1 : 0, t_first = std::min(P_lo, nk);  // valid k-mers that start before P_lo
            if (t_first >= w) {
                piece_lo[r] = t_first - w;
                piece_drop[r] = 3;
            }
        }
        if (piece_hi[r] <= piece_lo[r]) {
            piece_hi[r] = piece_lo[r] = 0;
            piece_drop[r] = 0;
        }
    }
    return MXG_OK;
}

int mxg_add_assembly_packed_device_pieces(mxg_handle *h, const char *name, double weight, const void *d_packed,
                                          const uint64_t *rec_start, const uint64_t *rec_len, const uint64_t *piece_lo,
                                          const uint64_t *piece_hi, const uint8_t *piece_drop, const char *const *ids,
                                          uint64_t n_records)
{
    Assembly *a;
    int rc = new_assembly(h, name, weight, &a);
    if (rc != MXG_OK) return rc;
    if (!d_packed || !rec_start || !rec_len || !piece_lo || !piece_hi || !piece_drop)
        return commit(h, a, set_err(h, MXG_EINVAL, "null argument"));
    const uint32_t k = h->cfg.k, w = h->cfg.w;
    uint64_t end = 0, lo_rec = n_records, hi_rec = 0;
    for (uint64_t r = 0; r < n_records && rc == MXG_OK; ++r) {
        Record rec;
        rec.id = (ids && ids[r]) ? std::string(ids[r]) : std::to_string(r);
        rec.len = rec_len[r];
        if (rec_len[r] >= (1ull << 32)) rc = set_err(h, MXG_ELIMIT, "record %llu too long", (unsigned long long)r);
        const uint64_t lo = piece_lo[r], hi = piece_hi[r];
        if (hi > lo) {
            if ((rec_start[r] & 15) || hi > rec_len[r]) rc = set_err(h, MXG_EINVAL, "bad piece of record %llu", (unsigned long long)r);
            // base b of the record (lo <= b < hi) sits at packed index rec_start + (b - (lo & ~15)); offsets stay "of base 0"
            rec.base_off = rec_start[r] - (lo & ~uint64_t(15));
            a->recs.push_back(rec);
            lo_rec = std::min(lo_rec, r);
            hi_rec = r + 1;
            end = std::max(end, rec_start[r] + (hi - (lo & ~uint64_t(15))));
            a->total_bases += hi - lo;
            if (hi - lo >= k && hi - lo - k + 1 >= w) {
                Run run;
                run.base_off = rec.base_off + lo;
                run.n_kmers = (uint32_t)(hi - lo - k + 1);
                run.contig = (uint32_t)a->ctg_rec.size();
                run.kidx0 = 0;
                run.pos0 = (uint32_t)lo;
                a->ctg_rec.push_back((uint32_t)r);
                a->ctg_nk.push_back(run.n_kmers);
                a->ctg_run0.push_back((uint32_t)a->runs.size());
                a->ctg_drop.push_back((piece_drop[r] & 1) ? 1 : 0);
                a->any_drop = a->any_drop || (piece_drop[r] & 1);
                a->runs.push_back(run);
                a->total_kmers += run.n_kmers;
            }
        } else {
            a->recs.push_back(rec);  // registered (global record index, id, length), held by another shard
        }
    }
    if (rc == MXG_OK) {
        a->ctg_run0.push_back((uint32_t)a->runs.size());
        a->d_packed = static_cast<const uint32_t *>(d_packed);
        a->packed_words = (end + 15) / 16 + 1;
        a->has_bases = true;
        a->shard_lo = std::min(lo_rec, hi_rec);
        a->shard_hi = hi_rec;
        // (the FASTA split route's rule, host_io.cpp: pieces[lo].cont = the shard's range starts inside the record)
        a->split_first_cont = hi_rec > lo_rec && (piece_lo[lo_rec] > 0 || (piece_drop[lo_rec] & 2));
    }
    return commit(h, a, rc);
}

static int adopt_host_sketch(mxg_handle *h, Assembly *a, const std::vector<uint64_t> &hash,
                             const std::vector<uint32_t> &pos, const std::vector<uint32_t> &rec)
{
    const uint64_t n = hash.size();
    for (uint64_t i = 0; i < n; ++i)
        if (rec[i] >= a->recs.size()) return set_err(h, MXG_EINVAL, "record index out of range at minimizer %llu", (unsigned long long)i);
    MXG_HIP(h, hipSetDevice(h->device));
    MXG_HIP(h, a->d_hash.ensure(std::max<uint64_t>(n * 8, 16)));
    MXG_HIP(h, a->d_pos.ensure(std::max<uint64_t>(n * 4, 16)));
    MXG_HIP(h, a->d_rec.ensure(std::max<uint64_t>(n * 4, 16)));
    MXG_HIP(h, a->d_fwd.ensure(std::max<uint64_t>(n, 16)));
    if (n) {
        MXG_HIP(h, hipMemcpyAsync(a->d_hash.p, hash.data(), n * 8, hipMemcpyHostToDevice, h->stream));
        MXG_HIP(h, hipMemcpyAsync(a->d_pos.p, pos.data(), n * 4, hipMemcpyHostToDevice, h->stream));
        MXG_HIP(h, hipMemcpyAsync(a->d_rec.p, rec.data(), n * 4, hipMemcpyHostToDevice, h->stream));
        MXG_HIP(h, hipMemsetAsync(a->d_fwd.p, 1, n, h->stream));
        MXG_HIP(h, hipStreamSynchronize(h->stream));
    }
    a->n_mx = n;
    a->h_hash = hash;
    a->h_pos = pos;
    a->h_rec = rec;
    a->h_fwd.assign(n, 1);
    build_rec_first(a);
    a->host_valid = true;
    a->fwd_valid = true;
    a->has_sketch = true;
    return MXG_OK;
}

int mxg_add_assembly_tsv(mxg_handle *h, const char *name, double weight, const char *tsv_path)
{
    Assembly *a;
    int rc = new_assembly(h, name, weight, &a);
    if (rc != MXG_OK) return rc;
    if (!tsv_path) return commit(h, a, set_err(h, MXG_EINVAL, "tsv_path is NULL"));
    std::vector<uint64_t> hash;
    std::vector<uint32_t> pos, rec;
    try {
        rc = load_tsv(h, a, tsv_path, hash, pos, rec);
        if (rc == MXG_OK) rc = adopt_host_sketch(h, a, hash, pos, rec);
    } catch (const std::bad_alloc &) {
        rc = set_err(h, MXG_ENOMEM, "out of host memory reading '%s'", tsv_path);
    }
    return commit(h, a, rc);
}

int mxg_add_assembly_bin(mxg_handle *h, const char *name, double weight, const char *bin_path)
{
    Assembly *a;
    int rc = new_assembly(h, name, weight, &a);
    if (rc != MXG_OK) return rc;
    if (!bin_path) return commit(h, a, set_err(h, MXG_EINVAL, "bin_path is NULL"));
    std::vector<uint64_t> hash;
    std::vector<uint32_t> pos, rec;
    try {
        rc = load_sketch_bin(h, a, bin_path, hash, pos, rec);
        if (rc == MXG_OK) rc = adopt_host_sketch(h, a, hash, pos, rec);
    } catch (const std::bad_alloc &) {
        rc = set_err(h, MXG_ENOMEM, "out of host memory reading '%s'", bin_path);
    }
    return commit(h, a, rc);
}

int mxg_add_assembly_minimizers(mxg_handle *h, const char *name, double weight, const uint64_t *out_hash,
                                const uint32_t *pos, const uint32_t *record, uint64_t n,
                                const char *const *record_ids, uint64_t n_records)
{
    Assembly *a;
    int rc = new_assembly(h, name, weight, &a);
    if (rc != MXG_OK) return rc;
    if (n && (!out_hash || !pos || !record)) return commit(h, a, set_err(h, MXG_EINVAL, "null argument"));
    for (uint64_t r = 0; r < n_records; ++r) {
        Record rec;
        rec.id = (record_ids && record_ids[r]) ? std::string(record_ids[r]) : std::to_string(r);
        a->recs.push_back(rec);
    }
    std::vector<uint64_t> hv(out_hash, out_hash + n);
    std::vector<uint32_t> pv(pos, pos + n), rv(record, record + n);
    for (uint64_t i = 1; i < n && rc == MXG_OK; ++i)
        if (rv[i] < rv[i - 1]) rc = set_err(h, MXG_EINVAL, "minimizers must be grouped by non-decreasing record index");
    if (rc == MXG_OK) rc = adopt_host_sketch(h, a, hv, pv, rv);
    return commit(h, a, rc);
}

int mxg_num_assemblies(const mxg_handle *h) { return h ? (int)h->asms.size() : MXG_EINVAL; }

static Assembly *get_asm(mxg_handle *h, int assembly)
{
    if (!h || assembly < 0 || (size_t)assembly >= h->asms.size()) {
        if (h) set_err(h, MXG_EINVAL, "assembly index %d out of range", assembly);
        return nullptr;
    }
    return h->asms[assembly];
}

const char *mxg_assembly_name(const mxg_handle *h, int assembly)
{
    if (!h || assembly < 0 || (size_t)assembly >= h->asms.size()) return nullptr;
    return h->asms[assembly]->name.c_str();
}

const char *mxg_record_id(const mxg_handle *h, int assembly, uint64_t record)
{
    if (!h || assembly < 0 || (size_t)assembly >= h->asms.size()) return nullptr;
    const Assembly *a = h->asms[assembly];
    if (record >= a->recs.size()) return nullptr;
    return a->recs[record].id.c_str();
}

uint64_t mxg_record_length(const mxg_handle *h, int assembly, uint64_t record)
{
    if (!h || assembly < 0 || (size_t)assembly >= h->asms.size()) return 0;
    const Assembly *a = h->asms[assembly];
    return record < a->recs.size() ? a->recs[record].len : 0;
}

uint64_t mxg_num_records(const mxg_handle *h, int assembly)
{
    if (!h || assembly < 0 || (size_t)assembly >= h->asms.size()) return 0;
    return h->asms[assembly]->recs.size();
}

double mxg_assembly_weight(const mxg_handle *h, int assembly)
{
    if (!h || assembly < 0 || (size_t)assembly >= h->asms.size()) return 0.0;
    return h->asms[assembly]->weight;
}

int mxg_sketch(mxg_handle *h, int assembly)
{
    if (!h) return MXG_EINVAL;
    try {
        if (assembly >= 0) {
            Assembly *a = get_asm(h, assembly);
            if (!a) return MXG_EINVAL;
            return sketch_assembly(h, a);
        }
        std::vector<Assembly *> todo;
        for (auto *a : h->asms)
            if (a->has_bases && (assembly == MXG_SKETCH_ALL || !a->has_sketch)) todo.push_back(a);
        if (!todo.empty()) return sketch_assemblies(h, todo.data(), todo.size());
    } catch (const std::bad_alloc &) {
        return set_err(h, MXG_ENOMEM, "out of host memory in mxg_sketch");
    }
    return MXG_OK;
}

int mxg_sketch_pack(mxg_handle *h, void *d_slot, uint64_t head_bytes, const uint64_t *caps)
{
    if (!h || !d_slot || !caps || head_bytes < 8 * h->asms.size()) return MXG_EINVAL;
    if (!h->pend_list.empty()) return set_err(h, MXG_EINVAL, "mxg_sketch_pack: the previous call was not finished (mxg_sketch_finish)");
    try {
        std::vector<Assembly *> todo;
        for (auto *a : h->asms) {
            if (!a->has_bases) return set_err(h, MXG_EINVAL, "mxg_sketch_pack: assembly '%s' has no bases", a->name.c_str());
            todo.push_back(a);
        }
        if (todo.empty()) return set_err(h, MXG_EINVAL, "mxg_sketch_pack: no assemblies");
        XchgPackReq xp{d_slot, head_bytes, caps};
        return sketch_assemblies(h, todo.data(), todo.size(), false, &xp);
    } catch (const std::bad_alloc &) {
        return set_err(h, MXG_ENOMEM, "out of host memory in mxg_sketch_pack");
    }
}

int mxg_sketch_pack_parts(mxg_handle *h, void *const *d_parts, const uint64_t *caps, const uint64_t *rcaps)
{
    if (!h || !d_parts || !caps || !rcaps) return MXG_EINVAL;
    if (!h->pend_list.empty()) return set_err(h, MXG_EINVAL, "mxg_sketch_pack_parts: the previous call was not finished (mxg_sketch_finish)");
    try {
        std::vector<Assembly *> todo;
        for (auto *a : h->asms) {
            if (!a->has_bases) return set_err(h, MXG_EINVAL, "mxg_sketch_pack_parts: assembly '%s' has no bases", a->name.c_str());
            if (!d_parts[todo.size()]) return MXG_EINVAL;
            todo.push_back(a);
        }
        if (todo.empty()) return set_err(h, MXG_EINVAL, "mxg_sketch_pack_parts: no assemblies");
        XchgPackReq xp{nullptr, 0, caps, d_parts, rcaps};
        return sketch_assemblies(h, todo.data(), todo.size(), false, &xp);
    } catch (const std::bad_alloc &) {
        return set_err(h, MXG_ENOMEM, "out of host memory in mxg_sketch_pack_parts");
    }
}

int mxg_sketch_dg_pack_slots(mxg_handle *h, uint32_t world, uint32_t n_asm, const uint32_t *cap, const uint32_t *rec_offsets, void *d_send)
{
    if (!h || !cap || !rec_offsets || !d_send) return MXG_EINVAL;
    if (n_asm != h->asms.size()) return set_err(h, MXG_EINVAL, "mxg_sketch_dg_pack_slots: the handle has %zu assemblies", h->asms.size());
    if (!h->pend_list.empty()) return set_err(h, MXG_EINVAL, "mxg_sketch_dg_pack_slots: the previous call was not finished (mxg_sketch_finish)");
    try {
        std::vector<Assembly *> todo;
        for (auto *a : h->asms) {
            if (!a->has_bases) return set_err(h, MXG_EINVAL, "mxg_sketch_dg_pack_slots: assembly '%s' has no bases", a->name.c_str());
            todo.push_back(a);
        }
        if (todo.empty()) return set_err(h, MXG_EINVAL, "mxg_sketch_dg_pack_slots: no assemblies");
        DgPackReq rq{world, n_asm, cap, rec_offsets, d_send};
        XchgPackReq xp{nullptr, 0, nullptr, nullptr, nullptr, &rq};
        return sketch_assemblies(h, todo.data(), todo.size(), false, &xp);
    } catch (const std::bad_alloc &) {
        return set_err(h, MXG_ENOMEM, "out of host memory in mxg_sketch_dg_pack_slots");
    }
}

int mxg_part_packed_wait(mxg_handle *h, int assembly, void *stream)
{
    if (!h || assembly < 0 || (size_t)assembly >= h->asms.size() || assembly >= MXG_MAX_ASSEMBLIES) return MXG_EINVAL;
    if (!h->ev_part[assembly]) return set_err(h, MXG_EINVAL, "mxg_part_packed_wait: no mxg_sketch_pack_parts / mxg_sketch_dg_pack_slots before");
    MXG_HIP(h, hipSetDevice(h->device));
    MXG_HIP(h, hipStreamWaitEvent(static_cast<hipStream_t>(stream), h->ev_part[assembly], 0));
    return MXG_OK;
}

int mxg_sketch_finish(mxg_handle *h)
{
    if (!h) return MXG_EINVAL;
    try {
        return sketch_finish(h);
    } catch (const std::bad_alloc &) {
        return set_err(h, MXG_ENOMEM, "out of host memory in mxg_sketch_finish");
    }
}

int mxg_sketch_graph(mxg_handle *h)
{
    if (!h) return MXG_EINVAL;
    try {
        std::vector<Assembly *> todo;
        bool all_bases = !h->asms.empty();
        for (auto *a : h->asms) {
            all_bases = all_bases && a->has_bases;
            if (a->has_bases) todo.push_back(a);
        }
        // an assembly that takes several batches (more k-mers than one launch of the slice kernel counts: 3600 Mi) cannot keep the graph
        // stage enqueued behind it -- the one-call mode would send it through the synchronous route (configs[4]: 56 ms per step instead
        // of 19).  Such handles get the two calls in one: every batch of every assembly through the streams, then the graph stage.
        bool several = false;
        for (auto *a : todo) several = several || a->total_kmers > knob_u64(h, "MXG_SEL_BATCH_KMERS", 3600ull << 20);
        if (all_bases && several) {
            int rc = sketch_assemblies(h, todo.data(), todo.size());
            return rc != MXG_OK ? rc : build_graph(h);
        }
        if (!all_bases) {  // some assembly came as a sketch (TSV, arrays): sketch what has bases, then the ordinary graph stage
            if (!todo.empty()) {
                int rc = sketch_assemblies(h, todo.data(), todo.size());
                if (rc != MXG_OK) return rc;
            }
            return build_graph(h);
        }
        return sketch_assemblies(h, todo.data(), todo.size(), true);
    } catch (const std::bad_alloc &) {
        return set_err(h, MXG_ENOMEM, "out of host memory in mxg_sketch_graph");
    }
}

int mxg_get_sketch(mxg_handle *h, int assembly, mxg_sketch_view *out)
{
    Assembly *a = get_asm(h, assembly);
    if (!a || !out) return MXG_EINVAL;
    int rc = sync_sketch_to_host(h, a);
    if (rc != MXG_OK) return rc;
    out->n = a->n_mx;
    out->out_hash = a->h_hash.data();
    out->pos = a->h_pos.data();
    out->record = a->h_rec.data();
    out->forward = a->h_fwd.data();
    out->n_records = a->recs.size();
    out->record_first = a->rec_first.data();
    return MXG_OK;
}

int mxg_get_sketch_device(mxg_handle *h, int assembly, mxg_sketch_dview *out)
{
    Assembly *a = get_asm(h, assembly);
    if (!a || !out) return MXG_EINVAL;
    if (!a->has_sketch) return set_err(h, MXG_EINVAL, "assembly '%s' has no sketch yet", a->name.c_str());
    out->n = a->n_mx;
    out->out_hash = a->d_hash.p;
    out->pos = a->d_pos.p;
    out->record = a->d_rec.p;
    out->forward = a->fwd_valid ? a->d_fwd.p : nullptr;
    return MXG_OK;
}

int mxg_compute_strands(mxg_handle *h, int assembly)
{
    Assembly *a = get_asm(h, assembly);
    if (!a) return MXG_EINVAL;
    if (!a->has_sketch) return set_err(h, MXG_EINVAL, "assembly '%s' has no sketch yet", a->name.c_str());
    return ensure_strand(h, a);
}

int mxg_pack_sketch_device(mxg_handle *h, int assembly, void *d_buf, uint64_t nmax)
{
    Assembly *a = get_asm(h, assembly);
    if (!a || !d_buf) return MXG_EINVAL;
    return pack_sketch(h, a, d_buf, nmax);
}

int mxg_set_sketch_gathered(mxg_handle *h, int assembly, const void *d_allbuf, uint32_t world, uint64_t nmax,
                            const uint64_t *counts, const uint64_t *rec_offsets)
{
    Assembly *a = get_asm(h, assembly);
    if (!a || !d_allbuf || !counts || !rec_offsets) return MXG_EINVAL;
    return unpack_gathered(h, a, d_allbuf, world, nmax, counts, rec_offsets);
}

int mxg_set_sketch_gathered_strided(mxg_handle *h, int assembly, const void *d_allbuf, uint32_t world, uint64_t stride_bytes,
                                    uint64_t nmax, const uint64_t *counts, const uint64_t *rec_offsets)
{
    Assembly *a = get_asm(h, assembly);
    if (!a || !d_allbuf || !counts || !rec_offsets || stride_bytes < 16 * nmax) return MXG_EINVAL;
    return unpack_gathered(h, a, d_allbuf, world, nmax, counts, rec_offsets, stride_bytes);
}

int mxg_xchg_pack(mxg_handle *h, void *d_slot, uint64_t head_bytes, const uint64_t *caps)
{
    if (!h || !d_slot || !caps || head_bytes < 8 * h->asms.size()) return MXG_EINVAL;
    return xchg_pack(h, d_slot, head_bytes, caps);
}

int mxg_xchg_unpack_graph(mxg_handle *h, const void *d_all, uint32_t world, uint64_t slot_bytes, uint64_t head_bytes,
                          const uint64_t *caps, const uint64_t *rec_offsets)
{
    if (!h || !d_all || !caps || !rec_offsets) return MXG_EINVAL;
    return xchg_unpack_graph(h, d_all, world, slot_bytes, head_bytes, caps, rec_offsets);
}

int mxg_xchg_unpack_graph_parts(mxg_handle *h, const void *const *d_all_parts, uint32_t world, const uint64_t *caps,
                                const uint64_t *rcaps, const uint64_t *rec_offsets)
{
    if (!h || !d_all_parts || !caps || !rcaps || !rec_offsets) return MXG_EINVAL;
    for (size_t a = 0; a < h->asms.size(); ++a)
        if (!d_all_parts[a] || caps[a] % 8 || rcaps[a] % 2) return MXG_EINVAL;  // (parts lie side by side: 8-byte columns)
    return xchg_unpack_graph(h, nullptr, world, 0, 0, caps, rec_offsets, d_all_parts, rcaps);
}

int mxg_set_sketch_device(mxg_handle *h, int assembly, const void *d_out_hash, const void *d_pos,
                          const void *d_record, const void *d_forward, uint64_t n)
{
    Assembly *a = get_asm(h, assembly);
    if (!a) return MXG_EINVAL;
    if (n && (!d_out_hash || !d_pos || !d_record)) return set_err(h, MXG_EINVAL, "null device pointer");
    MXG_HIP(h, hipSetDevice(h->device));
    // the sources must not alias this assembly's own sketch buffers (they are overwritten in place)
    const char *own[4] = {(const char *)a->d_hash.p, (const char *)a->d_pos.p, (const char *)a->d_rec.p, (const char *)a->d_fwd.p};
    const size_t own_sz[4] = {a->d_hash.bytes, a->d_pos.bytes, a->d_rec.bytes, a->d_fwd.bytes};
    const char *src[4] = {(const char *)d_out_hash, (const char *)d_pos, (const char *)d_record, (const char *)d_forward};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (src[i] && own[j] && src[i] >= own[j] && src[i] < own[j] + own_sz[j])
                return set_err(h, MXG_EINVAL, "mxg_set_sketch_device: source arrays alias the assembly's own sketch");
    MXG_HIP(h, a->d_hash.ensure(std::max<uint64_t>(n * 8, 16)));
    MXG_HIP(h, a->d_pos.ensure(std::max<uint64_t>(n * 4, 16)));
    MXG_HIP(h, a->d_rec.ensure(std::max<uint64_t>(n * 4, 16)));
    MXG_HIP(h, a->d_fwd.ensure(std::max<uint64_t>(n, 16)));
    if (n) {
        MXG_HIP(h, hipMemcpyAsync(a->d_hash.p, d_out_hash, n * 8, hipMemcpyDeviceToDevice, h->stream));
        MXG_HIP(h, hipMemcpyAsync(a->d_pos.p, d_pos, n * 4, hipMemcpyDeviceToDevice, h->stream));
        MXG_HIP(h, hipMemcpyAsync(a->d_rec.p, d_record, n * 4, hipMemcpyDeviceToDevice, h->stream));
        if (d_forward) MXG_HIP(h, hipMemcpyAsync(a->d_fwd.p, d_forward, n, hipMemcpyDeviceToDevice, h->stream));
        else MXG_HIP(h, hipMemsetAsync(a->d_fwd.p, 1, n, h->stream));
        MXG_HIP(h, hipStreamSynchronize(h->stream));  // the caller may release its arrays as soon as this returns
    }
    a->n_mx = n;
    a->has_sketch = true;
    a->fwd_valid = true;
    a->foreign_sketch = true;
    a->host_valid = false;
    a->flags_valid = false;
    h->graph.valid = false;
    return MXG_OK;
}

int mxg_write_tsv(mxg_handle *h, int assembly, const char *path, int with_pos, int with_strand, int with_seq)
{
    Assembly *a = get_asm(h, assembly);
    if (!a || !path) return MXG_EINVAL;
    try {
        return write_tsv(h, a, path, with_pos, with_strand, with_seq);
    } catch (const std::bad_alloc &) {
        return set_err(h, MXG_ENOMEM, "out of host memory in mxg_write_tsv");
    }
}

int mxg_write_sketch_bin(mxg_handle *h, int assembly, const char *path)
{
    Assembly *a = get_asm(h, assembly);
    if (!a || !path) return MXG_EINVAL;
    try {
        return write_sketch_bin(h, a, path);
    } catch (const std::bad_alloc &) {
        return set_err(h, MXG_ENOMEM, "out of host memory in mxg_write_sketch_bin");
    }
}

int mxg_build_graph(mxg_handle *h)
{
    if (!h) return MXG_EINVAL;
    try {
        return build_graph(h);
    } catch (const std::bad_alloc &) {
        return set_err(h, MXG_ENOMEM, "out of host memory in mxg_build_graph");
    }
}

int mxg_get_mx_flags(mxg_handle *h, int assembly, const uint8_t **flags, uint64_t *n)
{
    Assembly *a = get_asm(h, assembly);
    if (!a || !flags || !n) return MXG_EINVAL;
    int rc = flags_to_host(h, a);
    if (rc != MXG_OK) return rc;
    *flags = a->h_flags.data();
    *n = a->n_mx;
    return MXG_OK;
}

int mxg_get_graph(mxg_handle *h, mxg_graph_view *out)
{
    if (!h || !out) return MXG_EINVAL;
    int rc = graph_to_host(h);
    if (rc != MXG_OK) return rc;
    const Graph &g = h->graph;
    out->n_assemblies = g.n_asm;
    out->n_vertices = g.nv;
    out->vertex_hash = g.vhash.data();
    out->vertex_pos = g.vpos.data();
    out->vertex_record = g.vrec.data();
    out->n_edges = g.ne;
    out->edge_u = g.eu.data();
    out->edge_v = g.ev.data();
    out->edge_support = g.esup.data();
    out->edge_weight = g.ew.data();
    return MXG_OK;
}

int mxg_find_paths(mxg_handle *h, int64_t min_edge_weight, mxg_paths_view *out)
{
    if (!h || !out) return MXG_EINVAL;
    try {
        int rc = find_paths(h, min_edge_weight);
        if (rc != MXG_OK) return rc;
    } catch (const std::bad_alloc &) {
        return set_err(h, MXG_ENOMEM, "out of host memory in mxg_find_paths");
    }
    const Paths &P = h->paths;
    out->n_components = P.n_components;
    out->n_paths = P.component.size();
    out->path_first = P.first.data();
    out->path_vertex = P.vertex.data();
    out->path_component = P.component.data();
    return MXG_OK;
}

int mxg_path_segments(mxg_handle *h, int assembly, mxg_segments_view *out)
{
    if (!h || !out || assembly < 0) return MXG_EINVAL;
    try {
        int rc = path_segments(h, (uint32_t)assembly);
        if (rc != MXG_OK) return rc;
    } catch (const std::bad_alloc &) {
        return set_err(h, MXG_ENOMEM, "out of host memory in mxg_path_segments");
    }
    const Segments &S = h->segs;
    out->n_segments = S.path.size();
    out->seg_path = S.path.data();
    out->seg_record = S.record.data();
    out->seg_first = S.first.data();
    out->seg_stat = S.stat.data();
    return MXG_OK;
}

int mxg_mx_extremes(mxg_handle *h, int assembly, const uint32_t **min_pos, const uint32_t **max_pos, uint64_t *n_records)
{
    if (!h || !min_pos || !max_pos || !n_records || assembly < 0) return MXG_EINVAL;
    try {
        int rc = mx_extremes(h, (uint32_t)assembly);
        if (rc != MXG_OK) return rc;
    } catch (const std::bad_alloc &) {
        return set_err(h, MXG_ENOMEM, "out of host memory in mxg_mx_extremes");
    }
    *min_pos = h->segs.ext_min.data();
    *max_pos = h->segs.ext_max.data();
    *n_records = h->segs.ext_min.size();
    return MXG_OK;
}

// ---- distributed graph stage (dgraph.hip; the collectives are the caller's) -----------------------------------------
#define DG_TRY(expr)                                                                     \
    try {                                                                                \
        return (expr);                                                                   \
    } catch (const std::bad_alloc &) {                                                   \
        return set_err(h, MXG_ENOMEM, "out of host memory in the distributed graph stage"); \
    }

int mxg_dg_owner_counts(mxg_handle *h, uint32_t world, uint64_t *counts)
{
    if (!h || !counts) return MXG_EINVAL;
    DG_TRY(dg_owner_counts(h, world, counts))
}

int mxg_dg_pack_items(mxg_handle *h, int assembly, uint32_t world, uint32_t rec_offset, const uint64_t *starts, void *d_send)
{
    Assembly *a = get_asm(h, assembly);
    if (!a || !starts || !d_send || world == 0 || world > 64) return MXG_EINVAL;
    DG_TRY(dg_pack_items(h, a, world, rec_offset, starts, d_send))
}

int mxg_dg_set_items(mxg_handle *h, int assembly, const void *d_items, uint32_t world, const uint64_t *sec_start,
                     const uint64_t *sec_count)
{
    Assembly *a = get_asm(h, assembly);
    if (!a || !d_items || !sec_start || !sec_count) return MXG_EINVAL;
    DG_TRY(dg_set_items(h, a, d_items, world, sec_start, sec_count))
}

int mxg_dg_vertices(mxg_handle *h, void *d_n_vertices)
{
    if (!h || !d_n_vertices) return MXG_EINVAL;
    try {
        int rc = build_graph(h, GRAPH_DG_VERTICES, d_n_vertices, 0);
        if (rc == MXG_OK && h->own_stream && hipStreamSynchronize(h->stream) != hipSuccess) rc = MXG_EDEVICE;
        return rc;  // (own stream: the word feeds a collective on another stream)
    } catch (const std::bad_alloc &) {
        return set_err(h, MXG_ENOMEM, "out of host memory in mxg_dg_vertices");
    }
}

int mxg_dg_item_results(mxg_handle *h, int assembly, const void *d_gbase, uint32_t world, const uint64_t *sec_start,
                        const uint64_t *sec_count, void *d_out)
{
    Assembly *a = get_asm(h, assembly);
    if (!a || !d_gbase || !d_out || !sec_start || !sec_count) return MXG_EINVAL;
    DG_TRY(dg_item_results(h, a, d_gbase, world, sec_start, sec_count, d_out))
}

int mxg_dg_msg_counts(mxg_handle *h, uint32_t world, const void *d_ret, const void *d_bases, uint64_t *counts)
{
    if (!h || !d_ret || !d_bases || !counts || world == 0 || world > 64) return MXG_EINVAL;
    DG_TRY(dg_msg_counts(h, world, d_ret, d_bases, counts))
}

int mxg_dg_last_shared(mxg_handle *h, const void *d_ret, void *d_out)
{
    if (!h || !d_ret || !d_out) return MXG_EINVAL;
    DG_TRY(dg_last_shared(h, d_ret, d_out))
}

int mxg_dg_set_ghosts(mxg_handle *h, const void *d_all, uint32_t world, uint32_t rank)
{
    if (!h || world == 0 || world > 64) return MXG_EINVAL;
    DG_TRY(dg_set_ghosts(h, d_all, world, rank))
}

int mxg_dg_pack_msgs(mxg_handle *h, int assembly, uint32_t world, const void *d_bases, const uint64_t *starts, void *d_send)
{
    Assembly *a = get_asm(h, assembly);
    if (!a || !d_bases || !starts || !d_send || world == 0 || world > 64) return MXG_EINVAL;
    DG_TRY(dg_pack_msgs(h, a, (uint32_t)assembly, world, d_bases, starts, d_send))
}

int mxg_dg_edges(mxg_handle *h, const void *d_msgs, uint64_t n_msgs, uint64_t *n_vertices, uint64_t *n_edges)
{
    if (!h || (n_msgs && !d_msgs)) return MXG_EINVAL;
    try {
        int rc = build_graph(h, GRAPH_DG_EDGES, d_msgs, n_msgs);
        if (rc != MXG_OK) return rc;
    } catch (const std::bad_alloc &) {
        return set_err(h, MXG_ENOMEM, "out of host memory in mxg_dg_edges");
    }
    if (n_vertices) *n_vertices = h->graph.nv;
    if (n_edges) *n_edges = h->graph.ne;
    return MXG_OK;
}

int mxg_dg_pack_slots(mxg_handle *h, int assembly, uint32_t rec_offset, uint32_t world, uint32_t n_asm, const uint32_t *cap,
                      void *d_send)
{
    Assembly *a = get_asm(h, assembly);
    if (!a || !cap || !d_send) return MXG_EINVAL;
    DG_TRY(dg_pack_slots(h, a, (uint32_t)assembly, rec_offset, world, n_asm, cap, d_send))
}

int mxg_dg_owner_slots(mxg_handle *h, uint32_t world, uint32_t n_asm, const uint32_t *cap, const void *d_recv, void *d_n_vertices)
{
    if (!h || !cap || !d_recv || !d_n_vertices) return MXG_EINVAL;
    DG_TRY(dg_owner_slots(h, world, n_asm, cap, d_recv, d_n_vertices))
}

int mxg_dg_slot_results(mxg_handle *h, uint32_t world, uint32_t n_asm, const uint32_t *cap, const void *d_recv,
                        const void *d_gbase, void *d_out)
{
    if (!h || !cap || !d_recv || !d_gbase || !d_out) return MXG_EINVAL;
    DG_TRY(dg_slot_results(h, world, n_asm, cap, d_recv, d_gbase, d_out))
}

int mxg_dg_pack_msg_slots(mxg_handle *h, uint32_t world, uint32_t max_msgs, const void *d_ret, const void *d_bases, void *d_send)
{
    if (!h || !d_ret || !d_bases || !d_send || world == 0 || world > 64) return MXG_EINVAL;
    DG_TRY(dg_pack_msg_slots(h, world, max_msgs, d_ret, d_bases, d_send))
}

int mxg_dg_edges_slots(mxg_handle *h, const void *d_recv, uint32_t world, uint32_t max_msgs, uint64_t *n_vertices,
                       uint64_t *n_edges, uint32_t *overflow)
{
    if (!h || !d_recv || world == 0 || world > 64) return MXG_EINVAL;
    DG_TRY(dg_edges_slots(h, d_recv, world, max_msgs, n_vertices, n_edges, overflow))
}

int mxg_write_dot(mxg_handle *h, const char *path)
{
    if (!h || !path) return MXG_EINVAL;
    try {
        int rc = graph_to_host(h);
        if (rc != MXG_OK) return rc;
        return write_dot(h, path);
    } catch (const std::bad_alloc &) {
        return set_err(h, MXG_ENOMEM, "out of host memory in mxg_write_dot");
    }
}

int mxg_dot_part_format(mxg_handle *h, uint32_t part, uint32_t n_parts, uint64_t bytes[2])
{
    if (!h || !bytes) return MXG_EINVAL;
    try {
        int rc = graph_to_host(h);
        if (rc != MXG_OK) return rc;
        return dot_part_format(h, part, n_parts, bytes);
    } catch (const std::bad_alloc &) {
        return set_err(h, MXG_ENOMEM, "out of host memory in mxg_dot_part_format");
    }
}

int mxg_dot_part_write(mxg_handle *h, const char *path, uint64_t v_off, uint64_t e_off, int first, int last)
{
    if (!h || !path) return MXG_EINVAL;
    return dot_part_write(h, path, v_off, e_off, first, last);
}

int mxg_write_outputs(mxg_handle *h, const char *dot_path, const char *const *tsv_paths, int with_pos, int with_strand, int with_seq)
{
    if (!h || !dot_path || !tsv_paths) return MXG_EINVAL;
    try {
        const bool dbg_io = getenv("MXG_DEBUG_IO") != nullptr;
        auto now_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = now_s();
        int rc = graph_to_host(h);  // (device -> host copies on the handle's stream: before the TSV kernels use it)
        if (rc != MXG_OK) return rc;
        const double t1 = now_s();
        int rc_dot = MXG_OK;
        std::string dot_msg;  // the writer thread's error text: the handle's message is this thread's alone until the join
        struct Joiner {  // (whatever happens to the TSVs, the writer thread is waited for)
            std::thread t;
            ~Joiner()
            {
                if (t.joinable()) t.join();
            }
        } dot{std::thread([&]() {
            tl_err_sink = &dot_msg;
            try {
                rc_dot = write_dot(h, dot_path);  // host arrays only from here on
            } catch (const std::exception &) {
                rc_dot = MXG_ENOMEM;
                dot_msg.clear();
            }
            tl_err_sink = nullptr;
        })};
        int rc_tsv = MXG_OK;
        for (size_t a = 0; a < h->asms.size() && rc_tsv == MXG_OK; ++a)
            if (tsv_paths[a]) rc_tsv = mxg_write_tsv(h, (int)a, tsv_paths[a], with_pos, with_strand, with_seq);
        const double t2 = now_s();
        // MXG_FLAG_ONE_SHOT: nothing on the device is needed any more (the graph is on the host, the TSVs are written): the big
        // buffers go back to the driver while the .mx.dot is still being formatted and written, instead of all at once when the
        // process ends (0.15 s behind main() at 3 Gbp + 3 Gbp: the parent waits for it)
        double t_rel = 0;
        if ((h->cfg.flags & MXG_FLAG_ONE_SHOT) && rc_tsv == MXG_OK) {
            (void)hipSetDevice(h->device);
            (void)hipDeviceSynchronize();
            for (Assembly *a : h->asms) {
                a->d_text.release();
                a->d_packed_own.release();
                a->d_bs_out.release();
                a->d_bs_tail.release();
                a->d_ing_items.release(); a->d_ing_cnt.release(); a->d_ing_sub.release(); a->d_ing_pbase.release();
                a->d_packed = nullptr;
                a->has_bases = false;
                a->text_on_device = false;
            }
            for (auto &set : h->scratch)
                for (auto &b : set) b.release();
            t_rel = now_s() - t2;
        }
        dot.t.join();
        if (dbg_io && t_rel > 0) fprintf(stderr, "[mxg] write_outputs: device buffers released in %.3f s (beside the .mx.dot writer)\n", t_rel);
        if (dbg_io)
            fprintf(stderr, "[mxg] write_outputs: graph to host %.3f s, TSVs %.3f s, then %.3f s more for the .mx.dot\n", t1 - t0, t2 - t1,
                    now_s() - t2);
        if (rc_tsv != MXG_OK) return rc_tsv;  // (its message is the handle's)
        if (rc_dot != MXG_OK) {
            if (dot_msg.empty()) return set_err(h, MXG_ENOMEM, "out of host memory writing '%s'", dot_path);
            h->err = dot_msg;
        }
        return rc_dot;
    } catch (const std::bad_alloc &) {
        return set_err(h, MXG_ENOMEM, "out of host memory in mxg_write_outputs");
    } catch (const std::system_error &) {
        return set_err(h, MXG_ENOMEM, "cannot start a writer thread");
    }
}

static size_t copy_out(const std::string &r, char *buf, size_t cap)
{
    if (buf && cap) {
        size_t n = std::min(r.size(), cap - 1);
        memcpy(buf, r.data(), n);
        buf[n] = 0;
    }
    return r.size();
}

size_t mxg_py_repr_double(double v, char *buf, size_t cap) { return copy_out(py_repr_float(v), buf, cap); }

size_t mxg_py_repr_str(const char *s, char *buf, size_t cap) { return copy_out(py_repr_str(s ? s : ""), buf, cap); }

int mxg_get_stats(mxg_handle *h, mxg_stats *out)
{
    if (!h || !out) return MXG_EINVAL;
    if (out->struct_size != sizeof(mxg_stats)) return set_err(h, MXG_EINVAL, "mxg_stats.struct_size mismatch");
    mxg_stats s{};
    s.struct_size = sizeof(mxg_stats);
    s.n_assemblies = (uint32_t)h->asms.size();
    for (auto *a : h->asms) {
        s.bases += a->total_bases;
        s.kmers += a->total_kmers;
        if (a->has_sketch) s.minimizers += a->n_mx;
    }
    if (h->graph.valid && h->stat_unique == ~0ull) {  // lazily: needs the per-minimizer flags on the host
        uint64_t u = 0;
        for (auto *a : h->asms) {
            if (flags_to_host(h, a) != MXG_OK) return MXG_EDEVICE;
            for (uint8_t f : a->h_flags) u += (f & MXG_MX_UNIQUE) ? 1 : 0;
        }
        h->stat_unique = u;
    }
    if (flush_timers(h) != MXG_OK) return MXG_EDEVICE;
    s.candidates = h->stat_candidates;
    s.bs_filter_bases = h->stat_bs_bases;
    s.batches_redone = h->stat_batches_redone;
    s.sync_assemblies = h->stat_sync_assemblies;
    s.deferred_stretches = h->stat_deferred;
    s.select_slices = h->stat_sel_slices;
    s.slice_stretches = h->stat_slice_stretches;
    s.graph_join = h->stat_graph_join;
    s.retried_assemblies = h->stat_retries;
    s.dense_kmers = h->stat_dense_kmers;
    s.unique = h->graph.valid ? h->stat_unique : 0;
    s.vertices = h->graph.valid ? h->graph.nv : 0;
    s.edges = h->graph.valid ? h->graph.ne : 0;
    s.ms_hash = h->tm.ms_hash;
    s.ms_resolve = h->tm.ms_resolve;
    s.ms_graph = h->tm.ms_graph;
    s.launches_hash = h->tm.launches_hash;
    s.hash_kernel_bases = h->tm.hash_bases;
    s.ms_reorder = h->tm.ms_reorder;
    s.ms_resolve_kernel = h->tm.ms_resolve_k;
    s.ms_emit = h->tm.ms_emit;
    s.ms_join = h->tm.ms_join;
    s.ms_vertices = h->tm.ms_vertices;
    s.ms_edges = h->tm.ms_edges;
    *out = s;
    return MXG_OK;
}

size_t mxg_knobs(mxg_handle *h, char *buf, size_t cap)
{
    if (!h) return 0;
    std::string r;
    for (const auto &kv : h->knobs)  // (a std::map: sorted by name)
        if (kv.second.set) r += (r.empty() ? "" : " ") + kv.first + "=" + kv.second.raw;
    if (buf && cap) {
        const size_t n = std::min(r.size(), cap - 1);
        memcpy(buf, r.data(), n);
        buf[n] = 0;
    }
    return r.size();
}

int mxg_reset_timers(mxg_handle *h)
{
    if (!h) return MXG_EINVAL;
    int rc = flush_timers(h);
    if (rc != MXG_OK) return rc;
    h->tm = Timers();
    h->stat_candidates = h->stat_dense_kmers = h->stat_bs_bases = 0;
    h->stat_batches_redone = h->stat_sync_assemblies = h->stat_deferred = h->stat_retries = h->stat_sel_slices = h->stat_slice_stretches = 0;
    return MXG_OK;
}

}  // extern "C"
