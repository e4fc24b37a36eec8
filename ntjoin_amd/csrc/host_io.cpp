// host_io.cpp -- host side of libntjoin_mx.so: FASTA ingest (2-bit packing + valid-run table), TSV parse,
// TSV / .mx.dot writers.  Text formats follow the reference exactly:
//   TSV grammar           ntJoin:205 (`indexlr --seq --long --pos`), parsed at bin/ntjoin_utils.py:173-185
//   .mx.dot grammar       bin/ntjoin.py:25-62 (python repr() of (contig,pos) tuples and float weights)
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <thread>
#include <unordered_map>

#include "mxg_internal.h"

namespace mxg {

thread_local std::string *tl_err_sink = nullptr;

// ---- the handle's pinned pool (mxg_internal.h) ----
// MXG_PIN_MALLOC=1: one hipHostMalloc of the whole pool, as up to round 5 (A/B knob; also what happens when registering fails)
static hipError_t pin_pool_malloc(mxg_handle *h)
{
    void *q = nullptr;
    const hipError_t e = hipHostMalloc(&q, PIN_POOL_BYTES);
    if (e != hipSuccess) {
        h->pin_state = 3;
        h->pin_err = e;
        return e;
    }
    h->pin_pool = q;
    h->pin_registered = false;
    h->pin_ready = PIN_PIECES;
    h->pin_state = 2;
    return hipSuccess;
}
hipError_t pin_pool_start(mxg_handle *h)
{
    if (h->pin_state.load() == 3) {  // registering failed: the pieces go back and the pool is taken in one allocation.  The
        (void)hipSetDevice(h->device);  // operation that waited for the missing piece has ended, but copies it issued out of the
        (void)hipDeviceSynchronize();   // pieces that DID arrive may still be in flight: they end before their pages go away
        pin_pool_release(h);
        (void)hipGetLastError();
        return pin_pool_malloc(h);
    }
    if (h->pin_state.load() != 0) return hipSuccess;
    if (getenv("MXG_PIN_MALLOC")) return pin_pool_malloc(h);
    void *m = mmap(nullptr, PIN_POOL_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return pin_pool_malloc(h);
    h->pin_pool = m;
    h->pin_registered = true;
    h->pin_ready = 0;
    h->pin_state = 1;
    try {
        // (detached: a one-shot process ends without joining anybody; pin_pool_release waits for pin_state to leave 1)
        std::thread([h]() {
            hipError_t e = hipSetDevice(h->device);
            for (uint32_t b = 0; b < PIN_PIECES && e == hipSuccess; ++b) {
                e = hipHostRegister(static_cast<char *>(h->pin_pool) + (size_t)b * PIN_PIECE_BYTES, PIN_PIECE_BYTES, hipHostRegisterDefault);
                if (e == hipSuccess) h->pin_ready.store(b + 1, std::memory_order_release);
            }
            h->pin_err = e;
            h->pin_state.store(e == hipSuccess ? 2 : 3, std::memory_order_release);
        }).detach();
    } catch (...) {  // no thread to be had
        munmap(m, PIN_POOL_BYTES);
        h->pin_pool = nullptr;
        h->pin_registered = false;
        h->pin_state = 0;
        return pin_pool_malloc(h);
    }
    return hipSuccess;
}
hipError_t pin_pool_wait(mxg_handle *h, uint32_t pieces)
{
    pieces = std::min(pieces, PIN_PIECES);
    while (h->pin_ready.load(std::memory_order_acquire) < pieces && h->pin_state.load(std::memory_order_acquire) == 1)
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    if (h->pin_ready.load(std::memory_order_acquire) >= pieces) return hipSuccess;
    return h->pin_err != hipSuccess ? h->pin_err : hipErrorUnknown;  // (state 3: the next pin_pool_start allocates in one piece)
}
void pin_pool_release(mxg_handle *h)
{
    while (h->pin_state.load(std::memory_order_acquire) == 1) std::this_thread::sleep_for(std::chrono::microseconds(100));
    if (h->pin_pool) {
        if (h->pin_registered) {
            for (uint32_t b = 0; b < h->pin_ready.load(); ++b)
                (void)hipHostUnregister(static_cast<char *>(h->pin_pool) + (size_t)b * PIN_PIECE_BYTES);
            munmap(h->pin_pool, PIN_POOL_BYTES);
        } else {
            (void)hipHostFree(h->pin_pool);
        }
    }
    h->pin_pool = nullptr;
    h->pin_registered = false;
    h->pin_ready = 0;
    h->pin_state = 0;
}

int set_err(mxg_handle *h, int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (tl_err_sink) *tl_err_sink = buf;  // (a helper thread of the library: its caller decides what the handle reports)
    else if (h) h->err = buf;
    return code;
}

static const mxg_handle::Knob &knob_of(const mxg_handle *h, const char *name)
{
    auto &m = const_cast<mxg_handle *>(h)->knobs;  // (a cache: filling it does not change what the handle does)
    auto it = m.find(name);
    if (it == m.end()) {
        mxg_handle::Knob k;
        const char *e = getenv(name);
        if (e && *e) {
            k.set = true;
            k.raw = e;
            k.value = strtoull(e, nullptr, 10);
        }
        it = m.emplace(name, k).first;
    }
    return it->second;
}
uint64_t knob_u64(const mxg_handle *h, const char *name, uint64_t dflt)
{
    const mxg_handle::Knob &k = knob_of(h, name);
    return k.set ? k.value : dflt;
}
bool knob_set(const mxg_handle *h, const char *name) { return knob_of(h, name).set; }
const char *knob_raw(const mxg_handle *h, const char *name)
{
    const mxg_handle::Knob &k = knob_of(h, name);
    return k.set ? k.raw.c_str() : nullptr;
}

// ---- ntHash constants ---------------------------------------------------
static const uint64_t SEED[4] = {0x3c8bfbb395c60474ULL, 0x3193c18562a02b4cULL, 0x20323ed082572324ULL,
                                 0x295549f54be24456ULL};  // A C G T

static uint64_t srol_n(uint64_t x, unsigned n)
{
    const uint64_t LO = 0x1FFFFFFFFULL;
    uint64_t lo = x & LO, hi = x >> 33;
    unsigned a = n % 33, b = n % 31;
    if (a) lo = ((lo << a) | (lo >> (33 - a))) & LO;
    if (b) hi = ((hi << b) | (hi >> (31 - b))) & 0x7FFFFFFFULL;
    return (hi << 33) | lo;
}

void make_hash_tab(uint32_t k, HashTab *t)
{
    for (int o = 0; o < 5; ++o)
        for (int i = 0; i < 4; ++i) {
            uint64_t f = SEED[i], r = srol_n(SEED[3 - i], k);
            if (o < 4) {
                f ^= srol_n(SEED[o], k);
                r ^= SEED[3 - o];
            }
            t->e[o * 4 + i] = make_uint4((uint32_t)f, (uint32_t)(f >> 32), (uint32_t)r, (uint32_t)(r >> 32));
        }
}

// Byte table of the direct hash formula (sketch.hip init_direct): the "warm-up state" after the first m = 4*(k/4) bases
// is  F = XOR_j srol^{m-1-j}(seed[c_j]),  R = XOR_j srol^{k-m+j}(seed'[c_j]).  Entry v (one packed byte = bases 4q..4q+3,
// base i in bits 2i..2i+1) holds the four terms of a byte at position 0: f4 = XOR_i srol^{3-i} seed[c_i],
// r4 = XOR_i srol^{i} seed'[c_i]; the byte's position enters through Horner rotations by 4 on the device.
void make_init_tab(uint32_t k, std::vector<uint4> &out)
{
    (void)k;  // the byte table does not depend on k: position enters through the Horner rotations on the device
    out.assign(256 + 8 * 256, make_uint4(0, 0, 0, 0));
    for (uint32_t v = 0; v < 256; ++v) {
        uint64_t f = 0, r = 0;
        for (uint32_t i = 0; i < 4; ++i) {
            const uint32_t c = (v >> (2 * i)) & 3u;
            f ^= srol_n(SEED[c], 3 - i);
            r ^= srol_n(SEED[3 - c], i);
        }
        out[v] = make_uint4((uint32_t)f, (uint32_t)(f >> 32), (uint32_t)r, (uint32_t)(r >> 32));
        // position tables (sketch.hip init_pos): byte j of a group of 8 bytes (32 bases) contributes srol^{4(7-j)} f4 to the
        // forward hash and srol^{4j} r4 to the reverse-complement hash -- one lookup and four XORs per byte, no rotation
        for (uint32_t j = 0; j < 8; ++j) {
            const uint64_t fj = srol_n(f, 4 * (7 - j)), rj = srol_n(r, 4 * j);
            out[256 + j * 256 + v] = make_uint4((uint32_t)fj, (uint32_t)(fj >> 32), (uint32_t)rj, (uint32_t)(rj >> 32));
        }
    }
}

// ---- base code table -------------------------------------------------------------------------------
static const uint8_t *code_lut()
{
    static uint8_t lut[256];
    static bool init = false;
    if (!init) {
        memset(lut, 4, sizeof lut);
        lut[(int)'A'] = lut[(int)'a'] = 0;
        lut[(int)'C'] = lut[(int)'c'] = 1;
        lut[(int)'G'] = lut[(int)'g'] = 2;
        lut[(int)'T'] = lut[(int)'t'] = lut[(int)'U'] = lut[(int)'u'] = 3;
        init = true;
    }
    return lut;
}

// Incremental builder: packs bases and collects the valid-run table record by record.
struct Ingest {
    mxg_handle *h;
    Assembly *a;
    uint32_t k, w;
    uint64_t cur_base = 0;  // next free global base slot
    // current record
    uint64_t rec_start = 0, rec_pos = 0, run_len = 0;
    uint64_t fpos = 0;        // position one past the last base fed to the packer / run tracker (== rec_pos unless a piece)
    std::vector<std::pair<uint32_t, uint32_t>> rec_runs;  // (pos0, n_kmers) of the open record
    uint32_t cur_word = 0;
    bool skip = false;        // current record is outside this handle's shard: only its length is recorded
    bool lengths_only = false;  // first pass of a sharded load: record nothing but lengths
    uint64_t keep_lo = 0, keep_hi = ~0ull;
    std::vector<uint64_t> lengths;
    // split load (records may be cut between shards, see plan_pieces): pass 1 also keeps every record's valid runs,
    // pass 2 feeds only the bases [p_lo, p_hi) of a record
    struct Piece {
        uint64_t lo = 0, hi = 0;
        bool keep = false, drop = false;
        bool cont = false;  // the record began on an earlier shard (which printed its TSV line so far)
    };
    bool collect_runs = false, split = false;
    std::vector<std::vector<std::pair<uint64_t, uint64_t>>> all_runs;  // per record: (pos0, n_kmers)
    std::vector<Piece> pieces;
    uint64_t p_lo = 0, p_hi = ~0ull;
    bool p_drop = false;
    std::string stage;        // line bodies waiting to be packed when the text is not kept
    size_t pending = 0;       // bytes at the end of a->text / stage not packed yet
    bool pending_keep = false;

    Ingest(mxg_handle *h_, Assembly *a_) : h(h_), a(a_), k(h_->cfg.k), w(h_->cfg.w) {}

    void begin_record(const std::string &id)
    {
        if (lengths_only) {
            lengths.push_back(0);
            if (collect_runs) all_runs.emplace_back();
            rec_pos = 0;
            run_len = 0;
            skip = true;
            return;
        }
        const uint64_t ridx = a->recs.size();
        skip = ridx < keep_lo || ridx >= keep_hi;
        p_lo = 0;
        p_hi = ~0ull;
        p_drop = false;
        if (split) {
            const Piece &pc = pieces[ridx];
            skip = !pc.keep;
            p_lo = pc.lo;
            p_hi = pc.hi;
            p_drop = pc.drop;
        }
        cur_base = (cur_base + 15) & ~uint64_t(15);
        Record r;
        r.id = id;
        // a piece packs the bases [p_lo, p_hi) only; the offsets stay "of base 0" (they wrap; every use adds a position)
        r.base_off = cur_base - (p_lo & ~uint64_t(15));
        r.text_off = (uint64_t)a->text.size() - p_lo;
        a->recs.push_back(r);
        rec_start = cur_base;
        rec_pos = 0;
        fpos = p_lo;
        run_len = 0;
        rec_runs.clear();
        cur_word = 0;
    }
    inline void close_run()
    {
        if (run_len >= k) rec_runs.emplace_back((uint32_t)(fpos - run_len), (uint32_t)(run_len - k + 1));
        run_len = 0;
    }
    std::string no_cr;        // a line body with its carriage returns taken out
    void add_bases(const uint8_t *s, size_t n, bool keep_text)
    {
        // CRLF files: a '\r' is never a base, wherever the read buffer happened to cut the line (a "\r\n" split between two
        // 4 MiB reads used to leave the '\r' in the sequence: one invalid base, every later position off by one)
        if (memchr(s, '\r', n)) {
            no_cr.clear();
            for (size_t i = 0; i < n; ++i)
                if (s[i] != '\r') no_cr.push_back((char)s[i]);
            s = reinterpret_cast<const uint8_t *>(no_cr.data());
            n = no_cr.size();
            if (!n) return;
        }
        if (lengths_only && collect_runs) {  // pass 1 of a split load: valid runs of every record
            const uint8_t *lut = code_lut();
            for (size_t i = 0; i < n; ++i) {
                if (lut[s[i]] < 4) {
                    ++run_len;
                } else {
                    if (run_len >= k) all_runs.back().emplace_back(rec_pos + i - run_len, run_len - k + 1);
                    run_len = 0;
                }
            }
            rec_pos += n;
            return;
        }
        if (skip) {
            rec_pos += n;
            return;
        }
        // the part of [rec_pos, rec_pos + n) inside the piece (everything, for a whole record)
        const uint64_t s0 = rec_pos, s1 = rec_pos + n;
        rec_pos = s1;
        const uint64_t c0 = std::max(s0, p_lo), c1 = std::min(s1, p_hi);
        if (c0 >= c1) return;
        s += c0 - s0;
        n = (size_t)(c1 - c0);
        // the line bodies are first laid end to end (the record's text, or a staging buffer when the text is not kept) and
        // packed from there in long contiguous stretches, whatever the line length of the file
        std::string &dst = keep_text ? a->text : stage;
        dst.append(reinterpret_cast<const char *>(s), n);
        pending += n;
        pending_keep = keep_text;
        if (pending >= (size_t(1) << 20)) pack_pending();
    }
    void pack_pending()
    {
        if (!pending) return;
        std::string &src = pending_keep ? a->text : stage;
        const uint8_t *s = reinterpret_cast<const uint8_t *>(src.data()) + (src.size() - pending);
        const size_t n = pending;
        const uint8_t *lut = code_lut();
        auto one = [&](uint8_t ch) {
            const uint8_t c = lut[ch];
            const unsigned slot = (unsigned)(fpos & 15);
            if (c < 4) {
                cur_word |= (uint32_t)c << (2 * slot);
                ++run_len;
            } else {
                close_run();  // fpos = position of the invalid base = one past the run's last base
            }
            ++fpos;
            if (slot == 15) {
                a->h_packed.push_back(cur_word);
                cur_word = 0;
            }
        };
        size_t i = 0;
        while (i < n && (fpos & 15)) one(s[i++]);
        while (i + 16 <= n) {  // a whole packed word of valid bases at a time (no per-base branches)
            uint32_t wd = 0, bad = 0;
            for (unsigned j = 0; j < 16; ++j) {
                const uint32_t c = lut[s[i + j]];
                wd |= (c & 3u) << (2 * j);
                bad |= c;
            }
            if (bad & 4u) {
                for (unsigned j = 0; j < 16; ++j) one(s[i + j]);
            } else {
                a->h_packed.push_back(wd);
                run_len += 16;
                fpos += 16;
            }
            i += 16;
        }
        while (i < n) one(s[i++]);
        pending = 0;
        if (!pending_keep) stage.clear();
    }
    int end_record()
    {
        if (lengths_only) {
            lengths.back() = rec_pos;
            if (collect_runs && run_len >= k) all_runs.back().emplace_back(rec_pos - run_len, run_len - k + 1);
            return MXG_OK;
        }
        pack_pending();
        if (skip) {  // registered (global record index, id, length) but neither packed nor sketched here
            Record &r = a->recs.back();
            r.len = rec_pos;
            if (r.len >= (uint64_t(1) << 32))
                return set_err(h, MXG_ELIMIT, "record '%s' has %llu bases; the engine indexes positions with 32 bits",
                               r.id.c_str(), (unsigned long long)r.len);
            a->total_bases += 0;
            return MXG_OK;
        }
        close_run();
        if (fpos & 15) a->h_packed.push_back(cur_word);
        cur_word = 0;
        Record &r = a->recs.back();
        r.len = rec_pos;
        if (r.len >= (uint64_t(1) << 32))
            return set_err(h, MXG_ELIMIT, "record '%s' has %llu bases; the engine indexes positions with 32 bits",
                           r.id.c_str(), (unsigned long long)r.len);
        cur_base = rec_start + ((fpos - (p_lo & ~uint64_t(15)) + 15) & ~uint64_t(15));
        a->total_bases += fpos - p_lo;
        uint64_t nk = 0;
        for (auto &rr : rec_runs) nk += rr.second;
        if (nk >= w && nk > 0) {
            uint32_t ctg = (uint32_t)a->ctg_rec.size();
            a->ctg_rec.push_back((uint32_t)(a->recs.size() - 1));
            a->ctg_nk.push_back((uint32_t)nk);
            a->ctg_run0.push_back((uint32_t)a->runs.size());
            if (split) {
                a->ctg_drop.push_back(p_drop ? 1 : 0);
                a->any_drop = a->any_drop || p_drop;
            }
            uint32_t kidx = 0;
            for (auto &rr : rec_runs) {
                Run run;
                run.base_off = r.base_off + rr.first;
                run.n_kmers = rr.second;
                run.contig = ctg;
                run.kidx0 = kidx;
                run.pos0 = rr.first;
                kidx += rr.second;
                a->runs.push_back(run);
            }
            a->total_kmers += nk;
        }
        return MXG_OK;
    }
    void finish()
    {
        a->ctg_run0.push_back((uint32_t)a->runs.size());
        // the hash kernel may read up to one strip + k bases past a run's end: pad with 1 KiB + k/4 bytes
        size_t pad = 256 + (k + 15) / 16 + 16;
        a->h_packed.insert(a->h_packed.end(), pad, 0u);
        a->packed_words = a->h_packed.size();
        a->has_bases = true;
    }
};

static std::string header_id(const char *p, size_t n)
{
    size_t e = 0;
    while (e < n && p[e] != ' ' && p[e] != '\t' && p[e] != '\r' && p[e] != '\n') ++e;
    return std::string(p, e);
}

void shard_range(const uint64_t *lengths, uint64_t n, uint32_t shard, uint32_t n_shards, uint64_t *lo, uint64_t *hi)
{
    // a record belongs to the shard that the midpoint of its base range falls in (cumulative over the file)
    long double total = 0;
    for (uint64_t r = 0; r < n; ++r) total += (long double)lengths[r];
    uint64_t l = n, hgh = n;
    bool have_lo = false;
    long double cum = 0;
    for (uint64_t r = 0; r < n; ++r) {
        const long double mid = cum + (long double)lengths[r] / 2;
        uint32_t s = total > 0 ? (uint32_t)std::min<long double>((long double)n_shards - 1, mid * n_shards / total) : 0;
        if (!have_lo && s >= shard) {
            l = r;
            have_lo = true;
        }
        if (s > shard) {
            hgh = r;
            break;
        }
        cum += (long double)lengths[r];
    }
    if (!have_lo) l = n;
    if (hgh < l) hgh = l;
    *lo = l;
    *hi = hgh;
}

// plain or gzip-compressed text (indexlr reads both; B1 in SURVEY.md 8b lists `.gz` as optional): zlib's gzread passes
// uncompressed files through unchanged, so one reader serves both
struct TextReader {
    gzFile g = nullptr;
    bool open(const char *path)
    {
        g = gzopen(path, "rb");
        if (g) gzbuffer(g, 1 << 20);
        return g != nullptr;
    }
    size_t read(char *dst, size_t n)
    {
        const int got = gzread(g, dst, (unsigned)n);
        if (got < 0) bad = true;
        return got > 0 ? (size_t)got : 0;
    }
    void rewind() { gzrewind(g); }
    void close()
    {
        if (g) gzclose(g);
        g = nullptr;
    }
    bool bad = false;
};

static int parse_fasta_stream(mxg_handle *h, TextReader &f, const char *path, Ingest &in, bool keep)
{
    std::vector<char> buf(1 << 22);
    std::string carry;  // partial header line across buffer boundaries
    bool in_header = false, have_rec = false, at_line_start = true;
    int rc = MXG_OK;
    size_t got;
    while (rc == MXG_OK && (got = f.read(buf.data(), buf.size())) > 0) {
        size_t i = 0;
        while (i < got && rc == MXG_OK) {
            if (in_header) {
                const char *nl = (const char *)memchr(buf.data() + i, '\n', got - i);
                size_t e = nl ? (size_t)(nl - buf.data()) : got;
                carry.append(buf.data() + i, e - i);
                i = e;
                if (nl) {
                    if (have_rec) rc = in.end_record();
                    in.begin_record(header_id(carry.data(), carry.size()));
                    have_rec = true;
                    carry.clear();
                    in_header = false;
                    at_line_start = true;
                    ++i;
                }
            } else if (at_line_start && buf[i] == '>') {
                in_header = true;
                ++i;
            } else {
                const char *nl = (const char *)memchr(buf.data() + i, '\n', got - i);
                size_t e = nl ? (size_t)(nl - buf.data()) : got;
                size_t n = e - i;
                if (n && buf[e - 1] == '\r' && nl) --n;  // CRLF
                if (have_rec && n) in.add_bases((const uint8_t *)buf.data() + i, n, keep);
                at_line_start = nl != nullptr;
                i = nl ? e + 1 : e;
            }
        }
    }
    if (f.bad) rc = set_err(h, MXG_EIO, "read error on '%s'", path);
    if (rc != MXG_OK) return rc;
    if (in_header) {  // header without trailing newline at EOF
        if (have_rec) rc = in.end_record();
        in.begin_record(header_id(carry.data(), carry.size()));
        have_rec = true;
    }
    if (rc == MXG_OK && have_rec) rc = in.end_record();
    return rc;
}

// Split load: shard s of n owns the base range [total*s/n, total*(s+1)/n) of the concatenated records, and with it every
// window whose LAST k-mer starts inside that range.  For a record cut by the range's start it needs the w valid k-mers
// before its first own k-mer: w-1 of them complete its first own window, one more makes its first window the LAST
// window of the shard before it, whose minimizer that shard reports -- so this piece's first minimizer is always
// dropped (k_resolve, ctg_drop) and nothing is reported twice or lost, whatever the hashes are (the
// sketch is the sequence of distinct window arg-mins, non-decreasing in the window).  Pieces whose first own k-mer is
// among the record's first w valid k-mers start at the record's beginning instead (the shards before them then hold
// fewer than w k-mers of it and report nothing).
static void plan_pieces(Ingest &in, uint32_t shard, uint32_t n_shards, uint32_t k, uint32_t w)
{
    const size_t n = in.lengths.size();
    unsigned __int128 total = 0;
    for (size_t r = 0; r < n; ++r) total += in.lengths[r];
    const uint64_t cut_lo = (uint64_t)(total * shard / n_shards), cut_hi = (uint64_t)(total * (shard + 1) / n_shards);
    in.pieces.assign(n, Ingest::Piece());
    uint64_t cum = 0;
    for (size_t r = 0; r < n; ++r) {
        const uint64_t len = in.lengths[r], r0 = cum, r1 = cum + len;
        cum = r1;
        if (len == 0 || r1 <= cut_lo || r0 >= cut_hi) continue;
        const uint64_t P_lo = std::max(cut_lo, r0) - r0, P_hi = std::min(cut_hi, r1) - r0;  // own k-mer starts: [P_lo, P_hi)
        Ingest::Piece &pc = in.pieces[r];
        pc.keep = true;
        pc.hi = P_hi == len ? len : std::min<uint64_t>(len, P_hi + k - 1);
        pc.lo = 0;
        pc.cont = P_lo > 0;
        if (P_lo > 0) {
            uint64_t t_first = 0;  // valid k-mers that start before P_lo
            for (auto &run : in.all_runs[r])
                if (run.first < P_lo) t_first += std::min<uint64_t>(run.second, P_lo - run.first);
            if (t_first >= w) {
                uint64_t want = t_first - w, seen = 0;  // the piece starts at valid k-mer number t_first - w
                for (auto &run : in.all_runs[r]) {
                    if (want < seen + run.second) {
                        pc.lo = run.first + (want - seen);
                        break;
                    }
                    seen += run.second;
                }
                pc.drop = true;
            }
        }
    }
}

int load_fasta(mxg_handle *h, Assembly *a, const char *path, uint32_t shard, uint32_t n_shards, bool split)
{
    TextReader f;
    if (!f.open(path)) return set_err(h, MXG_EIO, "cannot open FASTA '%s'", path);
    const bool keep = !(h->cfg.flags & MXG_FLAG_DROP_SEQ);
    Ingest in(h, a);
    int rc = MXG_OK;
    if (n_shards > 1 && split) {  // pass 1: lengths and valid runs of every record -> this rank's pieces (same plan on every rank)
        in.lengths_only = in.collect_runs = true;
        rc = parse_fasta_stream(h, f, path, in, false);
        if (rc == MXG_OK) {
            plan_pieces(in, shard, n_shards, h->cfg.k, h->cfg.w);
            uint64_t lo = in.pieces.size(), hi = 0;
            for (size_t r = 0; r < in.pieces.size(); ++r)
                if (in.pieces[r].keep) {
                    lo = std::min<uint64_t>(lo, r);
                    hi = r + 1;
                }
            a->shard_lo = std::min(lo, hi);
            a->shard_hi = hi;
            a->split_first_cont = hi > lo && in.pieces[lo].cont;
            in.lengths_only = in.collect_runs = false;
            in.split = true;
            in.all_runs.clear();
            f.rewind();
        }
    } else if (n_shards > 1) {  // pass 1: record lengths -> this rank's contiguous record range (same on every rank)
        in.lengths_only = true;
        rc = parse_fasta_stream(h, f, path, in, false);
        if (rc == MXG_OK) {
            shard_range(in.lengths.data(), in.lengths.size(), shard, n_shards, &in.keep_lo, &in.keep_hi);
            a->shard_lo = in.keep_lo;
            a->shard_hi = in.keep_hi;
            in.lengths_only = false;
            f.rewind();
        }
    }
    if (rc == MXG_OK) rc = parse_fasta_stream(h, f, path, in, keep);
    f.close();
    if (rc != MXG_OK) return rc;
    in.finish();
    a->has_text = keep;
    return MXG_OK;
}

int load_buffers(mxg_handle *h, Assembly *a, const uint8_t *ascii, const uint64_t *offsets,
                 const char *const *ids, uint64_t n_records)
{
    const bool keep = !(h->cfg.flags & MXG_FLAG_DROP_SEQ);
    Ingest in(h, a);
    for (uint64_t r = 0; r < n_records; ++r) {
        if (offsets[r + 1] < offsets[r]) return set_err(h, MXG_EINVAL, "offsets must be non-decreasing");
        in.begin_record(ids && ids[r] ? std::string(ids[r]) : std::to_string(r));
        in.add_bases(ascii + offsets[r], (size_t)(offsets[r + 1] - offsets[r]), keep);
        int rc = in.end_record();
        if (rc != MXG_OK) return rc;
    }
    in.finish();
    a->has_text = keep;
    return MXG_OK;
}

void build_runs_from_lengths(mxg_handle *h, Assembly *a)
{
    const uint32_t k = h->cfg.k, w = h->cfg.w;
    for (size_t r = 0; r < a->recs.size(); ++r) {
        const Record &rec = a->recs[r];
        a->total_bases += rec.len;
        if (rec.len < k) continue;
        uint64_t nk = rec.len - k + 1;
        if (nk < w) continue;
        Run run;
        run.base_off = rec.base_off;
        run.n_kmers = (uint32_t)nk;
        run.contig = (uint32_t)a->ctg_rec.size();
        run.kidx0 = 0;
        run.pos0 = 0;
        a->ctg_rec.push_back((uint32_t)r);
        a->ctg_nk.push_back((uint32_t)nk);
        a->ctg_run0.push_back((uint32_t)a->runs.size());
        a->runs.push_back(run);
        a->total_kmers += nk;
    }
    a->ctg_run0.push_back((uint32_t)a->runs.size());
    a->has_bases = true;
}

// ---- TSV parse (the parse half of read_minimizers, bin/ntjoin_utils.py:173-185) -------------------
static inline bool is_py_space(char c)
{
    return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f';
}

// One line of the TSV as read_minimizers takes it apart (bin/ntjoin_utils.py:173-185); [b, e) excludes the '\n'.
// Returns 0, or -1 with *bad pointing at the offending entry (msg says why).
struct TsvPart {
    std::vector<Record> recs;
    std::vector<uint64_t> hash;
    std::vector<uint32_t> pos, rec;  // rec: index into this part's recs
    const char *bad = nullptr, *bad_end = nullptr;
    int bad_kind = 0;  // 1: not three fields, 2: not <u64>:<u32>:<seq>
};

static int parse_tsv_line(const char *b, const char *e, TsvPart &out)
{
    // line.strip()
    while (b < e && is_py_space(*b)) ++b;
    while (e > b && is_py_space(e[-1])) --e;
    // .split("\t"): need field 0 and field 1
    const char *t1 = (const char *)memchr(b, '\t', (size_t)(e - b));
    if (!t1) return 0;  // len(line) == 1 : record without minimizers is skipped (:176)
    const char *f1 = t1 + 1;
    const char *t2 = (const char *)memchr(f1, '\t', (size_t)(e - f1));
    const char *f1e = t2 ? t2 : e;
    Record r;
    r.id.assign(b, (size_t)(t1 - b));
    const uint32_t ridx = (uint32_t)out.recs.size();
    out.recs.push_back(r);
    const char *p = f1;  // field 1 .split(" ")
    while (true) {
        const char *sp = (const char *)memchr(p, ' ', (size_t)(f1e - p));
        const char *ee = sp ? sp : f1e;
        // entry.split(":") must have exactly three fields (mx, pos, seq) at HEAD (:181)
        const char *c1 = (const char *)memchr(p, ':', (size_t)(ee - p));
        const char *c2 = c1 ? (const char *)memchr(c1 + 1, ':', (size_t)(ee - c1 - 1)) : nullptr;
        const char *c3 = c2 ? (const char *)memchr(c2 + 1, ':', (size_t)(ee - c2 - 1)) : nullptr;
        if (!c1 || !c2 || c3) {
            out.bad = p;
            out.bad_end = ee;
            out.bad_kind = 1;
            return -1;
        }
        uint64_t hv = 0;
        auto r1 = std::from_chars(p, c1, hv);
        uint32_t pv = 0;
        auto r2 = std::from_chars(c1 + 1, c2, pv);
        if (r1.ec != std::errc() || r1.ptr != c1 || p == c1 || r2.ec != std::errc() || r2.ptr != c2 || c1 + 1 == c2) {
            out.bad = p;
            out.bad_end = ee;
            out.bad_kind = 2;
            return -1;
        }
        out.hash.push_back(hv);
        out.pos.push_back(pv);
        out.rec.push_back(ridx);
        if (!sp) break;
        p = sp + 1;
    }
    return 0;
}

// the parse half of read_minimizers on `host_threads` workers: the file is cut at line ends into one piece per worker
// (the reference parses ~65 bytes of text per minimizer in Python, 5.7 us each: SURVEY.md 6)
int load_tsv(mxg_handle *h, Assembly *a, const char *path, std::vector<uint64_t> &hash,
             std::vector<uint32_t> &pos, std::vector<uint32_t> &rec)
{
    FILE *f = fopen(path, "rb");
    if (!f) return set_err(h, MXG_EIO, "cannot open TSV '%s'", path);
    // a regular file is mapped (the workers then parse straight out of the page cache); anything else is read
    std::string text;
    const char *t0 = nullptr;
    size_t n = 0;
    void *map = MAP_FAILED;
    struct stat sb;
    if (fstat(fileno(f), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
        map = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fileno(f), 0);
        if (map != MAP_FAILED) {
            t0 = static_cast<const char *>(map);
            n = (size_t)sb.st_size;
        }
    }
    bool rerr = false;
    if (map == MAP_FAILED) {
        char buf[1 << 16];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, got);
        rerr = ferror(f) != 0;
        t0 = text.data();
        n = text.size();
    }
    fclose(f);
    struct Unmap {
        void *p;
        size_t n;
        ~Unmap()
        {
            if (p != MAP_FAILED) munmap(p, n);
        }
    } unmap{map, n};
    if (rerr) return set_err(h, MXG_EIO, "read error on '%s'", path);
    uint32_t T = std::max<uint32_t>(1, std::min<uint32_t>(host_threads(h), (uint32_t)(n / (1 << 20)) + 1));
    std::vector<size_t> cut(T + 1, n);
    cut[0] = 0;
    for (uint32_t t = 1; t < T; ++t) {
        size_t c = std::max(cut[t - 1], n * t / T);
        const char *nl = c < n ? (const char *)memchr(t0 + c, '\n', n - c) : nullptr;
        cut[t] = nl ? (size_t)(nl - t0) + 1 : n;
    }
    std::vector<TsvPart> parts(T);
    auto work = [&](uint32_t t) {
        const char *p = t0 + cut[t], *end = t0 + cut[t + 1];
        while (p < end) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
            const char *e = nl ? nl : end;
            if (parse_tsv_line(p, e, parts[t]) != 0) return;
            p = nl ? nl + 1 : end;
        }
    };
    {
        std::vector<std::thread> th;
        for (uint32_t t = 1; t < T; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    for (uint32_t t = 0; t < T; ++t) {
        const TsvPart &pt = parts[t];
        if (pt.bad) {
            const unsigned long long lineno = 1ull + (unsigned long long)std::count(t0, pt.bad, '\n');
            const int len = (int)std::min<ptrdiff_t>(pt.bad_end - pt.bad, 60);
            if (pt.bad_kind == 1)
                return set_err(h, MXG_EIO,
                               "%s:%llu: entry '%.*s' does not have exactly three ':'-separated fields "
                               "(hash:pos:seq), as ntJoin's read_minimizers requires",
                               path, lineno, len, pt.bad);
            return set_err(h, MXG_EIO, "%s:%llu: cannot parse '%.*s' as <u64 hash>:<u32 pos>:<seq>", path, lineno, len, pt.bad);
        }
    }
    size_t n_mx = 0, n_rec = 0;
    for (auto &pt : parts) {
        n_mx += pt.hash.size();
        n_rec += pt.recs.size();
    }
    hash.reserve(n_mx);
    pos.reserve(n_mx);
    rec.reserve(n_mx);
    a->recs.reserve(a->recs.size() + n_rec);
    for (auto &pt : parts) {
        const uint32_t r0 = (uint32_t)a->recs.size();
        for (auto &r : pt.recs) a->recs.push_back(std::move(r));
        hash.insert(hash.end(), pt.hash.begin(), pt.hash.end());
        pos.insert(pos.end(), pt.pos.begin(), pt.pos.end());
        for (uint32_t r : pt.rec) rec.push_back(r0 + r);
    }
    return MXG_OK;
}

void build_rec_first(Assembly *a)
{
    a->rec_first.assign(a->recs.size() + 1, 0);
    for (uint64_t i = 0; i < a->n_mx; ++i) a->rec_first[a->h_rec[i] + 1]++;
    for (size_t r = 0; r < a->recs.size(); ++r) a->rec_first[r + 1] += a->rec_first[r];
}

// ---- python repr() ------------------------------------------------------------------------------------
std::string py_repr_str(const std::string &s)
{
    bool has_sq = s.find('\'') != std::string::npos, has_dq = s.find('"') != std::string::npos;
    char q = (has_sq && !has_dq) ? '"' : '\'';
    std::string o(1, q);
    char tmp[8];
    for (unsigned char c : s) {
        if (c == (unsigned char)q || c == '\\') {
            o += '\\';
            o += (char)c;
        } else if (c == '\n') o += "\\n";
        else if (c == '\r') o += "\\r";
        else if (c == '\t') o += "\\t";
        else if (c < 0x20 || c == 0x7f) {
            snprintf(tmp, sizeof tmp, "\\x%02x", c);
            o += tmp;
        } else o += (char)c;  // bytes >= 0x80: printable UTF-8 passes through unchanged
    }
    o += q;
    return o;
}

std::string py_repr_float(double v)
{
    if (std::isnan(v)) return "nan";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    // shortest round-trip digits, then CPython's float_repr_style='short' layout:
    // fixed notation for -4 <= exp10 < 16, otherwise d.ddde+XX
    char buf[64];
    auto res = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
    std::string s(buf, res.ptr);  // [-]d[.ddd]e[+-]XX
    bool neg = s[0] == '-';
    if (neg) s.erase(0, 1);
    size_t epos = s.find('e');
    std::string mant = s.substr(0, epos);
    int exp10 = atoi(s.c_str() + epos + 1);
    std::string digits;
    for (char c : mant)
        if (c != '.') digits += c;
    std::string out;
    if (exp10 >= -4 && exp10 < 16) {
        if (exp10 >= 0) {
            if ((int)digits.size() <= exp10 + 1) {
                out = digits + std::string(exp10 + 1 - digits.size(), '0') + ".0";
            } else {
                out = digits.substr(0, exp10 + 1) + "." + digits.substr(exp10 + 1);
            }
        } else {
            out = "0." + std::string(-exp10 - 1, '0') + digits;
        }
    } else {
        out = digits.substr(0, 1);
        if (digits.size() > 1) out += "." + digits.substr(1);
        char eb[16];
        snprintf(eb, sizeof eb, "e%c%02d", exp10 < 0 ? '-' : '+', std::abs(exp10));
        out += eb;
    }
    return neg ? "-" + out : out;
}

// ---- writers ---------------------------------------------------------------------------------------------
// Several byte ranges -> consecutive places of one file, each by its own thread (pwrite; optionally copies into a shared mapping
// of the file's new range).
bool put_parallel(int fd, uint64_t off, const char *const *data, const size_t *len, uint32_t n_parts)
{
    std::vector<uint64_t> at(n_parts + 1, off);
    for (uint32_t t = 0; t < n_parts; ++t) at[t + 1] = at[t] + len[t];
    const uint64_t total = at[n_parts] - off;
    if (!total) return true;
    // (a shared mapping of the file filled by the workers was measured on the GPU boxes' /tmp and dropped: 1.2-1.35 s against
    // 0.8-0.9 s of pwrite for 1.9 GB of outputs -- write faults on a shared file mapping are no cheaper than the inode lock)
    std::atomic<bool> good{true};
    auto put = [&](uint32_t t) {
        size_t done = 0;
        while (done < len[t]) {
            const ssize_t wr = pwrite(fd, data[t] + done, len[t] - done, (off_t)(at[t] + done));
            if (wr <= 0) {
                good = false;
                return;
            }
            done += (size_t)wr;
        }
    };
    {
        std::vector<std::thread> th;
        for (uint32_t t = 1; t < n_parts; ++t) th.emplace_back(put, t);
        put(0);
        for (auto &x : th) x.join();
    }
    return good;
}

struct OutBuf {
    FILE *f;
    std::vector<char> b;
    size_t n = 0;
    bool ok = true;
    explicit OutBuf(FILE *f_, size_t cap = (size_t)1 << 22) : f(f_), b(cap) {}
    inline void room(size_t need)
    {
        if (!f) {  // memory only: grows (the parallel .mx.dot writer formats chunks into such buffers)
            if (n + need > b.size()) b.resize(std::max(b.size() * 2, n + need));
            return;
        }
        if (n + need > b.size()) flush();
        if (need > b.size()) b.resize(need * 2);
    }
    void flush()
    {
        if (n && fwrite(b.data(), 1, n, f) != n) ok = false;
        n = 0;
    }
    inline void put(char c)
    {
        room(1);
        b[n++] = c;
    }
    inline void put(const char *s, size_t len)
    {
        room(len);
        memcpy(b.data() + n, s, len);
        n += len;
    }
    inline void put(const std::string &s) { put(s.data(), s.size()); }
    inline void put_u64(uint64_t v)
    {
        room(24);
        auto r = std::to_chars(b.data() + n, b.data() + n + 24, v);
        n = (size_t)(r.ptr - b.data());
    }
};

int write_tsv(mxg_handle *h, Assembly *a, const char *path, int with_pos, int with_strand, int with_seq)
{
    // Text formatted on the device (ingest.hip) unless the k-mer column must come from text kept on the HOST (records
    // handed over in buffers, sharded loads) or nothing on the device can spell the k-mers.  MXG_HOST_TSV=1: host writer.
    const bool whole = a->shard_lo == 0 && a->shard_hi >= a->recs.size();
    if (a->has_sketch && !a->has_text && whole && h->cfg.k <= 200 && !knob_set(h, "MXG_HOST_TSV") &&
        (!with_seq || a->text_on_device || (a->has_bases && a->d_packed && !a->foreign_sketch)))
        return write_tsv_device(h, a, path, with_pos, with_strand, with_seq);
    int rc = sync_sketch_to_host(h, a);
    if (rc != MXG_OK) return rc;
    const uint32_t k = h->cfg.k;
    if (with_seq && !a->has_text) {
        if (!a->has_bases)
            return set_err(h, MXG_EINVAL, "assembly '%s' has no bases: cannot print k-mer sequences", a->name.c_str());
        if (a->h_packed.empty()) {  // fetch the packed bases back from HBM
            a->h_packed.resize(a->packed_words);
            MXG_HIP(h, hipMemcpy(a->h_packed.data(), a->d_packed, a->packed_words * 4, hipMemcpyDeviceToHost));
        }
    }
    FILE *f = strcmp(path, "-") == 0 ? stdout : fopen(path, "wb");
    if (!f) return set_err(h, MXG_EIO, "cannot open '%s' for writing", path);
    OutBuf o(f);
    std::string kmer(k, 'N');
    for (size_t r = 0; r < a->recs.size(); ++r) {
        if (r < a->shard_lo || r >= a->shard_hi) continue;  // sharded load: this rank prints its own records (rank-ordered parts concatenate to the full file)
        const Record &rec = a->recs[r];
        o.put(rec.id);
        o.put('\t');
        for (uint64_t i = a->rec_first[r]; i < a->rec_first[r + 1]; ++i) {
            if (i != a->rec_first[r]) o.put(' ');
            o.put_u64(a->h_hash[i]);
            if (with_pos) {
                o.put(':');
                o.put_u64(a->h_pos[i]);
            }
            if (with_strand) {
                o.put(':');
                o.put(a->h_fwd[i] ? '+' : '-');
            }
            if (with_seq) {
                o.put(':');
                if (a->has_text) {
                    o.put(a->text.data() + (size_t)(rec.text_off + a->h_pos[i]), k);
                } else {
                    uint64_t b0 = rec.base_off + a->h_pos[i];
                    for (uint32_t j = 0; j < k; ++j) {
                        uint64_t bi = b0 + j;
                        kmer[j] = "ACGT"[(a->h_packed[bi >> 4] >> (2 * (bi & 15))) & 3];
                    }
                    o.put(kmer);
                }
            }
        }
        o.put('\n');
    }
    o.flush();
    bool ok = o.ok;
    if (f != stdout) ok = (fclose(f) == 0) && ok;
    else fflush(f);
    if (!ok) return set_err(h, MXG_EIO, "write error on '%s'", path);
    return MXG_OK;
}

// the text of the .mx.dot: vertex lines and edge lines of the host copy of the graph (reference bin/ntjoin.py:25-67)
struct DotText {
    const mxg_handle *h;
    const Graph &g;
    uint32_t A;
    std::vector<std::vector<std::string>> rec_repr;  // python repr() of every record id, once
    explicit DotText(const mxg_handle *h_) : h(h_), g(h_->graph), A(h_->graph.n_asm), rec_repr(h_->graph.n_asm)
    {
        for (uint32_t a = 0; a < A; ++a) {
            rec_repr[a].resize(h->asms[a]->recs.size());
            for (size_t r = 0; r < rec_repr[a].size(); ++r) rec_repr[a][r] = py_repr_str(h->asms[a]->recs[r].id);
        }
    }
    // label line per assembly: f"{file_name}_{(contig, pos)}"  (bin/ntjoin.py:43-47)
    // (one capacity check per line, then plain stores: the text of a 3 Gbp + 3 Gbp graph is 1.1 GB, and formatting it -- not
    // writing it -- was what the output phase of the one-process route waited for)
    void vertices(uint64_t v0, uint64_t v1, OutBuf &o) const
    {
        size_t fixed = 2 * 20 + 16;
        for (uint32_t a = 0; a < A; ++a) fixed += h->asms[a]->name.size() + 16 + 20;
        for (uint64_t v = v0; v < v1; ++v) {
            size_t need = fixed;
            for (uint32_t a = 0; a < A; ++a) need += rec_repr[a][g.vrec[(uint64_t)a * g.nv + v]].size();
            o.room(need);
            char *p = o.b.data() + o.n;
            *p++ = '"';
            char *d0 = p;
            p = std::to_chars(p, p + 20, g.vhash[v]).ptr;
            const size_t nd = (size_t)(p - d0);
            memcpy(p, "\" [label=\"", 10);
            p += 10;
            memcpy(p, d0, nd);
            p += nd;
            for (uint32_t a = 0; a < A; ++a) {
                *p++ = '\n';
                const std::string &nm = h->asms[a]->name;
                memcpy(p, nm.data(), nm.size());
                p += nm.size();
                *p++ = '_';
                *p++ = '(';
                const std::string &rr = rec_repr[a][g.vrec[(uint64_t)a * g.nv + v]];
                memcpy(p, rr.data(), rr.size());
                p += rr.size();
                *p++ = ',';
                *p++ = ' ';
                p = std::to_chars(p, p + 20, (uint64_t)g.vpos[(uint64_t)a * g.nv + v]).ptr;
                *p++ = ')';
            }
            memcpy(p, "\"]\n", 3);
            p += 3;
            o.n = (size_t)(p - o.b.data());
        }
    }
    void edges(uint64_t e0, uint64_t e1, OutBuf &o) const
    {
        static const char *COLOURS[10] = {"red",       "green", "blue",   "purple", "orange",
                                          "turquoise", "pink",  "yellow", "orchid", "salmon"};
        // " [weight=W color=C]\n" is a function of the edge's support set: spelt once per set (the weight is the sum of the
        // supporting assemblies' weights in assembly order, g.ew: repr() of a double is not cheap)
        std::unordered_map<uint32_t, std::string> tails;
        uint32_t last_m = 0;
        const std::string *last_tail = nullptr;
        for (uint64_t e = e0; e < e1; ++e) {
            const uint32_t m = g.esup[e];
            if (!last_tail || m != last_m) {
                auto it = tails.find(m);
                if (it == tails.end()) {
                    const int pc = __builtin_popcount(m);
                    const char *col;
                    if (pc == 1) col = (A > 10) ? "red" : COLOURS[__builtin_ctz(m)];
                    else if (pc == 2) col = "lightgrey";
                    else col = "black";
                    it = tails.emplace(m, "\" [weight=" + py_repr_float(g.ew[e]) + " color=" + col + "]\n").first;
                }
                last_m = m;
                last_tail = &it->second;
            }
            o.room(2 * 20 + 8 + last_tail->size());
            char *p = o.b.data() + o.n;
            *p++ = '"';
            p = std::to_chars(p, p + 20, g.vhash[g.eu[e]]).ptr;
            memcpy(p, "\" --\"", 5);
            p += 5;
            p = std::to_chars(p, p + 20, g.vhash[g.ev[e]]).ptr;
            memcpy(p, last_tail->data(), last_tail->size());
            p += last_tail->size();
            o.n = (size_t)(p - o.b.data());
        }
    }
};

// One part of the .mx.dot, for runs in which every rank holds the whole graph (the union route of ntjoin_amd/dist.py): part p of
// n takes the vertex lines [nv p / n, nv (p + 1) / n) and the edge lines likewise.  The file is "graph G {\n", every part's vertex
// segment, every part's edge segment, "}\n": a part's two segments are formatted into memory first (their sizes decide where
// everybody's segments go), then written at the offsets the caller computed from all parts' sizes.
int dot_part_format(mxg_handle *h, uint32_t part, uint32_t n_parts, uint64_t bytes[2])
{
    const Graph &g = h->graph;
    if (!g.valid || !g.host_valid) return set_err(h, MXG_EINVAL, "mxg_dot_part_format: no graph on the host");
    if (n_parts == 0 || part >= n_parts) return set_err(h, MXG_EINVAL, "mxg_dot_part_format: part out of range");
    const DotText dt(h);
    for (int seg = 0; seg < 2; ++seg) {
        const uint64_t n = seg ? g.ne : g.nv, lo = n * part / n_parts, hi = n * (part + 1) / n_parts;
        // one worker per 16 Ki items at most, buffers sized for their share (they grow if a line is longer than the guess)
        const uint32_t T = (uint32_t)std::min<uint64_t>(std::min(64u, std::max(1u, host_threads(h))), std::max<uint64_t>(1, (hi - lo) >> 14));
        std::vector<OutBuf> bufs;
        bufs.reserve(T);
        for (uint32_t t = 0; t < T; ++t) bufs.emplace_back(nullptr, (size_t)((hi - lo) / T + 1) * (seg ? 80 : 224) + 4096);
        std::atomic<bool> failed{false};
        auto work = [&](uint32_t t) {
            try {
                const uint64_t a = lo + (hi - lo) * t / T, b = lo + (hi - lo) * (t + 1) / T;
                bufs[t].n = 0;
                if (seg) dt.edges(a, b, bufs[t]); else dt.vertices(a, b, bufs[t]);
            } catch (...) {
                failed = true;
            }
        };
        {
            std::vector<std::thread> th;
            for (uint32_t t = 1; t < T; ++t) th.emplace_back(work, t);
            work(0);
            for (auto &x : th) x.join();
        }
        if (failed) return set_err(h, MXG_ENOMEM, "out of host memory formatting a part of the .mx.dot");
        size_t total = 0;
        for (auto &b : bufs) total += b.n;
        h->dot_part[seg].resize(total);
        size_t at = 0;
        for (auto &b : bufs) {
            memcpy(h->dot_part[seg].data() + at, b.b.data(), b.n);
            at += b.n;
        }
        bytes[seg] = total;
    }
    return MXG_OK;
}

int dot_part_write(mxg_handle *h, const char *path, uint64_t v_off, uint64_t e_off, int first, int last)
{
    const int fd = open(path, O_WRONLY | O_CREAT, 0644);
    if (fd < 0) return set_err(h, MXG_EIO, "cannot open '%s' for writing", path);
    auto put = [&](const char *p, size_t n, uint64_t off) {
        size_t done = 0;
        while (done < n) {
            const ssize_t wr = pwrite(fd, p + done, n - done, (off_t)(off + done));
            if (wr <= 0) return false;
            done += (size_t)wr;
        }
        return true;
    };
    bool ok = true;
    if (first) ok = put("graph G {\n", 10, 0);
    ok = ok && put(h->dot_part[0].data(), h->dot_part[0].size(), v_off);
    ok = ok && put(h->dot_part[1].data(), h->dot_part[1].size(), e_off);
    if (last) ok = ok && put("}\n", 2, e_off + h->dot_part[1].size());
    ok = (close(fd) == 0) && ok;
    h->dot_part[0] = std::vector<char>();
    h->dot_part[1] = std::vector<char>();
    if (!ok) return set_err(h, MXG_EIO, "write error on '%s'", path);
    return MXG_OK;
}

int write_dot(mxg_handle *h, const char *path)
{
    const Graph &g = h->graph;
    if (!g.valid) return set_err(h, MXG_EINVAL, "mxg_write_dot: call mxg_build_graph first");
    FILE *f = fopen(path, "w+b");
    if (!f) return set_err(h, MXG_EIO, "cannot open '%s' for writing", path);
    const bool dbg_io = getenv("MXG_DEBUG_IO") != nullptr;  // timings on stderr
    auto now_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now_s();
    double t_wait = 0.0;
    const DotText dt(h);
    auto vertices = [&](uint64_t v0, uint64_t v1, OutBuf &o) { dt.vertices(v0, v1, o); };
    auto edges = [&](uint64_t e0, uint64_t e1, OutBuf &o) { dt.edges(e0, e1, o); };
    // ~200 bytes per vertex and ~75 per edge (1.2 GB at 3 Gbp + 3 Gbp).  `host_threads` workers format chunks of 16 Ki items
    // into memory (two buffers each); this thread writes the chunks out in order AS THEY COMPLETE, so formatting and the
    // (serial: one file) copy into the page cache overlap -- formatting rounds alternating with writing rounds took their sum.
    constexpr uint64_t CH = 1u << 14;
    const uint64_t v_chunks = (g.nv + CH - 1) / CH, e_chunks = (g.ne + CH - 1) / CH, n_chunks = v_chunks + e_chunks;
    // (no more workers than chunks: a small graph -- the overlap stage's calls, the tests -- does not pay for 64 threads and buffers)
    const uint32_t T = (uint32_t)std::min<uint64_t>(std::min(64u, std::max(1u, host_threads(h))), std::max<uint64_t>(1, n_chunks));
    struct Slot {
        OutBuf buf{nullptr, CH * 64};
        std::atomic<int> full{0};
    };
    std::vector<Slot> slots(2 * (size_t)T);
    bool ok = fwrite("graph G {\n", 1, 10, f) == 10;
    ok = fflush(f) == 0 && ok;
    const int fd = fileno(f);
    std::atomic<bool> stop{false}, failed{false};
    auto slot_of = [&](uint64_t c) -> Slot & { return slots[(size_t)(c % T) * 2 + (size_t)((c / T) & 1)]; };
    auto worker = [&](uint32_t t) {
        try {
            for (uint64_t c = t; c < n_chunks && !stop; c += T) {
                Slot &sl = slot_of(c);
                while (sl.full.load(std::memory_order_acquire) && !stop) std::this_thread::yield();
                sl.buf.n = 0;
                if (c < v_chunks) vertices(c * CH, std::min<uint64_t>(g.nv, (c + 1) * CH), sl.buf);
                else edges((c - v_chunks) * CH, std::min<uint64_t>(g.ne, (c - v_chunks + 1) * CH), sl.buf);
                sl.full.store(1, std::memory_order_release);
            }
        } catch (...) {  // (a buffer could not grow: the writer below stops waiting, the call reports MXG_ENOMEM)
            failed = true;
            stop = true;
        }
    };
    {
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < T; ++t) th.emplace_back(worker, t);
        for (uint64_t c = 0; c < n_chunks && ok; ++c) {
            Slot &sl = slot_of(c);
            const double tw0 = dbg_io ? now_s() : 0.0;
            while (!sl.full.load(std::memory_order_acquire) && !stop) std::this_thread::yield();
            if (stop) break;
            if (dbg_io) t_wait += now_s() - tw0;
            size_t done = 0;
            while (done < sl.buf.n) {
                const ssize_t wr = write(fd, sl.buf.b.data() + done, sl.buf.n - done);
                if (wr <= 0) {
                    ok = false;
                    break;
                }
                done += (size_t)wr;
            }
            sl.full.store(0, std::memory_order_release);
        }
        if (!ok) stop = true;
        for (auto &x : th) x.join();
    }
    ok = ok && !failed && write(fd, "}\n", 2) == 2;
    ok = (fclose(f) == 0) && ok;
    if (dbg_io)
        fprintf(stderr, "[mxg] write_dot: %.3f s, of which the writer waited %.3f s for formatted chunks (%u workers, %llu chunks)\n",
                now_s() - t_begin, t_wait, T, (unsigned long long)n_chunks);
    if (failed) return set_err(h, MXG_ENOMEM, "out of host memory formatting '%s'", path);
    if (!ok) return set_err(h, MXG_EIO, "write error on '%s'", path);
    return MXG_OK;
}

// ---- binary side-car of a sketch -------------------------------------------------------------------
// The TSV stays the interchange / checkpoint format (ntJoin:202 .SECONDARY); next to it the library can leave the same
// sketch as raw arrays, so that a later run (or the Python side) does not parse 65 bytes of text per minimizer again.
// Layout (little endian): magic "MXGSKB1\0"; u32 k, w, variant, 0; u64 n_records, n_mx, id_bytes;
// u64 record_len[n_records]; u64 record_first[n_records + 1]; char ids[id_bytes] ('\n'-terminated, in record order);
// padding to 8; u64 out_hash[n_mx]; u32 pos[n_mx].
static const char SKB_MAGIC[8] = {'M', 'X', 'G', 'S', 'K', 'B', '1', '\0'};

int write_sketch_bin(mxg_handle *h, Assembly *a, const char *path)
{
    int rc = sync_sketch_to_host(h, a);
    if (rc != MXG_OK) return rc;
    FILE *f = fopen(path, "wb");
    if (!f) return set_err(h, MXG_EIO, "cannot open '%s' for writing", path);
    const uint64_t nr = a->recs.size(), n = a->n_mx;
    std::string ids;
    std::vector<uint64_t> lens(nr);
    for (uint64_t r = 0; r < nr; ++r) {
        ids += a->recs[r].id;
        ids += '\n';
        lens[r] = a->recs[r].len;
    }
    const uint32_t head32[4] = {h->cfg.k, h->cfg.w, h->cfg.variant, 0u};
    const uint64_t head64[3] = {nr, n, (uint64_t)ids.size()};
    const char pad[8] = {0};
    bool ok = fwrite(SKB_MAGIC, 1, 8, f) == 8 && fwrite(head32, 4, 4, f) == 4 && fwrite(head64, 8, 3, f) == 3;
    ok = ok && (nr == 0 || fwrite(lens.data(), 8, nr, f) == nr);
    ok = ok && fwrite(a->rec_first.data(), 8, nr + 1, f) == nr + 1;
    ok = ok && (ids.empty() || fwrite(ids.data(), 1, ids.size(), f) == ids.size());
    ok = ok && fwrite(pad, 1, (8 - ids.size() % 8) % 8, f) == (8 - ids.size() % 8) % 8;
    ok = ok && (n == 0 || (fwrite(a->h_hash.data(), 8, n, f) == n && fwrite(a->h_pos.data(), 4, n, f) == n));
    ok = (fclose(f) == 0) && ok;
    return ok ? MXG_OK : set_err(h, MXG_EIO, "write error on '%s'", path);
}

int load_sketch_bin(mxg_handle *h, Assembly *a, const char *path, std::vector<uint64_t> &hash, std::vector<uint32_t> &pos,
                    std::vector<uint32_t> &rec)
{
    FILE *f = fopen(path, "rb");
    if (!f) return set_err(h, MXG_EIO, "cannot open '%s'", path);
    char magic[8];
    uint32_t head32[4];
    uint64_t head64[3];
    int rc = MXG_OK;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, SKB_MAGIC, 8) != 0 || fread(head32, 4, 4, f) != 4 ||
        fread(head64, 8, 3, f) != 3)
        rc = set_err(h, MXG_EIO, "'%s' is not a sketch side-car (bad header)", path);
    else if (head32[0] != h->cfg.k || head32[2] != h->cfg.variant)
        rc = set_err(h, MXG_EINVAL, "'%s' was written with k=%u variant=%u, this handle has k=%u variant=%u", path, head32[0],
                     head32[2], h->cfg.k, h->cfg.variant);
    if (rc != MXG_OK) {
        fclose(f);
        return rc;
    }
    const uint64_t nr = head64[0], n = head64[1], idb = head64[2];
    std::vector<uint64_t> lens(nr), first(nr + 1);
    std::string ids(idb, '\0');
    hash.resize(n);
    pos.resize(n);
    rec.resize(n);
    bool ok = (nr == 0 || fread(lens.data(), 8, nr, f) == nr) && fread(first.data(), 8, nr + 1, f) == nr + 1 &&
              (idb == 0 || fread(&ids[0], 1, idb, f) == idb) && fseek(f, (long)((8 - idb % 8) % 8), SEEK_CUR) == 0 &&
              (n == 0 || (fread(hash.data(), 8, n, f) == n && fread(pos.data(), 4, n, f) == n));
    fclose(f);
    if (!ok || first[0] != 0 || first[nr] != n) return set_err(h, MXG_EIO, "'%s' is truncated or inconsistent", path);
    // like the TSV route (read_minimizers skips lines without minimizers, bin/ntjoin_utils.py:176), records without
    // minimizers do not enter the record table
    size_t at = 0;
    for (uint64_t r = 0; r < nr; ++r) {
        const size_t e = ids.find('\n', at);
        if (e == std::string::npos || first[r + 1] < first[r]) return set_err(h, MXG_EIO, "'%s': bad record table", path);
        if (first[r + 1] > first[r]) {
            Record rc_;
            rc_.id = ids.substr(at, e - at);
            rc_.len = lens[r];
            for (uint64_t i = first[r]; i < first[r + 1]; ++i) rec[i] = (uint32_t)a->recs.size();
            a->recs.push_back(rc_);
        }
        at = e + 1;
    }
    return MXG_OK;
}

}  // namespace mxg
