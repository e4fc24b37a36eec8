// ingest.hip -- FASTA text -> 2-bit packed bases + valid-run table ON THE DEVICE, and the TSV text of a sketch formatted
// ON THE DEVICE: the file route of the sketch stage (replaces what `indexlr` does around its hashing loop: reading the
// FASTA with `-t` threads and printing `id \t hash:pos:seq ...`, reference ntJoin:204-205, bin/ntjoin_utils.py:195-202).
//
// Why on the device: a 3 Gbp assembly is 3 GB of text in and 0.4 GB of text out.  One host core parses ~1 GB/s and formats
// ~0.4 GB/s, which made the file route 0.5 Gbp/s while the kernels run at hundreds of Gbp/s.  Here the host only
//   (1) finds the header lines (memchr for '>' over the mmap'ed file, `threads` workers),
//   (2) copies the raw text through pinned staging buffers into HBM (`threads` workers feeding hipMemcpyAsync),
//   (3) adds up ~10^6 per-tile base counts into record lengths and packed offsets,
// and the device classifies every byte (line breaks skipped, ACGTU any case = a base, anything else = an invalid base),
// packs the bases, and reports where validity changes (the run table of SURVEY.md A.3: k-mers over invalid bases do not
// exist).  The raw text stays in HBM, so the k-mer column of the TSV is printed exactly as the file spells it.
//
// TSV: entry lengths -> exclusive scan -> every byte of the file has a known offset; the text is produced in windows of
// TSV_WIN bytes (double-buffered: the device formats window c+1 while the host writes window c).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "mxg_internal.h"
#include "scan_kernels.h"

namespace mxg {

constexpr uint32_t ING_TILE = 4096;  // text bytes per work item: one aligned tile of the file, cut at record borders

uint32_t host_threads(const mxg_handle *h)
{
    uint32_t t = h->cfg.host_threads;
    if (!t) t = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    return std::min(t, 256u);
}

struct IngItem {
    uint64_t lo;   // first text byte
    uint32_t len;  // bytes (all inside one tile)
    uint32_t rec;  // record
};

// byte class: 0..3 = A C G T(U) (either case), 4 = any other base character (invalid), 5 = line break (not a base)
__device__ __forceinline__ uint32_t byte_class(uint32_t b)
{
    if (b == '\n' || b == '\r') return 5u;
    const uint32_t u = b & 0xDFu;  // upper case
    return u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : (u == 'T' || u == 'U') ? 3u : 4u;
}

// thread t of an item's block owns text bytes [tile + 16 t, tile + 16 t + 16): -> base codes (2 bits each, in order),
// validity bits, count
__device__ __forceinline__ void classify16(const unsigned char *__restrict__ text, const IngItem it, uint32_t &bits, uint32_t &valid,
                                           uint32_t &cnt)
{
    const uint64_t tile = it.lo & ~(uint64_t)(ING_TILE - 1);
    const uint64_t a0 = tile + 16u * threadIdx.x;
    bits = valid = cnt = 0;
    if (a0 + 16 <= it.lo || a0 >= it.lo + it.len) return;
    const uint4 v = *reinterpret_cast<const uint4 *>(text + a0);  // (the text buffer is padded to whole tiles)
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (uint32_t j = 0; j < 16; ++j) {
        const uint64_t a = a0 + j;
        const uint32_t c = byte_class((w[j >> 2] >> (8u * (j & 3u))) & 255u);
        if (a >= it.lo && a < it.lo + it.len && c != 5u) {
            bits |= (c & 3u) << (2u * cnt);
            valid |= (c < 4u ? 1u : 0u) << cnt;
            ++cnt;
        }
    }
}

__global__ __launch_bounds__(256) void k_ing_count(const unsigned char *__restrict__ text, const IngItem *__restrict__ items,
                                                   uint32_t *__restrict__ item_cnt, uint16_t *__restrict__ item_sub)
{
    __shared__ uint32_t tot;
    if (threadIdx.x == 0) tot = 0;
    __syncthreads();
    uint32_t bits, valid, cnt;
    classify16(text, items[blockIdx.x], bits, valid, cnt);
    uint32_t s = cnt;  // sum over the 16 threads of a 256-byte sub-tile
    s += (uint32_t)__shfl_xor((int)s, 8, 64);
    s += (uint32_t)__shfl_xor((int)s, 4, 64);
    s += (uint32_t)__shfl_xor((int)s, 2, 64);
    s += (uint32_t)__shfl_xor((int)s, 1, 64);
    if ((threadIdx.x & 15u) == 0) {
        item_sub[(size_t)blockIdx.x * 16u + (threadIdx.x >> 4)] = (uint16_t)s;
        if (s) atomicAdd(&tot, s);
    }
    __syncthreads();
    if (threadIdx.x == 0) item_cnt[blockIdx.x] = tot;
}

struct IngEvent {
    uint64_t at;     // packed base index where validity changes
    uint32_t state;  // new state: 1 = valid
    uint32_t item;
};

__global__ __launch_bounds__(256) void k_ing_pack(const unsigned char *__restrict__ text, const IngItem *__restrict__ items,
                                                  const uint64_t *__restrict__ item_pbase, uint32_t *__restrict__ packed,
                                                  uint8_t *__restrict__ first_valid, uint8_t *__restrict__ last_valid,
                                                  IngEvent *__restrict__ events, uint32_t ev_cap, uint32_t *__restrict__ ev_count)
{
    __shared__ uint32_t sh[256];
    __shared__ uint32_t vb[ING_TILE / 32 + 1];  // validity of the item's bases by local index
    const IngItem it = items[blockIdx.x];
    uint32_t bits, valid, cnt;
    classify16(text, it, bits, valid, cnt);
    if (threadIdx.x <= ING_TILE / 32) vb[threadIdx.x] = 0;
    const bool all_ok = __syncthreads_and(valid == (cnt ? (0xFFFFFFFFu >> (32u - cnt)) : 0u)) != 0;
    const uint32_t l0 = block_exclusive_256(cnt, sh);
    const uint32_t total = sh[255];
    const uint64_t pbase = item_pbase[blockIdx.x];
    if (cnt) {
        const uint64_t g = pbase + l0;
        const uint64_t v = (uint64_t)bits << (2u * ((uint32_t)g & 15u));
        atomicOr(&packed[g >> 4], (uint32_t)v);
        if (v >> 32) atomicOr(&packed[(g >> 4) + 1], (uint32_t)(v >> 32));
    }
    if (all_ok) {  // (block-uniform) the common case: nothing but ACGTU in this tile
        if (threadIdx.x == 0) first_valid[blockIdx.x] = last_valid[blockIdx.x] = total ? 1 : 2;
        return;
    }
    if (cnt) {
        const uint64_t m = (uint64_t)valid << (l0 & 31u);
        atomicOr(&vb[l0 >> 5], (uint32_t)m);
        if (m >> 32) atomicOr(&vb[(l0 >> 5) + 1], (uint32_t)(m >> 32));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        first_valid[blockIdx.x] = total ? (uint8_t)(vb[0] & 1u) : 2;
        last_valid[blockIdx.x] = total ? (uint8_t)((vb[(total - 1) >> 5] >> ((total - 1) & 31u)) & 1u) : 2;
    }
    for (uint32_t u = 0; u < cnt; ++u) {
        const uint32_t i = l0 + u;
        if (i == 0) continue;  // (the host compares an item's first base with the item before it)
        const uint32_t cur = (vb[i >> 5] >> (i & 31u)) & 1u, prev = (vb[(i - 1) >> 5] >> ((i - 1) & 31u)) & 1u;
        if (cur != prev) {
            const uint32_t e = atomicAdd(ev_count, 1u);
            if (e < ev_cap) events[e] = IngEvent{pbase + i, cur, blockIdx.x};
        }
    }
}

static std::string header_token(const unsigned char *p, size_t n)
{
    size_t e = 0;
    while (e < n && p[e] != ' ' && p[e] != '\t' && p[e] != '\r' && p[e] != '\n') ++e;
    return std::string(reinterpret_cast<const char *>(p), e);
}

template <class F> static void parallel_for(uint32_t n_threads, F f)
{
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < n_threads; ++t) th.emplace_back(f, t);
    f(0u);
    for (auto &x : th) x.join();
}

// returns MXG_OK, a negative error, or 1: "not for this route" (the caller falls back to the host parser)
int load_fasta_device(mxg_handle *h, Assembly *a, const char *path, uint32_t n_threads)
{
    const uint32_t k = h->cfg.k, w = h->cfg.w;
    n_threads = std::max(1u, std::min(n_threads, 64u));
    const bool dbg_io = getenv("MXG_DEBUG_IO") != nullptr;  // phase timings on stderr
    auto now_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tp0 = now_s();
    double tp_alloc = 0, tp_pool = 0, tp_hdr = 0, tp_up = 0, tp_count = 0, tp_pack = 0;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return set_err(h, MXG_EIO, "cannot open FASTA '%s'", path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {  // pipes etc.: the streaming host parser
        close(fd);
        return 1;
    }
    const uint64_t fsz = (uint64_t)sb.st_size;
    unsigned char magic[2] = {0, 0};
    const bool gz = fsz >= 2 && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    if (fsz == 0 || gz) {  // (the host parser produces the empty assembly / inflates gzip input)
        close(fd);
        return 1;
    }
    const unsigned char *txt = static_cast<const unsigned char *>(mmap(nullptr, fsz, PROT_READ, MAP_PRIVATE, fd, 0));
    close(fd);
    if (txt == MAP_FAILED) return 1;
    (void)madvise(const_cast<unsigned char *>(txt), fsz, MADV_SEQUENTIAL);
    // (taking a 3 GB mapping apart costs ~25 ms: a handle that runs once -- MXG_FLAG_ONE_SHOT, the CLI -- leaves it to
    // mxg_destroy or, as mxgraph does, to the end of a process nobody waits for)
    struct Unmap {
        const unsigned char *p;
        uint64_t n;
        mxg_handle *keep;
        ~Unmap()
        {
            if (keep) {
                try {
                    keep->kept_maps.emplace_back(const_cast<unsigned char *>(p), (size_t)n);
                    return;
                } catch (...) {
                }
            }
            munmap(const_cast<unsigned char *>(p), n);
        }
    } unmap{txt, fsz, (h->cfg.flags & MXG_FLAG_ONE_SHOT) && !getenv("MXG_UNMAP_EARLY") ? h : nullptr};
    MXG_HIP(h, hipSetDevice(h->device));

    // ---- (2) raw text -> HBM through pinned staging buffers, started first so that it overlaps the header scan ----
    // A device allocation that fails here (the route keeps ~1 byte per base of text in HBM) is no error: the host parser
    // takes the file instead (return 1; nothing has been committed to `a` that the caller's delete / new does not undo).
    auto dev_fallback = [&](hipError_t e) {
        (void)hipGetLastError();
        return e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation;
    };
    const uint64_t text_alloc = ((fsz + ING_TILE - 1) / ING_TILE + 1) * ING_TILE;
    {
        const hipError_t e = a->d_text.ensure(text_alloc);
        if (e != hipSuccess) return dev_fallback(e) ? 1 : set_err(h, MXG_EDEVICE, "device allocation failed: %s", hipGetErrorString(e));
    }
    unsigned char *d_text = a->d_text.as<unsigned char>();
    tp_alloc = now_s() - tp0;
    constexpr uint64_t STAGE = 32ull << 20;
    constexpr int NB = 4;
    // staging resources + the uploader thread: joined and released on every way out of this function (an exception thrown
    // by the host-side section below must not unwind through a joinable std::thread)
    struct Staging {
        unsigned char *buf[NB] = {nullptr, nullptr, nullptr, nullptr};
        hipEvent_t ev[NB] = {nullptr, nullptr, nullptr, nullptr};
        hipStream_t cs = nullptr;
        std::thread uploader;
        void finish()
        {
            if (uploader.joinable()) uploader.join();
        }
        ~Staging()
        {
            finish();
            for (int b = 0; b < NB; ++b)
                if (ev[b]) (void)hipEventDestroy(ev[b]);  // (the buffers are the handle's: pin_pool)
            if (cs) (void)hipStreamDestroy(cs);
        }
    } sg;
    unsigned char **stage = sg.buf;
    hipEvent_t *sev = sg.ev;
    MXG_HIP(h, hipStreamCreateWithFlags(&sg.cs, hipStreamNonBlocking));
    hipStream_t cs = sg.cs;
    static_assert(NB * STAGE <= PIN_POOL_BYTES, "staging buffers come out of the handle's pinned pool");
    {
        // (the pool is pinned piece by piece by a thread of the handle; a staging buffer = a piece: the upload starts with the first)
        static_assert(STAGE == PIN_PIECE_BYTES && NB <= (int)PIN_PIECES, "a staging buffer is one piece of the pinned pool");
        hipError_t e = pin_pool_start(h);
        if (e == hipSuccess) e = pin_pool_wait(h, 1);
        if (e != hipSuccess && (e = pin_pool_start(h)) == hipSuccess) e = pin_pool_wait(h, 1);  // (registering failed: one allocation)
        if (e != hipSuccess) return dev_fallback(e) ? 1 : set_err(h, MXG_EDEVICE, "pinned allocation failed: %s", hipGetErrorString(e));
        unsigned char *pool = static_cast<unsigned char *>(h->pin_pool);
        tp_pool = now_s() - tp0;
        for (int b = 0; b < NB; ++b) {
            stage[b] = pool + (size_t)b * STAGE;
            MXG_HIP(h, hipEventCreateWithFlags(&sev[b], hipEventDisableTiming));
        }
    }
    hipError_t uerr = hipSuccess;
    sg.uploader = std::thread([&]() {
        (void)hipSetDevice(h->device);
        const uint32_t T = std::max(1u, n_threads - 1u);
        uint64_t c = 0;
        uint32_t ring = NB;  // staging buffers in use: all of them, unless pinning stopped part of the way (then what is pinned: >= 1)
        bool used[NB] = {false, false, false, false};
        for (uint64_t off = 0; off < fsz && uerr == hipSuccess; off += STAGE, ++c) {
            if (c < ring && pin_pool_wait(h, (uint32_t)c + 1u) != hipSuccess)  // (its piece of the pool may still be on its way)
                ring = std::max(1u, std::min((uint32_t)c, h->pin_ready.load()));
            const int b = (int)(c % ring);
            if (used[b]) uerr = hipEventSynchronize(sev[b]);
            used[b] = true;
            if (uerr != hipSuccess) break;
            const uint64_t n = std::min(STAGE, fsz - off);
            try {
                parallel_for(T, [&](uint32_t t) {
                    const uint64_t lo = n * t / T, hi = n * (t + 1) / T;
                    memcpy(stage[b] + lo, txt + off + lo, hi - lo);
                });
            } catch (const std::exception &) {  // (no thread to be had: copy alone)
                memcpy(stage[b], txt + off, n);
            }
            if (uerr == hipSuccess) uerr = hipMemcpyAsync(d_text + off, stage[b], n, hipMemcpyHostToDevice, cs);
            if (uerr == hipSuccess) uerr = hipEventRecord(sev[b], cs);
        }
        if (uerr == hipSuccess) uerr = hipMemsetAsync(d_text + fsz, '\n', text_alloc - fsz, cs);
        if (uerr == hipSuccess) uerr = hipStreamSynchronize(cs);
    });

    // ---- (1) header lines: '>' at the start of a line ----
    struct Hdr {
        uint64_t at, end;  // '>' and its line's '\n' (or the file's end)
    };
    std::vector<std::vector<Hdr>> found(n_threads);
    parallel_for(n_threads, [&](uint32_t t) {
        const uint64_t lo = fsz * t / n_threads, hi = fsz * (t + 1) / n_threads;
        uint64_t p = lo;
        while (p < hi) {
            const unsigned char *q = static_cast<const unsigned char *>(memchr(txt + p, '>', hi - p));
            if (!q) break;
            const uint64_t at = (uint64_t)(q - txt);
            if (at == 0 || txt[at - 1] == '\n') {
                const unsigned char *nl = static_cast<const unsigned char *>(memchr(q, '\n', fsz - at));
                const uint64_t end = nl ? (uint64_t)(nl - txt) : fsz;
                found[t].push_back(Hdr{at, end});
                p = end + 1;  // (a '>' inside the header line is not a header)
            } else {
                p = at + 1;
            }
        }
    });
    std::vector<Hdr> hdr;
    for (auto &v : found)
        for (auto &x : v)
            if (hdr.empty() || x.at > hdr.back().end) hdr.push_back(x);  // (a chunk border inside a header line)
    const size_t n_rec = hdr.size();
    if (n_rec >= (1ull << 32)) return set_err(h, MXG_ELIMIT, "too many records in '%s'", path);
    // ---- work items: the tiles every record's sequence text overlaps ----
    std::vector<IngItem> items;
    std::vector<uint64_t> rec_item0(n_rec + 1);
    a->recs.resize(n_rec);
    for (size_t r = 0; r < n_rec; ++r) {
        const uint64_t sb0 = std::min(hdr[r].end + 1, fsz), se = r + 1 < n_rec ? hdr[r + 1].at : fsz;
        a->recs[r].id = header_token(txt + hdr[r].at + 1, hdr[r].end - hdr[r].at - 1);
        a->recs[r].text_off = sb0;
        rec_item0[r] = items.size();
        for (uint64_t p = sb0; p < se;) {
            const uint64_t tile_end = (p / ING_TILE + 1) * ING_TILE, e = std::min(se, tile_end);
            items.push_back(IngItem{p, (uint32_t)(e - p), (uint32_t)r});
            p = e;
        }
    }
    rec_item0[n_rec] = items.size();
    const size_t n_items = items.size();
    if (n_items >= (1ull << 31)) return set_err(h, MXG_ELIMIT, "'%s' is too large for one handle", path);
    hipStream_t st = h->stream;
    DevBuf &d_items = a->d_ing_items, &d_cnt = a->d_ing_cnt, &d_sub = a->d_ing_sub, &d_pbase = a->d_ing_pbase;
    int rc = MXG_OK;
    if (n_items) {
        hipError_t e;
        if ((e = d_items.ensure(n_items * sizeof(IngItem))) != hipSuccess || (e = d_cnt.ensure(n_items * 4)) != hipSuccess ||
            (e = d_sub.ensure(n_items * 32)) != hipSuccess || (e = d_pbase.ensure(n_items * 8)) != hipSuccess)
            return dev_fallback(e) ? 1 : set_err(h, MXG_EDEVICE, "device allocation failed: %s", hipGetErrorString(e));
        if ((e = hipMemcpyAsync(d_items.p, items.data(), n_items * sizeof(IngItem), hipMemcpyHostToDevice, st)) != hipSuccess)
            return set_err(h, MXG_EDEVICE, "upload failed: %s", hipGetErrorString(e));
    }
    tp_hdr = now_s() - tp0;
    sg.finish();  // the text is in HBM
    tp_up = now_s() - tp0;
    if (uerr != hipSuccess) return set_err(h, MXG_EDEVICE, "text upload failed: %s", hipGetErrorString(uerr));
    std::vector<uint32_t> cnt(n_items);
    if (n_items) {
        hipLaunchKernelGGL(k_ing_count, dim3((uint32_t)n_items), dim3(256), 0, st, d_text, d_items.as<IngItem>(), d_cnt.as<uint32_t>(),
                           d_sub.as<uint16_t>());
        MXG_HIP(h, hipGetLastError());
        MXG_HIP(h, hipMemcpyAsync(cnt.data(), d_cnt.p, n_items * 4, hipMemcpyDeviceToHost, st));
        MXG_HIP(h, hipStreamSynchronize(st));
    }
    tp_count = now_s() - tp0;
    // ---- (3) record lengths, packed layout (every record starts at a multiple of 16 bases) ----
    std::vector<uint64_t> pbase(n_items);
    uint64_t cur = 0;
    for (size_t r = 0; r < n_rec; ++r) {
        Record &rec = a->recs[r];
        rec.base_off = cur;
        uint64_t len = 0;
        for (uint64_t q = rec_item0[r]; q < rec_item0[r + 1]; ++q) {
            pbase[q] = cur + len;
            len += cnt[q];
        }
        rec.len = len;
        if (len >= (1ull << 32))
            return set_err(h, MXG_ELIMIT, "record '%s' has %llu bases; the engine indexes positions with 32 bits", rec.id.c_str(),
                           (unsigned long long)len);
        a->total_bases += len;
        cur += (len + 15) & ~15ull;
    }
    const size_t pad = 256 + (k + 15) / 16 + 16;  // the hash kernel reads up to one strip + k bases past a run's end
    a->packed_words = cur / 16 + pad;
    {
        const hipError_t e = a->d_packed_own.ensure(a->packed_words * 4);
        if (e != hipSuccess) return dev_fallback(e) ? 1 : set_err(h, MXG_EDEVICE, "device allocation failed: %s", hipGetErrorString(e));
    }
    MXG_HIP(h, hipMemsetAsync(a->d_packed_own.p, 0, a->packed_words * 4, st));
    // changes between valid and invalid bases inside the tiles; MXG_INGEST_EV_CAP: test knob
    const uint32_t EV_CAP = (uint32_t)std::max<uint64_t>(1, knob_u64(h, "MXG_INGEST_EV_CAP", 4u << 20));
    std::vector<uint8_t> fv(n_items), lv(n_items);
    std::vector<IngEvent> events;
    if (n_items) {
        DevBuf d_fv, d_lv, d_ev, d_evn;
        MXG_HIP(h, d_fv.ensure(n_items));
        MXG_HIP(h, d_lv.ensure(n_items));
        MXG_HIP(h, d_ev.ensure((size_t)EV_CAP * sizeof(IngEvent)));
        MXG_HIP(h, d_evn.ensure(16));
        MXG_HIP(h, hipMemsetAsync(d_evn.p, 0, 16, st));
        MXG_HIP(h, hipMemcpyAsync(d_pbase.p, pbase.data(), n_items * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_ing_pack, dim3((uint32_t)n_items), dim3(256), 0, st, d_text, d_items.as<IngItem>(), d_pbase.as<uint64_t>(),
                           a->d_packed_own.as<uint32_t>(), d_fv.as<uint8_t>(), d_lv.as<uint8_t>(), d_ev.as<IngEvent>(), EV_CAP,
                           d_evn.as<uint32_t>());
        MXG_HIP(h, hipGetLastError());
        uint32_t n_ev = 0;
        MXG_HIP(h, hipMemcpyAsync(fv.data(), d_fv.p, n_items, hipMemcpyDeviceToHost, st));
        MXG_HIP(h, hipMemcpyAsync(lv.data(), d_lv.p, n_items, hipMemcpyDeviceToHost, st));
        MXG_HIP(h, hipMemcpyAsync(&n_ev, d_evn.p, 4, hipMemcpyDeviceToHost, st));
        MXG_HIP(h, hipStreamSynchronize(st));
        // more changes than the buffer holds (drafts with scattered N / IUPAC codes): the host parser has no such limit
        if (n_ev > EV_CAP) return 1;
        events.resize(n_ev);
        if (n_ev) {
            MXG_HIP(h, hipMemcpy(events.data(), d_ev.p, (size_t)n_ev * sizeof(IngEvent), hipMemcpyDeviceToHost));
            std::sort(events.begin(), events.end(), [](const IngEvent &x, const IngEvent &y) { return x.at < y.at; });
        }
    }
    a->d_packed = a->d_packed_own.as<uint32_t>();
    tp_pack = now_s() - tp0;
    // ---- the run table: maximal stretches of valid bases holding at least one k-mer ----
    size_t ev_i = 0;
    for (size_t r = 0; r < n_rec; ++r) {
        const Record &rec = a->recs[r];
        std::vector<std::pair<uint32_t, uint32_t>> rec_runs;  // (pos0, n_kmers)
        uint32_t state = 0;
        uint64_t run_start = 0;
        auto flip = [&](uint64_t at_rel, uint32_t to) {
            if (to == state) return;
            if (to) run_start = at_rel;
            else if (at_rel - run_start >= k) rec_runs.emplace_back((uint32_t)run_start, (uint32_t)(at_rel - run_start - k + 1));
            state = to;
        };
        for (uint64_t q = rec_item0[r]; q < rec_item0[r + 1]; ++q) {
            if (fv[q] == 2) continue;  // no base in this tile
            flip(pbase[q] - rec.base_off, fv[q]);
            while (ev_i < events.size() && events[ev_i].item < q) ++ev_i;  // (cannot happen: events are in item order)
            while (ev_i < events.size() && events[ev_i].item == q) {
                flip(events[ev_i].at - rec.base_off, events[ev_i].state);
                ++ev_i;
            }
            // (the tile's last state follows from its events; lv[] is only a cross-check)
            if (state != lv[q]) return set_err(h, MXG_EDEVICE, "internal error: validity tracking out of step in '%s'", path);
        }
        flip(rec.len, 0);
        uint64_t nk = 0;
        for (auto &rr : rec_runs) nk += rr.second;
        if (nk >= w && nk > 0) {
            const uint32_t ctg = (uint32_t)a->ctg_rec.size();
            a->ctg_rec.push_back((uint32_t)r);
            a->ctg_nk.push_back((uint32_t)nk);
            a->ctg_run0.push_back((uint32_t)a->runs.size());
            uint32_t kidx = 0;
            for (auto &rr : rec_runs) {
                Run run;
                run.base_off = rec.base_off + rr.first;
                run.n_kmers = rr.second;
                run.contig = ctg;
                run.kidx0 = kidx;
                run.pos0 = rr.first;
                kidx += rr.second;
                a->runs.push_back(run);
            }
            a->total_kmers += nk;
        }
    }
    a->ctg_run0.push_back((uint32_t)a->runs.size());
    a->has_bases = true;
    a->has_text = false;
    a->text_on_device = true;
    a->text_bytes = fsz;
    a->ing_item0.assign(rec_item0.begin(), rec_item0.end());
    if (dbg_io)
        fprintf(stderr, "[mxg] load_fasta_device %s: %.3f s = open + map + text buffer %.3f, first pinned buffer at %.3f, headers + items (beside the "
                        "upload) until %.3f, upload done at %.3f, base counts at %.3f, packed at %.3f, run table at %.3f (%.2f GB)\n", path, now_s() - tp0,
                tp_alloc, tp_pool, tp_hdr, tp_up, tp_count, tp_pack, now_s() - tp0, fsz / 1e9);
    if (h->cfg.flags & MXG_FLAG_DROP_SEQ) {  // the caller does not want the text kept: k-mers are then printed from the packed bases
        a->d_text.release();
        a->d_ing_items.release();
        a->d_ing_sub.release();
        a->d_ing_pbase.release();
        a->d_ing_cnt.release();
        a->text_on_device = false;
    }
    return rc;
}

// ======================================================================================================
// TSV text on the device
// ======================================================================================================
__device__ __forceinline__ uint32_t dec_digits(uint64_t v)
{
    uint32_t d = 1;
    while (v >= 10u) {
        v /= 10u;
        ++d;
    }
    return d;
}

struct TsvParams {
    uint64_t n;               // minimizers
    const uint64_t *hash;
    const uint32_t *pos, *rec;
    const uint8_t *fwd;
    uint32_t with_pos, with_strand, with_seq, k;
    const uint64_t *rec_first;   // [n_rec + 1]
    const uint64_t *rec_prefix;  // [n_rec]: bytes of the id columns (and of the lines of records without minimizers) before record r
    const uint32_t *id_off;      // [n_rec + 1] into ids
    const char *ids;
    uint64_t n_rec;
    uint8_t *len;             // [n] entry length incl. its separator
    const uint64_t *off;      // [n + 1] exclusive scan of len
    // k-mer text: the FASTA's own spelling (text in HBM) or decoded from the packed bases
    const unsigned char *text;
    const IngItem *items;
    const uint64_t *item_pbase;
    const uint16_t *item_sub;
    const uint64_t *rec_item0;
    const uint64_t *rec_base;    // [n_rec] packed base offset of record r
    const uint32_t *packed;
    // output window
    char *out;
    uint64_t win_lo, win_hi;
};

__global__ __launch_bounds__(256) void k_tsv_len(const TsvParams p)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= p.n) return;
    uint32_t l = dec_digits(p.hash[i]) + 1u;  // + separator (' ' or the line's '\n')
    if (p.with_pos) l += 1u + dec_digits(p.pos[i]);
    if (p.with_strand) l += 2u;
    if (p.with_seq) l += 1u + p.k;
    p.len[i] = (uint8_t)l;  // (k <= 200 on this route: the host writer takes longer k-mers)
}

// exclusive scan of u8 lengths into u64 offsets: tile sums -> one block scans them (u64) -> tiles add their base
__global__ __launch_bounds__(256) void k_tsv_tile_sum(const uint8_t *__restrict__ len, uint64_t n, uint32_t *__restrict__ tsum)
{
    __shared__ uint32_t sh[256];
    const uint64_t base = (uint64_t)blockIdx.x * TILE + (uint64_t)threadIdx.x * TILE_PER_THREAD;
    uint32_t c = 0;
    for (int u = 0; u < TILE_PER_THREAD; ++u)
        if (base + u < n) c += len[base + u];
    (void)block_exclusive_256(c, sh);
    if (threadIdx.x == 0) tsum[blockIdx.x] = sh[255];
}
__global__ __launch_bounds__(256) void k_tsv_scan_tiles(const uint32_t *__restrict__ tsum, uint32_t n_tiles, uint64_t *__restrict__ tbase,
                                                        uint64_t *__restrict__ total)
{
    __shared__ uint32_t sh[256];
    uint64_t carry = 0;
    for (uint32_t b0 = 0; b0 < n_tiles; b0 += 256u * 16u) {
        const uint32_t i0 = b0 + threadIdx.x * 16u;
        uint32_t v[16], c = 0;
        for (int u = 0; u < 16; ++u) {
            v[u] = i0 + u < n_tiles ? tsum[i0 + u] : 0u;
            c += v[u];  // (a tile holds at most 1024 x 255 bytes, 4096 tiles per pass: fits 32 bits)
        }
        uint64_t run = carry + block_exclusive_256(c, sh);
        const uint32_t pass = sh[255];
        for (int u = 0; u < 16; ++u) {
            if (i0 + u < n_tiles) tbase[i0 + u] = run;
            run += v[u];
        }
        carry += pass;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(256) void k_tsv_offsets(const uint8_t *__restrict__ len, uint64_t n, const uint64_t *__restrict__ tbase,
                                                     const uint64_t *__restrict__ total, uint64_t *__restrict__ off)
{
    __shared__ uint32_t sh[256];
    const uint64_t base = (uint64_t)blockIdx.x * TILE + (uint64_t)threadIdx.x * TILE_PER_THREAD;
    uint32_t v[TILE_PER_THREAD], c = 0;
    for (int u = 0; u < TILE_PER_THREAD; ++u) {
        v[u] = base + u < n ? len[base + u] : 0u;
        c += v[u];
    }
    uint64_t run = tbase[blockIdx.x] + block_exclusive_256(c, sh);
    for (int u = 0; u < TILE_PER_THREAD; ++u) {
        if (base + u < n) off[base + u] = run;
        run += v[u];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) off[n] = *total;
}

__device__ __forceinline__ void put_win(const TsvParams &p, uint64_t at, char c)
{
    if (at >= p.win_lo && at < p.win_hi) p.out[at - p.win_lo] = c;
}
__device__ __forceinline__ uint64_t put_dec(const TsvParams &p, uint64_t at, uint64_t v)
{
    const uint32_t d = dec_digits(v);
    for (uint32_t u = 0; u < d; ++u) {
        put_win(p, at + d - 1u - u, (char)('0' + (uint32_t)(v % 10u)));
        v /= 10u;
    }
    return at + d;
}

// text offset of base `pos` of record r: the tile by binary search over the tiles' first base indices, the 256-byte
// sub-tile by its 16 counts, then a scan over at most 256 bytes
__device__ __forceinline__ uint64_t text_of_base(const TsvParams &p, uint32_t r, uint32_t pos)
{
    const uint64_t want = p.rec_base[r] + pos;
    uint64_t lo = p.rec_item0[r], hi = p.rec_item0[r + 1];
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (p.item_pbase[mid] <= want) lo = mid; else hi = mid;
    }
    const IngItem it = p.items[lo];
    uint32_t local = (uint32_t)(want - p.item_pbase[lo]);
    const uint64_t tile = it.lo & ~(uint64_t)(ING_TILE - 1);
    uint32_t s = 0;
    for (; s < 15; ++s) {
        const uint32_t c = p.item_sub[lo * 16u + s];
        if (local < c) break;
        local -= c;
    }
    uint64_t a = tile + 256u * s;
    if (a < it.lo) a = it.lo;
    for (;; ++a) {  // (the base is there: the counts say so)
        const uint32_t c = byte_class(p.text[a]);
        if (c != 5u) {
            if (local == 0) return a;
            --local;
        }
    }
}

__global__ __launch_bounds__(256) void k_tsv_entries(const TsvParams p)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= p.n) return;
    const uint32_t r = p.rec[i];
    // global offset = everything the id columns before and including this record's take + the entries before this one
    uint64_t at = p.rec_prefix[r] + (p.id_off[r + 1] - p.id_off[r]) + 1u + p.off[i];
    const uint64_t end = at + p.len[i];
    if (end <= p.win_lo || at >= p.win_hi) return;
    at = put_dec(p, at, p.hash[i]);
    if (p.with_pos) {
        put_win(p, at++, ':');
        at = put_dec(p, at, p.pos[i]);
    }
    if (p.with_strand) {
        put_win(p, at++, ':');
        put_win(p, at++, p.fwd[i] ? '+' : '-');
    }
    if (p.with_seq) {
        put_win(p, at++, ':');
        if (p.text) {
            uint64_t a = text_of_base(p, r, p.pos[i]);
            for (uint32_t u = 0; u < p.k; ++a) {
                const unsigned char b = p.text[a];
                if (b == '\n' || b == '\r') continue;
                put_win(p, at++, (char)b);
                ++u;
            }
        } else {
            const uint64_t b0 = p.rec_base[r] + p.pos[i];
            for (uint32_t u = 0; u < p.k; ++u) {
                const uint64_t g = b0 + u;
                put_win(p, at++, "ACGT"[(p.packed[g >> 4] >> (2u * ((uint32_t)g & 15u))) & 3u]);
            }
        }
    }
    put_win(p, at, i + 1 == p.rec_first[r + 1] ? '\n' : ' ');
}

// one thread per record: its id, the tab, and the line break of a record without minimizers
__global__ __launch_bounds__(256) void k_tsv_ids(const TsvParams p)
{
    const uint64_t r = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (r >= p.n_rec) return;
    const uint64_t f = p.rec_first[r];
    uint64_t at = p.rec_prefix[r] + p.off[f];
    const uint32_t l = p.id_off[r + 1] - p.id_off[r];
    if (at + l + 2u <= p.win_lo || at >= p.win_hi) return;
    for (uint32_t u = 0; u < l; ++u) put_win(p, at++, p.ids[p.id_off[r] + u]);
    put_win(p, at++, '\t');
    if (p.rec_first[r + 1] == f) put_win(p, at, '\n');
}

int write_tsv_device(mxg_handle *h, Assembly *a, const char *path, int with_pos, int with_strand, int with_seq)
{
    if (!a->has_sketch) return set_err(h, MXG_EINVAL, "assembly '%s' has no sketch yet (call mxg_sketch)", a->name.c_str());
    const uint32_t k = h->cfg.k;
    const bool dbg_io = getenv("MXG_DEBUG_IO") != nullptr;  // timings on stderr
    auto now_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now_s();
    double t_tables = 0, t_alloc = 0, t_dev_wait = 0, t_put = 0;
    MXG_HIP(h, hipSetDevice(h->device));
    hipStream_t st = h->stream;
    int rc;
    if (with_strand && (rc = ensure_strand(h, a)) != MXG_OK) return rc;
    const uint64_t n = a->n_mx, n_rec = a->recs.size();
    // record table: first minimizer of every record (from the record column, 4 bytes per minimizer over PCIe)
    std::vector<uint32_t> h_rec(n);
    if (n) MXG_HIP(h, hipMemcpyAsync(h_rec.data(), a->d_rec.p, n * 4, hipMemcpyDeviceToHost, st));
    MXG_HIP(h, hipStreamSynchronize(st));
    const double t_rec = now_s() - t_begin;  // (the record column on the host)
    std::vector<uint64_t> rec_first(n_rec + 1, 0);
    for (uint64_t i = 0; i < n; ++i) rec_first[h_rec[i] + 1]++;
    for (uint64_t r = 0; r < n_rec; ++r) rec_first[r + 1] += rec_first[r];
    const uint64_t r_lo = std::min<uint64_t>(a->shard_lo, n_rec), r_hi = std::min<uint64_t>(a->shard_hi, n_rec);
    (void)r_lo;
    (void)r_hi;  // (sharded loads go through the host writer)
    std::vector<uint64_t> rec_prefix(n_rec), rec_base(n_rec);
    std::vector<uint32_t> id_off(n_rec + 1);
    std::string ids;
    uint64_t pre = 0;
    for (uint64_t r = 0; r < n_rec; ++r) {
        rec_prefix[r] = pre;
        id_off[r] = (uint32_t)ids.size();
        ids += a->recs[r].id;
        pre += a->recs[r].id.size() + 1 + (rec_first[r + 1] == rec_first[r] ? 1 : 0);
        rec_base[r] = a->recs[r].base_off;
    }
    id_off[n_rec] = (uint32_t)ids.size();
    if (ids.size() >= (1ull << 32)) return set_err(h, MXG_ELIMIT, "record ids too long");
    const double t_host = now_s() - t_begin - t_rec;  // (the record tables)
    DevBuf d_first, d_prefix, d_idoff, d_ids, d_len, d_off, d_tsum, d_tbase, d_total, d_recbase;
    auto up = [&](DevBuf &b, const void *src, size_t bytes) -> int {
        MXG_HIP(h, b.ensure(std::max<size_t>(bytes, 16)));
        if (bytes) MXG_HIP(h, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
        return MXG_OK;
    };
    if ((rc = up(d_first, rec_first.data(), (n_rec + 1) * 8)) != MXG_OK) return rc;
    if ((rc = up(d_prefix, rec_prefix.data(), n_rec * 8)) != MXG_OK) return rc;
    if ((rc = up(d_idoff, id_off.data(), (n_rec + 1) * 4)) != MXG_OK) return rc;
    if ((rc = up(d_ids, ids.data(), ids.size())) != MXG_OK) return rc;
    if ((rc = up(d_recbase, rec_base.data(), n_rec * 8)) != MXG_OK) return rc;
    const uint32_t n_tiles = (uint32_t)((n + TILE - 1) / TILE);
    MXG_HIP(h, d_len.ensure(std::max<uint64_t>(n, 16)));
    MXG_HIP(h, d_off.ensure((n + 1) * 8));
    MXG_HIP(h, d_tsum.ensure(std::max<size_t>((size_t)n_tiles * 4, 16)));
    MXG_HIP(h, d_tbase.ensure(std::max<size_t>((size_t)n_tiles * 8, 16)));
    MXG_HIP(h, d_total.ensure(16));
    TsvParams p;
    memset(&p, 0, sizeof p);
    p.n = n;
    p.hash = a->d_hash.as<uint64_t>();
    p.pos = a->d_pos.as<uint32_t>();
    p.rec = a->d_rec.as<uint32_t>();
    p.fwd = a->d_fwd.as<uint8_t>();
    p.with_pos = with_pos;
    p.with_strand = with_strand;
    p.with_seq = with_seq;
    p.k = k;
    p.rec_first = d_first.as<uint64_t>();
    p.rec_prefix = d_prefix.as<uint64_t>();
    p.id_off = d_idoff.as<uint32_t>();
    p.ids = d_ids.as<char>();
    p.n_rec = n_rec;
    p.len = d_len.as<uint8_t>();
    p.off = d_off.as<uint64_t>();
    p.rec_base = d_recbase.as<uint64_t>();
    p.packed = a->d_packed;
    if (a->text_on_device) {
        DevBuf &d_item0 = a->d_ing_item0;
        if (!d_item0.p) {
            if ((rc = up(d_item0, a->ing_item0.data(), a->ing_item0.size() * 8)) != MXG_OK) return rc;
        }
        p.text = a->d_text.as<unsigned char>();
        p.items = a->d_ing_items.as<IngItem>();
        p.item_pbase = a->d_ing_pbase.as<uint64_t>();
        p.item_sub = a->d_ing_sub.as<uint16_t>();
        p.rec_item0 = d_item0.as<uint64_t>();
    }
    uint64_t total_entries = 0;
    const double t_up = now_s() - t_begin - t_rec - t_host;  // (their upload, the arrays of the lengths)
    if (n) {
        hipLaunchKernelGGL(k_tsv_len, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, p);
        hipLaunchKernelGGL(k_tsv_tile_sum, dim3(n_tiles), dim3(256), 0, st, d_len.as<uint8_t>(), n, d_tsum.as<uint32_t>());
        hipLaunchKernelGGL(k_tsv_scan_tiles, dim3(1), dim3(256), 0, st, d_tsum.as<uint32_t>(), n_tiles, d_tbase.as<uint64_t>(),
                           d_total.as<uint64_t>());
        hipLaunchKernelGGL(k_tsv_offsets, dim3(n_tiles), dim3(256), 0, st, d_len.as<uint8_t>(), n, d_tbase.as<uint64_t>(),
                           d_total.as<uint64_t>(), d_off.as<uint64_t>());
        MXG_HIP(h, hipGetLastError());
        MXG_HIP(h, hipMemcpyAsync(&total_entries, d_total.p, 8, hipMemcpyDeviceToHost, st));
        MXG_HIP(h, hipStreamSynchronize(st));
    } else {
        MXG_HIP(h, hipMemsetAsync(d_off.p, 0, 8, st));
    }
    const uint64_t total = pre + total_entries;
    t_tables = now_s() - t_begin;
    FILE *f = strcmp(path, "-") == 0 ? stdout : fopen(path, "w+b");  // (read access too: put_parallel maps the file)
    if (!f) return set_err(h, MXG_EIO, "cannot open '%s' for writing", path);
    const int ofd = fileno(f);
    fflush(f);
    // a regular file takes the windows in parallel parts at absolute offsets (pwrite); anything else -- a FIFO, /dev/stdout, a
    // process substitution -- has no offsets: it is written in order at the descriptor's own position, and is never removed
    struct stat sb;
    const bool regular = f != stdout && fstat(ofd, &sb) == 0 && S_ISREG(sb.st_mode);
    // (the NAME may still be a symbolic link to a regular file -- /dev/stdout redirected into one -- and is then left alone too)
    const bool removable = regular && lstat(path, &sb) == 0 && S_ISREG(sb.st_mode);
    constexpr uint64_t WIN = 64ull << 20;
    // the file, the pinned windows and their events are released on every way out; a file left incomplete is removed
    struct Out {
        FILE *f;
        const char *path;
        hipStream_t st;
        bool removable;
        char *pin[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        bool complete = false, closed = false;
        bool close()
        {
            closed = true;
            return f == stdout ? fflush(f) == 0 : fclose(f) == 0;
        }
        ~Out()
        {
            (void)hipStreamSynchronize(st);
            for (int b = 0; b < 2; ++b)
                if (ev[b]) (void)hipEventDestroy(ev[b]);  // (the windows are the handle's: pin_pool, tsv_win)
            if (!closed) (void)close();
            if (!complete && removable) (void)remove(path);  // (only a regular file this call created or truncated)
        }
    } out{f, path, st, removable};
    char **pin = out.pin;
    hipEvent_t *ev = out.ev;
    bool ok = true;
    static_assert(2 * WIN <= PIN_POOL_BYTES && WIN % PIN_PIECE_BYTES == 0, "the windows come out of the handle's pinned pool, whole pieces each");
    {
        unsigned char *pool = nullptr;
        MXG_HIP(h, pin_pool_get(h, &pool));
        for (int b = 0; b < 2; ++b) {
            MXG_HIP(h, h->tsv_win[b].ensure(WIN));
            pin[b] = reinterpret_cast<char *>(pool) + (size_t)b * WIN;
            MXG_HIP(h, hipEventCreateWithFlags(&ev[b], hipEventDisableTiming));
        }
    }
    t_alloc = now_s() - t_begin - t_tables;
    auto enqueue = [&](uint64_t c) -> int {
        const int b = (int)(c & 1);
        p.out = h->tsv_win[b].as<char>();
        p.win_lo = c * WIN;
        p.win_hi = std::min(total, p.win_lo + WIN);
        if (n) hipLaunchKernelGGL(k_tsv_entries, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, p);
        if (n_rec) hipLaunchKernelGGL(k_tsv_ids, dim3((uint32_t)((n_rec + 255) / 256)), dim3(256), 0, st, p);
        MXG_HIP(h, hipGetLastError());
        // (the pool is pinned in pieces, each registered with HIP on its own: no copy may reach across two of them)
        for (uint64_t done = 0, n = p.win_hi - p.win_lo; done < n; done += PIN_PIECE_BYTES)
            MXG_HIP(h, hipMemcpyAsync(pin[b] + done, h->tsv_win[b].as<char>() + done, std::min<uint64_t>(PIN_PIECE_BYTES, n - done),
                                      hipMemcpyDeviceToHost, st));
        MXG_HIP(h, hipEventRecord(ev[b], st));
        return MXG_OK;
    };
    const uint64_t n_win = (total + WIN - 1) / WIN;
    rc = MXG_OK;
    if (n_win) rc = enqueue(0);
    for (uint64_t c = 0; c < n_win && rc == MXG_OK; ++c) {
        if (c + 1 < n_win) rc = enqueue(c + 1);  // the device formats the next window while this one is written out
        if (rc != MXG_OK) break;
        const int b = (int)(c & 1);
        const double tw0 = now_s();
        if (hipEventSynchronize(ev[b]) != hipSuccess) {
            rc = set_err(h, MXG_EDEVICE, "TSV formatting failed on the device");
            break;
        }
        const double tw1 = now_s();
        t_dev_wait += tw1 - tw0;
        const uint64_t bytes = std::min(total, (c + 1) * WIN) - c * WIN;
        if (!regular) {  // (a pipe, a FIFO, a device or the shell's redirection: in order, at the descriptor's own position)
            uint64_t done = 0;
            while (done < bytes) {
                const ssize_t wr = write(ofd, pin[b] + done, bytes - done);
                if (wr <= 0) {
                    ok = false;
                    break;
                }
                done += (uint64_t)wr;
            }
        } else {  // the window in `-t` parts, copied into the file's pages side by side
            const uint32_t T = (uint32_t)std::min<uint64_t>(std::min(16u, std::max(1u, host_threads(h))), (bytes + (1u << 20) - 1) >> 20);
            const char *src[16];
            size_t len[16];
            for (uint32_t t = 0; t < T; ++t) {
                const uint64_t lo = bytes * t / T, hi = bytes * (t + 1) / T;
                src[t] = pin[b] + lo;
                len[t] = hi - lo;
            }
            ok = put_parallel(ofd, c * WIN, src, len, T);
        }
        t_put += now_s() - tw1;
        if (!ok) break;
    }
    (void)hipStreamSynchronize(st);
    ok = out.close() && ok;
    if (dbg_io)
        fprintf(stderr, "[mxg] write_tsv_device %s: %.3f s = tables %.3f (record column to the host %.3f, record tables %.3f, uploads + allocations %.3f, lengths + offsets %.3f) + buffers %.3f + waiting for the device %.3f + writing %.3f (%llu MB)\n",
                a->name.c_str(), now_s() - t_begin, t_tables, t_rec, t_host, t_up, t_tables - t_rec - t_host - t_up, t_alloc, t_dev_wait, t_put,
                (unsigned long long)(total >> 20));
    if (rc != MXG_OK) return rc;
    if (!ok) return set_err(h, MXG_EIO, "write error on '%s'", path);
    out.complete = true;
    return MXG_OK;
}

}  // namespace mxg
