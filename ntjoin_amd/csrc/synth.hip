// synth.hip -- counter-based synthetic genomes produced straight into HBM as 2-bit packed bases (bench / test support of
// the hot path: SURVEY.md 8(d) configs 2-5 ask for inputs of 0.1-20 Gbp per assembly, "generated ON DEVICE from a
// counter-based RNG mirrored on the CPU, so nothing of 20 GB crosses PCIe").  No counterpart in the reference (its
// tests use four small FASTA files, tests/*.fa); ntjoin_amd/synth.py mirrors every formula in numpy for the oracle.
//
// The genome is a function of (seed, coordinate):
//     block(n)  = n-th output of splitmix64 seeded with `seed`  = mix(seed + (n + 1) * GOLDEN)        32 bases per block
//     base(g)   = (block(g >> 5) >> 2 * (g & 31)) & 3                                                  A0 C1 G2 T3
// and an assembly is a list of SEGMENTS {dst_base, src, len, rc}: output bases [dst_base, dst_base + len) are genome
// coordinates src .. src + len - 1 (rc = 0) or their reverse complement (rc = 1), each base first substituted with
// probability sub_per_65536 / 65536 by  (b + 1 + (h >> 16) % 3) & 3,  h = mix(sub_seed + (g + 1) * GOLDEN).
// A reference is one segment per record with rc = 0 and no substitutions; a target assembly derived from it (contigs
// cut, flipped, diverged, shuffled: SURVEY.md 8(d) config 2) is one segment per contig.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "mxg_internal.h"

namespace mxg {

static constexpr uint64_t GOLDEN = 0x9E3779B97F4A7C15ull;

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__host__ __device__ __forceinline__ uint32_t synth_base(uint64_t seed, uint64_t sub_seed, uint32_t sub_thr, uint64_t g)
{
    uint32_t b = (uint32_t)(mix64(seed + ((g >> 5) + 1) * GOLDEN) >> (2 * (g & 31))) & 3u;
    if (sub_thr) {
        const uint64_t h = mix64(sub_seed + (g + 1) * GOLDEN);
        if ((uint32_t)(h & 0xFFFFu) < sub_thr) b = (b + 1u + (uint32_t)((h >> 16) % 3u)) & 3u;
    }
    return b;
}

__host__ __device__ __forceinline__ uint32_t synth_word(const mxg_synth_seg &s, uint64_t j0, uint64_t seed, uint64_t sub_seed,
                                                        uint32_t sub_thr)
{
    uint32_t word = 0;
    for (uint32_t u = 0; u < 16; ++u) {
        const uint64_t j = j0 + u;
        if (j >= s.len) break;
        const uint64_t g = s.rc ? s.src + (s.len - 1 - j) : s.src + j;
        uint32_t b = synth_base(seed, sub_seed, sub_thr, g);
        if (s.rc) b = 3u - b;
        word |= b << (2 * u);
    }
    return word;
}

// one thread per output word; segments sorted by dst_base (multiples of 16), words outside every segment are zero
__global__ __launch_bounds__(256) void k_synth(uint32_t *__restrict__ out, uint64_t n_words, const mxg_synth_seg *__restrict__ segs,
                                               uint64_t n_segs, uint64_t seed, uint64_t sub_seed, uint32_t sub_thr)
{
    const uint64_t wi = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (wi >= n_words) return;
    const uint64_t b0 = wi * 16u;
    uint64_t lo = 0, hi = n_segs;  // last segment with dst_base <= b0
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (segs[mid].dst_base <= b0) lo = mid; else hi = mid;
    }
    uint32_t word = 0;
    if (n_segs && segs[lo].dst_base <= b0) {
        const mxg_synth_seg s = segs[lo];
        if (b0 - s.dst_base < s.len) word = synth_word(s, b0 - s.dst_base, seed, sub_seed, sub_thr);
    }
    out[wi] = word;
}

static int check_segs(const mxg_synth_seg *segs, uint64_t n_segs, uint64_t n_words)
{
    uint64_t end = 0;
    for (uint64_t i = 0; i < n_segs; ++i) {
        if ((segs[i].dst_base & 15u) || segs[i].dst_base < end) return MXG_EINVAL;
        end = segs[i].dst_base + segs[i].len;
        if (end > n_words * 16u) return MXG_EINVAL;
    }
    return MXG_OK;
}

}  // namespace mxg

using namespace mxg;

extern "C" {

int mxg_synth_fill_packed_device(void *d_out, uint64_t n_words, const mxg_synth_seg *segs, uint64_t n_segs, uint64_t seed,
                                 uint64_t sub_seed, uint32_t sub_per_65536, int device)
{
    if (!d_out || (!segs && n_segs) || sub_per_65536 > 65536u) return MXG_EINVAL;
    if (check_segs(segs, n_segs, n_words) != MXG_OK) return MXG_EINVAL;
    if (n_words == 0) return MXG_OK;
    if (n_words > (uint64_t)0x7FFFFFFFu * 256u) return MXG_ELIMIT;
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return MXG_EDEVICE;
    mxg_synth_seg *d_segs = nullptr;
    if (hipMalloc((void **)&d_segs, std::max<uint64_t>(n_segs, 1) * sizeof(mxg_synth_seg)) != hipSuccess) return MXG_ENOMEM;
    int rc = MXG_OK;
    if (n_segs && hipMemcpy(d_segs, segs, n_segs * sizeof(mxg_synth_seg), hipMemcpyHostToDevice) != hipSuccess) rc = MXG_EDEVICE;
    if (rc == MXG_OK) {
        hipLaunchKernelGGL(k_synth, dim3((uint32_t)((n_words + 255) / 256)), dim3(256), 0, nullptr, static_cast<uint32_t *>(d_out),
                           n_words, d_segs, n_segs, seed, sub_seed, sub_per_65536);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = MXG_EDEVICE;
    }
    (void)hipFree(d_segs);
    return rc;
}

// the same words on the host (no device needed): the CPU mirror the tests compare the kernel with
int mxg_synth_fill_packed_host(uint32_t *out, uint64_t n_words, const mxg_synth_seg *segs, uint64_t n_segs, uint64_t seed,
                               uint64_t sub_seed, uint32_t sub_per_65536, uint32_t n_threads)
{
    if (!out || (!segs && n_segs) || sub_per_65536 > 65536u) return MXG_EINVAL;
    if (check_segs(segs, n_segs, n_words) != MXG_OK) return MXG_EINVAL;
    memset(out, 0, n_words * 4);
    const uint32_t T = std::max(1u, std::min(n_threads ? n_threads : 1u, 256u));
    auto work = [&](uint32_t t) {
        for (uint64_t i = t; i < n_segs; i += T) {
            const mxg_synth_seg &s = segs[i];
            uint32_t *dst = out + s.dst_base / 16u;
            for (uint64_t j0 = 0; j0 < s.len; j0 += 16) dst[j0 / 16u] = synth_word(s, j0, seed, sub_seed, sub_per_65536);
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    return MXG_OK;
}

// FASTA text of packed records (2-bit, record r = bases [rec_start[r], rec_start[r] + rec_len[r]) of `packed`), `line`
// bases per line, ids "<prefix><r>": how the benchmark materialises its synthetic assemblies as files for the
// end-to-end (FASTA -> .tsv + .mx.dot) measurement.  Host only.
int mxg_synth_write_fasta(const char *path, const uint32_t *packed, const uint64_t *rec_start, const uint64_t *rec_len,
                          uint64_t n_records, const char *id_prefix, uint32_t line, uint32_t n_threads)
{
    if (!path || !packed || (!rec_start && n_records) || (!rec_len && n_records) || line == 0) return MXG_EINVAL;
    FILE *f = fopen(path, "wb");
    if (!f) return MXG_EIO;
    const uint32_t T = std::max(1u, std::min(n_threads ? n_threads : 1u, 64u));
    const uint64_t CH = (uint64_t)line * 65536u;  // bases per work item (whole lines)
    std::vector<std::vector<char>> bufs(T);
    bool ok = true;
    const char *pre = id_prefix ? id_prefix : "";
    for (uint64_t r = 0; r < n_records && ok; ++r) {
        ok = fprintf(f, ">%s%llu\n", pre, (unsigned long long)r) > 0;
        const uint64_t len = rec_len[r], b0 = rec_start[r];
        for (uint64_t c0 = 0; c0 < len && ok; c0 += CH * T) {
            auto work = [&](uint32_t t) {
                const uint64_t lo = c0 + (uint64_t)t * CH, hi = std::min(len, lo + CH);
                std::vector<char> &b = bufs[t];
                b.clear();
                if (lo >= hi) return;
                b.resize((size_t)(hi - lo) + (size_t)((hi - lo + line - 1) / line));
                char *o = b.data();
                for (uint64_t p = lo; p < hi; p += line) {
                    const uint64_t e = std::min(hi, p + line);
                    for (uint64_t q = p; q < e; ++q) {
                        const uint64_t g = b0 + q;
                        *o++ = "ACGT"[(packed[g >> 4] >> (2 * (g & 15))) & 3u];
                    }
                    *o++ = '\n';
                }
            };
            std::vector<std::thread> th;
            for (uint32_t t = 1; t < T; ++t) th.emplace_back(work, t);
            work(0);
            for (auto &x : th) x.join();
            for (uint32_t t = 0; t < T && ok; ++t)
                if (!bufs[t].empty()) ok = fwrite(bufs[t].data(), 1, bufs[t].size(), f) == bufs[t].size();
        }
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? MXG_OK : MXG_EIO;
}

}  // extern "C"
