"""
Seeded synthetic inputs shaped like BASELINE.json's configs: a reference of i.i.d.
uniform ACGT records, and a target made from it by cutting into contigs (length log-uniform), reverse-
complementing half of them, substituting 0.5 % of the bases, dropping 20-500 bp between contigs and shuffling
the order.  Bases are produced as 2-bit codes (A=0,C=1,G=2,T=3) so the same arrays feed the GPU (packed) and
the CPU baseline (decoded to ASCII); numpy's Generator makes them identical on every box.
"""
import numpy as np

LUT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_reference(seed, n_bases, n_records=1):
    rng = np.random.default_rng(seed)
    cuts = np.linspace(0, n_bases, n_records + 1).astype(np.int64)
    return [rng.integers(0, 4, size=int(cuts[i + 1] - cuts[i]), dtype=np.uint8) for i in range(n_records)]


def derive_target(ref_records, seed, min_len=10_000, max_len=2_000_000, sub_rate=0.005, gap=(20, 500)):
    rng = np.random.default_rng(seed)
    contigs = []
    for codes in ref_records:
        p, n = 0, len(codes)
        while p < n:
            ln = int(np.exp(rng.uniform(np.log(min_len), np.log(max_len))))
            seg = codes[p:p + ln].copy()
            p += ln + int(rng.integers(gap[0], gap[1] + 1))
            if len(seg) < 1000:
                continue
            n_sub = rng.binomial(len(seg), sub_rate)
            if n_sub:
                idx = rng.integers(0, len(seg), size=n_sub)
                seg[idx] = (seg[idx] + rng.integers(1, 4, size=n_sub, dtype=np.uint8)) & 3
            if rng.random() < 0.5:
                seg = (3 - seg)[::-1].copy()  # reverse complement in code space
            contigs.append(seg)
    order = rng.permutation(len(contigs))
    return [contigs[i] for i in order]


LUT5 = np.frombuffer(b"ACGTN", dtype=np.uint8)


def repeat_rich_records(seed, n_records, rec_len):
    """records with the structure real genomes add to random sequence (codes 0..3, 4 = N): 55 % unique stretches of 0.5-8 kbp,
    30 % copies of six repeat families (0.3 / 1.2 / 6 kbp consensus, 5-15 % diverged, either strand), 8 % tandem arrays of a
    171-base unit (20-120 copies), 5 % homopolymer / di- / trinucleotide runs of 30-900 units, 2 % N gaps of 1 / 20 / 300 bases.
    What it provokes in the sketch stage: k-mers below the candidate threshold that occur thousands of times (candidate floods
    beyond any estimate), windows whose every k-mer hashes alike (low-complexity: the rightmost-tie rule reports each), long
    candidate-free stretches, many short valid runs."""
    rng = np.random.default_rng(seed)
    fams = [rng.integers(0, 4, size=int(n), dtype=np.uint8) for n in rng.choice([300, 1200, 6000], size=6)]
    sat = rng.integers(0, 4, size=171, dtype=np.uint8)
    units = [np.array(u, dtype=np.uint8) for u in ([0], [3], [0, 1], [0, 3], [2, 2, 1])]
    recs = []
    for _ in range(n_records):
        out = np.empty(rec_len + 25000, dtype=np.uint8)
        n = 0
        while n < rec_len:
            kind = rng.random()
            if kind < 0.55:
                p = rng.integers(0, 4, size=int(rng.integers(500, 8001)), dtype=np.uint8)
            elif kind < 0.85:
                f = fams[int(rng.integers(0, len(fams)))].copy()
                m = rng.random(len(f)) < rng.uniform(0.05, 0.15)
                f[m] = rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8)
                p = (3 - f)[::-1] if rng.random() < 0.5 else f
            elif kind < 0.93:
                p = np.tile(sat, int(rng.integers(20, 121)))
            elif kind < 0.98:
                p = np.tile(units[int(rng.integers(0, len(units)))], int(rng.integers(30, 901)))
            else:
                p = np.full(int(rng.choice([1, 20, 300])), 4, dtype=np.uint8)
            m = min(len(p), len(out) - n)
            out[n:n + m] = p[:m]
            n += m
        recs.append(out[:rec_len].copy())
    return recs


def derive_target_with_n(ref_records, seed, **kw):
    """derive_target for records that hold N (code 4): the N positions travel with their contig, unchanged"""
    clean = []
    for codes in ref_records:
        c = codes.copy()
        c[c == 4] = 0
        clean.append(c | ((codes == 4).astype(np.uint8) << 7))  # bit 7 marks N through the cutting / reversal
    out = []
    rng_state = np.random.default_rng(seed)
    min_len, max_len = kw.get("min_len", 10_000), kw.get("max_len", 2_000_000)
    sub_rate, gap = kw.get("sub_rate", 0.005), kw.get("gap", (20, 500))
    for codes in clean:
        p, n = 0, len(codes)
        while p < n:
            ln = int(np.exp(rng_state.uniform(np.log(min_len), np.log(max_len))))
            seg = codes[p:p + ln].copy()
            p += ln + int(rng_state.integers(gap[0], gap[1] + 1))
            if len(seg) < 1000:
                continue
            isn = (seg & 0x80) != 0
            seg &= 3
            n_sub = rng_state.binomial(len(seg), sub_rate)
            if n_sub:
                idx = rng_state.integers(0, len(seg), size=n_sub)
                seg[idx] = (seg[idx] + rng_state.integers(1, 4, size=n_sub, dtype=np.uint8)) & 3
            if rng_state.random() < 0.5:
                seg, isn = (3 - seg)[::-1].copy(), isn[::-1].copy()
            seg[isn] = 4
            out.append(seg)
    order = rng_state.permutation(len(out))
    return [out[i] for i in order]


def to_ascii5(codes):
    return LUT5[codes].tobytes()


def pack_records(records, pad_words=512):
    """-> (packed uint32[...], rec_start uint64[n], rec_len uint64[n]); record r starts at a multiple of 16 bases."""
    lens = np.array([len(r) for r in records], dtype=np.uint64)
    padded = (lens + np.uint64(15)) // np.uint64(16) * np.uint64(16)
    starts = np.zeros(len(records), dtype=np.uint64)
    if len(records) > 1:
        starts[1:] = np.cumsum(padded[:-1])
    total = int(padded.sum())
    words = np.zeros(total // 16 + pad_words, dtype=np.uint32)
    shifts = (np.arange(16, dtype=np.uint32) * 2)[None, :]
    for r, codes in enumerate(records):
        n = len(codes)
        if n == 0:
            continue
        w0 = int(starts[r]) // 16
        step = 1 << 24
        for o in range(0, n, step):  # chunked to bound temporaries
            chunk = codes[o:o + step]
            m = len(chunk)
            full = (m // 16) * 16
            if full:
                words[w0 + o // 16: w0 + (o + full) // 16] = \
                    (chunk[:full].reshape(-1, 16).astype(np.uint32) << shifts).sum(axis=1, dtype=np.uint32)
            if m > full:  # tail of the record
                tail = np.zeros(16, dtype=np.uint32)
                tail[:m - full] = chunk[full:]
                words[w0 + (o + full) // 16] = np.uint32((tail << shifts[0]).sum())
    return words, starts, lens


def to_ascii(codes):
    return LUT[codes].tobytes()


def config2(seed=1, n_bases=100_000_000):
    """BASELINE.json configs[1]: 1 x n_bases reference + derived target."""
    ref = make_reference(seed, n_bases, 1)
    tgt = derive_target(ref, seed + 1)
    return ref, tgt


# ---- counter-based genomes (mirror of ntjoin_amd/csrc/synth.hip) ---------------------------
# The genome is a pure function of (seed, coordinate), so the GPU fills multi-Gbp assemblies straight into HBM
# (mxg_synth_fill_packed_device) and the CPU side (oracle, tests) reproduces any excerpt of them from the same formulas.
GOLDEN = np.uint64(0x9E3779B97F4A7C15)
SUB_PER_65536 = 328  # 0.5 % substitutions (config 2)


def mix64(z):
    """splitmix64's output function on a uint64 array (wrapping arithmetic)"""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def counter_codes(seed, g0, n, sub_seed=0, sub_per_65536=0):
    """2-bit codes of genome coordinates [g0, g0 + n) (substitutions applied when sub_per_65536 > 0)"""
    g0, n = int(g0), int(n)
    if n <= 0:
        return np.zeros(0, dtype=np.uint8)
    out = np.empty(n, dtype=np.uint8)
    step = 1 << 22
    with np.errstate(over="ignore"):
        for o in range(0, n, step):
            m = min(step, n - o)
            g = np.arange(g0 + o, g0 + o + m, dtype=np.uint64)
            b_lo, b_hi = (g0 + o) >> 5, (g0 + o + m - 1) >> 5
            blocks = mix64(np.uint64(seed) + (np.arange(b_lo, b_hi + 1, dtype=np.uint64) + np.uint64(1)) * GOLDEN)
            b = ((blocks[(g >> np.uint64(5)) - np.uint64(b_lo)] >> (np.uint64(2) * (g & np.uint64(31)))) & np.uint64(3)).astype(np.uint8)
            if sub_per_65536:
                h = mix64(np.uint64(sub_seed) + (g + np.uint64(1)) * GOLDEN)
                hit = (h & np.uint64(0xFFFF)) < np.uint64(sub_per_65536)
                add = (np.uint64(1) + (h >> np.uint64(16)) % np.uint64(3)).astype(np.uint8)
                b = np.where(hit, (b + add) & np.uint8(3), b).astype(np.uint8)
            out[o:o + m] = b
    return out


def segment_codes(seg, seed, sub_seed=0, sub_per_65536=0):
    """codes of one segment (dst_base, src, len, rc) in OUTPUT order"""
    _, src, ln, rc = (int(x) for x in seg)
    c = counter_codes(seed, src, ln, sub_seed, sub_per_65536)
    return (np.uint8(3) - c)[::-1].copy() if rc else c


def layout(lengths):
    """packed layout of records of the given lengths: (rec_start (multiples of 16), total words incl. read padding)"""
    lens = np.asarray(lengths, dtype=np.uint64)
    padded = (lens + np.uint64(15)) // np.uint64(16) * np.uint64(16)
    starts = np.zeros(len(lens), dtype=np.uint64)
    if len(lens) > 1:
        starts[1:] = np.cumsum(padded[:-1])
    return starts, int(padded.sum()) // 16 + 512


def reference_segments(lengths):
    """a reference: record r = genome coordinates [gstart[r], gstart[r] + len[r]), gstart multiples of 32"""
    lens = np.asarray(lengths, dtype=np.uint64)
    g = np.zeros(len(lens), dtype=np.uint64)
    if len(lens) > 1:
        g[1:] = np.cumsum((lens[:-1] + np.uint64(31)) // np.uint64(32) * np.uint64(32))
    starts, n_words = layout(lens)
    segs = np.zeros((len(lens), 4), dtype=np.uint64)
    segs[:, 0], segs[:, 1], segs[:, 2] = starts, g, lens
    return segs, n_words, g


def target_segments(ref_gstart, ref_lens, seed, min_len=10_000, max_len=2_000_000, gap=(20, 500), min_keep=1000):
    """a target derived from the reference (config 2's recipe, vectorised): every reference record cut into contigs of
    log-uniform length with 20-500 bp dropped between them, half of them reverse-complemented, order shuffled.
    -> (segs [n,4] = dst_base, src, len, rc; n_words)"""
    rng = np.random.default_rng(seed)
    mean = (max_len - min_len) / np.log(max_len / min_len)
    srcs, lens = [], []
    for g0, L in zip(np.asarray(ref_gstart).tolist(), np.asarray(ref_lens).tolist()):
        n_est = int(2.0 * L / mean) + 64
        while True:
            ln = np.exp(rng.uniform(np.log(min_len), np.log(max_len), size=n_est)).astype(np.int64)
            gp = rng.integers(gap[0], gap[1] + 1, size=n_est)
            start = np.concatenate(([0], np.cumsum(ln + gp)[:-1]))
            if start[-1] >= L:
                break
            n_est *= 2
        keep = start < L
        start, ln = start[keep], ln[keep]
        ln = np.minimum(ln, L - start)
        ok = ln >= min(min_keep, min_len)
        srcs.append(start[ok] + g0)
        lens.append(ln[ok])
    src = np.concatenate(srcs).astype(np.uint64)
    ln = np.concatenate(lens).astype(np.uint64)
    rc = (rng.random(len(src)) < 0.5).astype(np.uint64)
    order = rng.permutation(len(src))
    src, ln, rc = src[order], ln[order], rc[order]
    starts, n_words = layout(ln)
    segs = np.zeros((len(src), 4), dtype=np.uint64)
    segs[:, 0], segs[:, 1], segs[:, 2], segs[:, 3] = starts, src, ln, rc
    return segs, n_words


def record_lengths(seed, total, n_records, lo_frac=0.2):
    """n_records lengths adding up to `total`, spread like chromosomes (config 3: 24 records of 50-250 Mbp in 3 Gbp)"""
    rng = np.random.default_rng(seed)
    wts = rng.uniform(lo_frac, 1.0, size=n_records)
    lens = np.floor(wts / wts.sum() * total).astype(np.int64)
    lens[-1] += int(total) - int(lens.sum())
    return lens.astype(np.uint64)


def _segs_buffer(segs):
    """the mxg_synth_seg array as one numpy buffer (fast for millions of segments)"""
    s = np.asarray(segs, dtype=np.uint64).reshape(-1, 4)
    buf = np.zeros((len(s), 4), dtype=np.uint64)
    buf[:, 0:3] = s[:, 0:3]
    buf[:, 3] = s[:, 3] & np.uint64(0xFFFFFFFF)  # rc in the low 32 bits, reserved = 0
    return np.ascontiguousarray(buf)


def fill_host(segs, n_words, seed, sub_seed=0, sub_per_65536=0, n_threads=8):
    """packed words of an assembly on the host through the library's own CPU mirror of the kernel (no device needed)"""
    from . import capi
    lib = capi.load()
    out = np.zeros(int(n_words), dtype=np.uint32)
    buf = _segs_buffer(segs)
    rc = lib.mxg_synth_fill_packed_host(out.ctypes.data, int(n_words), buf.ctypes.data, len(buf), int(seed), int(sub_seed),
                                        int(sub_per_65536), int(n_threads))
    if rc != 0:
        raise RuntimeError(f"mxg_synth_fill_packed_host failed ({rc})")
    return out


def fill_device(segs, n_words, seed, sub_seed=0, sub_per_65536=0, device=0):
    """-> torch int32 tensor [n_words] on `device`, filled by the generator kernel (nothing crosses PCIe but the segment table)"""
    import torch
    from . import capi
    lib = capi.load()
    t = torch.empty(int(n_words), dtype=torch.int32, device=torch.device("cuda", device))
    buf = _segs_buffer(segs)
    rc = lib.mxg_synth_fill_packed_device(t.data_ptr(), int(n_words), buf.ctypes.data, len(buf), int(seed), int(sub_seed),
                                          int(sub_per_65536), int(device))
    if rc != 0:
        raise RuntimeError(f"mxg_synth_fill_packed_device failed ({rc})")
    return t


def genome_config(total_bases, n_records, seed=1, target=True, min_len=10_000, max_len=2_000_000, rec_seed=None):
    """segment tables of a reference of n_records records totalling total_bases and (optionally) a target derived from it.
    -> dict(ref_segs, ref_words, tgt_segs, tgt_words, seed, sub_seed)"""
    lens = record_lengths(rec_seed if rec_seed is not None else seed + 7, total_bases, n_records) if n_records > 1 \
        else np.array([int(total_bases)], dtype=np.uint64)
    ref_segs, ref_words, g = reference_segments(lens)
    out = {"ref_segs": ref_segs, "ref_words": ref_words, "seed": int(seed), "sub_seed": int(seed) * 7919 + 13}
    if target:
        out["tgt_segs"], out["tgt_words"] = target_segments(g, lens, seed + 1, min_len=min_len, max_len=max_len)
    return out
