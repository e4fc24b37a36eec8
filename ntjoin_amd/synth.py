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
