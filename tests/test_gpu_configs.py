"""GPU parity at the sizes and shapes BASELINE.json names beyond configs[1] (tests/test_gpu_fullsize.py covers that one
bit for bit).  The assemblies are born in HBM from the counter-based generator (ntjoin_amd/csrc/synth.hip), whose numpy
mirror lets the CPU oracle re-create any excerpt.  What is checked at sizes the oracle cannot sketch in seconds:
  * the GRAPH stage completely, against an independent checker written with numpy set operations on the sketch arrays
    (unique / intersect / adjacent pairs: nothing shared with the engine's hash joins or with the oracle's dictionaries);
  * the SKETCH stage through exact comparison with the oracle on whole contigs and on interior windows of excerpts of the
    long records, plus the size-independent properties (sorted, duplicate-free, density, window coverage).
configs[2]  3 Gbp reference (24 records) + 3 Gbp target, k=32 w=1000, one GPU, the join that runs by default at 12 M minimizers
configs[4]  its shape: > 10^5 contigs of 0.8-200 kbp (some shorter than k+w-1) and a record > 1.5 Gbp; sub-record split, 8 shards
configs[1]  variant with 0.1 % N-runs at 100 Mbp through the ASCII route, bit-exact
(configs[3]: four assemblies, w=500, 4 ranks: tests/test_gpu_dist2.py::test_four_and_eight_ranks_on_one_gpu and
 tests/test_gpu_fullsize.py::test_config4_shape_four_assemblies)"""
import os

import numpy as np
import pytest

from ntjoin_amd import capi, synth
from ntjoin_amd.engine import MxEngine
from tests import _oracle

pytestmark = pytest.mark.gpu
K = 32


def numpy_graph(hashes, recs, weights):
    """independent checker of uniqueness -> intersection -> adjacency edges on arrays sorted by (record, position).
    -> (sorted vertex hashes, {(min hash, max hash): (support mask, weight)})"""
    uniq = []
    for h in hashes:
        v, c = np.unique(h, return_counts=True)
        uniq.append(v[c == 1])
    shared = uniq[0]
    for u in uniq[1:]:
        shared = np.intersect1d(shared, u, assume_unique=True)
    keys, masks = [], []
    for a, (h, r) in enumerate(zip(hashes, recs)):
        m = np.isin(h, shared)
        f, fr = h[m], r[m]
        same = fr[1:] == fr[:-1]
        u, v = f[:-1][same], f[1:][same]
        lo, hi = np.minimum(u, v), np.maximum(u, v)
        keys.append(np.stack([lo, hi], axis=1))
        masks.append(np.full(len(lo), 1 << a, dtype=np.uint32))
    allk = np.concatenate(keys) if keys else np.zeros((0, 2), dtype=np.uint64)
    allm = np.concatenate(masks)
    order = np.lexsort((allk[:, 1], allk[:, 0]))
    allk, allm = allk[order], allm[order]
    new = np.ones(len(allk), dtype=bool)
    new[1:] = np.any(allk[1:] != allk[:-1], axis=1)
    starts = np.flatnonzero(new)
    sup = np.bitwise_or.reduceat(allm, starts) if len(starts) else np.zeros(0, dtype=np.uint32)
    ek = allk[starts]
    wts = np.zeros(len(starts))
    for a, wt in enumerate(weights):  # python: sum(weights[f] for f in support), support in assembly order
        wts = np.where((sup >> a) & 1, wts + wt, wts)
    return shared, ek, sup, wts


def check_graph_against_numpy(eng, weights):
    A = eng.n_assemblies
    sks = [eng.get_sketch(a) for a in range(A)]
    shared, ek, sup, wts = numpy_graph([s["out_hash"] for s in sks], [s["record"] for s in sks], weights)
    g = eng.get_graph()
    assert np.array_equal(np.sort(g["vertex_hash"]), shared)
    hu, hv = g["vertex_hash"][g["edge_u"]], g["vertex_hash"][g["edge_v"]]
    lo, hi = np.minimum(hu, hv), np.maximum(hu, hv)
    order = np.lexsort((hi, lo))
    assert len(lo) == len(ek)
    assert np.array_equal(lo[order], ek[:, 0]) and np.array_equal(hi[order], ek[:, 1])
    assert np.array_equal(g["edge_support"][order], sup)
    assert np.array_equal(g["edge_weight"][order], wts)
    # where every vertex lies in every assembly = the position of its one occurrence there
    for a, s in enumerate(sks):
        o = np.argsort(s["out_hash"], kind="stable")
        at = o[np.searchsorted(s["out_hash"][o], g["vertex_hash"])]
        assert np.array_equal(s["pos"][at], g["vertex_pos"][a]) and np.array_equal(s["record"][at], g["vertex_record"][a])
    # first-seen orientation: an edge runs in the direction of its first supporting assembly's record order
    first_sup = np.zeros(len(hu), dtype=np.int64)
    for a in range(A - 1, -1, -1):
        first_sup = np.where((g["edge_support"] >> a) & 1, a, first_sup)
    for a in range(A):
        sel = first_sup == a
        assert np.all(g["vertex_pos"][a][g["edge_u"][sel]] < g["vertex_pos"][a][g["edge_v"][sel]])
    # flags restate the same sets per minimizer
    for a, s in enumerate(sks):
        fl = eng.get_mx_flags(a)
        v, c = np.unique(s["out_hash"], return_counts=True)
        assert np.array_equal((fl & capi.MX_UNIQUE) != 0, np.isin(s["out_hash"], v[c == 1]))
        assert np.array_equal((fl & capi.MX_SHARED) != 0, np.isin(s["out_hash"], shared))
    return sks, g


def sketch_properties(sk, lens, w):
    key = (sk["record"].astype(np.uint64) << np.uint64(32)) | sk["pos"].astype(np.uint64)
    assert np.all(key[1:] > key[:-1])  # sorted by (record, pos), no duplicates
    first = sk["record_first"]
    nk = lens.astype(np.int64) - K + 1
    cnt = np.diff(first.astype(np.int64))
    assert np.all(cnt[nk < w] == 0) and np.all(cnt[nk >= w] > 0)  # records without a full window yield nothing
    elig = np.flatnonzero(nk >= w)
    p = sk["pos"].astype(np.int64)
    f0, f1 = first[elig].astype(np.int64), first[elig + 1].astype(np.int64)
    assert np.all(p[f0] < w) and np.all(nk[elig] - p[f1 - 1] <= w)  # first / last window covered
    d = np.diff(p)
    inside = np.ones(len(p) - 1, dtype=bool)
    inside[f1[:-1] - 1] = False  # pairs that straddle two records
    assert np.all(d[inside] <= w) and np.all(d[inside] > 0)


def excerpt_check(orc, sk, seg, r, seed, sub_seed, sub, w, lo, n):
    """interior windows of the excerpt [lo, lo+n) of record r are sketched identically with or without the rest of it"""
    codes = synth.segment_codes((0, int(seg[1]) + (int(seg[2]) - lo - n if seg[3] else lo), n, int(seg[3])), seed, sub_seed, sub)
    want = [(h, p + lo) for h, p, _, _ in orc.sketch(synth.to_ascii(codes), K, w) if 2 * w <= p <= n - 3 * w]
    s0, s1 = int(sk["record_first"][r]), int(sk["record_first"][r + 1])
    pr = sk["pos"][s0:s1]
    sel = (pr >= lo + 2 * w) & (pr <= lo + n - 3 * w)
    assert list(zip(sk["out_hash"][s0:s1][sel].tolist(), pr[sel].tolist())) == want
    assert len(want) > 0


def whole_record_check(orc, sk, seg, r, seed, sub_seed, sub, w):
    codes = synth.segment_codes(seg, seed, sub_seed, sub)
    want = orc.sketch(synth.to_ascii(codes), K, w)
    s0, s1 = int(sk["record_first"][r]), int(sk["record_first"][r + 1])
    assert sk["out_hash"][s0:s1].tolist() == [x[0] for x in want] and sk["pos"][s0:s1].tolist() == [x[1] for x in want]
    assert sk["forward"][s0:s1].tolist() == [x[2] for x in want]


def paths_properties(eng, g, n):
    """what find_paths(n) returns on a graph too large for the Python oracle: every vertex on at most one path, consecutive path
    vertices joined by an edge of weight >= n, every path inside one component, no path of a single vertex unless the
    reference would print one (it does not: bin/ntjoin.py:163-170 keeps paths of any length, vertices without an edge of
    weight >= n are components of their own and are skipped by the degree test), and most of the graph covered (the synthetic
    target is the reference cut into contigs: long chains)"""
    paths = eng.find_paths(n)
    assert len(paths) >= (24 if n == 1 else 1000)  # n = 1: the reference's edges chain each of its 24 records; n = 3: the target's contigs
    nv = len(g["vertex_hash"])
    lens = np.array([len(v) for _, v in paths], dtype=np.int64)
    verts = np.concatenate([np.asarray(v, dtype=np.int64) for _, v in paths])
    assert verts.min() >= 0 and verts.max() < nv
    assert len(np.unique(verts)) == len(verts)
    ends = np.cumsum(lens)
    a, b = verts[:-1], verts[1:]
    inside = np.ones(len(verts) - 1, dtype=bool)
    inside[ends[:-1] - 1] = False
    a, b = a[inside], b[inside]
    keep = np.asarray(g["edge_weight"]) >= n
    eu, ev = np.asarray(g["edge_u"], dtype=np.int64)[keep], np.asarray(g["edge_v"], dtype=np.int64)[keep]
    keys = np.sort(np.minimum(eu, ev) * nv + np.maximum(eu, ev))
    want = np.minimum(a, b) * nv + np.maximum(a, b)
    at = np.searchsorted(keys, want)
    assert np.all(at < len(keys)) and np.all(keys[np.minimum(at, len(keys) - 1)] == want)
    comp = np.array([c for c, _ in paths])
    assert len(np.unique(comp)) <= len(comp)
    assert len(verts) > 0.5 * nv if n == 1 else len(verts) > 0
    # format_path's inputs (8 f4): per record the extreme positions of its graph vertices, per assembly the runs of consecutive
    # path vertices on one record with their position range and orientation tallies
    for a in range(len(g["vertex_pos"])):
        rec, pos = np.asarray(g["vertex_record"][a], dtype=np.int64), np.asarray(g["vertex_pos"][a], dtype=np.int64)
        n_rec = int(rec.max()) + 1
        lo = np.full(n_rec, np.iinfo(np.int64).max)
        hi = np.full(n_rec, -1)
        np.minimum.at(lo, rec, pos)
        np.maximum.at(hi, rec, pos)
        ext = eng.mx_extremes(a)
        got = [(i, e) for i, e in enumerate(ext) if e is not None]
        assert [i for i, _ in got] == np.flatnonzero(hi >= 0).tolist()
        assert [e for _, e in got] == list(zip(lo[hi >= 0].tolist(), hi[hi >= 0].tolist()))
        sg = eng.path_segments(a)
        assert int(sg["n"].sum()) == len(verts)  # the runs tile the concatenated paths
        first = np.asarray(sg["first"], dtype=np.int64)
        assert np.array_equal(first, np.concatenate([[0], np.cumsum(sg["n"].astype(np.int64))[:-1]]))
        # every run lies on one record, and its position range is the range of its vertices
        run_of = np.repeat(np.arange(len(first)), sg["n"].astype(np.int64))
        assert np.array_equal(rec[verts], np.asarray(sg["record"], dtype=np.int64)[run_of])
        rmin = np.full(len(first), np.iinfo(np.int64).max)
        rmax = np.full(len(first), -1)
        np.minimum.at(rmin, run_of, pos[verts])
        np.maximum.at(rmax, run_of, pos[verts])
        assert np.array_equal(rmin, sg["min_pos"].astype(np.int64)) and np.array_equal(rmax, sg["max_pos"].astype(np.int64))
        step = np.diff(pos[verts])
        same_run = run_of[1:] == run_of[:-1]
        inc = np.bincount(run_of[1:][same_run & (step > 0)], minlength=len(first))
        dec = np.bincount(run_of[1:][same_run & (step < 0)], minlength=len(first))
        assert np.array_equal(inc, sg["inc"].astype(np.int64)) and np.array_equal(dec, sg["dec"].astype(np.int64))


def _born_in_hbm(eng, cfg, which, name, weight):
    segs, n_words = cfg[which + "_segs"], cfg[which + "_words"]
    sub = synth.SUB_PER_65536 if which == "tgt" else 0
    d = synth.fill_device(segs, n_words, cfg["seed"], cfg["sub_seed"], sub)
    eng.add_packed_device(name, weight, d.data_ptr(), segs[:, 0], segs[:, 2], keepalive=d)
    return segs, sub


def test_configs2_whole_path_3gbp_plus_3gbp():
    """BASELINE configs[2] at full size, exactly what bench.py times: both assemblies, the graph stage on the join that runs
    BY DEFAULT at 12 M minimizers"""
    orc = _oracle.load()
    w = 1000
    cfg = synth.genome_config(3_000_000_000, 24, seed=1, min_len=3000, max_len=600_000)
    with MxEngine(k=K, w=w) as eng:
        rsegs, _ = _born_in_hbm(eng, cfg, "ref", "ref", 2.0)
        tsegs, tsub = _born_in_hbm(eng, cfg, "tgt", "tgt", 1.0)
        eng.sketch(-2)
        eng.build_graph()
        st = eng.stats()
        assert st["minimizers"] > 11_000_000 and st["bases"] > 5_900_000_000
        sks, g = check_graph_against_numpy(eng, [2.0, 1.0])
        assert len(g["vertex_hash"]) > 4_000_000 and len(g["edge_u"]) > 4_000_000
        for sk, segs in zip(sks, (rsegs, tsegs)):
            sketch_properties(sk, segs[:, 2], w)
            assert abs(len(sk["pos"]) / float(segs[:, 2].sum()) - 2.0 / (w + 1)) < 0.02 * 2.0 / (w + 1)
        # the first "next" row at this size: linear paths over 4.5 M vertices (reference bin/ntjoin.py:69-176)
        for n in (1, 3):
            paths_properties(eng, g, n)
        # sketch stage, exact: excerpts of the long reference records (first, last, across the second sparse batch) ...
        for r, lo in ((0, 0), (0, 50_000_000), (23, int(rsegs[23][2]) - 300_000), (16, 1_000_000), (17, 77_777)):
            excerpt_check(orc, sks[0], rsegs[r], r, cfg["seed"], cfg["sub_seed"], 0, w, lo, 300_000)
        # ... and whole target contigs: the shortest, some of middle size, both strands, the last one
        order = np.argsort(tsegs[:, 2])
        pick = list(order[:3]) + list(order[len(order) // 2: len(order) // 2 + 3]) + [len(tsegs) - 1, 0]
        assert {int(tsegs[i][3]) for i in pick} == {0, 1}
        for r in pick:
            whole_record_check(orc, sks[1], tsegs[r], int(r), cfg["seed"], cfg["sub_seed"], tsub, w)
        # sampled known-answer hashes all over both assemblies
        rng = np.random.default_rng(5)
        for sk, segs, sub in ((sks[0], rsegs, 0), (sks[1], tsegs, tsub)):
            for i in rng.integers(0, len(sk["pos"]), size=150):
                r, p = int(sk["record"][i]), int(sk["pos"][i])
                seg = segs[r]
                src = int(seg[1]) + (int(seg[2]) - p - K if seg[3] else p)
                kmer = synth.segment_codes((0, src, K, int(seg[3])), cfg["seed"], cfg["sub_seed"], sub)
                _, oh, fw, ok = orc.kmer_hashes(synth.to_ascii(kmer), K)
                assert ok[0] and int(oh[0]) == int(sk["out_hash"][i]) and int(fw[0]) == int(sk["forward"][i])


def _fragmented_config(seed, n_contigs, long_len):
    """configs[4]'s shape: a reference with one very long record (+ three shorter ones) and a target of n_contigs contigs of
    0.8-200 kbp cut from it (log-uniform, so most are shorter than 100 kbp; below 1031 bases there is no window)"""
    lens = np.array([long_len, 1_000_000_000, 900_000_000, 900_000_000], dtype=np.uint64)
    ref_segs, ref_words, g = synth.reference_segments(lens)
    tgt_segs, tgt_words = synth.target_segments(g, lens, seed + 1, min_len=800, max_len=200_000, min_keep=800)
    tgt_segs = tgt_segs[:n_contigs]
    starts, tgt_words = synth.layout(tgt_segs[:, 2])
    tgt_segs[:, 0] = starts
    return {"ref_segs": ref_segs, "ref_words": ref_words, "tgt_segs": tgt_segs, "tgt_words": tgt_words, "seed": seed,
            "sub_seed": seed * 7919 + 13}


def test_configs4_shape_fragmented_target_and_very_long_record():
    """> 10^5 contigs (some shorter than k+w-1) against a reference holding a 1.6 Gbp record: whole path, graph against the
    numpy checker, sketches against the oracle on whole contigs and on excerpts around the 2^31-st base of the long record"""
    orc = _oracle.load()
    w = 1000
    cfg = _fragmented_config(41, 110_000, 1_600_000_000)
    assert len(cfg["tgt_segs"]) >= 100_000
    tl = cfg["tgt_segs"][:, 2].astype(np.int64)
    assert (tl < K + w - 1).sum() > 100 and (tl < 100_000).mean() > 0.5
    with MxEngine(k=K, w=w) as eng:
        rsegs, _ = _born_in_hbm(eng, cfg, "ref", "ref", 1.0)
        tsegs, tsub = _born_in_hbm(eng, cfg, "tgt", "tgt", 1.0)
        eng.sketch(-2)
        eng.build_graph()
        sks, g = check_graph_against_numpy(eng, [1.0, 1.0])
        assert len(g["vertex_hash"]) > 1_000_000
        for sk, segs in zip(sks, (rsegs, tsegs)):
            sketch_properties(sk, segs[:, 2], w)
        for lo in (0, 1_073_000_000, 1_600_000_000 - 300_000):
            excerpt_check(orc, sks[0], rsegs[0], 0, cfg["seed"], cfg["sub_seed"], 0, w, lo, 300_000)
        order = np.argsort(tsegs[:, 2])
        n_short = int((tl < K + w - 1).sum())
        pick = list(order[n_short - 2: n_short + 4]) + list(order[-2:]) + list(np.random.default_rng(2).integers(0, len(tsegs), 6))
        for r in pick:
            whole_record_check(orc, sks[1], tsegs[r], int(r), cfg["seed"], cfg["sub_seed"], tsub, w)


def test_configs4_long_record_split_over_eight_shards(tmp_path):
    """mxg_add_assembly_fasta_split on a FASTA holding a record > 1.5 Gbp followed by short contigs: the rank-ordered
    concatenation of the 8 shards' sketches is the single handle's sketch, bit for bit"""
    w = 1000
    lens = np.concatenate([[1_550_000_000], np.random.default_rng(8).integers(500, 150_000, size=1500)]).astype(np.uint64)
    segs, n_words, _ = synth.reference_segments(lens)
    words = synth.fill_device(segs, n_words, 77).cpu().numpy().view(np.uint32)
    fa = str(tmp_path / "big.fa")
    lib = capi.load()
    rs, rl = np.ascontiguousarray(segs[:, 0]), np.ascontiguousarray(segs[:, 2])
    assert lib.mxg_synth_write_fasta(fa.encode(), words.ctypes.data, rs.ctypes.data, rl.ctypes.data, len(rl), b"c", 80, 16) == 0
    del words
    with MxEngine(k=K, w=w, drop_seq=True) as eng:
        eng.add_fasta("whole", 1.0, fa)
        eng.sketch()
        whole = eng.get_sketch(0)
    assert len(whole["pos"]) > 3_000_000
    parts = {k_: [] for k_ in ("out_hash", "pos", "record")}
    n_cut = 0
    for s in range(8):
        with MxEngine(k=K, w=w, drop_seq=True) as eng:
            eng.add_fasta_split("part", 1.0, fa, s, 8)
            eng.sketch()
            sk = eng.get_sketch(0)
            n_cut += int(eng.assembly_continues(0))
            for k_ in parts:
                parts[k_].append(sk[k_])
    assert n_cut >= 3  # the long record was really cut into pieces
    for k_ in parts:
        assert np.array_equal(np.concatenate(parts[k_]), whole[k_]), k_


def test_configs1_variant_with_n_runs_100mbp(oracle):
    """SURVEY config 2's parity variant: 0.1 % of the bases in N-runs of 1-1000 bp, 100 Mbp reference + derived target,
    through the ASCII route (N cannot travel 2-bit packed), bit-exact sketches and canonical .mx.dot"""
    w = 1000
    ref, tgt = synth.config2(seed=3, n_bases=100_000_000)
    rng = np.random.default_rng(12)

    def with_n(codes):
        s = bytearray(synth.to_ascii(codes))
        n_runs = max(1, int(0.001 * len(s) / 500))
        for st in rng.integers(0, max(1, len(s) - 1000), size=n_runs):
            ln = int(rng.integers(1, 1001))
            s[int(st):int(st) + ln] = b"N" * min(ln, len(s) - int(st))
        return bytes(s)
    ref_s = [with_n(c) for c in ref]
    tgt_s = [with_n(c) if len(c) > 50_000 else synth.to_ascii(c) for c in tgt]
    assert sum(s.count(b"N") for s in ref_s) > 50_000
    with MxEngine(k=K, w=w) as eng:
        eng.add_records("ref", 2.0, [(f"r{i}", s) for i, s in enumerate(ref_s)])
        eng.add_records("tgt", 1.0, [(f"t{i}", s) for i, s in enumerate(tgt_s)])
        eng.sketch()
        for a, seqs in enumerate((ref_s, tgt_s)):
            sk = eng.get_sketch(a)
            first = sk["record_first"]
            for r, s in enumerate(seqs):
                want = oracle.sketch(s, K, w)
                lo, hi = int(first[r]), int(first[r + 1])
                assert sk["out_hash"][lo:hi].tolist() == [x[0] for x in want], (a, r)
                assert sk["pos"][lo:hi].tolist() == [x[1] for x in want], (a, r)
        eng.build_graph()
        check_graph_against_numpy(eng, [2.0, 1.0])


def _bench_workload(eng, name, mbp, w):
    """the assemblies bench.py times for `--workload name`, born in HBM the same way (whole records, one GPU)"""
    import bench
    cfg, asms, _ = bench.workload_tables(name, mbp, w, seed=1)
    for aname, weight, segs, n_words, sub, sub_seed in asms:
        d = synth.fill_device(segs, n_words, cfg["seed"], sub_seed, sub)
        eng.add_packed_device(aname, weight, d.data_ptr(), segs[:, 0], segs[:, 2], keepalive=d)
    return cfg, asms


def _density_ok(sk, segs, w, tol=0.02):
    lens = segs[:, 2].astype(np.int64)
    elig = lens[lens - K + 1 >= w].sum()
    return abs(len(sk["pos"]) / float(elig) - 2.0 / (w + 1)) < tol * 2.0 / (w + 1)


def test_configs3_full_size_target_and_three_references_w500():
    """BASELINE configs[3] at full size on one GPU (`bench.py --workload configs3`): a 3 Gbp target + three 3 Gbp references
    of 0.5 / 1 / 2 % divergence, w = 500, weights 2/2/2/1 -- 12 Gbp and 48 M minimizers per step.  Graph stage against the
    numpy checker (four assemblies: support masks and summed weights), sketches against the oracle on excerpts of every
    reference (each has its own substitution stream) and on whole target contigs, plus the size-independent properties."""
    orc = _oracle.load()
    w = 500
    with MxEngine(k=K, w=w) as eng:
        cfg, asms = _bench_workload(eng, "configs3", 3000.0, w)
        eng.sketch(-2)
        eng.build_graph()
        st = eng.stats()
        assert st["bases"] > 11_900_000_000 and st["minimizers"] > 47_000_000
        assert st["bs_filter_bases"] == st["bases"] or os.environ.get("MXG_BS") == "0"   # (MXG_BS=0: the rolling-hash route is under test at size)  # the k = 32 route took every assembly
        weights = [a[1] for a in asms]
        assert weights == [2.0, 2.0, 2.0, 1.0]
        sks, g = check_graph_against_numpy(eng, weights)
        assert len(g["vertex_hash"]) > 1_500_000
        assert set(np.unique(g["edge_weight"]).tolist()) <= {1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0}
        for sk, a in zip(sks, asms):
            sketch_properties(sk, a[2][:, 2], w)
            assert _density_ok(sk, a[2], w)
        for ai in range(3):  # the references: same records, different substitutions
            _, _, segs, _, sub, sub_seed = asms[ai]
            for r, lo in ((0, 0), (11, 20_000_000), (23, int(segs[23][2]) - 200_000)):
                excerpt_check(orc, sks[ai], segs[r], r, cfg["seed"], sub_seed, sub, w, lo, 200_000)
        _, _, tsegs, _, tsub, tsub_seed = asms[3]
        order = np.argsort(tsegs[:, 2])
        for r in list(order[:2]) + list(order[len(order) // 2: len(order) // 2 + 2]) + [0, len(tsegs) - 1]:
            whole_record_check(orc, sks[3], tsegs[r], int(r), cfg["seed"], tsub_seed, tsub, w)


def test_configs4_full_size_40gbp_on_one_gpu():
    """BASELINE configs[4] at full size on ONE GPU (`bench.py --workload configs4`): a 20 Gbp reference of 12 records of
    ~1.67 Gbp + a ~20 Gbp target of > 5 x 10^5 contigs of 1-200 kbp, w = 1000: 4 x 10^10 k-mers per step (beyond 2^32 per
    job, beyond 2^32 per assembly), 79 M minimizers, ~30 M vertices.  Graph stage against the numpy checker; sketches against the
    oracle on excerpts around the 2^32-nd base of the assembly and of the packed layout, whole target contigs (the shortest
    that hold a window, the longest, both strands), known-answer hashes, size-independent properties."""
    orc = _oracle.load()
    w = 1000
    with MxEngine(k=K, w=w) as eng:
        cfg, asms = _bench_workload(eng, "configs4", 20000.0, w)
        eng.sketch(-2)
        eng.build_graph()
        st = eng.stats()
        assert st["bases"] > 39_000_000_000 and st["kmers"] > (1 << 32) * 9
        assert st["bs_filter_bases"] == st["bases"] or os.environ.get("MXG_BS") == "0"   # (MXG_BS=0: the rolling-hash route is under test at size)
        assert st["minimizers"] > 78_000_000
        sks, g = check_graph_against_numpy(eng, [a[1] for a in asms])
        assert len(g["vertex_hash"]) > 25_000_000 and len(g["edge_u"]) > 25_000_000
        for sk, a in zip(sks, asms):
            sketch_properties(sk, a[2][:, 2], w)
            assert _density_ok(sk, a[2], w)
        _, _, rsegs, _, rsub, rsub_seed = asms[0]
        assert len(rsegs) == 12 and int(rsegs[:, 2].min()) > 500_000_000 and int(rsegs[:, 2].max()) > 2_500_000_000
        # record 2 holds the assembly's 2^32-nd base; the last record ends near base 2 x 10^10
        cum = np.cumsum(rsegs[:, 2].astype(np.int64))
        r32 = int(np.searchsorted(cum, 1 << 32))
        lo32 = int((1 << 32) - (cum[r32 - 1] if r32 else 0)) - 150_000
        for r, lo in ((0, 0), (r32, max(lo32, 0)), (7, 1_000_000_000), (11, int(rsegs[11][2]) - 300_000)):
            excerpt_check(orc, sks[0], rsegs[r], r, cfg["seed"], rsub_seed, rsub, w, lo, 300_000)
        _, _, tsegs, _, tsub, tsub_seed = asms[1]
        tl = tsegs[:, 2].astype(np.int64)
        order = np.argsort(tl)
        n_short = int((tl < K + w - 1).sum())
        pick = list(order[max(n_short - 2, 0): n_short + 3]) + list(order[-2:]) + [0, len(tsegs) - 1] + \
            list(np.random.default_rng(4).integers(0, len(tsegs), 6))
        assert {int(tsegs[i][3]) for i in pick} == {0, 1}
        for r in pick:
            whole_record_check(orc, sks[1], tsegs[r], int(r), cfg["seed"], tsub_seed, tsub, w)
        rng = np.random.default_rng(6)
        for sk, segs, sub, sub_seed in ((sks[0], rsegs, rsub, rsub_seed), (sks[1], tsegs, tsub, tsub_seed)):
            for i in rng.integers(0, len(sk["pos"]), size=100):
                r, p = int(sk["record"][i]), int(sk["pos"][i])
                seg = segs[r]
                src = int(seg[1]) + (int(seg[2]) - p - K if seg[3] else p)
                kmer = synth.segment_codes((0, src, K, int(seg[3])), cfg["seed"], sub_seed, sub)
                _, oh, fw, ok = orc.kmer_hashes(synth.to_ascii(kmer), K)
                assert ok[0] and int(oh[0]) == int(sk["out_hash"][i]) and int(fw[0]) == int(sk["forward"][i])


@pytest.mark.parametrize("w", [1000, 500])
def test_fallback_routes_agree_with_the_default_route_at_1_gbp(w):
    """the routes that real input can send a batch down -- behind the filter's bitmap the batch kernels instead of the slice kernel
    (MXG_BS_SELECT=0: second attempts, run tables with short runs), the rolling-hash filter (MXG_BS=0: other k, caller's layouts) --
    on bench.py's configs[2] workload at 1 Gbp + 1 Gbp: every array of both sketches and the graph's counts equal the default
    route's, bit for bit.  (The same routes pass the whole config-size suite: profiles/r05/configs_select_off.txt, configs_bs_off.txt.)"""
    saved = {k_: os.environ.get(k_) for k_ in ("MXG_BS_SELECT", "MXG_BS")}
    res = {}
    try:
        for route, env in (("default", {}), ("batch kernels", {"MXG_BS_SELECT": "0"}), ("rolling hash", {"MXG_BS": "0"})):
            for k_ in saved:
                os.environ.pop(k_, None)
            os.environ.update(env)
            with MxEngine(k=K, w=w) as eng:
                _bench_workload(eng, "configs2", 1000.0, w)
                eng.sketch(-2)
                eng.build_graph()
                st = eng.stats()
                sks = [eng.get_sketch(a) for a in range(2)]
                res[route] = ([{f: sk[f].copy() for f in ("out_hash", "pos", "record")} for sk in sks], st["vertices"], st["edges"])
                assert (st["select_slices"] > 0) == (route == "default") and (st["bs_filter_bases"] > 0) == (route != "rolling hash")
    finally:
        for k_, v in saved.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v
    ref = res["default"]
    assert ref[1] > 500_000 and ref[2] > 500_000
    for route in ("batch kernels", "rolling hash"):
        got = res[route]
        assert got[1:] == ref[1:], route
        for a in range(2):
            for f in ("out_hash", "pos", "record"):
                assert np.array_equal(got[0][a][f], ref[0][a][f]), (route, a, f)
