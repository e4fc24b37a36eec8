"""GPU parity at BASELINE.json's sizes.
configs[1] (100 Mbp reference + derived target, k=32, w=1000): bit-exact against the CPU oracle, sketch and graph.
Beyond what the oracle finishes in seconds (0.5 Gbp here; the 3-20 Gbp configs are the same code on more batches):
size-independent properties -- the two independent GPU formulations (sparse candidates + gap fix-up vs dense)
agree bit for bit, sketches are sorted and duplicate-free, density ~ 2/(w+1), every reported hash is the oracle's
hash of the k-mer at that position, windows are covered, rank-count-independence of the sharded path."""
import os

import numpy as np
import pytest

from ntjoin_amd import synth
from oracle import graph_oracle as go
from tests import _oracle

pytestmark = pytest.mark.gpu
K, W = 32, 1000


def _engine(**kw):
    from ntjoin_amd.engine import MxEngine
    return MxEngine(**kw)


def _add_packed(eng, name, weight, recs):
    import torch
    words, starts, lens = synth.pack_records(recs)
    d = torch.from_numpy(words.view(np.int32)).cuda()
    eng.add_packed_device(name, weight, d.data_ptr(), starts, lens, keepalive=d)


def test_config2_full_size_bit_exact(oracle, tmp_path):
    ref, tgt = synth.config2(seed=1, n_bases=100_000_000)
    names = [str(tmp_path / "ref.fa.k32.w1000.tsv"), str(tmp_path / "tgt.fa.k32.w1000.tsv")]
    with _engine(k=K, w=W) as eng:
        _add_packed(eng, names[0], 2.0, ref)
        _add_packed(eng, names[1], 1.0, tgt)
        eng.sketch()
        for a, recs in enumerate((ref, tgt)):
            sk = eng.get_sketch(a)
            first = sk["record_first"]
            for r, codes in enumerate(recs):
                want = oracle.sketch(synth.to_ascii(codes), K, W)
                lo, hi = int(first[r]), int(first[r + 1])
                assert hi - lo == len(want), (a, r)
                wh = np.array([x[0] for x in want], dtype=np.uint64)
                wp = np.array([x[1] for x in want], dtype=np.uint32)
                wf = np.array([x[2] for x in want], dtype=np.uint8)
                assert np.array_equal(sk["out_hash"][lo:hi], wh) and np.array_equal(sk["pos"][lo:hi], wp)
                assert np.array_equal(sk["forward"][lo:hi], wf)
            eng.write_tsv(a, names[a], with_seq=True)  # k-mers decoded from the packed bases in HBM
        eng.build_graph()
        eng.write_dot(str(tmp_path / "o.mx.dot"))
        st = eng.stats()
    state = go.load_and_build([names[0]], [2.0], names[1], 1.0)
    got = go.canonical_dot_from_text((tmp_path / "o.mx.dot").read_text(encoding="utf-8"))
    assert got == go.canonical_dot_from_state(state)
    assert st["vertices"] == len(state["vertices"]) and st["edges"] == len(state["edges"])
    # the TSV's sequence column is the k-mer at that position
    line = open(names[0], encoding="ascii").readline().split("\t")[1].split(" ")[0].split(":")
    assert line[2] == synth.to_ascii(ref[0][int(line[1]):int(line[1]) + K]).decode()


def test_half_gbp_properties(oracle):
    n = 500_000_000
    ref = synth.make_reference(11, n, n_records=3)
    # a low-complexity island and an N-free homopolymer: forces candidate-free stretches and dense fix-ups
    ref[1][1_000_000:1_003_000] = 0
    ref[1][5_000_000:5_004_000] = np.tile(np.array([0, 1], dtype=np.uint8), 2000)
    results = {}
    for mode in ("sparse", "dense"):
        with _engine(k=K, w=W, dense_only=(mode == "dense")) as eng:
            _add_packed(eng, "a", 1.0, ref)
            eng.sketch()
            results[mode] = eng.get_sketch(0)
            if mode == "sparse":
                st = eng.stats()
                assert st["candidates"] > 0 and st["dense_kmers"] > 0  # fix-ups did run
    a, b = results["sparse"], results["dense"]
    for key in ("out_hash", "pos", "record", "forward", "record_first"):
        assert np.array_equal(a[key], b[key]), key
    sk = a
    key = (sk["record"].astype(np.uint64) << np.uint64(32)) | sk["pos"].astype(np.uint64)
    assert np.all(key[1:] > key[:-1])  # sorted by (record, pos), no duplicates
    dens = len(key) / float(n)
    assert abs(dens - 2.0 / (W + 1)) < 0.05 * 2.0 / (W + 1)
    # window coverage: consecutive minimizers of a record are never more than w k-mers apart, ends within w
    first = sk["record_first"]
    for r, codes in enumerate(ref):
        p = sk["pos"][int(first[r]):int(first[r + 1])].astype(np.int64)
        assert p[0] < W and (len(codes) - K + 1) - p[-1] <= W and np.all(np.diff(p) <= W)
    # sampled known-answer check: the reported hash/strand is the oracle's for the k-mer at that position
    rng = np.random.default_rng(0)
    for i in rng.integers(0, len(key), size=300):
        r, p = int(sk["record"][i]), int(sk["pos"][i])
        mh, oh, fw, ok = oracle.kmer_hashes(synth.to_ascii(ref[r][p:p + K]), K)
        assert ok[0] and int(oh[0]) == int(sk["out_hash"][i]) and int(fw[0]) == int(sk["forward"][i])
    # local exactness around the forced low-complexity islands (oracle on a 60 kbp excerpt, interior windows)
    for lo in (980_000, 4_980_000):
        seg = synth.to_ascii(ref[1][lo:lo + 60_000])
        want = [(h, p + lo) for h, p, _, _ in oracle.sketch(seg, K, W) if 2000 <= p <= 56_000]
        s0 = int(first[1])
        pr = sk["pos"][s0:int(first[2])]
        sel = (pr >= lo + 2000) & (pr <= lo + 56_000)
        got = list(zip(sk["out_hash"][s0:int(first[2])][sel].tolist(), pr[sel].tolist()))
        assert got == want


def test_config4_shape_four_assemblies(oracle, tmp_path):
    """configs[3] shape: target + 3 references (independent 0.5-2 % divergence), k=32, w=500, weights 1/2/2/2 --
    at 20 Mbp per assembly the oracle checks everything: sketches bit-exact, canonical .mx.dot identical
    (support lists of up to 4 assemblies, colours black/lightgrey/..., weights 2.0..7.0)."""
    w = 500
    base = synth.make_reference(40, 20_000_000, n_records=4)
    names, recs_all, weights = [], [], []
    for i, div in enumerate((0.005, 0.01, 0.02)):
        rng = np.random.default_rng(100 + i)
        recs = []
        for codes in base:
            c = codes.copy()
            idx = rng.integers(0, len(c), size=int(len(c) * div))
            c[idx] = (c[idx] + rng.integers(1, 4, size=len(idx), dtype=np.uint8)) & 3
            recs.append(c)
        names.append(str(tmp_path / f"ref{i}.fa.k32.w500.tsv"))
        recs_all.append(recs)
        weights.append(2.0)
    names.append(str(tmp_path / "tgt.fa.k32.w500.tsv"))
    recs_all.append(synth.derive_target(base, 7, min_len=10_000, max_len=500_000))
    weights.append(1.0)
    with _engine(k=K, w=w) as eng:
        for nm, wt, recs in zip(names, weights, recs_all):
            _add_packed(eng, nm, wt, recs)
        eng.sketch()
        for a, recs in enumerate(recs_all):
            sk = eng.get_sketch(a)
            first = sk["record_first"]
            with open(names[a], "w", encoding="ascii") as fh:
                for r, codes in enumerate(recs):
                    want = oracle.sketch(synth.to_ascii(codes), K, w)
                    lo, hi = int(first[r]), int(first[r + 1])
                    got = list(zip(sk["out_hash"][lo:hi].tolist(), sk["pos"][lo:hi].tolist()))
                    assert got == [(h, p) for h, p, _, _ in want], (a, r)
                    fh.write(f"{r}\t" + " ".join(f"{h}:{p}:N" for h, p in got) + "\n")
        eng.build_graph()
        eng.write_dot(str(tmp_path / "o.mx.dot"))
    state = go.load_and_build(names[:3], weights[:3], names[3], weights[3])
    got = go.canonical_dot_from_text((tmp_path / "o.mx.dot").read_text(encoding="utf-8"))
    want = go.canonical_dot_from_state(state)
    assert got == want
    colours = {e[2].split("color=")[1].rstrip("]") for e in want["edges"]}
    assert "black" in colours and len(want["nodes"]) > 10_000


def test_config3_scale_properties(oracle):
    """configs[2] scale: one 3 Gbp assembly (24 records of 50-250 Mbp) on one GPU.  Beyond what the oracle finishes
    in seconds, so size-independent properties: sorted + duplicate-free, density 2/(w+1), window coverage, sampled
    known-answer hashes, and exact agreement with the oracle on excerpts sketched as stand-alone records."""
    rng = np.random.default_rng(3)
    lens = rng.integers(50_000_000, 250_000_000, size=24).astype(np.float64)
    lens = (lens * (3.0e9 / lens.sum())).astype(np.int64)
    recs = [np.random.default_rng(1000 + i).integers(0, 4, size=int(n), dtype=np.uint8) for i, n in enumerate(lens)]
    with _engine(k=K, w=W) as eng:
        _add_packed(eng, "hs", 1.0, recs)
        eng.sketch()
        sk = eng.get_sketch(0)
        st = eng.stats()
    n = int(sum(lens))
    key = (sk["record"].astype(np.uint64) << np.uint64(32)) | sk["pos"].astype(np.uint64)
    assert np.all(key[1:] > key[:-1])
    assert abs(len(key) / n - 2.0 / (W + 1)) < 0.02 * 2.0 / (W + 1)
    assert st["kmers"] == n - 24 * (K - 1)
    first = sk["record_first"]
    for r in range(24):
        p = sk["pos"][int(first[r]):int(first[r + 1])].astype(np.int64)
        assert p[0] < W and (lens[r] - K + 1) - p[-1] <= W and np.all(np.diff(p) <= W)
    pick = np.random.default_rng(9).integers(0, len(key), size=200)
    for i in pick:
        r, p = int(sk["record"][i]), int(sk["pos"][i])
        mh, oh, fw, ok = oracle.kmer_hashes(synth.to_ascii(recs[r][p:p + K]), K)
        assert ok[0] and int(oh[0]) == int(sk["out_hash"][i]) and int(fw[0]) == int(sk["forward"][i])
    # interior windows of a 300 kbp excerpt are sketched identically whether or not the rest of the record is there
    for r, lo in ((0, 10_000_000), (23, int(lens[23]) - 400_000), (11, 0)):
        seg = synth.to_ascii(recs[r][lo:lo + 300_000])
        want = [(h, p + lo) for h, p, _, _ in oracle.sketch(seg, K, W) if 2 * W <= p <= 300_000 - 3 * W]
        s0, s1 = int(first[r]), int(first[r + 1])
        pr = sk["pos"][s0:s1]
        sel = (pr >= lo + 2 * W) & (pr <= lo + 300_000 - 3 * W)
        assert list(zip(sk["out_hash"][s0:s1][sel].tolist(), pr[sel].tolist())) == want
