"""GPU test of sub-record sharding (mxg_add_assembly_fasta_split, SURVEY.md 8e "chunk with halo"): the shards are equal
base ranges of the file, long records are sketched in pieces, and the rank-ordered concatenation of the shards' sketches
(and of their TSV parts) must be EXACTLY the sketch (TSV) of the whole file on one handle -- for any number of shards,
with N runs and lower case at the cuts, through the sparse path, its gap fix-up and the dense path."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _genome(path, seed, lens, n_frac=0.002):
    rng = random.Random(seed)
    with open(path, "w") as f:
        for i, n in enumerate(lens):
            seq = rng.choices("ACGT", k=n)
            for _ in range(int(n * n_frac / 20) + (1 if n > 5000 else 0)):       # N runs of 1..60
                p = rng.randrange(n)
                for q in range(p, min(n, p + rng.randint(1, 60))):
                    seq[q] = "N"
            if i % 2:
                seq = [c.lower() if rng.random() < 0.3 else c for c in seq]
            s = "".join(seq)
            f.write(f">rec{i} some comment\n")
            for j in range(0, n, 70):
                f.write(s[j:j + 70] + "\n")


def _whole(path, k, w, **kw):
    from ntjoin_amd.engine import MxEngine
    with MxEngine(k=k, w=w, **kw) as eng:
        eng.add_fasta("x", 1.0, path)
        eng.sketch()
        sk = eng.get_sketch(0)
        return {key: np.asarray(sk[key]).copy() for key in ("out_hash", "pos", "record")}


def _pieces(path, k, w, n_shards, tmp=None, **kw):
    from ntjoin_amd.engine import MxEngine
    parts = {"out_hash": [], "pos": [], "record": []}
    tsvs, cont, bases = [], [], []
    for s in range(n_shards):
        with MxEngine(k=k, w=w, **kw) as eng:
            eng.add_fasta_split("x", 1.0, path, s, n_shards)
            eng.sketch()
            sk = eng.get_sketch(0)
            for key in parts:
                parts[key].append(np.asarray(sk[key]).copy())
            bases.append(eng.stats()["bases"])
            cont.append(eng.assembly_continues(0))
            if tmp is not None:
                t = os.path.join(tmp, f"part{s}.tsv")
                eng.write_tsv(0, t, with_pos=True, with_strand=False, with_seq=True)
                tsvs.append(t)
    return {key: np.concatenate(v) for key, v in parts.items()}, tsvs, cont, bases


LENS = [30_000, 500, 260_000, 80_000, 31, 1_000, 150_000, 40, 90_000]


@pytest.mark.parametrize("k,w,kw", [(32, 500, {}), (32, 50, {}), (15, 10, {}), (32, 500, {"variant": "v1"}),
                                    (32, 300, {"cand_per_window": 2}), (32, 200, {"dense_only": True}),
                                    (32, 400, {"drop_seq": True})])
@pytest.mark.parametrize("n_shards", [2, 3, 7])
def test_split_shards_concatenate_to_whole(tmp_path, k, w, kw, n_shards):
    fa = str(tmp_path / "g.fa")
    _genome(fa, 5, LENS)
    whole = _whole(fa, k, w, **kw)
    got, _, _, bases = _pieces(fa, k, w, n_shards, **kw)
    assert len(whole["pos"]) > 1000
    for key in whole:
        assert np.array_equal(whole[key], got[key]), key
    assert max(bases) < 1.35 * sum(LENS) / n_shards + 2 * (w + k) + 70_000 / n_shards   # balanced by bases, halo aside


def test_split_many_small_shards_and_tsv(tmp_path):
    """shards shorter than a window's halo (pieces reach back over several shards) and the stitched TSV"""
    from ntjoin_amd.dist import concat_tsv_parts
    from ntjoin_amd.engine import MxEngine
    fa = str(tmp_path / "g.fa")
    lens = [9_000, 200, 25_000, 3_000, 12_000]
    _genome(fa, 9, lens)
    k, w = 32, 400
    whole = _whole(fa, k, w)
    for n_shards in (4, 40):
        got, tsvs, cont, _ = _pieces(fa, k, w, n_shards, tmp=str(tmp_path))
        for key in whole:
            assert np.array_equal(whole[key], got[key]), (n_shards, key)
        out = str(tmp_path / f"joined{n_shards}.tsv")
        concat_tsv_parts(tsvs, cont, out)
        with MxEngine(k=k, w=w) as eng:
            eng.add_fasta("x", 1.0, fa)
            eng.sketch()
            ref = str(tmp_path / "whole.tsv")
            eng.write_tsv(0, ref, with_pos=True, with_strand=False, with_seq=True)
        assert open(out, "rb").read() == open(ref, "rb").read(), n_shards


def test_split_graph_of_exchanged_pieces_equals_whole(tmp_path):
    """the multi-GPU data path on one GPU: per-shard handles -> packed sketches in rank order -> union handle's graph"""
    import torch
    from ntjoin_amd.engine import MxEngine
    fa_r, fa_t = str(tmp_path / "ref.fa"), str(tmp_path / "tgt.fa")
    _genome(fa_r, 21, [400_000], n_frac=0.0)
    ref = "".join(line.strip() for line in open(fa_r) if not line.startswith(">"))
    rng = random.Random(3)
    with open(fa_t, "w") as f:                                   # target: pieces of the reference, some reversed
        comp = str.maketrans("ACGT", "TGCA")
        p, i = 0, 0
        while p < len(ref):
            n = rng.randint(20_000, 90_000)
            s = ref[p:p + n]
            if rng.random() < 0.5:
                s = s.translate(comp)[::-1]
            f.write(f">t{i}\n{s}\n")
            p += n + rng.randint(20, 300)
            i += 1
    k, w, world = 32, 250, 3
    with MxEngine(k=k, w=w) as one:
        one.add_fasta("ref", 2.0, fa_r)
        one.add_fasta("tgt", 1.0, fa_t)
        one.sketch()
        one.build_graph()
        g1 = {key: np.asarray(v).copy() for key, v in one.get_graph().items()}
        n_rec = [one.n_records(a) for a in range(2)]
        ids = [one.record_ids(a, n_rec[a]) for a in range(2)]
    engs = [MxEngine(k=k, w=w) for _ in range(world)]
    try:
        for r, eng in enumerate(engs):
            eng.add_fasta_split("ref", 2.0, fa_r, r, world)
            eng.add_fasta_split("tgt", 1.0, fa_t, r, world)
            eng.sketch()
        with MxEngine(k=k, w=w) as union:
            for a, name, wt in ((0, "ref", 2.0), (1, "tgt", 1.0)):
                union.add_minimizers(name, wt, np.zeros(0, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.uint32), ids[a])
                counts = np.array([e.sketch_size(a) for e in engs], dtype=np.uint64)
                nmax = (max(int(counts.max()), 1) + 7) // 8 * 8
                recv = torch.empty(world * 16 * nmax, dtype=torch.uint8, device="cuda")
                for r, e in enumerate(engs):
                    e.pack_sketch_device(a, recv.data_ptr() + r * 16 * nmax, nmax)
                torch.cuda.synchronize()
                union.set_sketch_gathered(a, recv.data_ptr(), nmax, counts, np.zeros(world, dtype=np.uint64))
            union.build_graph()
            g2 = {key: np.asarray(v).copy() for key, v in union.get_graph().items()}
    finally:
        for e in engs:
            e.close()
    assert g1.keys() == g2.keys() and len(g1["vertex_hash"]) > 1000
    for key in g1:
        assert np.array_equal(g1[key], g2[key]), key


def test_split_with_small_batches(tmp_path, monkeypatch):
    """pieces and whole records spread over several sketch batches (the batch boundaries fall between contigs)"""
    monkeypatch.setenv("MXG_SPARSE_BATCH_KMERS", "120000")
    monkeypatch.setenv("MXG_DENSE_BATCH_KMERS", "50000")
    fa = str(tmp_path / "g.fa")
    _genome(fa, 13, LENS)
    for kw in ({}, {"cand_per_window": 2}):
        whole = _whole(fa, 32, 300, **kw)
        got, _, _, _ = _pieces(fa, 32, 300, 5, **kw)
        for key in whole:
            assert np.array_equal(whole[key], got[key]), (kw, key)


def test_fuzz_split_equals_whole(tmp_path):
    """random record sets (plain, short-unit repeats, N runs, long homopolymer stretches: the records of tests/test_gpu_fuzz.py),
    2-9 shards, k = 32 route with batches and stretch routes forced small: the shards' sketches concatenate to the whole file's"""
    from tests.test_gpu_fuzz import _rand_record
    trials = int(os.environ.get("MXG_FUZZ_TRIALS", "15"))
    rng = random.Random(int(os.environ.get("MXG_FUZZ_SEED", "808")))
    knobs = ("MXG_SPARSE_S", "MXG_SPARSE_BATCH_KMERS", "MXG_DEV_GAPS")
    saved = {k_: os.environ.get(k_) for k_ in knobs}
    try:
        for t in range(trials):
            w = rng.choice([50, 200, 500, 1000])
            c = rng.choice([2, 4, 10, 18])
            while c / w > 0.125:
                c //= 2
            os.environ["MXG_SPARSE_S"] = str(rng.choice([64, 128, 320, 512]))
            os.environ["MXG_SPARSE_BATCH_KMERS"] = str(rng.choice([50_000, 200_000, 10**9]))
            os.environ["MXG_DEV_GAPS"] = str(rng.choice([0, 1]))
            fa = str(tmp_path / f"g{t}.fa")
            with open(fa, "w") as f:
                for i in range(rng.randint(1, 7)):
                    s = _rand_record(rng, rng.choice([31, 40, 1500, 20000, 65536, 70000, 140000]))
                    f.write(f">rec{i}\n")
                    for j in range(0, len(s), 80):
                        f.write(s[j:j + 80] + "\n")
            kw = {"cand_per_window": c}
            whole = _whole(fa, 32, w, **kw)
            n_shards = rng.randint(2, 9)
            got, _, _, _ = _pieces(fa, 32, w, n_shards, **kw)
            for key in whole:
                assert np.array_equal(whole[key], got[key]), (t, key, w, c, n_shards, {k_: os.environ[k_] for k_ in knobs})
    finally:
        for k_, v in saved.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v
