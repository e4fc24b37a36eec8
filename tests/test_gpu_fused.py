"""GPU test of mxg_sketch_graph (sketch + graph stage in one call, one host sync in the common case): same sketches, flags
and graph as mxg_sketch + mxg_build_graph, on the goldens, on configs[1]-shaped input, and on the inputs that leave the
common case (candidate-free stretches, arena overflow, dense path, an assembly that came as a TSV)."""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN, golden_cases, load_case

pytestmark = pytest.mark.gpu
FASTA = os.path.join(GOLDEN, "fasta")
CASES = [m["name"] for m in golden_cases()]


def _same(e1, e2, n_asm):
    for a in range(n_asm):
        s1, s2 = e1.get_sketch(a), e2.get_sketch(a)
        for key in ("out_hash", "pos", "record", "forward", "record_first"):
            assert np.array_equal(s1[key], s2[key]), (a, key)
        assert np.array_equal(e1.get_mx_flags(a), e2.get_mx_flags(a)), a
    g1, g2 = e1.get_graph(), e2.get_graph()
    for key in g1:
        assert np.array_equal(np.asarray(g1[key]), np.asarray(g2[key])), key
    st1, st2 = e1.stats(), e2.stats()
    for key in ("minimizers", "unique", "vertices", "edges"):
        assert st1[key] == st2[key], key


@pytest.mark.parametrize("name", CASES)
def test_fused_equals_two_calls_on_goldens(name):
    from ntjoin_amd.engine import MxEngine
    meta = load_case(name)["meta"]
    asms = meta["refs"] + [meta["target"]]
    with MxEngine(k=meta["k"], w=meta["w"], variant=meta["variant"]) as e1, MxEngine(k=meta["k"], w=meta["w"], variant=meta["variant"]) as e2:
        for a in asms:
            e1.add_fasta(a["tsv"], a["weight"], os.path.join(FASTA, a["fasta"]))
            e2.add_fasta(a["tsv"], a["weight"], os.path.join(FASTA, a["fasta"]))
        e1.sketch_graph()
        e1.sketch_graph()          # again on the same handle (buffers, counters and flags of the first run are stale)
        e2.sketch()
        e2.build_graph()
        _same(e1, e2, len(asms))


def _packed(eng, name, weight, recs):
    import torch
    from ntjoin_amd import synth
    words, starts, lens = synth.pack_records(recs)
    d = torch.from_numpy(words.view(np.int32)).cuda()
    eng.add_packed_device(name, weight, d.data_ptr(), starts, lens, keepalive=d)


@pytest.mark.parametrize("kw", [dict(), dict(cand_per_window=2), dict(dense_only=True)])
def test_fused_on_synthetic_genome(kw):
    """common case (one sync), candidate-free stretches everywhere (fallback), dense path (fallback)"""
    from ntjoin_amd import synth
    from ntjoin_amd.engine import MxEngine
    ref, tgt = synth.config2(seed=3, n_bases=20_000_000)
    with MxEngine(k=32, w=1000, **kw) as e1, MxEngine(k=32, w=1000, **kw) as e2:
        for e in (e1, e2):
            _packed(e, "ref", 2.0, ref)
            _packed(e, "tgt", 1.0, tgt)
        e1.sketch_graph()
        e2.sketch(-2)
        e2.build_graph()
        _same(e1, e2, 2)
        assert e1.stats()["vertices"] > 10000


@pytest.mark.parametrize("cand", [10, 3])
def test_fused_with_stretches_on_the_device(monkeypatch, cand):
    """the stretches stay on the device in the one-call mode too (MXG_DEV_GAPS=1: as on genome-scale input): one host sync,
    nothing redone; a stretch the device route hands to the host (a homopolymer of 9000) sends the call the long way round"""
    import random
    from ntjoin_amd import synth
    from ntjoin_amd.engine import MxEngine
    monkeypatch.setenv("MXG_DEV_GAPS", "1")
    ref, tgt = synth.config2(seed=6, n_bases=20_000_000)
    for extra in (False, True):
        with MxEngine(k=32, w=1000, cand_per_window=cand) as e1, MxEngine(k=32, w=1000, cand_per_window=cand) as e2:
            for e in (e1, e2):
                _packed(e, "ref", 2.0, ref)
                _packed(e, "tgt", 1.0, tgt)
                if extra:
                    rng = random.Random(3)
                    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
                    e.add_records("odd", 1.5, [("x", (rnd(30000) + "A" * 9000 + rnd(30000)).encode())])
            e1.sketch_graph()
            e2.sketch(-2)
            e2.build_graph()
            _same(e1, e2, 3 if extra else 2)
            st = e1.stats()
            if not extra and cand == 10:  # (3 candidates per window: more stretches than one batch holds, and the call needs one batch)
                assert st["batches_redone"] == 0 and st["sync_assemblies"] == 0 and st["deferred_stretches"] == 0, st


def test_fused_after_arena_overflow(monkeypatch):
    from ntjoin_amd import synth
    from ntjoin_amd.engine import MxEngine
    monkeypatch.setenv("MXG_WAVE_CAP", "8")      # every wave overflows its arena slice: the batch is redone
    ref, tgt = synth.config2(seed=4, n_bases=3_000_000)
    with MxEngine(k=32, w=200) as e1, MxEngine(k=32, w=200) as e2:
        for e in (e1, e2):
            _packed(e, "ref", 2.0, ref)
            _packed(e, "tgt", 1.0, tgt)
        e1.sketch_graph()
        e2.sketch(-2)
        e2.build_graph()
        _same(e1, e2, 2)


def test_fused_with_an_assembly_from_tsv():
    from ntjoin_amd.engine import MxEngine
    meta = load_case("synth3_w50")["meta"]
    asms = meta["refs"] + [meta["target"]]
    cdir = os.path.join(GOLDEN, "cases", "synth3_w50")
    with MxEngine(k=meta["k"], w=meta["w"]) as e1, MxEngine(k=meta["k"], w=meta["w"]) as e2:
        for i, a in enumerate(asms):
            for e in (e1, e2):
                if i == 0:
                    e.add_tsv(a["tsv"], a["weight"], os.path.join(cdir, a["tsv"]))
                else:
                    e.add_fasta(a["tsv"], a["weight"], os.path.join(FASTA, a["fasta"]))
        e1.sketch_graph()
        e2.sketch()
        e2.build_graph()
        g1, g2 = e1.get_graph(), e2.get_graph()
        for key in g1:
            assert np.array_equal(np.asarray(g1[key]), np.asarray(g2[key])), key


def test_fused_call_on_assemblies_of_several_batches():
    """an assembly with more k-mers than one launch of the slice kernel takes (MXG_SEL_BATCH_KMERS; configs[4]'s 20 Gbp at the
    default) cannot keep the graph stage enqueued behind it: mxg_sketch_graph then runs every batch of every assembly through
    the streams and the graph stage behind them -- the same result as the two calls, not the synchronous route (round 6: the
    bench line of configs[4] took 56 ms per step through it instead of 19)"""
    from ntjoin_amd import synth
    from ntjoin_amd.engine import MxEngine
    ref, tgt = synth.config2(seed=5, n_bases=12_000_000)
    saved = os.environ.get("MXG_SEL_BATCH_KMERS")
    os.environ["MXG_SEL_BATCH_KMERS"] = "3000000"
    try:
        with MxEngine(k=32, w=1000) as e1, MxEngine(k=32, w=1000) as e2:
            for e in (e1, e2):
                _packed(e, "ref", 2.0, ref)
                _packed(e, "tgt", 1.0, tgt)
            e1.sketch_graph()
            e1.sketch_graph()
            e2.sketch(-2)
            e2.build_graph()
            _same(e1, e2, 2)
            assert e1.stats()["sync_assemblies"] == 0 and e1.stats()["select_slices"] > 0
    finally:
        if saved is None:
            os.environ.pop("MXG_SEL_BATCH_KMERS", None)
        else:
            os.environ["MXG_SEL_BATCH_KMERS"] = saved
