"""GPU tests of the code paths that normally only run at multi-Gbp scale, forced at small sizes through the
library's test knobs (environment): many batches per assembly (sparse and dense), the arena-overflow retry,
output-array growth, gap fix-ups in several batches.  Always compared with the CPU oracle, bit for bit."""
import os
import random

import pytest

from tests import _oracle

pytestmark = pytest.mark.gpu


def _records(seed):
    rng = random.Random(seed)
    recs = []
    for r in range(40):
        n = rng.choice([300, 1200, 5000, 20000, 60000])
        s = [rng.choice("ACGT") for _ in range(n)]
        if r % 5 == 0 and n > 4000:  # low-complexity island -> candidate-free stretch or candidate flood
            s[1000:3500] = list((rng.choice(["A", "AC", "ACG", "AAT"]) * 2500)[:2500])
        if r % 7 == 0 and n > 2000:
            s[500:540] = "N" * 40
        recs.append((f"rec{r}", "".join(s)))
    return recs


def _check(oracle, recs, k, w, **kw):
    from ntjoin_amd.engine import MxEngine
    with MxEngine(k=k, w=w, **kw) as eng:
        eng.add_records("x", 1.0, recs)
        eng.sketch()
        sk = eng.get_sketch(0)
        st = eng.stats()
    for r, (rid, seq) in enumerate(recs):
        lo, hi = int(sk["record_first"][r]), int(sk["record_first"][r + 1])
        want = oracle.sketch(seq, k, w, _oracle.V1_MIN if kw.get("variant") == "v1" else _oracle.V2_SUM)
        got = list(zip(sk["out_hash"][lo:hi].tolist(), sk["pos"][lo:hi].tolist(), sk["forward"][lo:hi].tolist()))
        assert got == [(h, p, f) for h, p, f, _ in want], (rid, k, w)
    return st


@pytest.fixture
def knobs():
    saved = {k: os.environ.get(k) for k in ("MXG_SPARSE_BATCH_KMERS", "MXG_DENSE_BATCH_KMERS", "MXG_WAVE_CAP", "MXG_SPARSE_S",
                                             "MXG_RING_SLACK")}
    yield os.environ
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_many_sparse_batches(oracle, knobs):
    knobs["MXG_SPARSE_BATCH_KMERS"] = "30000"
    st = _check(oracle, _records(1), 32, 200)
    assert st["candidates"] > 0


def test_many_dense_batches(oracle, knobs):
    knobs["MXG_DENSE_BATCH_KMERS"] = "25000"
    _check(oracle, _records(2), 32, 200, dense_only=True)
    _check(oracle, _records(2), 20, 7)  # c/w > 1/8: dense kernel chosen automatically


def test_arena_overflow_retry(oracle, knobs):
    knobs["MXG_WAVE_CAP"] = "8"      # every wave overflows its arena slice -> the batch is redone with the needed size
    knobs["MXG_SPARSE_S"] = "128"
    st = _check(oracle, _records(3), 32, 200)
    assert st["candidates"] > 0


def test_large_arena_slices_hash_per_entry(oracle, knobs):
    knobs["MXG_WAVE_CAP"] = "9000"   # beyond the LDS queue of k_reorder: the one-thread-per-entry path
    knobs["MXG_SPARSE_S"] = "256"
    st = _check(oracle, _records(5), 32, 200)
    assert st["candidates"] > 0


@pytest.mark.parametrize("slack", ["7", "60", "400"])
def test_ring_filter_false_positives_are_absent(oracle, knobs, slack):
    """The sparse kernel decides "hash < tau" on the top 31 bits of the two strand hashes and cannot see the carry out
    of the low 33 bits: a captured k-mer can turn out >= tau (about one in 10^9).  MXG_RING_SLACK widens the ring
    threshold so that 7 % / 60 % / 400 % more k-mers are captured than are candidates: k_resolve must treat every one
    of them as absent (never selected, never blocking, stepped over by the stretch detection), with and without
    candidate-free stretches."""
    knobs["MXG_SPARSE_BATCH_KMERS"] = "60000"
    base = _check(oracle, _records(6), 32, 200)["candidates"]
    knobs["MXG_RING_SLACK"] = slack
    st = _check(oracle, _records(6), 32, 200)
    assert st["candidates"] > base * (1 + int(slack) / 100) * 0.9   # the captured-but-absent entries are really there
    st = _check(oracle, _records(7), 32, 500, cand_per_window=2)   # few real candidates: stretches between absent entries
    assert st["dense_kmers"] > 0
    _check(oracle, _records(8), 21, 300, cand_per_window=4, variant="v1")  # min(fwd,rev): the ring test is exact here


def test_gaps_in_several_batches(oracle, knobs):
    knobs["MXG_SPARSE_BATCH_KMERS"] = "70000"
    # very few candidates per window -> candidate-free stretches everywhere -> dense fix-ups in every batch
    st = _check(oracle, _records(4), 32, 500, cand_per_window=2)
    assert st["dense_kmers"] > 0


def test_output_growth(oracle, knobs):
    # homopolymers: every k-mer is a minimizer (rightmost tie rule), far more than the 3x density estimate reserved
    recs = [("polyA", "A" * 30000), ("mix", "ACGT" * 5000), ("polyT", "T" * 12000)]
    _check(oracle, recs, 16, 50)
    _check(oracle, recs, 16, 50, dense_only=True)


@pytest.fixture
def dev_knobs(knobs):
    saved = {k: os.environ.get(k) for k in ("MXG_DEV_GAPS", "MXG_DEV_CAND")}
    yield knobs
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _plain_records(seed, n_rec, lo, hi):
    rng = random.Random(seed)
    return [(f"r{r}", "".join(rng.choice("ACGT") for _ in range(rng.randint(lo, hi)))) for r in range(n_rec)]


@pytest.mark.parametrize("cand,w,variant", [(5, 200, "v2"), (3, 500, "v2"), (8, 100, "v2"), (4, 300, "v1")])
def test_stretches_fixed_up_on_the_device(oracle, dev_knobs, cand, w, variant):
    """MXG_DEV_GAPS=1: the route genome-scale assemblies take (k_gap_fix / k_gap_post / k_merge_fin), forced on small random
    records with few candidates per window so that hundreds of candidate-free stretches occur -- bit-exact against the
    oracle, and the stretches were really sketched on the device (dense_kmers counts the k-mers hashed there)"""
    dev_knobs["MXG_DEV_GAPS"] = "1"
    recs = _plain_records(31, 12, 20_000, 90_000) + [("short", "ACGT" * 40)]
    st = _check(oracle, recs, 32 if variant == "v2" else 25, w, cand_per_window=cand, variant=variant)
    # (k = 32: the stretches between two candidates of a slice are sketched by k_sel_stretch right behind the slice kernel)
    assert (st["dense_kmers"] > 0 or st["slice_stretches"] > 0) and st["candidates"] > 0


def test_device_stretch_route_in_chained_batches(oracle, dev_knobs):
    """several batches per assembly enqueued back to back on the two streams (the sum of the earlier batches travels
    through a device word), with and without stretches, plus what the device route must hand back to the host: a
    low-complexity island (more minimizers in one stretch than its region holds) and N-runs inside stretches"""
    dev_knobs["MXG_DEV_GAPS"] = "1"
    dev_knobs["MXG_SPARSE_BATCH_KMERS"] = "150000"
    recs = _plain_records(32, 30, 15_000, 70_000)
    st = _check(oracle, recs, 32, 200, cand_per_window=5)
    assert st["dense_kmers"] > 0 or st["slice_stretches"] > 0
    st = _check(oracle, recs, 32, 200)            # 18 candidates per window: (almost) no stretch, chained all the same
    assert st["candidates"] > 0
    _check(oracle, _records(4), 32, 500, cand_per_window=2)   # islands + N-runs: falls back to the general route
    dev_knobs["MXG_DEV_GAPS"] = "0"
    _check(oracle, recs, 32, 200, cand_per_window=5)           # chained batches, stretches through the host route


def test_two_assemblies_many_batches_and_graph(oracle, dev_knobs):
    """two assemblies x several batches interleaved on the two streams, then the graph stage: same result as one batch each"""
    from ntjoin_amd.engine import MxEngine
    import numpy as np
    recs_a = _plain_records(41, 25, 10_000, 50_000)
    recs_b = [(f"b{i}", s[::-1].translate(str.maketrans("ACGT", "TGCA")) if i % 2 else s) for i, (_, s) in enumerate(recs_a)][::-1]
    results = []
    for batch, dev in (("1000000000", "0"), ("120000", "1"), ("90000", "0")):
        dev_knobs["MXG_SPARSE_BATCH_KMERS"] = batch
        dev_knobs["MXG_DEV_GAPS"] = dev
        with MxEngine(k=32, w=150, cand_per_window=6) as eng:
            eng.add_records("a", 2.0, recs_a)
            eng.add_records("b", 1.0, recs_b)
            eng.sketch(-2)
            eng.build_graph()
            results.append(([eng.get_sketch(a) for a in range(2)], eng.get_graph()))
    for sks, g in results[1:]:
        for a in range(2):
            for key in ("out_hash", "pos", "record", "forward"):
                assert np.array_equal(sks[a][key], results[0][0][a][key]), key
        for key in ("vertex_hash", "edge_u", "edge_v", "edge_support", "edge_weight"):
            assert np.array_equal(g[key], results[0][1][key]), key
    first = results[0][0][0]["record_first"]
    for r, (_, seq) in enumerate(recs_a[:5]):
        want = oracle.sketch(seq, 32, 150)
        lo, hi = int(first[r]), int(first[r + 1])
        assert results[0][0][0]["out_hash"][lo:hi].tolist() == [x[0] for x in want]


@pytest.mark.parametrize("variant", ["v2", "v1"])
def test_exact_hash_from_position_tables_every_group_shape(oracle, variant):
    """k_reorder_w takes the exact hash of a candidate from position tables: groups of 8 packed bytes (32 bases) with no
    rotation, combined by rotations of 32 for k > 35, a partial last group, and k % 4 ordinary steps -- every shape of
    (full groups, leftover bytes, leftover bases), and the old block-per-slice kernel (MXG_REORDER_W=0) on the same input"""
    recs = _records(11)[:12]
    saved = os.environ.get("MXG_REORDER_W")
    try:
        for k in (3, 4, 7, 8, 29, 31, 32, 33, 35, 36, 39, 40, 61, 63, 64, 65, 67, 68, 96, 99, 100, 131, 200):
            for rw in ("1", "0") if k in (32, 47, 100) or k % 32 == 3 else ("1",):
                os.environ["MXG_REORDER_W"] = rw
                st = _check(oracle, recs, k, 100, variant=variant, cand_per_window=6)
                assert st["candidates"] > 0 or k < 8, k
    finally:
        if saved is None:
            os.environ.pop("MXG_REORDER_W", None)
        else:
            os.environ["MXG_REORDER_W"] = saved


def test_more_candidates_than_the_launch_was_sized_for(oracle, knobs):
    """Batches enqueued without a host sync launch k_resolve / k_emit for the expected number of candidates + 30 %; a batch
    with more never reports and the host redoes the assembly.  MXG_GRID_BY_ESTIMATE=2 sizes the launch for HALF the
    expectation, so every batch takes that way out -- with one batch per assembly, with several, with stretches."""
    saved = os.environ.get("MXG_GRID_BY_ESTIMATE")
    os.environ["MXG_GRID_BY_ESTIMATE"] = "2"
    try:
        _check(oracle, _records(21), 32, 200)
        knobs["MXG_SPARSE_BATCH_KMERS"] = "50000"
        _check(oracle, _records(22), 32, 200)
        st = _check(oracle, _records(23), 32, 500, cand_per_window=2)
        assert st["dense_kmers"] > 0
    finally:
        if saved is None:
            os.environ.pop("MXG_GRID_BY_ESTIMATE", None)
        else:
            os.environ["MXG_GRID_BY_ESTIMATE"] = saved


def _repeat_rich_records(seed, n_rec, rec_len):
    """records with the structure real genomes add to random sequence: interspersed copies of a few repeat families
    (5-15 % diverged), tandem arrays of a 171-base unit, homopolymer / dinucleotide runs and a few N gaps"""
    rng = random.Random(seed)
    fams = ["".join(rng.choice("ACGT") for _ in range(rng.choice([300, 1200, 6000]))) for _ in range(6)]
    sat = "".join(rng.choice("ACGT") for _ in range(171))
    recs = []
    for r in range(n_rec):
        parts, n = [], 0
        while n < rec_len:
            kind = rng.random()
            if kind < 0.55:
                p = "".join(rng.choice("ACGT") for _ in range(rng.randint(500, 8000)))
            elif kind < 0.85:
                f = list(rng.choice(fams))
                div = rng.uniform(0.05, 0.15)
                for i in range(len(f)):
                    if rng.random() < div:
                        f[i] = rng.choice("ACGT")
                p = "".join(f)
                if rng.random() < 0.5:
                    p = p[::-1].translate(str.maketrans("ACGT", "TGCA"))
            elif kind < 0.93:
                p = sat * rng.randint(20, 120)
            elif kind < 0.98:
                p = rng.choice(["A", "T", "AC", "AT", "GGC"]) * rng.randint(30, 900)
            else:
                p = "N" * rng.choice([1, 20, 300])
            parts.append(p)
            n += len(p)
        recs.append((f"chr{r}", "".join(parts)[:rec_len]))
    return recs


@pytest.mark.parametrize("w,cand", [(200, 6), (500, 10)])
def test_repeat_rich_genome_on_the_device_route(oracle, dev_knobs, w, cand):
    """what random test genomes lack: repeat families, satellite arrays, low-complexity runs -- candidate floods (one k-mer
    below tau occurs thousands of times), stretches with more minimizers than a region holds, arena slices that overflow.
    Through the route genome-scale assemblies take, in several chained batches; bit-exact against the oracle."""
    dev_knobs["MXG_DEV_GAPS"] = "1"
    dev_knobs["MXG_SPARSE_BATCH_KMERS"] = "400000"
    recs = _repeat_rich_records(5, 6, 250_000)
    st = _check(oracle, recs, 32, w, cand_per_window=cand)
    assert st["candidates"] > 0


def _repeat_rich_numpy(seed, n_rec, rec_len):
    from ntjoin_amd import synth
    return [(f"chr{i}", synth.to_ascii5(c).decode()) for i, c in enumerate(synth.repeat_rich_records(seed, n_rec, rec_len))]


@pytest.fixture
def route_knobs(dev_knobs):
    saved = {k: os.environ.get(k) for k in ("MXG_GAP_BUDGET", "MXG_GRID_BY_ESTIMATE", "MXG_STRETCH_DENSE", "MXG_BS_SELECT", "MXG_GAP_WHOLE",
                                            "MXG_GAP_DEV_CAP")}
    yield dev_knobs
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_stretches_left_to_the_tile_kernel(oracle, route_knobs):
    """candidate-free stretches longer than the device route's 4096 k-mers (homopolymers, short-unit arrays, some of them
    across N gaps) -- handed to k_stretch_tiles and merged into the sketch; the same through the dense pipeline
    (MXG_STRETCH_DENSE=1).  (Stretches of up to 4096 k-mers stay with k_gap_fix, however many minimizers they hold and
    whatever invalid bases they cross: the repeat-rich records are full of them.)"""
    route_knobs["MXG_DEV_GAPS"] = "1"
    route_knobs["MXG_SPARSE_BATCH_KMERS"] = "600000"
    import random
    rng = random.Random(17)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    recs = _repeat_rich_numpy(3, 4, 700_000)
    recs += [("long_a", rnd(4000) + "A" * 9000 + rnd(30000) + "AC" * 3500 + rnd(2000)),
             ("long_n", rnd(9000) + "TTG" * 1500 + "N" * 7 + "TTG" * 1500 + rnd(20000) + "C" * 5000 + "N" + "C" * 4000 + rnd(800)),
             ("unit7", rnd(600) + "ACGGTCA" * 3000), ("unit5_start", "GATTA" * 2500 + rnd(12000)),
             ("long_b", rnd(100) + "GA" * 2600 + rnd(60000) + "T" * 4700 + rnd(3000) + "AGC" * 2000 + rnd(10))]
    # round 5: the slice kernel reports a long stretch in pieces of 4096 k-mers that overlap by one window (a piece leaves out the
    # arg-min of the window it shares with the piece before it), so the long stretches stay with k_gap_fix as well
    # (their thousands of minimizers now count towards the batch's output: in records as small as these they outgrow the arrays
    # sized for five times the i.i.d. density and the assembly takes another round -- the result must be the oracle's either way)
    st0 = _check(oracle, recs, 32, 500, cand_per_window=10)
    st0b = _check(oracle, recs, 32, 200, cand_per_window=6)
    route_knobs["MXG_GAP_WHOLE"] = "1"     # ... reported whole: the route of rounds 3 and 4
    st = _check(oracle, recs, 32, 500, cand_per_window=10)
    assert st["deferred_stretches"] > 5 and st["batches_redone"] == 0 and st["sync_assemblies"] == 0
    assert st0["deferred_stretches"] < st["deferred_stretches"] and st0b["deferred_stretches"] < st["deferred_stretches"]
    st2 = _check(oracle, recs, 32, 200, cand_per_window=6)
    assert st2["deferred_stretches"] > 5
    route_knobs["MXG_STRETCH_DENSE"] = "1"
    st3 = _check(oracle, recs, 32, 500, cand_per_window=10)
    assert st3["deferred_stretches"] == st["deferred_stretches"]


def test_tile_borders_of_long_stretches(oracle, route_knobs):
    """one record that is a single candidate-free stretch of many tiles (a 7-base unit repeated: 7 distinct k-mers, none below
    the threshold at 2 candidates per window), another of a homopolymer (every k-mer equal): every tile border decides which
    tile reports the minimizer the windows on both sides share"""
    route_knobs["MXG_DEV_GAPS"] = "1"
    rng = random.Random(9)
    unit = "ACGGTCA"
    recs = [("sat", unit * 3000), ("polyA", "A" * 9000),
            ("mixed", "".join(rng.choice("ACGT") for _ in range(30000)) + "AC" * 4000 + "".join(rng.choice("ACGT") for _ in range(20000)))]
    for w in (100, 1000, 2048):
        st = _check(oracle, recs, 32, w, cand_per_window=2)
        assert st["deferred_stretches"] > 0 or st["dense_kmers"] > 0


def test_second_attempt_with_resized_batches(oracle, route_knobs):
    """more stretches in a batch than the device route holds (forced: budget of 2 expected stretches per batch is what the
    i.i.d. estimate plans for; the repeat-rich records hold hundreds): the assembly goes through the streams a second time with
    batches cut for the density the first attempt met, and nothing is left to the synchronous route"""
    route_knobs["MXG_DEV_GAPS"] = "1"
    recs = _repeat_rich_numpy(4, 6, 900_000)
    from ntjoin_amd.engine import MxEngine
    with MxEngine(k=32, w=300, cand_per_window=8) as eng:
        eng.add_records("x", 1.0, recs)
        os.environ["MXG_SPARSE_BATCH_KMERS"] = "3000000"
        eng.sketch()
        st = eng.stats()
        sk = eng.get_sketch(0)
        # a second sketch of the same handle starts from what the first one learnt
        eng.reset_timers()
        eng.sketch()
        st_warm = eng.stats()
    for r, (rid, seq) in enumerate(recs):
        lo, hi = int(sk["record_first"][r]), int(sk["record_first"][r + 1])
        want = oracle.sketch(seq, 32, 300)
        assert list(zip(sk["out_hash"][lo:hi].tolist(), sk["pos"][lo:hi].tolist())) == [(h, p) for h, p, _, _ in want], rid
    if st["retried_assemblies"]:  # (the device route's capacity was exceeded: GAP_DEV_MAX stretches in a batch)
        assert st["sync_assemblies"] == 0 and st["batches_redone"] == 0
        assert st_warm["retried_assemblies"] == 0 and st_warm["batches_redone"] == 0


def test_batches_redone_one_by_one(oracle, route_knobs):
    """a batch that never reports (its grids sized below its candidate count: MXG_GRID_BY_ESTIMATE=2) is redone, with the
    batches behind it, through the synchronous route -- after the second pipelined attempt, which sizes the grids in full"""
    route_knobs["MXG_DEV_GAPS"] = "1"
    route_knobs["MXG_SPARSE_BATCH_KMERS"] = "150000"
    route_knobs["MXG_GRID_BY_ESTIMATE"] = "2"
    route_knobs["MXG_BS_SELECT"] = "0"   # (grids sized by an estimate exist on the count -> reorder -> resolve route only)
    st = _check(oracle, _records(41), 32, 200)
    assert st["retried_assemblies"] + st["batches_redone"] + st["sync_assemblies"] > 0


def test_kept_batches_keep_their_deferred_stretches(oracle, route_knobs):
    """batch 0 hands a long homopolymer stretch to the tile kernel, the last batch (a short-unit repeat: no candidate at all)
    never reports the common way, so the assembly ends in the per-batch redo with batches 0 and 1 kept: the stretch batch 0
    deferred is merged all the same (found by the k=32 fuzz test: 117 minimizers for 80 445)"""
    route_knobs["MXG_DEV_GAPS"] = "1"
    route_knobs["MXG_SPARSE_BATCH_KMERS"] = "150000"
    route_knobs["MXG_SPARSE_S"] = "512"
    rng = random.Random(44)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    recs = [("tiny", rnd(33)), ("polyA", rnd(46000) + "".join(rng.choice("Aa") for _ in range(81000)) + rnd(13000)),
            ("plain", rnd(140000)), ("unit", ("ACGGTCA" * 10000)[:65536])]
    st = _check(oracle, recs, 32, 1000, cand_per_window=4)
    if os.environ.get("MXG_BS_SELECT") == "0" and os.environ.get("MXG_BS", "1") != "0":  # (the default route's bookkeeping)
        assert st["deferred_stretches"] >= 1
        assert st["batches_redone"] >= 1


def test_dense_batch_of_records_without_a_kmer(oracle, knobs):
    """records shorter than k around records larger than the dense batch budget: no batch is launched with an empty grid, the
    records without a k-mer come out with empty sketches"""
    knobs["MXG_DENSE_BATCH_KMERS"] = "2000"
    rng = random.Random(12)
    recs = [("t1", "ACGT"), ("t2", "AC"), ("big", "".join(rng.choice("ACGT") for _ in range(6000))), ("t3", "A" * 31),
            ("big2", "".join(rng.choice("ACGT") for _ in range(5000)))]
    _check(oracle, recs, 32, 10, dense_only=True)
    _check(oracle, recs, 32, 100, dense_only=True)
