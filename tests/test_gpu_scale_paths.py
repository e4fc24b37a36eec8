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
