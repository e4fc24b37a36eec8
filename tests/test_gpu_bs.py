"""GPU tests of the k = 32 route of the sketch stage (csrc/bs_kernels.h, sketch_bs.hip: k_bs_select; k_bs_count / k_bs_reorder_w):

  * the generated filter kernel against the direct ring formula on random bases (standalone binary built with the library:
    layout kernel + filter over whole chunks, ragged tail, first / middle / last chunks compared word for word);
  * the three routes of the library -- bit-sliced filter + the slice kernel k_bs_select (default), bit-sliced filter + the batch
    kernels count -> reorder -> resolve (MXG_BS_SELECT=0), the rolling-hash kernel (MXG_BS=0) -- against the CPU oracle on the
    same records, bit for bit, with the statistics saying which route ran;
  * slices that outgrow their LDS queue (MXG_SEL_QCAP) work in global memory, and a batch without a region left is redone;
  * inputs the filter does not take (k != 32, the min(fwd, rev) variant) still go the old way.
"""
import os
import random
import subprocess

import pytest

from tests import _oracle
from tests.test_gpu_scale_paths import _check, _records

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUTE_KNOBS = ("MXG_GAP_POOL", "MXG_BS", "MXG_BS_SELECT", "MXG_SEL_QCAP", "MXG_SEL_RK", "MXG_GAP_WHOLE", "MXG_GAP_DEV_CAP", "MXG_SPARSE_BATCH_KMERS", "MXG_WAVE_CAP", "MXG_SPARSE_S", "MXG_DEV_GAPS", "MXG_SEL_INLINE")


@pytest.fixture
def env():
    saved = {k: os.environ.get(k) for k in ROUTE_KNOBS}
    yield os.environ
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("mbp,tt", [(1, 164), (3, 40), (2, 0), (1, 16383)])
def test_filter_kernel_against_direct_formula(mbp, tt):
    exe = os.path.join(REPO, "ntjoin_amd", "bin", "bs_check")
    assert os.path.exists(exe), "ntjoin_amd/bin/bs_check missing: run __graft_entry__.build()"
    r = subprocess.run([exe, str(mbp), str(tt)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "verify: ok" in r.stdout


def _bases(recs):
    return sum(len(s) for _, s in recs)


@pytest.mark.parametrize("seed,w", [(11, 200), (12, 1000), (13, 150)])
def test_three_routes_agree_with_the_oracle(oracle, env, seed, w):
    recs = _records(seed)
    env["MXG_BS"] = "1"
    st = _check(oracle, recs, 32, w)
    assert st["bs_filter_bases"] == _bases(recs), "the k = 32 route did not run"
    assert st["candidates"] > 0
    env["MXG_BS_SELECT"] = "0"
    st = _check(oracle, recs, 32, w)
    assert st["bs_filter_bases"] == _bases(recs)
    env["MXG_BS_SELECT"] = "1"
    env["MXG_BS"] = "0"
    st = _check(oracle, recs, 32, w)
    assert st["bs_filter_bases"] == 0


@pytest.mark.parametrize("S,w", [(320, 1000), (64, 200), (128, 500), (512, 1000), (1024, 777)])
def test_select_route_runs_and_agrees(oracle, env, S, w):
    """k_bs_select on records with N runs, low-complexity islands and short records, strips of every template size"""
    env["MXG_SPARSE_S"] = str(S)
    env["MXG_BS"] = "1"
    env["MXG_BS_SELECT"] = "1"
    st = _check(oracle, _records(100 + S), 32, w)
    assert st["select_slices"] > 0, "k_bs_select did not run"
    env["MXG_DEV_GAPS"] = "1"      # candidate-free stretches stay on the device: few candidates per window make many
    st = _check(oracle, _records(200 + S), 32, w, cand_per_window=3)
    assert st["select_slices"] > 0
    env["MXG_SPARSE_BATCH_KMERS"] = "50000"
    st = _check(oracle, _records(300 + S), 32, w, cand_per_window=5)
    assert st["select_slices"] > 0


@pytest.mark.parametrize("w", [1000, 2500, 200])
def test_select_route_is_the_default_for_small_inputs(oracle, env, w):
    """no strip-length knob: an input of a few hundred kbp takes the same route as a genome (strips long enough for the
    slice kernel's halo at any w it covers)"""
    env.pop("MXG_SPARSE_S", None)
    st = _check(oracle, _records(400 + w), 32, w)
    assert st["select_slices"] > 0


def test_select_slices_beyond_their_queue(oracle, env):
    """MXG_SEL_QCAP=64: nearly every slice has more raw candidates than its LDS queue holds and works in a region of global
    memory; the results do not change"""
    env["MXG_SPARSE_S"] = "320"
    env["MXG_SEL_QCAP"] = "64"
    env["MXG_DEV_GAPS"] = "1"
    st = _check(oracle, _records(51), 32, 1000)
    assert st["select_slices"] > 0
    st = _check(oracle, _records(52), 32, 300, cand_per_window=12)
    assert st["select_slices"] > 0


@pytest.mark.parametrize("dev_gaps", ["0", "1"])
def test_select_slice_that_gives_up_is_redone(oracle, env, dev_gaps):
    """MXG_SEL_RK=4: a slice has room for four selected candidates, so nearly every slice gives up (its output would be
    truncated); the kernel's flag sends the batch through the other route, with and without the device's stretch route"""
    env["MXG_SPARSE_S"] = "320"
    env["MXG_SEL_RK"] = "4"
    env["MXG_DEV_GAPS"] = dev_gaps
    _check(oracle, _records(61), 32, 1000)
    _check(oracle, _records(62), 32, 300, cand_per_window=12)


def test_select_stretch_ends_behind_the_slice(oracle, env):
    """candidate-free stretches far longer than a slice's halo (a satellite-like array, a homopolymer, a contig without any
    candidate): the reporting slice walks on to the stretch's end"""
    rng = random.Random(5)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    unit = rnd(171)
    recs = [("sat", rnd(30000) + unit * 400 + rnd(30000)), ("polyA", rnd(5000) + "A" * 90000 + rnd(2000)),
            ("unit7", ("ACGGTCA" * 20000)[:100000]), ("plain", rnd(200000)), ("polyT_end", rnd(3000) + "T" * 50000)]
    env["MXG_SPARSE_S"] = "320"
    env["MXG_DEV_GAPS"] = "1"
    for c in (2, 10):
        st = _check(oracle, recs, 32, 1000, cand_per_window=c)
        assert st["select_slices"] > 0


@pytest.mark.parametrize("dev_gaps", ["0", "1"])
def test_stretches_are_sketched_behind_the_slice_kernel(oracle, env, dev_gaps):
    """few candidates per window on plain random sequence: hundreds of candidate-free stretches just over a window long, between two
    candidates of a slice, at a contig's start and end, reaching past the slice's strips -- k_sel_stretch sketches them one wave per
    slice and puts their minimizers into the slice's row (several stretches in one slice: the last one first); with a window of
    2500 and 3 candidates a stretch is taken in pieces; MXG_SEL_INLINE=0: the same sketch with every stretch through k_gap_fix"""
    rng = random.Random(77)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    recs = [("a", rnd(700000)), ("b", rnd(1500)), ("c", rnd(1031)), ("d", rnd(300000)), ("n_inside", rnd(40000) + "N" + rnd(50000) + "NNNN" + rnd(800)),
            ("e", rnd(2200)), ("f", rnd(450000))]
    env["MXG_SPARSE_S"] = "320"
    env["MXG_DEV_GAPS"] = dev_gaps
    for w, c in ((1000, 4), (1000, 7), (300, 3), (2500, 3), (64, 2)):
        st = _check(oracle, recs, 32, w, cand_per_window=c)
        assert st["select_slices"] > 0 and st["deferred_stretches"] == 0, (w, c)
        assert st["slice_stretches"] > 5, (w, c, st["slice_stretches"])
    env["MXG_SEL_INLINE"] = "0"
    st = _check(oracle, recs, 32, 1000, cand_per_window=4)
    assert st["slice_stretches"] == 0


def test_low_complexity_stretches_stay_on_the_device(oracle, env):
    """di- and trinucleotide runs longer than a window: every second / third k-mer of the run is a minimizer (the rightmost of
    equal hashes wins each time the window moves on) -- hundreds per candidate-free stretch, all of them through k_gap_fix"""
    rng = random.Random(8)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    recs = [("di", rnd(20000) + "AC" * 900 + rnd(7000) + "AT" * 1400 + rnd(9000)),
            ("tri", rnd(3000) + "GGC" * 800 + rnd(30000) + "GGC" * 1000 + rnd(100)),
            ("plain", rnd(150000)), ("di_end", rnd(9000) + "TG" * 1200),
            # invalid bases inside the stretch: its k-mers come from two or three valid runs
            ("di_n", rnd(12000) + "AC" * 500 + "N" + "AC" * 600 + rnd(8000) + "AT" * 450 + "N" * 20 + "AT" * 450 + "N" * 300 + "AT" * 300
             + rnd(5000)),
            ("tri_n", rnd(700) + "N" * 5 + "GGC" * 700 + "NN" + rnd(40000))]
    env["MXG_DEV_GAPS"] = "1"
    for w in (1000, 600):
        st = _check(oracle, recs, 32, w)
        assert st["select_slices"] > 0
        assert st["deferred_stretches"] == 0, st["deferred_stretches"]
    # a pool of 300 entries: the stretches that find it used up go to the host and through the tile kernel, same sketch
    env["MXG_GAP_POOL"] = "300"
    st = _check(oracle, recs, 32, 1000)
    assert st["deferred_stretches"] > 0
    env.pop("MXG_GAP_POOL")
    # more invalid bases inside a stretch than the block's words take: handed over as well
    recs2 = [("wide_n", rnd(6000) + "AT" * 900 + "N" * 3000 + "AT" * 900 + rnd(9000)), ("plain", rnd(90000))]
    st = _check(oracle, recs2, 32, 1000)
    assert st["deferred_stretches"] > 0


def test_long_stretches_are_reported_in_pieces(oracle, env):
    """candidate-free stretches longer than k_gap_fix's 4096 k-mers (a homopolymer of 90 kb, a satellite-like array, dinucleotide
    runs, one of them across N): the slice kernel reports them in pieces that overlap by one window, every piece after the first
    leaving out the arg-min of the window it shares with the piece before it -- nothing goes to the host.  Then with a stretch
    pool of 300 entries: pieces that find the pool used up are handed over, marked pieces among them (the tile kernel starts those
    at their second window), same sketch."""
    rng = random.Random(23)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    unit = rnd(171)
    recs = [("polyA", rnd(5000) + "A" * 90000 + rnd(2000)), ("sat", rnd(30000) + unit * 300 + rnd(30000)),
            ("di", rnd(800) + "AC" * 9000 + rnd(7000) + "TG" * 2300 + "N" * 3 + "TG" * 2500 + rnd(900)),
            ("unit7_whole", ("ACGGTCA" * 6000)[:40000]), ("plain", rnd(120000)), ("polyT_end", rnd(3000) + "T" * 20000)]
    env["MXG_SPARSE_S"] = "320"
    env["MXG_DEV_GAPS"] = "1"
    for w, c in ((1000, 10), (2048, 10), (300, 4), (1000, 2)):
        st = _check(oracle, recs, 32, w, cand_per_window=c)
        assert st["select_slices"] > 0 and st["deferred_stretches"] == 0, (w, c, st["deferred_stretches"])
    env["MXG_GAP_POOL"] = "300"
    st = _check(oracle, recs, 32, 1000, cand_per_window=10)
    assert st["deferred_stretches"] > 3
    env.pop("MXG_GAP_POOL")
    env["MXG_GAP_WHOLE"] = "1"      # the long ones whole, to the host: rounds 3 and 4
    st = _check(oracle, recs, 32, 1000, cand_per_window=10)
    assert st["deferred_stretches"] > 3
    env.pop("MXG_GAP_WHOLE")
    env["MXG_GAP_DEV_CAP"] = "64"   # more stretches than the arrays hold: the batch is enqueued again, then redone
    _check(oracle, recs + [(f"x{i}", rnd(9000)) for i in range(40)], 32, 1000, cand_per_window=2)


def test_stretch_counts_around_the_rank_kernels_last_block(oracle, env):
    """k_gap_post ranks sixteen stretches per block: with per-stretch arrays (and placing blocks) for a number of stretches that is
    no multiple of sixteen, a batch whose stretch count falls into the last, partial block must still have every stretch ranked
    (the grid rounded down until round 6: those stretches' minimizers landed at stale offsets).  A sweep of record lengths
    walks the count through 50 ... 90 stretches against arrays for 72: counts beyond 72 are enqueued again / redone, all others
    are placed on the device."""
    env["MXG_DEV_GAPS"] = "1"
    env["MXG_GAP_DEV_CAP"] = "72"
    rng = random.Random(66)
    placed = 0
    for L in range(60000, 124000, 2000):
        rec = [("r%d" % L, "".join(rng.choice("ACGT") for _ in range(L)))]
        st = _check(oracle, rec, 32, 200, cand_per_window=3)
        assert st["select_slices"] > 0
        placed += 1 if st["batches_redone"] == 0 and st["retried_assemblies"] == 0 else 0
    assert placed >= 4, placed      # (the sweep did reach counts the device route keeps)


def test_many_batches_and_overflow_on_the_bs_route(oracle, env):
    env["MXG_SPARSE_BATCH_KMERS"] = "30000"
    st = _check(oracle, _records(21), 32, 200)
    assert st["bs_filter_bases"] > 0 or os.environ.get("MXG_BS") == "0"  # (MXG_BS=0: the rolling-hash route is under test)
    env["MXG_WAVE_CAP"] = "8"       # every slice outgrows its queue: the batch is redone with the capacity it asked for
    env["MXG_SPARSE_S"] = "128"
    st = _check(oracle, _records(22), 32, 200)
    assert st["bs_filter_bases"] > 0 or os.environ.get("MXG_BS") == "0"  # (MXG_BS=0: the rolling-hash route is under test)
    env["MXG_WAVE_CAP"] = "9000"    # queues beyond LDS: such a batch takes the rolling-hash kernel instead
    env["MXG_SPARSE_S"] = "256"
    _check(oracle, _records(23), 32, 200)


def test_inputs_the_filter_does_not_take(oracle, env):
    st = _check(oracle, _records(31), 31, 200)
    assert st["bs_filter_bases"] == 0
    st = _check(oracle, _records(32), 32, 300, cand_per_window=4, variant="v1")
    assert st["bs_filter_bases"] == 0


def test_run_borders_and_short_records(oracle, env):
    """N runs every few hundred bases, records shorter than a window, records of exactly k .. k + w bases: positions whose
    32-mer crosses a run border are set in the bitmap (the filter sees bases, not runs) and must be masked by the batch kernels"""
    rng = random.Random(77)
    recs = []
    for r in range(60):
        n = rng.choice([31, 32, 33, 100, 231, 232, 233, 700, 3000, 9000])
        s = [rng.choice("ACGT") for _ in range(n)]
        for _ in range(n // 400):
            p = rng.randrange(0, n)
            s[p:p + rng.choice([1, 2, 31, 32, 33, 90])] = "N" * min(rng.choice([1, 2, 31, 32, 33, 90]), n - p)
        recs.append((f"r{r}", "".join(s)[:n]))
    recs.append(("long", "".join(rng.choice("ACGT") for _ in range(150000))))
    st = _check(oracle, recs, 32, 200)
    assert st["bs_filter_bases"] > 0 or os.environ.get("MXG_BS") == "0"  # (MXG_BS=0: the rolling-hash route is under test)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_select_halo_where_runs_meet(oracle, env, seed):
    """records made of many valid runs between N gaps, run lengths just under / at / over whole strips and just under / over a
    window: the halo of a slice (strips on either side of its own ones) must still hold w k-mers of the contig where a run's
    short last strip falls into it (bs_select_halo), or the route must decline the assembly -- either way the sketch is the
    oracle's"""
    rng = random.Random(seed)
    S, w, k = 320, 1000, 32
    env["MXG_SPARSE_S"] = str(S)
    env["MXG_DEV_GAPS"] = "1"
    recs = []
    for r in range(6):
        parts = []
        for _ in range(rng.randrange(8, 40)):
            n_kmers = rng.choice([S - 1, S, S + 1, 2 * S - 1, 2 * S + 3, 3 * S, w - 2, w, w + 1, 5, 40, 4 * S + 7, rng.randrange(1, 6000)])
            parts.append("".join(rng.choice("ACGT") for _ in range(n_kmers + k - 1)))
            parts.append("N" * rng.choice([1, 1, 2, 31, 32, 33, 100]))
        recs.append((f"r{r}", "".join(parts)))
    recs.append(("plain", "".join(rng.choice("ACGT") for _ in range(120000))))
    st = _check(oracle, recs, k, w)
    st2 = _check(oracle, recs, k, w, cand_per_window=4)
    assert st["bs_filter_bases"] > 0 and st2["bs_filter_bases"] > 0


def test_knobs_are_read_once_per_handle_and_reported(oracle, env):
    """mxg_knobs: the MXG_* switches a handle read and found set, as it first saw them (a later change of the environment does
    not reach a handle that has already run)"""
    from ntjoin_amd.engine import MxEngine
    env["MXG_SPARSE_S"] = "128"
    env["MXG_BS_SELECT"] = "0"
    recs = _records(61)
    with MxEngine(k=32, w=200) as eng:
        eng.add_records("x", 1.0, recs)
        eng.sketch()
        first = eng.get_sketch(0)["out_hash"].copy()
        kn = dict(kv.split("=", 1) for kv in eng.knobs().split())
        assert kn.get("MXG_SPARSE_S") == "128" and kn.get("MXG_BS_SELECT") == "0"
        assert eng.stats()["select_slices"] == 0
        env["MXG_BS_SELECT"] = "1"     # too late for this handle
        eng.sketch()
        assert eng.stats()["select_slices"] == 0
        assert (eng.get_sketch(0)["out_hash"] == first).all()
    with MxEngine(k=32, w=200) as eng:
        eng.add_records("x", 1.0, recs)
        eng.sketch()
        assert eng.stats()["select_slices"] > 0 and "MXG_BS_SELECT=1" in eng.knobs()
