"""GPU test of the two joins behind mxg_build_graph: per-partition LDS tables (default for the whole-stage call) and the
global table (MXG_GRAPH_JOIN=global; also the fallback when a partition's table overflows, forced here with
MXG_PJ_FORCE_FAIL=1).  Flags and graph must be identical; the golden parity tests pin the default against the reference."""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN, golden_cases, load_case

pytestmark = pytest.mark.gpu
FASTA = os.path.join(GOLDEN, "fasta")
CASES = [m["name"] for m in golden_cases()]


def _graph_state(eng, n_asm):
    out = {f"flags{a}": eng.get_mx_flags(a).copy() for a in range(n_asm)}
    for key, val in eng.get_graph().items():
        out[key] = np.asarray(val).copy()
    return out


def _three_ways(monkeypatch, build):
    res = []
    # default (LDS tables per hash partition), the global table, the fallback after a failed LDS join, and the two-level LDS
    # join that genome-scale inputs (> 5 M minimizers) take, forced here
    # (MXG_PJ_SKEW=1: every block of the second level collapses equal records, as the blocks of a skewed partition do)
    for env in ({}, {"MXG_GRAPH_JOIN": "global"}, {"MXG_PJ_FORCE_FAIL": "1"}, {"MXG_PJ_TWO_LEVEL": "1"},
                {"MXG_PJ_TWO_LEVEL": "1", "MXG_PJ_FORCE_FAIL": "1"}, {"MXG_PJ_TWO_LEVEL": "1", "MXG_PJ_SKEW": "1"}):
        for key in ("MXG_GRAPH_JOIN", "MXG_PJ_FORCE_FAIL", "MXG_PJ_TWO_LEVEL", "MXG_PJ_SKEW"):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        res.append(build())
    for other in res[1:]:
        assert other.keys() == res[0].keys()
        for key in res[0]:
            assert np.array_equal(res[0][key], other[key]), key
    return res[0]


@pytest.mark.parametrize("name", CASES)
def test_joins_agree_on_goldens(name, monkeypatch):
    from ntjoin_amd.engine import MxEngine
    meta = load_case(name)["meta"]
    asms = meta["refs"] + [meta["target"]]

    def build():
        with MxEngine(k=meta["k"], w=meta["w"], variant=meta["variant"]) as eng:
            for a in asms:
                eng.add_fasta(a["tsv"], a["weight"], os.path.join(FASTA, a["fasta"]))
            eng.sketch()
            eng.build_graph()
            eng.build_graph()  # twice on one handle: the partition counters must come back clean
            return _graph_state(eng, len(asms))
    _three_ways(monkeypatch, build)


def test_joins_agree_on_repeats_and_three_assemblies(monkeypatch):
    """minimizer lists with heavy duplication inside an assembly (one key 5000 times: a single partition takes them all),
    keys missing from one assembly, the key 2^64-1, three assemblies"""
    from ntjoin_amd.engine import MxEngine
    rng = np.random.default_rng(11)
    base = rng.integers(0, 2**63, size=60000, dtype=np.int64).astype(np.uint64)
    base[0] = np.uint64(0xFFFFFFFFFFFFFFFF)

    def mk(seed, drop, extra_dups):
        r = np.random.default_rng(seed)
        keep = r.random(base.size) >= drop
        hs = base[keep].copy()
        r.shuffle(hs)
        hs = np.concatenate([hs, np.repeat(hs[:3], extra_dups), r.integers(0, 2**63, size=3000, dtype=np.int64).astype(np.uint64)])
        n_rec = 40
        rec = np.sort(r.integers(0, n_rec, size=hs.size)).astype(np.uint32)
        pos = np.zeros(hs.size, np.uint32)
        for c in range(n_rec):
            m = rec == c
            pos[m] = np.arange(int(m.sum()), dtype=np.uint32) * 37
        return hs, pos, rec, [f"c{i}" for i in range(n_rec)]

    sets = [mk(1, 0.02, 5000), mk(2, 0.05, 2), mk(3, 0.0, 700)]

    def build():
        with MxEngine(k=32, w=1000) as eng:
            for i, (hs, pos, rec, ids) in enumerate(sets):
                eng.add_minimizers(f"a{i}", float(i + 1), hs, pos, rec, ids)
            eng.build_graph()
            st = eng.stats()
            assert st["vertices"] > 40000
            return _graph_state(eng, len(sets))
    _three_ways(monkeypatch, build)


def test_joins_agree_on_synthetic_genome(monkeypatch):
    import torch
    from ntjoin_amd import synth
    from ntjoin_amd.engine import MxEngine
    ref, tgt = synth.config2(seed=5, n_bases=30_000_000)

    def build():
        with MxEngine(k=32, w=1000) as eng:
            for name, wt, recs in (("ref", 2.0, ref), ("tgt", 1.0, tgt)):
                words, starts, lens = synth.pack_records(recs)
                d = torch.from_numpy(words.view(np.int32)).cuda()
                eng.add_packed_device(name, wt, d.data_ptr(), starts, lens, keepalive=d)
            eng.sketch_graph()     # device-side counts
            st1 = _graph_state(eng, 2)
            eng.sketch()
            eng.build_graph()
            st2 = _graph_state(eng, 2)
            for key in st1:
                assert np.array_equal(st1[key], st2[key]), key
            return st2
    g = _three_ways(monkeypatch, build)
    assert len(g["flags0"]) > 50000


def test_joins_agree_beyond_256_regions(monkeypatch):
    """1.5 million minimizers: more than 256 bucketing regions, so k_pj_join lays its segments out in several rounds"""
    from ntjoin_amd.engine import MxEngine
    rng = np.random.default_rng(5)
    base = rng.integers(0, 2**63, size=760_000, dtype=np.int64).astype(np.uint64)

    def mk(seed, drop):
        r = np.random.default_rng(seed)
        hs = base[r.random(base.size) >= drop].copy()
        r.shuffle(hs)
        hs = np.concatenate([hs, np.repeat(hs[:2], 1500)])
        rec = np.sort(r.integers(0, 64, size=hs.size)).astype(np.uint32)
        pos = np.arange(hs.size, dtype=np.uint32)
        return hs, pos, rec, [f"c{i}" for i in range(64)]

    sets = [mk(1, 0.01), mk(2, 0.03)]

    def build():
        with MxEngine(k=32, w=1000) as eng:
            for i, (hs, pos, rec, ids) in enumerate(sets):
                eng.add_minimizers(f"a{i}", float(i + 1), hs, pos, rec, ids)
            eng.build_graph()
            assert eng.stats()["vertices"] > 700_000
            return _graph_state(eng, len(sets))
    _three_ways(monkeypatch, build)


@pytest.mark.parametrize("adjacent", [True, False])
def test_two_level_join_one_key_400000_times(monkeypatch, adjacent):
    """one key 400 000 times among 900 000 minimizers.  In one run (a tandem array): the run travels as one record and nothing
    overflows.  Scattered among the others: the key's coarse partition outgrows the capacity reserved for it (25 % above the
    mean); the stage notices, sizes the partitions by what the cursors counted and runs the join again.  Either way the result
    is the global table's."""
    from ntjoin_amd.engine import MxEngine
    rng = np.random.default_rng(9)
    base = rng.integers(0, 2**63, size=500_000, dtype=np.int64).astype(np.uint64)

    def mk(seed, heavy):
        r = np.random.default_rng(seed)
        hs = base.copy()
        r.shuffle(hs)
        if adjacent:
            hs = np.concatenate([hs, np.full(heavy, base[7], dtype=np.uint64)])
        elif heavy:
            out = np.empty(hs.size + heavy, dtype=np.uint64)  # every other minimizer of the first 800 000 is the heavy key
            out[:2 * heavy:2] = base[7]
            out[1:2 * heavy:2] = hs[:heavy]
            out[2 * heavy:] = hs[heavy:]
            hs = out
        rec = np.sort(r.integers(0, 16, size=hs.size)).astype(np.uint32)
        return hs, np.arange(hs.size, dtype=np.uint32), rec, [f"c{i}" for i in range(16)]

    sets = [mk(1, 400_000), mk(2, 0)]
    res = []
    for env in ({"MXG_PJ_TWO_LEVEL": "1"}, {"MXG_GRAPH_JOIN": "global"}):
        for key in ("MXG_GRAPH_JOIN", "MXG_PJ_TWO_LEVEL"):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        with MxEngine(k=32, w=1000) as eng:
            for i, (hs, pos, rec, ids) in enumerate(sets):
                eng.add_minimizers(f"a{i}", float(i + 1), hs, pos, rec, ids)
            eng.build_graph()
            st = eng.stats()
            assert st["vertices"] == 499_999
            if "MXG_PJ_TWO_LEVEL" in env:
                assert st["graph_join"] == (2 if adjacent else 2 | 0x100), hex(st["graph_join"])
                eng.build_graph()  # the handle remembers the capacity: no second attempt
                assert eng.stats()["graph_join"] == 2
            else:
                assert st["graph_join"] == 3
            res.append(_graph_state(eng, 2))
    for key in res[0]:
        assert np.array_equal(res[0][key], res[1][key]), key


def _flag_truth(sets):
    """UNIQUE / SHARED / INALL per minimizer from numpy (ntjoin_utils.py:152-193: seen once here; once everywhere; everywhere)"""
    cnt = []
    for hs, *_ in sets:
        u, c = np.unique(hs, return_counts=True)
        cnt.append(dict(zip(u.tolist(), c.tolist())))
    out = []
    for a, (hs, *_) in enumerate(sets):
        fl = np.zeros(hs.size, np.uint8)
        for i, hv in enumerate(hs.tolist()):
            cs = [c.get(hv, 0) for c in cnt]
            inall = all(cs)
            fl[i] = (1 if cs[a] == 1 else 0) | (2 if inall and max(cs) == 1 else 0) | (4 if inall else 0)
        out.append(fl)
    return out


def test_joins_agree_on_runs_of_equal_hashes(monkeypatch):
    """runs of equal hashes in sketch order (what a tandem array leaves): lengths 2 ... 5000, across waves, 256-blocks and
    bucketing regions, at an assembly's first and last minimizer, one run right behind another, of keys that the other
    assemblies hold once / not at all / also as a run, and of the key 2^64 - 1; flags against numpy as well"""
    from ntjoin_amd.engine import MxEngine
    rng = np.random.default_rng(21)
    base = rng.integers(0, 2**63, size=30000, dtype=np.int64).astype(np.uint64)
    base[5] = np.uint64(0xFFFFFFFFFFFFFFFF)

    def mk(seed, runs, first_run=0, last_run=0):
        r = np.random.default_rng(seed)
        hs = base[r.random(base.size) >= 0.03].copy()
        r.shuffle(hs)
        parts, at = [], 0
        if first_run:
            parts.append(np.full(first_run, base[1], dtype=np.uint64))
        for j, (key, ln) in enumerate(runs):
            step = int(r.integers(1, 900))
            parts.append(hs[at:at + step])
            at += step
            parts.append(np.full(ln, base[key], dtype=np.uint64))
            if j % 3 == 0:  # a second run right behind the first
                parts.append(np.full(1 + ln // 2, base[key + 1], dtype=np.uint64))
        parts.append(hs[at:])
        if last_run:
            parts.append(np.full(last_run, base[2], dtype=np.uint64))
        hs = np.concatenate(parts)
        rec = np.sort(r.integers(0, 9, size=hs.size)).astype(np.uint32)
        return hs, np.arange(hs.size, dtype=np.uint32), rec, [f"c{i}" for i in range(9)]

    lens = [2, 3, 63, 64, 65, 127, 129, 255, 256, 257, 700, 5000, 2, 2, 70, 70]
    sets = [mk(1, [(10 + 2 * j, ln) for j, ln in enumerate(lens)], first_run=300, last_run=2),
            mk(2, [(10 + 2 * j, 1 + ln % 7) for j, ln in enumerate(lens)] + [(5, 80)], last_run=400),
            mk(3, [(5, 3), (200, 4100), (12, 1)], first_run=2)]
    truth = _flag_truth(sets)

    def build():
        with MxEngine(k=32, w=1000) as eng:
            for i, (hs, pos, rec, ids) in enumerate(sets):
                eng.add_minimizers(f"a{i}", float(i + 1), hs, pos, rec, ids)
            eng.build_graph()
            for a in range(len(sets)):
                assert np.array_equal(eng.get_mx_flags(a), truth[a]), a
            return _graph_state(eng, len(sets))
    _three_ways(monkeypatch, build)


@pytest.mark.parametrize("n_asm", [16, 17, 32])
def test_joins_agree_with_many_assemblies(monkeypatch, n_asm):
    """16 assemblies: the LDS join keeps the seen and the duplicate mask of a key in one word; 17 and 32 (the maximum):
    two words.  Keys missing from some assemblies, duplicated in others."""
    from ntjoin_amd.engine import MxEngine
    rng = np.random.default_rng(5)
    base = rng.integers(0, 2**63, size=6000, dtype=np.int64).astype(np.uint64)

    def mk(a):
        r = np.random.default_rng(100 + a)
        hs = base[r.random(base.size) >= (0.01 if a % 3 else 0.0)].copy()
        r.shuffle(hs)
        if a % 4 == 1:
            hs = np.concatenate([hs, hs[:40]])  # 40 keys twice in this assembly
        rec = np.sort(r.integers(0, 12, size=hs.size)).astype(np.uint32)
        return hs, np.arange(hs.size, dtype=np.uint32), rec, [f"c{i}" for i in range(12)]
    sets = [mk(a) for a in range(n_asm)]

    def build():
        with MxEngine(k=32, w=1000) as eng:
            for i, (hs, pos, rec, ids) in enumerate(sets):
                eng.add_minimizers(f"a{i}", 1.0 + i / 8, hs, pos, rec, ids)
            eng.build_graph()
            assert eng.stats()["vertices"] > 3000
            return _graph_state(eng, len(sets))
    _three_ways(monkeypatch, build)
