"""GPU parity for the first "next" row: mxg_find_paths (global edge filter, branch filtering, cycle
opening, source/target choice, linear paths) against (1) the reference's own find_paths() output committed in the goldens
(reference.json["paths_by_n"]) and (2) oracle/paths_oracle.py on random graphs with branches, cycles and fractional weights."""
import os
import random

import pytest

from oracle import graph_oracle as go
from oracle import paths_oracle as po
from tests import _path_literals
from tests.conftest import GOLDEN, golden_cases, load_case

pytestmark = pytest.mark.gpu

CASES = [m["name"] for m in golden_cases()]


def _canonical_gpu(eng, n):
    names = [str(h) for h in eng.get_graph()["vertex_hash"].tolist()]
    by_comp = {}
    for comp, verts in eng.find_paths(n):
        by_comp.setdefault(comp, []).append([names[v] for v in verts])
    return po.canonical(by_comp.values())


@pytest.mark.parametrize("name", CASES)
def test_paths_match_reference_goldens(name):
    from ntjoin_amd.engine import MxEngine
    case = load_case(name)
    meta, ref = case["meta"], case["reference"]
    cdir = os.path.join(GOLDEN, "cases", name)
    with MxEngine(k=meta["k"], w=meta["w"], variant=meta["variant"]) as eng:
        for a in meta["refs"] + [meta["target"]]:
            eng.add_tsv(a["tsv"], a["weight"], os.path.join(cdir, a["tsv"]))
        eng.build_graph()
        for n, ref_paths in ref["paths_by_n"].items():
            assert _canonical_gpu(eng, int(n)) == po.canonical(ref_paths), (name, n)
            assert eng.n_components == len(ref_paths), (name, n)   # the reference returns one entry per component


def test_paths_orientation_and_order():
    """paths run source -> target with the source at the smaller position of the highest-weight assembly
    (bin/ntjoin.py:95-103) and come ordered by source vertex index"""
    from ntjoin_amd.engine import MxEngine
    case = load_case("f-f_w1000")
    meta = case["meta"]
    cdir = os.path.join(GOLDEN, "cases", "f-f_w1000")
    with MxEngine(k=meta["k"], w=meta["w"], variant=meta["variant"]) as eng:
        asms = meta["refs"] + [meta["target"]]
        for a in asms:
            eng.add_tsv(a["tsv"], a["weight"], os.path.join(cdir, a["tsv"]))
        eng.build_graph()
        g = eng.get_graph()
        found = eng.find_paths(2)
        assert len(found) == 1 and len(found[0][1]) == 5
        best = max(range(len(asms)), key=lambda i: asms[i]["weight"])
        pos = [int(g["vertex_pos"][best][v]) for v in found[0][1]]
        assert pos == sorted(pos)
        srcs = [p[1][0] for p in eng.find_paths(1)]
        assert srcs == sorted(srcs)


def test_find_paths_requires_graph():
    from ntjoin_amd.engine import MxEngine, MxError
    with MxEngine(k=32, w=10) as eng:
        eng.add_minimizers("a", 1.0, [1, 2], [0, 5], [0, 0], ["c"])
        with pytest.raises(MxError):
            eng.find_paths(1)


def _write_random_assemblies(rng, t, tie_free=True):
    A = rng.randint(1, 5)
    universe = [rng.getrandbits(64) for _ in range(rng.choice([6, 12, 40, 300, 3000]))]
    names, weights = [], []
    for a in range(A):
        name = f"p{t}_asm{a}.k32.w10.tsv"
        recs = []
        for r in range(rng.randint(1, 10)):
            n = rng.choice([0, 1, 2, 3, 8, 40, 400])
            mode = rng.random()
            if mode < 0.4:      # a run of the universe in order (collinear with the other assemblies)
                s = rng.randrange(len(universe))
                picks = universe[s:s + n]
                if rng.random() < 0.3:
                    picks = picks[::-1]
            elif mode < 0.7:    # collinear with random drop-outs and a few strays
                s = rng.randrange(len(universe))
                picks = [h for h in universe[s:s + 2 * n] if rng.random() < 0.6]
                if picks and rng.random() < 0.5:
                    picks.insert(rng.randrange(len(picks) + 1), rng.choice(universe))
            else:
                picks = rng.sample(universe, min(n, len(universe)))
            recs.append(picks)
        total = sum(len(p) for p in recs)
        pool = sorted(rng.sample(range(10 ** 6), total))    # unique positions per assembly: the reference leaves ties
        rng.shuffle(recs)                                   # between equal positions to python's set order
        with open(name, "w", encoding="ascii") as fh:
            at = 0
            for r, picks in enumerate(recs):
                pos = pool[at:at + len(picks)]
                at += len(picks)
                fh.write(f"ctg{r}\t" + " ".join(f"{h}:{p}:ACGT" for h, p in zip(picks, pos)) + "\n")
        names.append(name)
        weights.append(rng.choice([1, 1, 2, 2, 0.5, 1.5, 3]))
    return names, weights


def test_fuzz_paths_vs_oracle(tmp_path):
    from ntjoin_amd.engine import MxEngine
    trials = int(os.environ.get("MXG_FUZZ_TRIALS", "60"))
    rng = random.Random(int(os.environ.get("MXG_FUZZ_SEED", "4242")))
    os.chdir(tmp_path)
    n_paths = 0
    for t in range(trials):
        names, weights = _write_random_assemblies(rng, t)
        state = go.load_and_build(names[:-1], weights[:-1], names[-1], weights[-1])
        with MxEngine(k=32, w=10) as eng:
            for nm, wt in zip(names, weights):
                eng.add_tsv(nm, wt, nm)
            eng.build_graph()
            for n in sorted({1, 2, 3, int(sum(weights)), int(sum(weights)) + 1}):
                want = po.canonical(po.find_paths(state, n))
                assert _canonical_gpu(eng, n) == want, (t, n, names, weights)
                n_paths += sum(len(c) for c in want)
    assert n_paths > trials // 2  # the generator does produce chains (about one per trial)


def test_cycle_is_opened(tmp_path):
    """a circular component (all degrees 2) is opened at its smallest-position vertex (bin/ntjoin.py:113-135)"""
    from ntjoin_amd.engine import MxEngine
    os.chdir(tmp_path)
    ring = [101, 202, 303, 404, 505, 606]
    with open("ref.tsv", "w", encoding="ascii") as fh:     # weight 2: a ring split over two records
        fh.write("r0\t" + " ".join(f"{h}:{10 * (i + 1)}:A" for i, h in enumerate(ring)) + "\n")
        fh.write("r1\t" + f"{ring[-1] + 1}:5:A\n")
    with open("tgt.tsv", "w", encoding="ascii") as fh:     # closes the ring 606 - 101
        fh.write("t0\t" + f"{ring[-1]}:7:A {ring[0]}:9:A\n")
        fh.write("t1\t" + " ".join(f"{h}:{100 + i}:A" for i, h in enumerate(ring[1:-1])) + "\n")
    state = go.load_and_build(["ref.tsv"], [2], "tgt.tsv", 1)
    with MxEngine(k=32, w=10) as eng:
        eng.add_tsv("ref.tsv", 2, "ref.tsv")
        eng.add_tsv("tgt.tsv", 1, "tgt.tsv")
        eng.build_graph()
        got = _canonical_gpu(eng, 1)
    assert got == po.canonical(po.find_paths(state, 1))
    assert got == {frozenset({tuple(str(h) for h in ring)})}


def _state_from_graph(eng, weights):
    """oracle state from the graph arrays of the library (their parity with the reference is test_gpu_parity's job)"""
    g = eng.get_graph()
    names = [str(h) for h in g["vertex_hash"].tolist()]
    asms = [f"asm{a}" for a in range(len(weights))]
    info = {nm: {v: (int(r), int(p)) for v, r, p in zip(names, g["vertex_record"][a].tolist(), g["vertex_pos"][a].tolist())}
            for a, nm in enumerate(asms)}
    edges = [(names[u], names[v], None, w) for u, v, w in
             zip(g["edge_u"].tolist(), g["edge_v"].tolist(), g["edge_weight"].tolist())]
    return {"list_mx_info": info, "weights": dict(zip(asms, map(float, weights))), "vertices": names, "edges": edges}, g


@pytest.mark.parametrize("weights", [(2.0, 1.0), (1.0, 1.0)])
def test_paths_on_synthetic_genome(weights):
    """configs[1]-shaped input (reference + derived target with rearranged, reverse-complemented, mutated contigs):
    the paths equal the oracle's, cover only existing edges, and no vertex is used twice"""
    import numpy as np
    import torch
    from ntjoin_amd import synth
    from ntjoin_amd.engine import MxEngine
    ref, tgt = synth.config2(seed=5, n_bases=30_000_000)
    with MxEngine(k=32, w=500) as eng:
        for nm, wt, recs in (("ref", weights[0], ref), ("tgt", weights[1], tgt)):
            words, starts, lens = synth.pack_records(recs)
            d = torch.from_numpy(words.view(np.int32)).cuda()
            eng.add_packed_device(nm, wt, d.data_ptr(), starts, lens, keepalive=d)
        eng.sketch()
        eng.build_graph()
        state, g = _state_from_graph(eng, weights)
        edge_set = {frozenset(e) for e in zip(g["edge_u"].tolist(), g["edge_v"].tolist())}
        for n in (1, 2, 3):
            found = eng.find_paths(n)
            used = [v for _c, p in found for v in p]
            assert len(used) == len(set(used))
            assert all(frozenset(e) in edge_set for _c, p in found for e in zip(p, p[1:]))
            assert _canonical_gpu(eng, n) == po.canonical(po.find_paths(state, n)), n
        assert len(eng.find_paths(1)) >= 1


@pytest.mark.parametrize("length,n_chains", [(5000, 2), (70000, 1), (200, 400), (3, 20000)])
def test_long_and_many_chains(length, n_chains):
    """sizes at which the concurrent component labelling and the pointer jumping run many blocks deep: every record is
    one chain shared by both assemblies, so the paths are exactly the records, in position order"""
    import numpy as np
    from ntjoin_amd.engine import MxEngine
    rng = np.random.default_rng(length + n_chains)
    hs = np.unique(rng.integers(1, 2 ** 63, size=length * n_chains + 1000, dtype=np.uint64))[:length * n_chains]
    rng.shuffle(hs)
    rec = np.repeat(np.arange(n_chains, dtype=np.uint32), length)
    pos = np.tile(np.arange(length, dtype=np.uint32) * 7, n_chains)
    ids = [f"c{i}" for i in range(n_chains)]
    with MxEngine(k=32, w=10) as eng:
        eng.add_minimizers("a", 2.0, hs, pos, rec, ids)
        eng.add_minimizers("b", 1.0, hs, pos, rec, ids)
        eng.build_graph()
        vh = eng.get_graph()["vertex_hash"]
        for _rep in range(3):  # the labelling is racy by construction; its result must not be
            found = eng.find_paths(1)
            assert len(found) == n_chains
            got = sorted((vh[np.array(p)].tolist() for _c, p in found), key=lambda p: p[0])
            want = sorted((hs[i * length:(i + 1) * length].tolist() for i in range(n_chains)), key=lambda p: p[0])
            assert got == want


def test_ntjoin_class_find_paths(tmp_path, capsys):
    """the Ntjoin counterpart class: load -> graph -> find_paths returns one list per component of (path, _) tuples with
    vertex NAMES, as bin/ntjoin.py:137-176 does, and prints the component count"""
    import argparse
    from ntjoin_amd.ntjoin import Ntjoin
    name = "f-f_w1000"
    meta, ref = load_case(name)["meta"], load_case(name)["reference"]
    os.chdir(os.path.join(GOLDEN, "cases", name))
    args = argparse.Namespace(FILES=[r["tsv"] for r in meta["refs"]], s=meta["target"]["tsv"], l=meta["target"]["weight"],
                              p=str(tmp_path / "out"), k=meta["k"], n=2, t=1)
    nj = Ntjoin(args)
    try:
        nj.weights_list = [r["weight"] for r in meta["refs"]]
        nj.load_minimizers_scaffold()
        nj.make_minimizer_graph()
        paths = nj.find_paths()
    finally:
        nj.close()
    assert po.canonical([[p for p, _g in comp] for comp in paths]) == po.canonical(ref["paths_by_n"]["2"])
    assert len(paths) == len(ref["paths_by_n"]["2"])      # one entry per component, empty ones included
    assert f"Total number of components in graph: {len(paths)}" in capsys.readouterr().out


def _fasta_lengths(path):
    lens, rid = {}, None
    for line in open(path, encoding="ascii"):
        if line.startswith(">"):
            rid = line[1:].split()[0]
            lens[rid] = 0
        elif rid is not None:
            lens[rid] += len(line.strip())
    return lens


@pytest.mark.parametrize("name", CASES)
def test_format_paths_match_reference_goldens(name):
    """row f4: Ntjoin.find_mx_min_max / format_paths (segments and extremes from the GPU) against the reference's own
    find_mx_min_max and format_path output for the target assembly, every golden case, every -n"""
    import argparse
    import contextlib
    import io
    from ntjoin_amd.ntjoin import Ntjoin
    case = load_case(name)
    meta, ref = case["meta"], case["reference"]
    fa = ref["format_args"]
    lengths = _fasta_lengths(os.path.join(GOLDEN, "fasta", meta["target"]["fasta"]))
    cwd = os.getcwd()
    os.chdir(os.path.join(GOLDEN, "cases", name))
    try:
        for n, want in ref["format_by_n"].items():
            args = argparse.Namespace(FILES=[r["tsv"] for r in meta["refs"]], s=meta["target"]["tsv"],
                                      l=meta["target"]["weight"], p="/tmp/mxg_fmt_" + name, k=meta["k"], n=int(n), t=1)
            nj = Ntjoin(args, variant=meta["variant"])
            try:
                nj.weights_list = [r["weight"] for r in meta["refs"]]
                with contextlib.redirect_stdout(io.StringIO()):
                    nj.load_minimizers_scaffold()
                    nj.make_minimizer_graph(materialize=False)
                    nj.find_paths()
                assert {c: list(v) for c, v in nj.find_mx_min_max(args.s).items()} == ref["mx_extremes_by_n"][n], (name, n)
                got = nj.format_paths(lengths, g=fa["g"], G=fa["G"], m=fa["m"])
            finally:
                nj.close()
            key = lambda nodes: tuple(tuple(x) for x in nodes)
            assert sorted(map(key, got)) == sorted(map(key, want)), (name, n)
    finally:
        os.chdir(cwd)


@pytest.mark.parametrize("name", sorted(_path_literals.EXPECTED))
def test_path_strings_the_reference_tests_assert(name):
    """reference tests/ntjoin_test.py:85,97,104,111,120,133,148,157,165: the path strings (contig, orientation, cut coordinates,
    gap sizes) built from Ntjoin.find_paths + format_paths on the GPU, against the literals typed in from the reference's test file"""
    import argparse
    import contextlib
    import io
    from ntjoin_amd.ntjoin import Ntjoin
    n, expected = _path_literals.EXPECTED[name]
    case = load_case(name)
    meta, fa = case["meta"], case["reference"]["format_args"]
    lengths = _fasta_lengths(os.path.join(GOLDEN, "fasta", meta["target"]["fasta"]))
    cwd = os.getcwd()
    os.chdir(os.path.join(GOLDEN, "cases", name))
    try:
        args = argparse.Namespace(FILES=[r["tsv"] for r in meta["refs"]], s=meta["target"]["tsv"], l=meta["target"]["weight"],
                                  p="/tmp/mxg_lit_" + name, k=meta["k"], n=n, t=1)
        nj = Ntjoin(args, variant=meta["variant"])
        try:
            nj.weights_list = [r["weight"] for r in meta["refs"]]
            with contextlib.redirect_stdout(io.StringIO()):
                nj.load_minimizers_scaffold()
                nj.make_minimizer_graph(materialize=False)
                nj.find_paths()
            got = {_path_literals.path_string(nodes) for nodes in nj.format_paths(lengths, g=fa["g"], G=fa["G"], m=fa["m"])}
        finally:
            nj.close()
        assert got == expected
    finally:
        os.chdir(cwd)


def test_fuzz_segments_vs_oracle(tmp_path):
    """random multi-assembly graphs: per-path runs (contig, n, min, max, inc, dec) of every assembly against a plain
    walk over the oracle's paths"""
    from ntjoin_amd.engine import MxEngine
    trials = int(os.environ.get("MXG_FUZZ_TRIALS", "40"))
    rng = random.Random(99)
    os.chdir(tmp_path)
    for t in range(trials):
        names, weights = _write_random_assemblies(rng, 1000 + t)
        state = go.load_and_build(names[:-1], weights[:-1], names[-1], weights[-1])
        with MxEngine(k=32, w=10) as eng:
            for nm, wt in zip(names, weights):
                eng.add_tsv(nm, wt, nm)
            eng.build_graph()
            vnames = [str(h) for h in eng.get_graph()["vertex_hash"].tolist()]
            found = eng.find_paths(1)
            for a, nm in enumerate(names):
                seg = eng.path_segments(a)
                ids = eng.record_ids(a, eng.n_records(a))
                info = state["list_mx_info"][nm]
                want = []
                for p, (_c, verts) in enumerate(found):
                    runs = []
                    for v in verts:
                        ctg, pos = info[vnames[v]]
                        if runs and runs[-1][0] == ctg:
                            runs[-1][1].append(pos)
                        else:
                            runs.append((ctg, [pos]))
                    for ctg, ps in runs:
                        pairs = list(zip(ps, ps[1:]))
                        want.append((p, ctg, len(ps), min(ps), max(ps), sum(x < y for x, y in pairs), sum(x > y for x, y in pairs)))
                got = [(p, ids[r], n, mn, mx, i, d) for p, r, n, mn, mx, i, d in
                       zip(*[seg[c].tolist() for c in ("path", "record", "n", "min_pos", "max_pos", "inc", "dec")])]
                assert got == want, (t, a)
                ext = eng.mx_extremes(a)
                exp = po.mx_extremes(state, nm)
                assert {ids[r]: e for r, e in enumerate(ext) if e is not None} == exp, (t, a)
