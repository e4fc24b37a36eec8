"""The pin the build image cannot give: python-igraph itself (tests/golden/real_igraph/README.md).  Skipped unless `import igraph`
finds the real package.  (1) every golden case's graph built in python-igraph and in tests/golden/igraph_standin.py, and everything
the reference's code asks of its container (reference bin/ntjoin_utils.py:37-47,83-141; bin/ntjoin.py:25-176) compared call by call;
(2) where /root/reference is there too: the reference's own functions re-run on the committed TSVs with python-igraph as the
container, diffed against the committed goldens (whose container was the stand-in)."""
import importlib
import json
import os
import shutil
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


def _real_igraph():
    if "igraph" in sys.modules and getattr(sys.modules["igraph"], "__file__", "").endswith("igraph_standin.py"):
        return None
    try:
        ig = importlib.import_module("igraph")
    except Exception:
        return None
    return ig if hasattr(ig, "__version__") and hasattr(ig, "Graph") and "standin" not in (getattr(ig, "__file__", "") or "") else None


REAL = _real_igraph()
pytestmark = pytest.mark.skipif(REAL is None, reason="python-igraph is not installed (tests/golden/real_igraph/README.md)")
CASES = [m["name"] for m in json.load(open(os.path.join(GOLDEN, "cases", "index.json")))]


def _standin():
    sys.path.insert(0, GOLDEN)
    import igraph_standin
    return igraph_standin


def _build(mod, edges):
    names = []
    seen = set()
    for u, v, _s, _w in edges:
        for x in (u, v):
            if x not in seen:
                seen.add(x)
                names.append(x)
    g = mod.Graph()
    g.add_vertices(names)
    g.add_edges([(u, v) for u, v, _s, _w in edges])
    g.es()["support"] = [s for _u, _v, s, _w in edges]
    g.es()["weight"] = [w for _u, _v, _s, w in edges]
    return g, names


def _tuples(g):
    return [(e.source, e.target) if hasattr(e, "source") else tuple(e.tuple) for e in g.es()]


@pytest.mark.parametrize("case", CASES)
def test_standin_agrees_with_python_igraph_on_the_golden_graphs(case):
    edges = json.load(open(os.path.join(GOLDEN, "cases", case, "reference.json")))["reference"]["edges"]
    if not edges:
        pytest.skip("a case without edges")
    sg, names = _build(_standin(), edges)
    rg, names_r = _build(REAL, edges)
    assert names == names_r and [v["name"] for v in rg.vs()] == [v["name"] for v in sg.vs()]
    assert _tuples(rg) == _tuples(sg)                                              # C1
    for i, (u, v, _s, w) in enumerate(edges[:500]):
        assert rg.get_eid(u, v) == sg.get_eid(u, v) == i and rg.get_eid(v, u) == sg.get_eid(v, u) == i   # C6 (names, either way)
    comps_r, comps_s = [list(c) for c in rg.components()], [list(c) for c in sg.components()]
    assert comps_r == comps_s                                                      # C4
    for v in range(min(len(names), 2000)):                                         # C7
        assert rg.vs()[v].degree() == sg.vs()[v].degree()
        assert list(rg.incident(v)) == list(sg.incident(v)) and list(rg.neighbors(v)) == list(sg.neighbors(v))
    for comp in comps_r[:200]:                                                     # C3 + C5
        sr, ss = rg.subgraph(comp), sg.subgraph(comp)
        assert [v["name"] for v in sr.vs()] == [v["name"] for v in ss.vs()] and _tuples(sr) == _tuples(ss)
        assert [e["weight"] for e in sr.es()] == [e["weight"] for e in ss.es()] and [list(e["support"]) for e in sr.es()] == [list(e["support"]) for e in ss.es()]
        leaves = [v.index for v in sr.vs() if v.degree() == 1]
        if len(leaves) == 2 and all(v.degree() <= 2 for v in sr.vs()):
            assert [list(p) for p in sr.get_shortest_paths(leaves[0], to=leaves[1])] == [list(p) for p in ss.get_shortest_paths(leaves[0], to=leaves[1])]
    for n in (1, 2, 3):                                                            # C2: what filter_graph_global does (bin/ntjoin.py:80-89)
        r2, s2 = rg.copy(), sg.copy()
        r2.delete_edges([e.index for e in r2.es() if e["weight"] < n])
        s2.delete_edges([e.index for e in s2.es() if e["weight"] < n])
        assert _tuples(r2) == _tuples(s2) and [e["weight"] for e in r2.es()] == [e["weight"] for e in s2.es()]
        assert [list(c) for c in r2.components()] == [list(c) for c in s2.components()]


@pytest.mark.skipif(not os.path.isdir("/root/reference/bin"), reason="needs the reference tree (build container)")
@pytest.mark.parametrize("case", CASES)
def test_reference_code_over_python_igraph_reproduces_the_goldens(case, tmp_path):
    sys.path.insert(0, GOLDEN)
    for m in ("igraph", "ntjoin_utils", "ntjoin"):     # (a stand-in installed by an earlier import must not be what the reference sees)
        if m in sys.modules and (m != "igraph" or sys.modules[m] is not REAL):
            del sys.modules[m]
    sys.modules["igraph"] = REAL
    import make_golden
    gold = json.load(open(os.path.join(GOLDEN, "cases", case, "reference.json")))
    meta, ref = gold["meta"], gold["reference"]
    d = tmp_path / case
    shutil.copytree(os.path.join(GOLDEN, "cases", case), d)
    tsvs = [a["tsv"] for a in meta["refs"]] + [meta["target"]["tsv"]]
    got = make_golden.run_reference(str(d), tsvs[:-1], [a["weight"] for a in meta["refs"]], tsvs[-1], meta["target"]["weight"], "out", k=meta["k"],
                                    target_fasta=os.path.join(GOLDEN, "fasta", meta["target"]["fasta"]))
    got = json.loads(json.dumps(got))
    key = lambda e: (min(e[0], e[1]), max(e[0], e[1]))
    assert sorted(map(key, got["edges"])) == sorted(map(key, ref["edges"]))
    assert {key(e): (sorted(e[2]), e[3]) for e in got["edges"]} == {key(e): (sorted(e[2]), e[3]) for e in ref["edges"]}
    # (the reference numbers its vertices in the order of a Python set of strings, which changes from process to process: paths and
    # path descriptions are compared as sets, the way tests/test_oracle_paths.py compares the oracle with the goldens)
    from oracle import paths_oracle as po
    nodes_key = lambda nodes: tuple(tuple(x) for x in nodes)
    for n in ref["paths_by_n"]:
        assert po.canonical(got["paths_by_n"][n]) == po.canonical(ref["paths_by_n"][n]), ("paths_by_n", n)
        assert sorted(map(nodes_key, got["format_by_n"][n])) == sorted(map(nodes_key, ref["format_by_n"][n])), ("format_by_n", n)
    assert got["mx_extremes_by_n"] == ref["mx_extremes_by_n"]
