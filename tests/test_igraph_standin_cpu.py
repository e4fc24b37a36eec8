"""Property tests of tests/golden/igraph_standin.py, the container the golden generator (tests/golden/make_golden.py) gives the
reference's Python in place of python-igraph (not installed in the build image).  Every golden `reference.json` /
`reference.mx.dot` went through it, so the contracts of python-igraph that the reference's code relies on are stated here and
checked on random graphs against straightforward restatements:

  C1  Graph.add_edges / Edge.tuple: an undirected edge is reported as (smaller vertex id, larger vertex id); ids are assigned in
      insertion order (python-igraph API reference, Graph.add_edges, EdgeSeq).
  C2  Graph.delete_edges: the remaining edges keep their relative order and are renumbered 0..m-1 ("edge IDs are always
      continuous", igraph reference manual, igraph_delete_edges) -- attributes travel with their edge.
  C3  Graph.subgraph / induced_subgraph: vertices are renumbered in ascending order of their original ids, edges keep their
      relative order, attributes are copied (igraph_induced_subgraph, "the vertex IDs ... are mapped in increasing order").
  C4  Graph.components(): a membership clustering whose clusters are numbered by first appearance, i.e. ordered by their lowest
      vertex id; each cluster lists its vertices ascending (VertexClustering iteration).
  C5  Graph.get_shortest_paths(v, to=t) on a tree / chain: the unique path, as vertex ids from v to t; [[]] when unreachable
      (igraph_get_shortest_paths: an empty vector for unreachable targets).
  C6  Graph.get_eid(a, b): the id of the edge between a and b in either order (directed=True is irrelevant on undirected
      graphs); vertex NAMES are accepted wherever ids are (python-igraph resolves string names through the `name` attribute).
  C7  Vertex.degree(), Graph.incident(v), Graph.neighbors(v): consistent with the edge list at every moment, also after
      delete_edges.
  C8  es["attr"] = list assigns element-wise in edge-id order; copy() is deep with respect to structure and attributes.
"""
import os
import random
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
import igraph_standin as ig  # noqa: E402


def _random_graph(rng, n, m):
    g = ig.Graph()
    names = [f"v{i}" for i in range(n)]
    g.add_vertices(names)
    pairs = set()
    while len(pairs) < m:
        a, b = rng.randrange(n), rng.randrange(n)
        if a != b:
            pairs.add((min(a, b), max(a, b)))
    pairs = list(pairs)
    rng.shuffle(pairs)
    flipped = [(b, a) if rng.random() < 0.5 else (a, b) for a, b in pairs]
    # half by id, half by name (C6)
    g.add_edges([(names[a], names[b]) if i % 2 else (a, b) for i, (a, b) in enumerate(flipped)])
    g.es["weight"] = [float(i) for i in range(len(pairs))]
    return g, names, pairs


def test_edges_are_stored_low_high_in_insertion_order():  # C1, C8
    rng = random.Random(1)
    for _ in range(20):
        g, names, pairs = _random_graph(rng, 30, 60)
        assert [(e.source, e.target) for e in g.es] == pairs
        assert [e.index for e in g.es] == list(range(len(pairs)))
        assert [e["weight"] for e in g.es] == [float(i) for i in range(len(pairs))]
        for i, (a, b) in enumerate(pairs):  # C6
            assert g.get_eid(a, b) == i and g.get_eid(b, a) == i and g.get_eid(names[b], names[a]) == i


def test_delete_edges_keeps_order_and_compacts_ids():  # C2, C7
    rng = random.Random(2)
    for _ in range(20):
        g, names, pairs = _random_graph(rng, 25, 70)
        kill = set(rng.sample(range(len(pairs)), 25))
        by_obj = [g.es[i] for i in sorted(kill)][:10]
        g.delete_edges(by_obj + [i for i in kill if i not in {e.index for e in by_obj}])
        left = [(p, float(i)) for i, p in enumerate(pairs) if i not in kill]
        assert [(e.source, e.target) for e in g.es] == [p for p, _ in left]
        assert [e["weight"] for e in g.es] == [w for _, w in left]
        assert [e.index for e in g.es] == list(range(len(left)))
        for v in range(25):
            inc = [i for i, (p, _) in enumerate(left) if v in p]
            assert sorted(g.incident(v)) == inc and g.vs[v].degree() == len(inc)
            assert sorted(g.neighbors(v)) == sorted(p[0] + p[1] - v for p, _ in left if v in p)
        g.delete_edges(0)  # a single id
        assert [(e.source, e.target) for e in g.es] == [p for p, _ in left[1:]]


def test_subgraph_renumbers_in_ascending_original_order():  # C3
    rng = random.Random(3)
    for _ in range(20):
        g, names, pairs = _random_graph(rng, 30, 80)
        pick = rng.sample(range(30), 12)
        rng.shuffle(pick)
        sub = g.subgraph([names[v] if i % 2 else v for i, v in enumerate(pick)])
        keep = sorted(pick)
        assert [v["name"] for v in sub.vs] == [names[v] for v in keep]
        new = {old: i for i, old in enumerate(keep)}
        want = [((new[a], new[b]), float(i)) for i, (a, b) in enumerate(pairs) if a in new and b in new]
        assert [(e.source, e.target) for e in sub.es] == [p for p, _ in want]
        assert [e["weight"] for e in sub.es] == [w for _, w in want]
        assert [v.index for v in sub.vs] == list(range(len(keep)))


def test_components_ordered_by_lowest_vertex():  # C4
    rng = random.Random(4)
    for _ in range(30):
        g, names, pairs = _random_graph(rng, 40, 25)
        comps = g.components()
        assert sorted(v for c in comps for v in c) == list(range(40))
        assert all(c == sorted(c) for c in comps)
        assert [c[0] for c in comps] == sorted(c[0] for c in comps)
        label = {}
        for ci, c in enumerate(comps):
            for v in c:
                label[v] = ci
        assert all(label[a] == label[b] for a, b in pairs)
        # two components are never joined by a chain of edges: union-find restatement
        parent = list(range(40))

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x
        for a, b in pairs:
            parent[find(a)] = find(b)
        assert len({find(v) for v in range(40)}) == len(comps)


def test_shortest_path_on_chains_and_trees():  # C5
    rng = random.Random(5)
    for _ in range(20):
        n = rng.randrange(2, 40)
        order = list(range(n))
        rng.shuffle(order)
        g = ig.Graph()
        g.add_vertices([f"v{i}" for i in range(n + 3)])
        chain = list(zip(order, order[1:]))
        rng.shuffle(chain)
        g.add_edges(chain)
        assert g.get_shortest_paths(order[0], to=order[-1]) == [order]
        assert g.get_shortest_paths(f"v{order[-1]}", to=f"v{order[0]}") == [order[::-1]]
        assert g.get_shortest_paths(order[0], to=n + 1) == [[]]   # unreachable: an empty path
        i, j = sorted(rng.sample(range(n), 2)) if n > 2 else (0, n - 1)
        assert g.get_shortest_paths(order[i], to=order[j]) == [order[i:j + 1]]
    # a tree: the path is the unique one
    g = ig.Graph()
    g.add_vertices(list("abcdefg"))
    g.add_edges([("a", "b"), ("b", "c"), ("b", "d"), ("d", "e"), ("e", "f"), ("d", "g")])
    assert [[g.vs[v]["name"] for v in p] for p in g.get_shortest_paths("c", to="f")] == [list("cbdef")]


def test_copy_is_independent():  # C8
    rng = random.Random(6)
    g, names, pairs = _random_graph(rng, 15, 30)
    h = g.copy()
    h.delete_edges([0, 1, 2])
    h.es["weight"] = [0.0] * len(h.es)
    assert len(g.es) == 30 and [e["weight"] for e in g.es] == [float(i) for i in range(30)]
    assert [v["name"] for v in h.vs] == names and h.vcount() == 15
