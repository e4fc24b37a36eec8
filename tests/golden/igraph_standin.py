"""
A small stand-in for the python-igraph CONTAINER API, used only by tests/golden/make_golden.py in the build container
(python-igraph is not installed there).  It implements exactly the calls the reference makes on the hot path and on the
path-extraction step that follows it (reference bin/ntjoin_utils.py:37-47,83-141; bin/ntjoin.py:25-176):
Graph(), add_vertices, add_edges, get_eid, vs / vs(), es / es(), es()[attr] = list, copy, delete_edges, components,
subgraph, incident, neighbors, get_shortest_paths, Vertex.degree().  Semantics follow igraph's documented behaviour:
undirected simple graph, an edge is reported as (lower vertex id, higher vertex id), subgraph() renumbers vertices
and edges preserving their relative order, components() lists components by their lowest vertex id.
The contracts it models (C1-C8: edge tuples low/high in insertion order; edge ids compact after delete_edges; subgraph renumbers in
ascending original id; components by lowest vertex; unique shortest path on a chain / tree, empty when unreachable; get_eid either
way round and by name; degree / incident / neighbors consistent after deletions; element-wise attribute assignment, deep copy) are
listed with their igraph documentation references in tests/test_igraph_standin_cpu.py, which checks each on random graphs.
"""
import sys
import types
from collections import deque


class _Vertex(dict):
    def __init__(self, graph, index, name):
        super().__init__(name=name)
        self._graph, self.index = graph, index

    def degree(self):
        return len(self._graph._adj()[self.index])


class _Edge(dict):
    def __init__(self, index, source, target, attrs=None):
        super().__init__(attrs or {})
        self.index, self.source, self.target = index, source, target


class _Seq(list):
    """vs / es sequence: callable (graph.vs()), indexable, attribute assignment by name for whole-sequence lists"""

    def __call__(self):
        return self

    def __setitem__(self, key, values):
        if isinstance(key, str):
            assert len(values) == len(self)
            for e, v in zip(self, values):
                dict.__setitem__(e, key, v)
        else:
            list.__setitem__(self, key, values)

    def find(self, name):
        for v in self:
            if v["name"] == name:
                return v
        raise ValueError(f"no such vertex: {name}")


class Graph:
    def __init__(self):
        self.vs, self.es = _Seq(), _Seq()
        self._idx = {}
        self._adj_cache = None

    # -- construction ------------------------------------------------------------------------------------
    def _vid(self, x):
        return x if isinstance(x, int) else self._idx[x]

    def add_vertices(self, names):
        for n in names:
            self._idx[n] = len(self.vs)
            self.vs.append(_Vertex(self, len(self.vs), n))
        self._adj_cache = None

    def add_edges(self, pairs):
        for s, t in pairs:
            a, b = self._vid(s), self._vid(t)
            self.es.append(_Edge(len(self.es), min(a, b), max(a, b)))
        self._adj_cache = None

    def copy(self):
        g = Graph()
        g.add_vertices([v["name"] for v in self.vs])
        for e in self.es:
            g.es.append(_Edge(len(g.es), e.source, e.target, dict(e)))
        return g

    def delete_edges(self, which):
        ids = {which} if isinstance(which, int) else {w if isinstance(w, int) else w.index for w in which}
        kept = [e for e in self.es if e.index not in ids]
        self.es = _Seq()
        for e in kept:
            self.es.append(_Edge(len(self.es), e.source, e.target, dict(e)))
        self._adj_cache = None

    # -- queries -----------------------------------------------------------------------------------------
    def _adj(self):
        if self._adj_cache is None:
            adj = [[] for _ in self.vs]
            for e in self.es:
                adj[e.source].append((e.target, e.index))
                adj[e.target].append((e.source, e.index))
            self._adj_cache = adj
        return self._adj_cache

    def vcount(self):
        return len(self.vs)

    def get_eid(self, s, t):
        a, b = self._vid(s), self._vid(t)
        for n, eid in self._adj()[a]:
            if n == b:
                return eid
        raise ValueError("no such edge")

    def incident(self, v):
        return [eid for _, eid in self._adj()[self._vid(v)]]

    def neighbors(self, v):
        return [n for n, _ in self._adj()[self._vid(v)]]

    def components(self):
        seen, comps, adj = [False] * len(self.vs), [], self._adj()
        for s in range(len(self.vs)):
            if seen[s]:
                continue
            seen[s], comp, dq = True, [], deque([s])
            while dq:
                u = dq.popleft()
                comp.append(u)
                for n, _ in adj[u]:
                    if not seen[n]:
                        seen[n] = True
                        dq.append(n)
            comps.append(sorted(comp))
        return comps

    def subgraph(self, vertices):
        keep = sorted(self._vid(v) for v in vertices)
        new_id = {old: i for i, old in enumerate(keep)}
        g = Graph()
        g.add_vertices([self.vs[old]["name"] for old in keep])
        for e in self.es:
            if e.source in new_id and e.target in new_id:
                a, b = new_id[e.source], new_id[e.target]
                g.es.append(_Edge(len(g.es), min(a, b), max(a, b), dict(e)))
        return g

    def get_shortest_paths(self, source, to=None, output="vpath"):
        s, t = self._vid(source), self._vid(to)
        prev, adj, dq = {s: None}, self._adj(), deque([s])
        while dq:
            u = dq.popleft()
            if u == t:
                break
            for n, _ in adj[u]:
                if n not in prev:
                    prev[n] = u
                    dq.append(n)
        if t not in prev:
            return [[]]
        path, u = [], t
        while u is not None:
            path.append(u)
            u = prev[u]
        return [path[::-1]]


def install():
    mod = types.ModuleType("igraph")
    mod.Graph = Graph
    sys.modules["igraph"] = mod
