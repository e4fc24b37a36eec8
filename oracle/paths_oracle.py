"""
paths_oracle.py -- CPU oracle for the first "next" row: global edge filter, per-component branch
filtering and linear-path extraction on the minimizer graph.  TEST INFRASTRUCTURE ONLY (see graph_oracle.py).

Restates, on plain dict/list state produced by graph_oracle.load_and_build:
  filter_graph_global     reference bin/ntjoin.py:80-89
  filter_graph            reference bin/ntjoin.py:69-77
  is_graph_linear         reference bin/ntjoin.py:105-111
  check_circularity       reference bin/ntjoin.py:113-135
  determine_source_vertex reference bin/ntjoin.py:91-103
  find_paths_process      reference bin/ntjoin.py:137-161
Pinned by tests/golden/cases/*/reference.json["paths_by_n"], produced by the reference's own find_paths().
Vertex/component ORDER is not defined by the reference (python set order), so results are compared as sets of paths,
grouped per component of the globally filtered graph.
"""
from collections import defaultdict, deque


def _components(vertices, edges):
    adj = defaultdict(list)
    for s, t, _w in edges:
        adj[s].append(t)
        adj[t].append(s)
    seen, comps = set(), []
    for v in vertices:
        if v in seen:
            continue
        seen.add(v)
        comp, dq = [], deque([v])
        while dq:
            u = dq.popleft()
            comp.append(u)
            for x in adj[u]:
                if x not in seen:
                    seen.add(x)
                    dq.append(x)
        comps.append(comp)
    return comps


def _degrees(vertices, edges):
    deg = {v: 0 for v in vertices}
    for s, t, _w in edges:
        deg[s] += 1
        deg[t] += 1
    return deg


def find_paths(state, n):
    """-> list (one entry per component of the globally filtered graph) of lists of paths (vertex names, source->target)"""
    weights = state["weights"]
    vertices = list(state["vertices"])
    edges = [(s, t, w) for s, t, _sup, w in state["edges"]]
    if not n <= min(weights.values()):                      # filter_graph_global
        edges = [e for e in edges if not e[2] < n]
    max_w = max(weights.values())
    first_max = [a for a, wt in weights.items() if wt == max_w][0]   # sorted(..., reverse=True)[0]: stable -> first
    last_max = [a for a, wt in weights.items() if wt == max_w][-1]   # [...].pop() -> last
    info = state["list_mx_info"]
    out = []
    for comp in _components(vertices, edges):
        cset = set(comp)
        cedges = [e for e in edges if e[0] in cset]
        min_w, total_w = n, sum(weights.values())
        while True:                                           # find_paths_process: branch filtering
            deg = _degrees(comp, cedges)
            if all(d < 3 for d in deg.values()) or not min_w <= total_w:
                break
            branch = {v for v, d in deg.items() if d > 2}
            cedges = [e for e in cedges if not ((e[0] in branch or e[1] in branch) and e[2] < min_w)]
            min_w += 1
        paths = []
        for sub in _components(comp, cedges):
            sset = set(sub)
            sedges = [e for e in cedges if e[0] in sset]
            deg = _degrees(sub, sedges)
            sources = [v for v in sub if deg[v] == 1]
            if not sources:                                   # check_circularity
                if all(d == 2 for d in deg.values()):
                    mv = min(sub, key=lambda v: info[first_max][v][1])
                    nbrs = [t if s == mv else s for s, t, _w in sedges if mv in (s, t)]
                    hn = max(nbrs, key=lambda v: info[first_max][v][1])
                    sedges = [e for e in sedges if {e[0], e[1]} != {mv, hn}]
                    sources = [mv, hn]
            if len(sources) != 2:
                continue
            pos = {v: info[last_max][v][1] for v in sources}  # determine_source_vertex
            source = [v for v in sources if pos[v] == min(pos.values())][-1]
            target = [v for v in sources if pos[v] == max(pos.values())][-1]
            adj = defaultdict(list)
            for s, t, _w in sedges:
                adj[s].append(t)
                adj[t].append(s)
            prev, dq = {source: None}, deque([source])        # shortest path
            while dq:
                u = dq.popleft()
                for x in adj[u]:
                    if x not in prev:
                        prev[x] = u
                        dq.append(x)
            if target not in prev:
                continue
            path, u = [], target
            while u is not None:
                path.append(u)
                u = prev[u]
            path.reverse()
            if len(path) == len(sub) and len(path) - 1 == len(sedges) and len(path) == len(set(path)):
                paths.append(path)
        out.append(paths)
    return out


def canonical(paths_by_component):
    """order-free form: set of components, each a frozenset of path tuples (components without paths dropped)"""
    return {frozenset(tuple(p) for p in comp) for comp in paths_by_component if comp}


# ---- row f4: what the scaffolder derives from a path for the target assembly --------------------------------------
def mx_extremes(state, target):
    """find_mx_min_max (reference bin/ntjoin_assemble.py:688-702): contig -> (min, max) position over the target's
    minimizers that are graph vertices"""
    out = {}
    vertices = set(state["vertices"])
    for mx, (ctg, pos) in state["list_mx_info"][target].items():
        if mx not in vertices:
            continue
        lo, hi = out.get(ctg, (pos, pos))
        out[ctg] = (min(lo, pos), max(hi, pos))
    return out


def determine_orientation(positions, m=90):
    """reference bin/ntjoin_assemble.py:30-50 without --mkt"""
    if len(positions) > 1:
        pairs = list(zip(positions, positions[1:]))
        if all(x < y for x, y in pairs):
            return "+"
        if all(x > y for x, y in pairs):
            return "-"
        positive = sum(1 for x, y in pairs if x < y) / float(len(positions) - 1) * 100
        if positive >= m:
            return "+"
        if 100 - positive >= m:
            return "-"
    return "?"


def format_path(state, path, target, lengths, k, g=20, G=0, m=90):
    """format_path + calc_start/end_coord + calculate_gap_size (reference bin/ntjoin_assemble.py:175-218,52-65,68-112):
    -> [[contig, ori, start, end, contig_size, first_mx, terminal_mx, gap_size, raw_gap_size], ...]"""
    info = state["list_mx_info"]
    extremes = mx_extremes(state, target)
    support = {frozenset((s, t)): sup for s, t, sup, _w in state["edges"]}
    index = {mx: i for i, mx in enumerate(path)}
    groups, cur = [], None
    for mx in path:                                         # runs of minimizers on the same target contig
        ctg, pos = info[target][mx]
        if cur is not None and cur["ctg"] == ctg:
            cur["pos"].append(pos)
        else:
            cur = {"ctg": ctg, "pos": [pos], "first": mx}
            groups.append(cur)
        cur["last"] = mx
    nodes = []
    for grp in groups:
        ori = determine_orientation(grp["pos"], m)
        if ori == "?":
            continue
        lo, hi = min(grp["pos"]), max(grp["pos"])
        start = 0 if lo == extremes[grp["ctg"]][0] else lo
        end = lengths[grp["ctg"]] if hi == extremes[grp["ctg"]][1] else hi + k
        nodes.append([grp["ctg"], ori, start, end, lengths[grp["ctg"]], grp["first"], grp["last"], 0, 0])
    for u, v in zip(nodes, nodes[1:]):
        u_mx, v_mx = u[6], v[5]
        between = path[index[u_mx]:index[v_mx] + 1]          # the chain IS the shortest path
        sups = [set(support[frozenset(e)]) for e in zip(between, between[1:])]
        common = set.intersection(*sups)
        if not common:
            u[7], u[8] = g, g
            continue
        dists = [abs(info[a][v_mx][1] - info[a][u_mx][1]) for a in common]
        mean_dist = int(sum(dists) / len(dists)) - k
        a = (u[3] - info[target][u_mx][1] - k) if u[1] == "+" else (info[target][u_mx][1] - u[2])
        b = (info[target][v_mx][1] - v[2]) if v[1] == "+" else (v[3] - info[target][v_mx][1] - k)
        assert a >= 0 and b >= 0
        gap = max(mean_dist - a - b, g)
        if G > 0:
            gap = min(gap, G)
        u[7], u[8] = gap, mean_dist - a - b
    return nodes
