"""
graph_oracle.py -- CPU oracle for the minimizer-graph stage.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and
only as the checker / reported CPU baseline.  The product path (ntjoin_amd) never imports it.

It restates, with plain dict/list/set, what the reference computes between the `.tsv` sketches and
the `.mx.dot` file:

  read_minimizers      reference bin/ntjoin_utils.py:167-193
  filter_minimizers    reference bin/ntjoin_utils.py:152-165
  build_graph          reference bin/ntjoin_utils.py:83-115,121,129,132-137 (fresh-graph branch),
                       calc_total_weight :54-56
  load order / weights reference bin/ntjoin.py:178-186, bin/ntjoin_assemble.py:799-807
  print_graph          reference bin/ntjoin.py:25-62

Pinned by tests/golden/cases/*/reference.json, which were produced by importing the reference's own
bin/ntjoin_utils.py and bin/ntjoin.py in the build container (tests/golden/make_golden.py).
"""

COLOURS = ["red", "green", "blue", "purple", "orange",
           "turquoise", "pink", "yellow", "orchid", "salmon"]


def read_minimizers(tsv_filename):
    """TSV -> (mx_info, mxs): hashes seen exactly once in the assembly, and per-contig ordered lists with
    every duplicated hash removed (empty lists kept).  Follows ntjoin_utils.py:167-193."""
    mx_info = {}
    mxs = []
    dup = set()
    with open(tsv_filename, "r", encoding="utf-8") as tsv:
        for line in tsv:
            fields = line.strip().split("\t")
            if len(fields) > 1:
                contig = fields[0]
                entries = fields[1].split(" ")
                mxs.append([e.split(":")[0] for e in entries])
                for e in entries:
                    mx, pos, _seq = e.split(":")  # exactly three fields, as HEAD's parser demands
                    if mx in mx_info:
                        dup.add(mx)
                    else:
                        mx_info[mx] = (contig, int(pos))
    mx_info = {mx: v for mx, v in mx_info.items() if mx not in dup}
    mxs_filt = [[mx for mx in lst if mx not in dup] for lst in mxs]
    return mx_info, mxs_filt


def filter_minimizers(list_mxs):
    """Keep only hashes present in every assembly; order preserved (ntjoin_utils.py:152-165)."""
    sets = [{mx for lst in list_mxs[a] for mx in lst} for a in list_mxs]
    inter = set.intersection(*sets)
    return {a: [[mx for mx in lst if mx in inter] for lst in list_mxs[a]] for a in list_mxs}


def build_edges(list_mxs, weights):
    """Edge dictionary exactly as ntjoin_utils.build_graph builds it before handing it to igraph:
    returns (vertices:set, edges:list of (s, t, support:list, weight:float)) with (s,t) in the
    reference's `formatted_edges` order (first-seen orientation; grouped by first-seen source)."""
    vertices = set()
    edges = {}  # source -> {target -> [assembly, ...]}
    for assembly in list_mxs:
        for lst in list_mxs[assembly]:
            for i in range(len(lst) - 1):
                a, b = lst[i], lst[i + 1]
                if a in edges and b in edges[a]:
                    edges[a][b].append(assembly)
                elif b in edges and a in edges[b]:
                    edges[b][a].append(assembly)
                else:
                    edges.setdefault(a, {})[b] = [assembly]
                vertices.add(a)
            if lst:
                vertices.add(lst[-1])
    out = []
    for s in edges:
        for t in edges[s]:
            support = edges[s][t]
            out.append((s, t, list(support), sum(weights[f] for f in support)))
    return vertices, out


def load_and_build(ref_tsvs, ref_weights, target_tsv, target_weight):
    """Reference call order: refs in CLI order (ntjoin.py:178-186), then target
    (ntjoin_assemble.py:799-807); filter (ntjoin.py:198); build (ntjoin.py:201)."""
    list_mx_info, list_mxs, weights = {}, {}, {}
    for tsv, wt in zip(ref_tsvs, ref_weights):
        info, mxs = read_minimizers(tsv)
        list_mx_info[tsv], list_mxs[tsv], weights[tsv] = info, mxs, float(wt)
    info, mxs = read_minimizers(target_tsv)
    list_mx_info[target_tsv], list_mxs[target_tsv], weights[target_tsv] = info, mxs, float(target_weight)
    filtered = filter_minimizers(list_mxs)
    vertices, edges = build_edges(filtered, weights)
    return {"list_mx_info": list_mx_info, "list_mxs": list_mxs, "weights": weights,
            "filtered": filtered, "vertices": vertices, "edges": edges}


def dot_lines(state):
    """(node_lines, edge_lines) of the .mx.dot text (ntjoin.py:25-62); vertex/edge ORDER is not
    defined by the reference (python set order + igraph ids), so callers compare canonical forms."""
    files = list(state["list_mx_info"].keys())
    colours = COLOURS if len(files) <= len(COLOURS) else ["red"] * len(files)
    nodes = []
    for name in state["vertices"]:
        labels = "\n".join(f"{f}_{state['list_mx_info'][f][name]}" for f in files)
        nodes.append(f"\"{name}\" [label=\"{name}\n{labels}\"]")
    edges = []
    for s, t, support, weight in state["edges"]:
        if len(support) == 1:
            colour = colours[files.index(support[0])]
        elif len(support) == 2:
            colour = "lightgrey"
        else:
            colour = "black"
        edges.append((s, t, f"[weight={weight} color={colour}]"))
    return nodes, edges


def canonical_dot_from_state(state):
    nodes, edges = dot_lines(state)
    cedges = sorted((min(int(s), int(t)), max(int(s), int(t)), attr) for s, t, attr in edges)
    return {"nodes": sorted(nodes), "edges": [[str(a), str(b), attr] for a, b, attr in cedges]}


def canonical_dot_from_text(text):
    """Canonical form of a `.mx.dot` file: node statements sorted; each edge
    as (min endpoint, max endpoint, attribute text), sorted.  Accepts `--"v"` and `-- "v"`."""
    import re
    body = text.strip()
    assert body.startswith("graph G {") and body.endswith("}"), "not a graph G { ... } file"
    body = body[len("graph G {"):-1].strip("\n")
    # a node statement spans several physical lines (the label holds real newlines)
    stmts, cur = [], ""
    for line in body.split("\n"):
        cur = line if not cur else cur + "\n" + line
        if cur.endswith("]"):
            stmts.append(cur)
            cur = ""
    assert cur == "", f"trailing junk in dot: {cur!r}"
    nodes, edges = [], []
    edge_re = re.compile(r'^"(\d+)" -- ?"(\d+)" (\[weight=\S+ color=\S+\])$')
    for s in stmts:
        m = edge_re.match(s)
        if m:
            a, b = int(m.group(1)), int(m.group(2))
            edges.append((min(a, b), max(a, b), m.group(3)))
        else:
            nodes.append(s)
    return {"nodes": sorted(nodes), "edges": [[str(a), str(b), attr] for a, b, attr in sorted(edges)]}
