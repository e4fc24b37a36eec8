"""
Drop-in counterparts of the reference's hot-path helpers (same names, argument meaning and error
behaviour), computing on the GPU through libntjoin_mx:

    read_minimizers(tsv_filename, repeat_bf=False) -> (mx_info, mxs)     reference bin/ntjoin_utils.py:167-193
    filter_minimizers(list_mxs) -> dict                                   reference bin/ntjoin_utils.py:152-165
    build_graph(list_mxs, weights, graph=None, black_list=None) -> MxGraph   reference bin/ntjoin_utils.py:83-141
    calc_total_weight(list_files, weights)                                reference bin/ntjoin_utils.py:54-56
    run_indexlr(assembly, k, w, t, **kwargs) -> tsv name                  reference bin/ntjoin_utils.py:195-202

Hashes stay decimal strings at this boundary, exactly as in the reference (its vertex `name`s).  These
functions convert between Python containers and arrays and therefore pay Python costs per minimizer;
the fused path (ntjoin_amd.ntjoin.Ntjoin, which keeps everything in HBM between the stages) is the fast one.
python-igraph is not a dependency: build_graph returns an MxGraph (struct-of-arrays + name index) that
offers the small part of the igraph API the reference touches on this path.
"""
import datetime
import sys
from collections.abc import Mapping, Sequence

import numpy as np

from .engine import MxEngine, MxError
from . import capi


class MxInfo(Mapping):
    """`mx_info` of the reference (hash string -> (contig, position), bin/ntjoin_utils.py:187-192) as a read-only mapping
    over numpy arrays: at 6 M minimizers per assembly a real dict of Python strings and tuples costs seconds and ~1 GB;
    this costs one argsort.  Keys may be given as decimal strings (the reference's spelling) or ints.  Picklable."""

    def __init__(self, out_hash, pos, record, record_ids):
        order = np.argsort(out_hash, kind="stable")
        self.hash = np.ascontiguousarray(out_hash[order])
        self.pos = np.ascontiguousarray(pos[order])
        self.record = np.ascontiguousarray(record[order])
        self.record_ids = list(record_ids)

    def _find(self, key):
        try:
            v = np.uint64(int(key))
        except (TypeError, ValueError, OverflowError):
            return -1
        i = int(np.searchsorted(self.hash, v))
        return i if i < len(self.hash) and self.hash[i] == v else -1

    def __getitem__(self, key):
        i = self._find(key)
        if i < 0:
            raise KeyError(key)
        return (self.record_ids[int(self.record[i])], int(self.pos[i]))

    def __contains__(self, key):
        return self._find(key) >= 0

    def __len__(self):
        return len(self.hash)

    def __iter__(self):
        return (str(h) for h in self.hash.tolist())

    def to_dict(self):
        return {str(h): (self.record_ids[r], p) for h, r, p in zip(self.hash.tolist(), self.record.tolist(), self.pos.tolist())}


class MxLists(Sequence):
    """`mxs` of the reference (per contig the ordered list of hash strings, bin/ntjoin_utils.py:178,193) over one hash
    array + offsets; item i materialises contig i's list of decimal strings on demand.  .hashes / .offsets are the arrays."""

    def __init__(self, hashes, offsets):
        self.hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)

    def __len__(self):
        return len(self.offsets) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return [str(x) for x in self.hashes[self.offsets[i]:self.offsets[i + 1]].tolist()]

    def to_lists(self):
        return [self[i] for i in range(len(self))]


class MxGraph:
    """Undirected minimizer graph: vertices = decimal-string hashes, edge attributes support / weight."""

    @classmethod
    def from_arrays(cls, vertex_hash, edge_u, edge_v, edge_support, edge_weight, assembly_names):
        """the same container over the engine's arrays: names, index and support lists are built on first use, so a
        4.5 M-vertex graph (3 Gbp + 3 Gbp) costs nothing until somebody asks for strings"""
        g = cls.__new__(cls)
        g.vertex_hash = np.ascontiguousarray(vertex_hash, dtype=np.uint64)
        g.edge_u, g.edge_v = np.ascontiguousarray(edge_u), np.ascontiguousarray(edge_v)
        g.edge_support_mask = np.ascontiguousarray(edge_support)
        g.edge_weight = np.ascontiguousarray(edge_weight, dtype=np.float64)
        g.assembly_names = list(assembly_names)
        g._lazy = True
        return g

    def __getattr__(self, name):
        # (only reached for attributes not set yet: the lazily built views of from_arrays)
        if name.startswith("__") or not self.__dict__.get("_lazy"):
            raise AttributeError(name)
        if name == "names":
            self.names = [str(h) for h in self.vertex_hash.tolist()]
        elif name == "_index":
            self._index = {n: i for i, n in enumerate(self.names)}
        elif name == "edges":
            self.edges = list(zip(self.edge_u.tolist(), self.edge_v.tolist()))
        elif name == "support":
            nm = self.assembly_names
            self.support = [[nm[b] for b in range(len(nm)) if m >> b & 1] for m in self.edge_support_mask.tolist()]
        elif name == "weight":
            self.weight = self.edge_weight.tolist()
        elif name == "_eid":
            self._eid = {(min(s, t), max(s, t)): e for e, (s, t) in enumerate(self.edges)}
        else:
            raise AttributeError(name)
        return self.__dict__[name]

    def __init__(self, names, edges, support, weight):
        self._lazy = False
        self.names = list(names)                  # vertex id -> name
        self._index = {n: i for i, n in enumerate(self.names)}
        self.edges = [(int(s), int(t)) for s, t in edges]  # (source id, target id), first-seen orientation
        self.support = [list(s) for s in support]  # per edge: assembly names in load order
        self.weight = [float(x) for x in weight]
        self._eid = {}
        for e, (s, t) in enumerate(self.edges):
            self._eid[(min(s, t), max(s, t))] = e

    def vcount(self):
        return len(self.vertex_hash) if self.__dict__.get("_lazy") else len(self.names)

    def ecount(self):
        return len(self.edge_u) if self.__dict__.get("_lazy") else len(self.edges)

    def vertex_index(self, name):
        return self._index[name]

    def get_eid(self, source, target):
        s = source if isinstance(source, int) else self._index[source]
        t = target if isinstance(target, int) else self._index[target]
        return self._eid[(min(s, t), max(s, t))]

    def edge_list_named(self):
        """[(source name, target name, support list, weight)] in edge-id order."""
        return [(self.names[s], self.names[t], sup, w) for (s, t), sup, w in zip(self.edges, self.support, self.weight)]

    def degree(self):
        deg = [0] * len(self.names)
        for s, t in self.edges:
            deg[s] += 1
            deg[t] += 1
        return deg

    def to_igraph(self):
        import igraph as ig  # optional
        g = ig.Graph()
        g.add_vertices(self.names)
        g.add_edges(self.edges)
        g.es["support"] = self.support
        g.es["weight"] = self.weight
        return g


def calc_total_weight(list_files, weights):
    "Calculate the total weight of an edge given the assembly support"
    return sum((weights[f] for f in list_files))


def _lists_from_sketch(sk, keep):
    out = []
    first = sk["record_first"]
    hashes = sk["out_hash"]
    for r in range(len(sk["record_ids"])):
        lo, hi = int(first[r]), int(first[r + 1])
        if hi > lo:  # only records with at least one entry appear in the reference's `mxs` (:176)
            sel = keep[lo:hi]
            out.append([str(x) for x in hashes[lo:hi][sel].tolist()])
    return out


def read_minimizers(tsv_filename, repeat_bf=False, k=32, views=False):
    """Read the minimizers from a file, removing duplicate minimizers.
    views=True: the same two objects as array-backed views (MxInfo, MxLists) instead of a dict and lists of Python strings --
    what a 3 Gbp assembly needs (6 M minimizers: ~0.2 s instead of ~15 s; SURVEY.md B2 "numpy views at scale")."""
    print(datetime.datetime.today(), ": Reading minimizers", tsv_filename, file=sys.stdout)
    if repeat_bf:
        raise NotImplementedError("repeat_bf is never supplied on ntJoin's own path (bin/ntjoin.py:178)")
    try:
        with MxEngine(k=k, w=1) as eng:
            eng.add_tsv(tsv_filename, 1.0, tsv_filename)
            eng.build_graph()  # one assembly: MXG_MX_UNIQUE = "seen exactly once in this assembly"
            sk = eng.get_sketch(0)
            flags = eng.get_mx_flags(0)
    except MxError as e:
        if e.code == capi.MXG_EIO and "three" in str(e):
            raise ValueError(str(e)) from None  # the reference's tuple-unpack ValueError (:181)
        if e.code == capi.MXG_EIO:
            raise FileNotFoundError(str(e)) from None
        raise
    uniq = (flags & capi.MX_UNIQUE) != 0
    ids = sk["record_ids"]
    if views:
        return sketch_views(sk, uniq)
    mx_info = {str(h): (ids[r], int(p)) for h, p, r in
               zip(sk["out_hash"][uniq].tolist(), sk["pos"][uniq].tolist(), sk["record"][uniq].tolist())}
    return mx_info, _lists_from_sketch(sk, uniq)


def sketch_views(sk, uniq):
    """(MxInfo, MxLists) of one assembly from its sketch arrays and the "occurs once in this assembly" mask"""
    first = sk["record_first"].astype(np.int64)
    kept_before = np.concatenate(([0], np.cumsum(uniq, dtype=np.int64)))
    has = first[1:] > first[:-1]  # only records with at least one entry appear in the reference's `mxs` (:176)
    offs = np.concatenate((kept_before[first[:-1]][has], [kept_before[-1]])) if has.any() else np.zeros(1, dtype=np.int64)
    return (MxInfo(sk["out_hash"][uniq], sk["pos"][uniq], sk["record"][uniq], sk["record_ids"]),
            MxLists(sk["out_hash"][uniq], offs))


def _engine_from_lists(list_mxs, weights=None):
    eng = MxEngine(k=32, w=1)
    for assembly in list_mxs:
        lists = list_mxs[assembly]
        hashes = np.fromiter((int(mx) for lst in lists for mx in lst), dtype=np.uint64)
        rec = np.repeat(np.arange(len(lists), dtype=np.uint32), [len(lst) for lst in lists]) if lists else \
            np.zeros(0, dtype=np.uint32)
        pos = np.zeros(len(hashes), dtype=np.uint32)
        eng.add_minimizers(str(assembly), 1.0 if weights is None else weights[assembly], hashes, pos, rec,
                           [str(i) for i in range(len(lists))])
    return eng


def filter_minimizers(list_mxs):
    "Filters out minimizers that are not found in all assemblies"
    print(datetime.datetime.today(), ": Filtering minimizers", file=sys.stdout)
    if not list_mxs:
        raise TypeError("unbound method set.intersection() needs an argument")  # what the reference raises
    return_mxs = {}
    with _engine_from_lists(list_mxs) as eng:
        eng.build_graph()
        for a, assembly in enumerate(list_mxs):
            flags = eng.get_mx_flags(a)
            keep = (flags & capi.MX_INALL) != 0
            out, i = [], 0
            for lst in list_mxs[assembly]:
                n = len(lst)
                out.append([mx for mx, kf in zip(lst, keep[i:i + n].tolist()) if kf])
                i += n
            return_mxs[assembly] = out
    return return_mxs


def build_graph(list_mxs, weights, graph=None, black_list=None):
    "Builds an undirected graph: nodes=minimizers; edges=between adjacent minimizers"
    print(datetime.datetime.today(), ": Building graph", file=sys.stdout)
    if graph is not None or black_list is not None:
        raise NotImplementedError("incremental build_graph (graph=/black_list=) is not used by ntJoin's CLI "
                                  "(bin/ntjoin.py:201) and is out of scope")
    names_in_order = list(list_mxs.keys())
    with _engine_from_lists(list_mxs, weights) as eng:
        eng.build_graph()
        for a in range(len(names_in_order)):
            flags = eng.get_mx_flags(a)
            if not np.all((flags & capi.MX_SHARED) != 0):
                raise ValueError("build_graph on the GPU expects filter_minimizers' output: every minimizer exactly "
                                 "once in every assembly (the reference's call site, bin/ntjoin.py:198-201)")
        g = eng.get_graph()
    print(datetime.datetime.today(), ": Adding vertices", file=sys.stdout)
    print(datetime.datetime.today(), ": Adding edges", file=sys.stdout)
    print(datetime.datetime.today(), ": Adding attributes", file=sys.stdout)
    names = [str(h) for h in g["vertex_hash"].tolist()]
    support = [[names_in_order[b] for b in range(len(names_in_order)) if m >> b & 1]
               for m in g["edge_support"].tolist()]
    return MxGraph(names, zip(g["edge_u"].tolist(), g["edge_v"].tolist()), support, g["edge_weight"].tolist())


# ---- many independent calls at once -----------------------------------------------------------------------------------------
# The overlap stage calls filter_minimizers and build_graph once per pair of adjacent contigs, on two lists of ~15 minimizers,
# thousands of times per run (reference bin/ntjoin_overlap.py:28,132).  One engine, one upload and one pass of the graph-stage
# kernels per call is all overhead at that size; the *_many forms take the calls' inputs together: every item's hashes are
# renumbered (item << 32 | rank of the hash among the item's distinct hashes: a bijection, so items can never meet in the join),
# the items' lists become the records of A assemblies of ONE handle, the graph stage runs once, and the results are split per
# item.  Results are exactly those of the one-at-a-time functions (tests/test_gpu_parity.py).
def _compose_items(items):
    n_asm = {len(it) for it in items}
    if len(n_asm) != 1:
        raise ValueError("*_many: every item must hold the same number of assemblies")
    A = n_asm.pop()
    if len(items) >= (1 << 32):
        raise ValueError("*_many: too many items")
    keys = [[] for _ in range(A)]      # per assembly slot: composed keys of every item's lists, in item order
    recs = [[] for _ in range(A)]
    n_rec = [0] * A
    originals, shapes = [], []
    for i, it in enumerate(items):
        lists_per_asm = [it[name] for name in it]
        flat = np.fromiter((int(mx) for lists in lists_per_asm for lst in lists for mx in lst), dtype=np.uint64)
        uniq = np.unique(flat)
        originals.append(uniq)
        local = (np.uint64(i) << np.uint64(32)) | np.searchsorted(uniq, flat).astype(np.uint64)
        at = 0
        shape = []
        for a, lists in enumerate(lists_per_asm):
            lens = [len(lst) for lst in lists]
            n = int(sum(lens))
            keys[a].append(local[at:at + n])
            recs[a].append(np.repeat(np.arange(n_rec[a], n_rec[a] + len(lists), dtype=np.uint32), lens))
            n_rec[a] += len(lists)
            at += n
            shape.append(lens)
        shapes.append(shape)
    cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dtype=dt)  # noqa: E731
    return A, [cat(k_, np.uint64) for k_ in keys], [cat(r_, np.uint32) for r_ in recs], n_rec, originals, shapes


def _engine_from_composed(A, keys, recs, n_rec, weights=None):
    eng = MxEngine(k=32, w=1)
    for a in range(A):
        eng.add_minimizers(f"slot{a}", 1.0 if weights is None else float(weights[a]), keys[a], np.zeros(len(keys[a]), dtype=np.uint32),
                           recs[a], [str(r) for r in range(n_rec[a])])
    return eng


def _by_assembly_count(items, fn, *per_item):
    """items of 2, 3, ... assemblies are taken group by group (a handle has one number of assemblies); results in input order"""
    out = [None] * len(items)
    for n_asm in sorted({len(it) for it in items}):
        idx = [i for i, it in enumerate(items) if len(it) == n_asm]
        res = fn([items[i] for i in idx], *[[p[i] for i in idx] for p in per_item])
        for i, r in zip(idx, res):
            out[i] = r
    return out


def filter_minimizers_many(items):
    """[filter_minimizers(item) for item in items] with one handle and one pass of the graph-stage kernels (per number of
    assemblies among the items)"""
    items = list(items)
    if not items:
        return []
    if any(not it for it in items):
        raise TypeError("unbound method set.intersection() needs an argument")  # what the reference raises for an empty dict
    return _by_assembly_count(items, _filter_many_group)


def _filter_many_group(items):
    A, keys, recs, n_rec, _, shapes = _compose_items(items)
    with _engine_from_composed(A, keys, recs, n_rec) as eng:
        eng.build_graph()
        keep = [(eng.get_mx_flags(a) & capi.MX_INALL) != 0 for a in range(A)]
    out, at = [], [0] * A
    for it, shape in zip(items, shapes):
        res = {}
        for a, (name, lens) in enumerate(zip(it, shape)):
            lists = []
            for lst, n in zip(it[name], lens):
                kf = keep[a][at[a]:at[a] + n].tolist()
                lists.append([mx for mx, k_ in zip(lst, kf) if k_])
                at[a] += n
            res[name] = lists
        out.append(res)
    return out


def build_graph_many(items, weights):
    """[build_graph(item, w) for item, w in zip(items, weights)] with one handle and one pass of the graph-stage kernels.
    `weights`: one dict per item (assembly -> weight), or one dict for all items.  Every item must be filter_minimizers' output."""
    items = list(items)
    if not items:
        return []
    wts = [weights] * len(items) if isinstance(weights, dict) else list(weights)
    return _by_assembly_count(items, _graph_many_group, wts)


def _graph_many_group(items, wts):
    A, keys, recs, n_rec, originals, _ = _compose_items(items)
    with _engine_from_composed(A, keys, recs, n_rec) as eng:
        eng.build_graph()
        for a in range(A):
            if not np.all((eng.get_mx_flags(a) & capi.MX_SHARED) != 0):
                raise ValueError("build_graph_many expects filter_minimizers' output: every minimizer exactly once in every assembly "
                                 "of its item (the reference's call site, bin/ntjoin.py:198-201)")
        g = eng.get_graph()
    vkey = g["vertex_hash"]
    v_item = (vkey >> np.uint64(32)).astype(np.int64)          # vertices come in assembly 0's order: items are contiguous
    v_first = np.searchsorted(v_item, np.arange(len(items) + 1))
    e_item = v_item[g["edge_u"]] if len(g["edge_u"]) else np.zeros(0, dtype=np.int64)
    graphs = []
    for i, (it, w_i) in enumerate(zip(items, wts)):
        names_in_order = list(it.keys())
        v0, v1 = int(v_first[i]), int(v_first[i + 1])
        names = [str(h) for h in originals[i][(vkey[v0:v1] & np.uint64(0xFFFFFFFF)).astype(np.int64)].tolist()]
        sel = np.flatnonzero(e_item == i)
        masks = g["edge_support"][sel].tolist()
        support = [[names_in_order[b] for b in range(A) if m >> b & 1] for m in masks]
        weight = [float(sum(w_i[nm] for nm in sup)) for sup in support]   # (calc_total_weight: sum in support order)
        graphs.append(MxGraph(names, zip((g["edge_u"][sel] - v0).tolist(), (g["edge_v"][sel] - v0).tolist()), support, weight))
    return graphs


def run_indexlr(assembly, k, w, t, **kwargs):
    "Run indexlr on the given assembly with the specified k and w"
    out = f"{assembly}.k{k}.w{w}.tsv"
    with MxEngine(k=int(k), w=int(w)) as eng:
        eng.add_fasta(out, 1.0, assembly)
        eng.sketch()
        eng.write_tsv(0, out, with_pos=True, with_strand=False, with_seq=True)
    return out
