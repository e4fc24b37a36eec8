"""
Counterpart of the reference's `Ntjoin` base class for the hot path only (reference bin/ntjoin.py:
load_minimizers :178-186, make_minimizer_graph :189-204, print_graph :25-67) plus the target-loading step of
NtjoinScaffolder.load_minimizers_scaffold (reference bin/ntjoin_assemble.py:799-807).

Fused: all assemblies live in ONE engine handle; sketches stay in HBM between load and graph build; the
`.mx.dot` is written by the library.  `args` carries the same attributes the reference reads:
    args.FILES  reference minimizer TSVs (CLI order)      args.s  target TSV        args.l  target weight
    args.p      output prefix                               args.k  k-mer size
Instead of TSVs the engine can sketch FASTA directly: pass fasta={tsv_name: fasta_path} and the TSVs are
WRITTEN (checkpoint files, ntJoin:202 `.SECONDARY`) rather than read.
"""
import datetime
import sys

import numpy as np

from . import capi
from .engine import MxEngine
from .ntjoin_utils import MxGraph, sketch_views

COLOURS = ["red", "green", "blue", "purple", "orange", "turquoise", "pink", "yellow", "orchid", "salmon"]


class Ntjoin:
    "ntJoin hot path: minimizer sketches -> minimizer graph, on the GPU"

    def __init__(self, args, fasta=None, w=None, variant="v2"):
        self.list_mx_info = {}  # assembly -> {mx: (contig, position)}
        self.list_mxs = {}      # assembly -> [lists of mx]
        self._graph = None
        self._graph_pending = False
        self.args = args
        self.weights = {}
        self.weights_list = []
        self._fasta = dict(fasta or {})
        if self._fasta and w is None:
            # (w is only unused on the TSV route; sketching with a default of 1 would write a huge, wrong checkpoint file)
            raise ValueError("Ntjoin(fasta=...): the window size w must be given when assemblies are sketched from FASTA")
        self._engine = MxEngine(k=int(getattr(args, "k", 32)), w=int(w if w is not None else 1), variant=variant)
        self._order = []

    def close(self):
        self._engine.close()

    @property
    def graph(self):
        if self._graph is None and getattr(self, "_graph_pending", False):
            g = self._engine.get_graph()
            self._graph = MxGraph.from_arrays(g["vertex_hash"], g["edge_u"], g["edge_v"], g["edge_support"], g["edge_weight"],
                                              self._order)
            self._graph_pending = False
        return self._graph

    @graph.setter
    def graph(self, value):
        self._graph = value
        self._graph_pending = False

    # -- loading (reference order: refs in FILES order, then target) --------------------------------------
    def _add(self, assembly, weight):
        if assembly in self._fasta:
            a = self._engine.add_fasta(assembly, weight, self._fasta[assembly])
            self._engine.sketch(a)
            self._engine.write_tsv(a, assembly, with_pos=True, with_strand=False, with_seq=True)
        else:
            print(datetime.datetime.today(), ": Reading minimizers", assembly, file=sys.stdout)
            a = self._engine.add_tsv(assembly, weight, assembly)
        self._order.append(assembly)
        self.weights[assembly] = weight
        return a

    def load_minimizers(self, repeat_bf=False):
        "Load in minimizers for ntJoin scaffolding mode"
        if repeat_bf:
            raise NotImplementedError("repeat_bf is never supplied on ntJoin's own path")
        for assembly in self.args.FILES:
            self._add(assembly, float(self.weights_list.pop(0)))

    def load_minimizers_scaffold(self):
        "Load in minimizers for ntJoin scaffolding mode (references, then the target)"
        self.load_minimizers()
        self._add(self.args.s, float(self.args.l))

    # -- graph ----------------------------------------------------------------------------------------------
    def make_minimizer_graph(self, materialize=True):
        "Run ntJoin graph stage"
        print(datetime.datetime.today(), ": Generating minimizer graph ...\n")
        weight_str = "\n".join([f"{assembly}: {asm_weight}" for assembly, asm_weight in self.weights.items()])
        print("\nWeights of assemblies:\n", weight_str, "\n", sep="", flush=True)
        print(datetime.datetime.today(), ": Filtering minimizers", file=sys.stdout)
        print(datetime.datetime.today(), ": Building graph", file=sys.stdout)
        eng = self._engine
        eng.build_graph()
        if not materialize:
            # only the .mx.dot is wanted (ntjoin_amd.run): the library writes it from its own arrays; no Python object
            # per vertex or edge is made (4.5 M vertices at 3 Gbp + 3 Gbp: seconds of str() and dict inserts) unless
            # somebody asks for self.graph afterwards (then: the array-backed container, built on first use)
            self._graph = None
            self._graph_pending = True
            self.print_graph(None)
            return
        g = eng.get_graph()
        if materialize == "views":
            # array-backed state for genome-scale inputs: same objects to index and iterate, no per-minimizer Python work
            self.graph = MxGraph.from_arrays(g["vertex_hash"], g["edge_u"], g["edge_v"], g["edge_support"], g["edge_weight"],
                                             self._order)
            for a, assembly in enumerate(self._order):
                sk = eng.get_sketch(a)
                uniq = (eng.get_mx_flags(a) & capi.MX_UNIQUE) != 0
                self.list_mx_info[assembly], self.list_mxs[assembly] = sketch_views(sk, uniq)
            self.print_graph(self.graph)
            return
        names = [str(h) for h in g["vertex_hash"].tolist()]
        support = [[self._order[b] for b in range(len(self._order)) if m >> b & 1] for m in g["edge_support"].tolist()]
        self.graph = MxGraph(names, zip(g["edge_u"].tolist(), g["edge_v"].tolist()), support,
                             g["edge_weight"].tolist())
        if materialize:
            # the state later stages of the reference read
            for a, assembly in enumerate(self._order):
                sk = eng.get_sketch(a)
                flags = eng.get_mx_flags(a)
                ids = sk["record_ids"]
                uniq = (flags & capi.MX_UNIQUE) != 0
                self.list_mx_info[assembly] = {
                    str(h): (ids[r], int(p)) for h, p, r in
                    zip(sk["out_hash"][uniq].tolist(), sk["pos"][uniq].tolist(), sk["record"][uniq].tolist())}
                first = sk["record_first"]
                lists = []
                for r in range(len(ids)):
                    lo, hi = int(first[r]), int(first[r + 1])
                    if hi > lo:
                        lists.append([str(x) for x in sk["out_hash"][lo:hi][uniq[lo:hi]].tolist()])
                self.list_mxs[assembly] = lists
        self.print_graph(self.graph)

    def find_paths(self):
        "Finds paths through the minimizer graph (global filter with args.n, branch filtering, linear paths)"
        print(datetime.datetime.today(), ": Finding paths", file=sys.stdout)
        found = self._engine.find_paths(int(getattr(self.args, "n", 1)))
        self._found = found
        by_comp = {}
        for comp, verts in found:
            by_comp.setdefault(comp, []).append(([self.graph.names[v] for v in verts], None))
        n_comp = self._engine.n_components
        print("\nTotal number of components in graph:", n_comp, "\n", sep=" ", file=sys.stdout, flush=True)
        # one list per component, as the reference returns (components without an accepted path give [])
        return list(by_comp.values()) + [[] for _ in range(n_comp - len(by_comp))]

    # -- what the scaffolder derives from the paths (reference bin/ntjoin_assemble.py) -----------------------------
    def find_mx_min_max(self, target):
        "Given the target assembly, find the min/max position of its graph-vertex minimizers per contig (:688-702)"
        a = self._order.index(target)
        ids = self._engine.record_ids(a, self._engine.n_records(a))
        return {ids[r]: e for r, e in enumerate(self._engine.mx_extremes(a)) if e is not None}

    @staticmethod
    def determine_orientation(n, inc, dec, m=90):
        "orientation of a run of n minimizers with inc / dec increasing / decreasing consecutive pairs (:30-50, no --mkt)"
        if n > 1:
            if dec == 0 and inc == n - 1:
                return "+"
            if inc == 0 and dec == n - 1:
                return "-"
            positive = inc / float(n - 1) * 100
            if positive >= m:
                return "+"
            if 100 - positive >= m:
                return "-"
        return "?"

    def format_paths(self, lengths=None, g=20, G=0, m=90):
        """format_path (:175-218) for every path of the last find_paths() and the target assembly: one list per path of
        [contig, ori, start, end, contig_size, first_mx, terminal_mx, gap_size, raw_gap_size].  The per-minimizer work
        (grouping by contig, min/max, orientation tallies) is the library's (mxg_path_segments, mxg_mx_extremes); the
        gap estimate between two oriented runs (calculate_gap_size :68-112) reads a handful of graph entries here."""
        eng, k = self._engine, int(getattr(self.args, "k", 32))
        tgt = len(self._order) - 1
        ids = eng.record_ids(tgt, eng.n_records(tgt))
        if lengths is None:
            lens = eng.record_lengths(tgt)
            if ids and not any(lens):
                # a target loaded from its TSV carries no record lengths (the reference takes them from the FASTA index,
                # calc_end_coord / scaffolds[ctg].length): silently using 0 would put every contig end at 0
                raise ValueError("format_paths: the target was loaded from a minimizer TSV, which holds no contig lengths; "
                                 "pass lengths={contig: length}")
            lengths = dict(zip(ids, lens))
        ext = eng.mx_extremes(tgt)
        seg = eng.path_segments(tgt)
        gr = eng.get_graph()
        vpos, names = gr["vertex_pos"], self.graph.names
        masks = {(min(u, v), max(u, v)): int(s) for u, v, s in
                 zip(gr["edge_u"].tolist(), gr["edge_v"].tolist(), gr["edge_support"].tolist())}
        offsets, at = [], 0
        for _comp, verts in self._found:
            offsets.append(at)
            at += len(verts)
        out = [[] for _ in self._found]
        kept = [[] for _ in self._found]
        cols = [seg[c].tolist() for c in ("path", "record", "first", "n", "min_pos", "max_pos", "inc", "dec")]
        for p, rec, first, n, mn, mx, inc, dec in zip(*cols):
            ori = self.determine_orientation(n, inc, dec, m)
            if ori == "?":
                continue
            ctg, verts, lo = ids[rec], self._found[p][1], first - offsets[p]
            start = 0 if mn == ext[rec][0] else mn
            end = lengths[ctg] if mx == ext[rec][1] else mx + k
            out[p].append([ctg, ori, start, end, lengths[ctg], names[verts[lo]], names[verts[lo + n - 1]], 0, 0])
            kept[p].append((lo, lo + n - 1))
        for p, nodes in enumerate(out):
            verts = self._found[p][1]
            for i in range(len(nodes) - 1):
                u, v = nodes[i], nodes[i + 1]
                iu, iv = kept[p][i][1], kept[p][i + 1][0]
                common = -1
                for a_, b_ in zip(verts[iu:iv], verts[iu + 1:iv + 1]):
                    common &= masks[(min(a_, b_), max(a_, b_))]
                sup = [b for b in range(len(self._order)) if common >> b & 1]
                if not sup:
                    u[7], u[8] = g, g
                    continue
                um, vm = verts[iu], verts[iv]
                dists = [abs(int(vpos[b][vm]) - int(vpos[b][um])) for b in sup]
                mean_dist = int(sum(dists) / len(dists)) - k
                upos, vpos_t = int(vpos[tgt][um]), int(vpos[tgt][vm])
                a_over = (u[3] - upos - k) if u[1] == "+" else (upos - u[2])
                b_over = (vpos_t - v[2]) if v[1] == "+" else (v[3] - vpos_t - k)
                if a_over < 0 or b_over < 0:
                    raise ValueError(f"Gap distance estimation less than 0 between {u} and {v}")
                gap = max(mean_dist - a_over - b_over, g)
                if G > 0:
                    gap = min(gap, G)
                u[7], u[8] = gap, mean_dist - a_over - b_over
        return out

    def print_graph(self, graph, out_prefix=None):
        "Prints the minimizer graph in dot format"
        out_graph = (self.args.p + ".mx.dot") if out_prefix is None else (out_prefix + "mx.dot")
        print(datetime.datetime.today(), ": Printing graph", out_graph, sep=" ", file=sys.stdout)
        self._engine.write_dot(out_graph)
        list_files = list(self._order)
        colours = COLOURS if len(list_files) <= len(COLOURS) else ["red"] * len(list_files)
        print("\nfile_name\tnumber\tcolour")
        for i, filename in enumerate(list_files):
            print(filename, i, colours[i], sep="\t")
        print("", flush=True)
