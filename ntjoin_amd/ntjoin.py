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
from .ntjoin_utils import MxGraph

COLOURS = ["red", "green", "blue", "purple", "orange", "turquoise", "pink", "yellow", "orchid", "salmon"]


class Ntjoin:
    "ntJoin hot path: minimizer sketches -> minimizer graph, on the GPU"

    def __init__(self, args, fasta=None, w=None, variant="v2"):
        self.list_mx_info = {}  # assembly -> {mx: (contig, position)}
        self.list_mxs = {}      # assembly -> [lists of mx]
        self.graph = None
        self.args = args
        self.weights = {}
        self.weights_list = []
        self._fasta = dict(fasta or {})
        self._engine = MxEngine(k=int(getattr(args, "k", 32)), w=int(w if w is not None else 1), variant=variant)
        self._order = []

    def close(self):
        self._engine.close()

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
        g = eng.get_graph()
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
        by_comp = {}
        for comp, verts in found:
            by_comp.setdefault(comp, []).append(([self.graph.names[v] for v in verts], None))
        n_comp = self._engine.n_components
        print("\nTotal number of components in graph:", n_comp, "\n", sep=" ", file=sys.stdout, flush=True)
        # one list per component, as the reference returns (components without an accepted path give [])
        return list(by_comp.values()) + [[] for _ in range(n_comp - len(by_comp))]

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
