#!/usr/bin/env python3
"""
make_golden.py -- regenerate tests/golden/ in the BUILD container (needs /root/reference).

What it does, per case:
  1. copies the reference's test FASTA inputs (data files) into tests/golden/fasta/ ;
  2. sketches them with the C oracle (oracle/_build/mx_oracle, `--pos --seq`) into
     tests/golden/cases/<case>/<fasta>.k<k>.w<w>.tsv ;
  3. imports the reference's OWN bin/ntjoin_utils.py and bin/ntjoin.py from /root/reference/bin (with a
     small stand-in for the python-igraph container, which is not installed here), runs
     read_minimizers -> filter_minimizers -> build_graph -> print_graph on those TSVs in the
     reference's call order (bin/ntjoin.py:178-204, bin/ntjoin_assemble.py:799-807), and dumps what they
     return to tests/golden/cases/<case>/reference.json and reference.mx.dot .

The reference's source never enters this repository: only its outputs do.  Nothing in tests/, bench.py
or the package reads /root/reference at run time; this script is the only place that does.
"""
import contextlib
import io
import json
import os
import shutil
import subprocess
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
ORACLE_BIN = os.path.join(REPO, "oracle", "_build", "mx_oracle")

# (case name, k, w, variant, [(ref fasta, weight)...], (target fasta, weight))
CASES = [
    ("f-f_w1000", 32, 1000, "v2", [("ref.fa", 2)], ("scaf.f-f.fa", 1)),
    ("f-f_w1000_v1", 32, 1000, "v1", [("ref.fa", 2)], ("scaf.f-f.fa", 1)),
    ("f-f_w100_config1", 32, 100, "v2", [("ref.fa", 2)], ("scaf.f-f.fa", 1)),
    ("f-f_termN_w1000", 32, 1000, "v2", [("ref.fa", 2)], ("scaf.f-f.termN.fa", 1)),
    ("f-f_termN_unassigned_w1000", 32, 1000, "v2", [("ref.fa", 2)], ("scaf.f-f.termN.unassigned.fa", 1)),
    ("f-r_w1000", 32, 1000, "v2", [("ref.fa", 2)], ("scaf.f-r.fa", 1)),
    ("r-f_w1000", 32, 1000, "v2", [("ref.fa", 2)], ("scaf.r-f.fa", 1)),
    ("r-r_w1000", 32, 1000, "v2", [("ref.fa", 2)], ("scaf.r-r.fa", 1)),
    ("gap-dist_w500", 32, 500, "v2", [("ref.multiple.fa", 2)], ("scaf.multiple.fa", 1)),
    ("regions-ff-rr_w500", 32, 500, "v2", [("ref.multiple.fa", 2)], ("scaf.misassembled.f-f.r-r.fa", 1)),
    ("regions-fr-rf_w500", 32, 500, "v2", [("ref.multiple.fa", 2)], ("scaf.misassembled.f-r.r-f.fa", 1)),
    ("f-f-f_w1000", 32, 1000, "v2", [("ref.fa", 2), ("scaf.f-f.copy.fa", 2)], ("scaf.f-f.fa", 1)),
    ("f-f-f_w100_weights", 32, 100, "v2", [("ref.fa", 0.1), ("scaf.f-f.copy.fa", 0.2)], ("scaf.f-f.fa", 1.5)),
    ("overlap_k15_w10", 15, 10, "v2", [("ref.fa", 1)], ("scaf.f-f.overlapping.fa", 1)),
    ("more_seqs_pieces_w500", 32, 500, "v2", [("scaf.more_seqs.fa", 2)], ("more_seqs.pieces.fa", 1)),
    ("synth_w100", 32, 100, "v2", [("synth.ref.fa", 2)], ("synth.tgt.fa", 1)),
    ("synth_w1000", 32, 1000, "v2", [("synth.ref.fa", 2)], ("synth.tgt.fa", 1)),
    ("synth3_w50", 32, 50, "v2", [("synth.ref.fa", 2), ("synth.ref2.fa", 1.5)], ("synth.tgt.fa", 1)),
    ("synth_k15_w60", 15, 60, "v2", [("synth.ref.fa", 1)], ("synth.tgt.fa", 1)),
    ("synth_v1_w200", 32, 200, "v1", [("synth.ref2.fa", 3)], ("synth.ref.fa", 1)),
]


def install_igraph_standin():
    """python-igraph is not installed here: tests/golden/igraph_standin.py provides the container API the reference
    uses on this path (and on the path-extraction step that follows it)."""
    sys.path.insert(0, HERE)
    import igraph_standin
    igraph_standin.install()


def fasta_lengths(path):
    lens, rid = {}, None
    with open(path, encoding="ascii") as fh:
        for line in fh:
            if line.startswith(">"):
                rid = line[1:].split()[0]
                lens[rid] = 0
            elif rid is not None:
                lens[rid] += len(line.strip())
    return lens


def import_scaffolder():
    """bin/ntjoin_assemble.py imports pybedtools, pymannkendall and btllib at module level; none of them is installed
    here and none is touched by the methods called below (format_path, find_mx_min_max, determine_orientation without
    --mkt, calc_*_coord, calculate_gap_size), so empty modules stand in for the three names."""
    for missing in ("pybedtools", "pymannkendall", "btllib"):
        sys.modules.setdefault(missing, types.ModuleType(missing))
    import ntjoin_assemble  # noqa: the reference's own module
    return ntjoin_assemble


def run_reference(case_dir, ref_tsvs, ref_weights, target_tsv, target_weight, prefix, k=32, target_fasta=None):
    sys.path.insert(0, os.path.join(REF, "bin"))
    import ntjoin_utils  # noqa: the reference's own module
    import ntjoin        # noqa: the reference's own module
    cwd = os.getcwd()
    os.chdir(case_dir)
    try:
        args = types.SimpleNamespace(p=prefix, FILES=list(ref_tsvs), t=1, s=target_tsv, l=float(target_weight))
        nj = ntjoin.Ntjoin(args)
        nj.weights_list = [float(x) for x in ref_weights]            # ntjoin_assemble.py:788-797,811
        with contextlib.redirect_stdout(io.StringIO()):
            nj.load_minimizers()                                     # ntjoin.py:178-186
            # the four lines of ntjoin_assemble.py:803-807 (that module is not importable here:
            # it needs pybedtools / pymannkendall / btllib)
            info, mxs = ntjoin_utils.read_minimizers(args.s)
            nj.list_mx_info[args.s] = info
            nj.list_mxs[args.s] = mxs
            nj.weights[args.s] = args.l
            filtered = ntjoin_utils.filter_minimizers(nj.list_mxs)   # ntjoin.py:198
            nj.make_minimizer_graph()                                # ntjoin.py:189-204 (writes <prefix>.mx.dot)
        g = nj.graph
        # next row: the reference's own global filter + path extraction, for several -n values
        # (reference bin/ntjoin.py:80-89,137-176; called at bin/ntjoin_assemble.py:759,779)
        paths_by_n = {}
        # row f4: what the reference's scaffolder derives from each path for the TARGET assembly
        # (bin/ntjoin_assemble.py:688-702 find_mx_min_max, :175-218 format_path incl. determine_orientation :30-50,
        #  calc_start/end_coord :52-65, calculate_gap_size :68-120), with its default -g 20 -G 0 -m 90 and no --mkt
        asm_mod = import_scaffolder()
        sc = object.__new__(asm_mod.NtjoinScaffolder)
        sc.args = types.SimpleNamespace(k=k, g=20, G=0, m=90, mkt=False, s=args.s)
        sc.list_mx_info = nj.list_mx_info
        lens = fasta_lengths(target_fasta)
        sc.scaffolds = {c: ntjoin_utils.Scaffold(id=c, length=n, sequence="") for c, n in lens.items()}
        format_by_n, extremes_by_n = {}, {}
        total_w = sum(nj.weights.values())
        for n_min in sorted({1, 2, 3, int(total_w), int(total_w) + 1}):
            nj.args.n = n_min
            nj.graph = nj.filter_graph_global(g.copy())
            with contextlib.redirect_stdout(io.StringIO()):
                found = nj.find_paths()
            paths_by_n[str(n_min)] = [[list(path) for path, _sub in comp] for comp in found]
            sc.graph = nj.graph
            sc.mx_extremes = sc.find_mx_min_max(args.s)
            extremes_by_n[str(n_min)] = {c: list(v) for c, v in sc.mx_extremes.items()}
            formatted = []
            for comp in found:
                for path, sub in comp:
                    nodes = sc.format_path(path, args.s, sub)
                    formatted.append([[nd.contig, nd.ori, nd.start, nd.end, nd.contig_size, nd.first_mx, nd.terminal_mx,
                                       nd.gap_size, nd.raw_gap_size] for nd in nodes])
            format_by_n[str(n_min)] = formatted
        nj.graph = g
        names = [v["name"] for v in g.vs()]
        edges = []
        for e in g.es():
            edges.append([names[e.source], names[e.target], list(e["support"]), e["weight"]])
        return {
            "assemblies": list(nj.list_mx_info.keys()),
            "weights": nj.weights,
            "mx_info": {a: {mx: [c, p] for mx, (c, p) in d.items()} for a, d in nj.list_mx_info.items()},
            "mxs": nj.list_mxs,
            "filtered": filtered,
            "vertices": sorted(names, key=int),
            "edges": edges,
            "paths_by_n": paths_by_n,
            "mx_extremes_by_n": extremes_by_n,
            "format_by_n": format_by_n,
            "format_args": {"g": 20, "G": 0, "m": 90, "mkt": False},
        }
    finally:
        os.chdir(cwd)


def main():
    if not os.path.isdir(REF):
        sys.exit("make_golden.py needs /root/reference (build container only)")
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle")])
    if os.environ.get("MXG_GOLDEN_REAL_IGRAPH") == "1":  # a box that has python-igraph: the real container (tests/golden/real_igraph/README.md)
        import igraph  # noqa: F401
        print("make_golden.py: python-igraph", igraph.__version__, "is the reference's container", file=sys.stderr)
    else:
        install_igraph_standin()
    fasta_dir = os.path.join(HERE, "fasta")
    os.makedirs(fasta_dir, exist_ok=True)
    # the reference's own expected outputs for this path (data files its tests hold)
    exp_dir = os.path.join(HERE, "reference_expected_outputs")
    os.makedirs(exp_dir, exist_ok=True)
    for f in ("ref.fa.k32.w1000.tsv", "scaf.f-f.fa.k32.w1000.tsv", "f-f_test.mx.dot"):
        shutil.copyfile(os.path.join(REF, "tests", "expected_outputs", f), os.path.join(exp_dir, f))
    # synthetic fixtures (ours) need scaf.more_seqs.fa in place first
    src = os.path.join(fasta_dir, "scaf.more_seqs.fa")
    if not os.path.exists(src):
        shutil.copyfile(os.path.join(REF, "tests", "scaf.more_seqs.fa"), src)
    subprocess.check_call([sys.executable, os.path.join(HERE, "make_synth_fasta.py")])
    index = []
    for name, k, w, variant, refs, target in CASES:
        case_dir = os.path.join(HERE, "cases", name)
        shutil.rmtree(case_dir, ignore_errors=True)
        os.makedirs(case_dir)
        tsvs = []
        for fa, _ in refs + [target]:
            dst = os.path.join(fasta_dir, fa)
            if not os.path.exists(dst):
                shutil.copyfile(os.path.join(REF, "tests", fa), dst)  # a data file the reference's tests hold
            tsv = f"{fa}.k{k}.w{w}.tsv"
            subprocess.check_call([ORACLE_BIN, "-k", str(k), "-w", str(w), "--variant", variant,
                                   "--pos", "--seq", "-o", os.path.join(case_dir, tsv), dst])
            tsvs.append(tsv)
        prefix = "out"
        result = run_reference(case_dir, tsvs[:-1], [wt for _, wt in refs], tsvs[-1], target[1], prefix, k=k,
                               target_fasta=os.path.join(fasta_dir, target[0]))
        os.replace(os.path.join(case_dir, prefix + ".mx.dot"), os.path.join(case_dir, "reference.mx.dot"))
        meta = {"name": name, "k": k, "w": w, "variant": variant,
                "refs": [{"fasta": fa, "weight": wt, "tsv": t} for (fa, wt), t in zip(refs, tsvs[:-1])],
                "target": {"fasta": target[0], "weight": target[1], "tsv": tsvs[-1]}}
        with open(os.path.join(case_dir, "reference.json"), "w", encoding="utf-8") as fh:
            json.dump({"meta": meta, "reference": result}, fh, indent=0, sort_keys=True)
        index.append(meta)
        print(f"{name}: |V|={len(result['vertices'])} |E|={len(result['edges'])}")
    with open(os.path.join(HERE, "cases", "index.json"), "w", encoding="utf-8") as fh:
        json.dump(index, fh, indent=1)


if __name__ == "__main__":
    main()
