#!/usr/bin/env python3
"""
Counterpart of the reference's bin/ntjoin_run.py (argument surface: reference bin/ntjoin_run.py:10-60) for the
hot path only: load the minimizer TSVs (references in CLI order, then the target -s), run the graph stage on the
GPU and write <prefix>.mx.dot, exactly the state the reference has after `make_minimizer_graph()`
(reference bin/ntjoin_assemble.py:751-757).  Everything downstream (path finding, scaffolding, AGP) stays the
reference's Python and is out of scope here; flags that only matter downstream are accepted and ignored.
"""
import argparse
import re
import sys

from .ntjoin import Ntjoin


def parse_arguments(argv=None):
    parser = argparse.ArgumentParser(description="ntJoin hot path on MI355X: minimizer TSVs -> <prefix>.mx.dot")
    parser.add_argument("FILES", nargs="+", help="Minimizer TSV files of references")
    parser.add_argument("-s", help="Target scaffolds minimizer TSV file", required=True)
    parser.add_argument("-l", help="Weight of target genome assembly [1]", required=False, default=1, type=float)
    parser.add_argument("-r", help="List of reference assembly weights (in quotes, separated by spaces, "
                                   "in same order as minimizer TSV files)", required=True, type=str)
    parser.add_argument("-p", help="Output prefix [out]", default="out", type=str, required=False)
    parser.add_argument("-n", help="Minimum edge weight [1]", default=1, type=int)
    parser.add_argument("-k", help="Kmer size used for minimizer step", required=True, type=int)
    parser.add_argument("-g", help="Minimum gap size (bp)", required=False, default=20, type=int)
    parser.add_argument("-G", help="Maximum gap size (bp) (0 if no maximum threshold)", required=False, default=0, type=int)
    parser.add_argument("--mkt", action="store_true")
    parser.add_argument("-m", type=int, default=50, required=False)
    parser.add_argument("-t", type=int, default=1)
    parser.add_argument("--agp", action="store_true")
    parser.add_argument("--no_cut", action="store_true")
    parser.add_argument("--overlap", action="store_true")
    parser.add_argument("--overlap_gap", type=int, default=20)
    parser.add_argument("--overlap_k", type=int, default=15)
    parser.add_argument("--overlap_w", type=int, default=10)
    parser.add_argument("--btllib_t", type=int, default=4)
    parser.add_argument("-v", "--version", action="version", version="ntjoin_amd hot path (ntJoin v1.1.5 compatible)")
    return parser.parse_args(argv)


def set_weights(args):
    "Parse the supplied weights (reference bin/ntjoin_assemble.py:788-797)"
    weights = [float(w) for w in re.split(r"\s+", args.r.strip())]
    if len(weights) != len(args.FILES):
        print("ERROR: The length of supplied reference weights (-r) and "
              "number of assembly minimizer TSV inputs must be equal.")
        print("Supplied lengths of arguments:")
        print("Weights (-r):", len(weights), "Minimizer TSV files:", len(args.FILES), sep=" ")
        sys.exit(1)
    return weights


def main(argv=None):
    args = parse_arguments(argv)
    nj = Ntjoin(args)
    nj.weights_list = set_weights(args)
    try:
        nj.load_minimizers_scaffold()
        nj.make_minimizer_graph(materialize=False)
    finally:
        nj.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
