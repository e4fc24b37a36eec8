#!/usr/bin/env python3
"""
Counterpart of the reference's bin/ntjoin_run.py (argument surface: reference bin/ntjoin_run.py:10-60) for the
hot path only: load the minimizer TSVs (references in CLI order, then the target -s), run the graph stage on the
GPU and write <prefix>.mx.dot, exactly the state the reference has after `make_minimizer_graph()`
(reference bin/ntjoin_assemble.py:751-757).  Everything downstream (path finding, scaffolding, AGP) stays the
reference's Python and is out of scope here; flags that only matter downstream are accepted and ignored.
"""
import argparse
import sys

from .ntjoin import Ntjoin


def parse_arguments(argv=None):
    """same flag names, defaults and required-ness as the reference's parser (bin/ntjoin_run.py:10-60), so that its Makefile
    recipe runs unchanged; the descriptions are ours"""
    parser = argparse.ArgumentParser(description="ntJoin hot path on MI355X: minimizer TSVs -> <prefix>.mx.dot")
    parser.add_argument("FILES", nargs="+", help="sketches of the reference assemblies (indexlr TSVs), in the order their weights are given")
    parser.add_argument("-s", required=True, help="sketch (indexlr TSV) of the assembly to be scaffolded")
    parser.add_argument("-l", required=False, default=1, type=float, help="edge weight contributed by the target assembly (default 1)")
    parser.add_argument("-r", required=True, type=str, help="one edge weight per reference, blank-separated inside one quoted argument")
    parser.add_argument("-p", default="out", type=str, required=False, help="prefix of the files written (default: out)")
    parser.add_argument("-n", default=1, type=int, help="edges lighter than this are dropped downstream (default 1; not used by the graph build)")
    parser.add_argument("-k", required=True, type=int, help="the k the sketches were computed with")
    parser.add_argument("-g", required=False, default=20, type=int, help="downstream only: smallest gap written between joined pieces")
    parser.add_argument("-G", required=False, default=0, type=int, help="downstream only: largest gap allowed, 0 = unlimited")
    # flags of the stages behind the graph build: accepted so that the reference's command line parses, otherwise unused here
    parser.add_argument("--mkt", action="store_true")
    parser.add_argument("-m", type=int, default=90, required=False)
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
    """-r as a list of floats, one per reference TSV; a count mismatch ends the run with status 1, as in the reference
    (bin/ntjoin_assemble.py:788-797: same condition, same exit status)"""
    weights = [float(tok) for tok in args.r.split()]
    if len(weights) == len(args.FILES):
        return weights
    sys.stdout.write(f"ERROR: -r lists {len(weights)} weight(s) but {len(args.FILES)} reference sketch file(s) were given; "
                     "there must be exactly one weight per reference, in the same order.\n")
    sys.exit(1)


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
