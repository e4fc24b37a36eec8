"""CPU test: ntjoin_amd/run.py parses the reference's command line (reference bin/ntjoin_run.py:10-53) with the reference's
defaults -- the table below is typed in from that file (flag -> default when the flag is absent)."""
from ntjoin_amd import run

REFERENCE_DEFAULTS = {  # reference bin/ntjoin_run.py:15-53
    "l": 1, "p": "out", "n": 1, "g": 20, "G": 0, "mkt": False, "m": 90, "t": 1, "agp": False, "no_cut": False,
    "overlap": False, "overlap_gap": 20, "overlap_k": 15, "overlap_w": 10, "btllib_t": 4,
}


def test_defaults_equal_the_reference_parser():
    args = run.parse_arguments(["ref.tsv", "-s", "tgt.tsv", "-r", "2", "-k", "32"])
    for flag, want in REFERENCE_DEFAULTS.items():
        assert getattr(args, flag) == want, flag
    assert args.FILES == ["ref.tsv"] and args.s == "tgt.tsv" and args.r == "2" and args.k == 32


def test_every_flag_of_the_reference_recipe_parses():
    # the recipe of reference ntJoin:228-230 with every optional flag set
    argv = ["-s", "t.tsv", "-l", "1", "-r", "2 1", "-k", "32", "-p", "pre", "-n", "2", "-g", "20", "-G", "100", "-m", "80", "-t", "4",
            "--mkt", "--agp", "--no_cut", "--overlap", "--overlap_gap", "10", "--overlap_k", "15", "--overlap_w", "10",
            "--btllib_t", "2", "a.tsv", "b.tsv"]
    args = run.parse_arguments(argv)
    assert run.set_weights(args) == [2.0, 1.0]
    assert (args.m, args.G, args.mkt, args.no_cut, args.btllib_t) == (80, 100, True, True, 2)
