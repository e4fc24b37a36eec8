"""CPU tests: oracle/paths_oracle.py against the reference's own find_paths() output (tests/golden, paths_by_n)."""
import os

import pytest

from oracle import graph_oracle as go
from oracle import paths_oracle as po
from tests import _path_literals
from tests.conftest import GOLDEN, golden_cases, load_case

CASES = [m["name"] for m in golden_cases()]


def _state(meta):
    cwd = os.getcwd()
    os.chdir(os.path.join(GOLDEN, "cases", meta["name"]))
    try:
        return go.load_and_build([r["tsv"] for r in meta["refs"]], [r["weight"] for r in meta["refs"]],
                                 meta["target"]["tsv"], meta["target"]["weight"])
    finally:
        os.chdir(cwd)


@pytest.mark.parametrize("name", CASES)
def test_paths_oracle_matches_reference(name):
    case = load_case(name)
    state = _state(case["meta"])
    assert case["reference"]["paths_by_n"], "golden lacks paths_by_n: rerun tests/golden/make_golden.py"
    for n, ref_paths in case["reference"]["paths_by_n"].items():
        assert po.canonical(po.find_paths(state, int(n))) == po.canonical(ref_paths), (name, n)


def test_reference_expectation_f_f():
    """reference tests/ntjoin_test.py:85 `1_f+:0-1981 20N 2_f+:0-2329`: one path through all five minimizers, ordered as
    the reference assembly (positions increase along the path in the highest-weight assembly)"""
    case = load_case("f-f_w1000")
    state = _state(case["meta"])
    paths = [p for comp in po.find_paths(state, 2) for p in comp]
    assert len(paths) == 1 and len(paths[0]) == 5
    ref_asm = case["meta"]["refs"][0]["tsv"]
    pos = [state["list_mx_info"][ref_asm][v][1] for v in paths[0]]
    assert pos == sorted(pos)


def _fasta_lengths(path):
    lens, rid = {}, None
    for line in open(path, encoding="ascii"):
        if line.startswith(">"):
            rid = line[1:].split()[0]
            lens[rid] = 0
        elif rid is not None:
            lens[rid] += len(line.strip())
    return lens


@pytest.mark.parametrize("name", CASES)
def test_format_path_oracle_matches_reference(name):
    """row f4: find_mx_min_max and format_path of the reference's scaffolder (goldens: mx_extremes_by_n, format_by_n)"""
    case = load_case(name)
    meta, ref = case["meta"], case["reference"]
    state = _state(meta)
    target = meta["target"]["tsv"]
    lengths = _fasta_lengths(os.path.join(GOLDEN, "fasta", meta["target"]["fasta"]))
    fa = ref["format_args"]
    for n, want in ref["format_by_n"].items():
        filt = dict(state)
        if not int(n) <= min(state["weights"].values()):      # the graph the scaffolder holds is the globally filtered one
            filt["edges"] = [e for e in state["edges"] if not e[3] < int(n)]
        assert {c: list(v) for c, v in po.mx_extremes(filt, target).items()} == ref["mx_extremes_by_n"][n], (name, n)
        got = [po.format_path(filt, p, target, lengths, meta["k"], fa["g"], fa["G"], fa["m"])
               for comp in po.find_paths(state, int(n)) for p in comp]
        key = lambda nodes: tuple(tuple(x) for x in nodes)
        assert sorted(map(key, got)) == sorted(map(key, want)), (name, n)


@pytest.mark.parametrize("name", sorted(_path_literals.EXPECTED))
def test_path_strings_the_reference_tests_assert(name):
    """the literals of reference tests/ntjoin_test.py (contig, orientation, cut coordinates, gap sizes) from the oracle's
    find_paths + format_path, and from what the reference itself returned for the case (goldens)"""
    n, expected = _path_literals.EXPECTED[name]
    case = load_case(name)
    meta, ref = case["meta"], case["reference"]
    assert {_path_literals.path_string(nodes) for nodes in ref["format_by_n"][str(n)]} == expected
    state = _state(meta)
    target = meta["target"]["tsv"]
    lengths = _fasta_lengths(os.path.join(GOLDEN, "fasta", meta["target"]["fasta"]))
    fa = ref["format_args"]
    filt = dict(state)
    if not n <= min(state["weights"].values()):
        filt["edges"] = [e for e in state["edges"] if not e[3] < n]
    got = {_path_literals.path_string(po.format_path(filt, p, target, lengths, meta["k"], fa["g"], fa["G"], fa["m"]))
           for comp in po.find_paths(state, n) for p in comp}
    assert got == expected
