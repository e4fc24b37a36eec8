"""CPU tests: oracle/paths_oracle.py against the reference's own find_paths() output (tests/golden, paths_by_n)."""
import os

import pytest

from oracle import graph_oracle as go
from oracle import paths_oracle as po
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
