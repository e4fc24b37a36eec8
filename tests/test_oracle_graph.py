"""CPU tests: the pure-Python graph oracle against what the reference's own functions returned
(tests/golden/cases/*/reference.json, produced by tests/golden/make_golden.py)."""
import os
import re

import pytest

from oracle import graph_oracle as go
from tests.conftest import GOLDEN, golden_cases, load_case

CASES = [m["name"] for m in golden_cases()]


def _run(meta):
    cdir = os.path.join(GOLDEN, "cases", meta["name"])
    cwd = os.getcwd()
    os.chdir(cdir)  # assembly names are the bare TSV file names, as in the reference's tests
    try:
        return go.load_and_build([r["tsv"] for r in meta["refs"]], [r["weight"] for r in meta["refs"]],
                                 meta["target"]["tsv"], meta["target"]["weight"])
    finally:
        os.chdir(cwd)


@pytest.mark.parametrize("name", CASES)
def test_graph_oracle_matches_reference(name):
    case = load_case(name)
    ref, state = case["reference"], _run(case["meta"])
    assert list(state["list_mx_info"].keys()) == ref["assemblies"]
    assert state["weights"] == ref["weights"]
    for a in ref["assemblies"]:
        assert {mx: list(v) for mx, v in state["list_mx_info"][a].items()} == ref["mx_info"][a]
        assert state["list_mxs"][a] == ref["mxs"][a]
        assert state["filtered"][a] == ref["filtered"][a]
    assert sorted(state["vertices"], key=int) == ref["vertices"]
    # edges as unordered pairs; support order and float weight must match exactly
    mine = {frozenset((s, t)): (sup, w) for s, t, sup, w in state["edges"]}
    theirs = {frozenset((s, t)): (sup, w) for s, t, sup, w in ref["edges"]}
    assert mine == theirs
    for (sup, w) in mine.values():
        assert isinstance(w, float)


@pytest.mark.parametrize("name", CASES)
def test_dot_canonical_matches_reference(name):
    case = load_case(name)
    state = _run(case["meta"])
    with open(os.path.join(GOLDEN, "cases", name, "reference.mx.dot"), encoding="utf-8") as fh:
        theirs = go.canonical_dot_from_text(fh.read())
    assert go.canonical_dot_from_state(state) == theirs


def test_stale_golden_dot_edges():
    """reference tests/expected_outputs/f-f_test.mx.dot (older label/edge syntax, variant V1): vertex set,
    edge set, weights 3.0/3.0/2.0/3.0/3.0 and colours must still agree with the V1 case."""
    state = _run(load_case("f-f_w1000_v1")["meta"])
    text = open(os.path.join(GOLDEN, "reference_expected_outputs", "f-f_test.mx.dot"), encoding="utf-8").read()
    edges = sorted((min(int(a), int(b)), max(int(a), int(b)), attr)
                   for a, b, attr in re.findall(r'^"(\d+)" -- "(\d+)" (\[.*\])$', text, flags=re.M))
    assert [list(map(str, e[:2])) + [e[2]] for e in edges] == go.canonical_dot_from_state(state)["edges"]
    names = set(re.findall(r'^"(\d+)" \[label=', text, flags=re.M))
    assert names == set(state["vertices"])
    assert "1177713728801312737" not in names  # unique to ref.fa: dropped by the intersection
    # labels: same (contig, pos) tuples, minus the `<file>_` prefix HEAD adds
    for name in names:
        m = re.search(r'^"%s" \[label="%s\n(.*?)"\]' % (name, name), text, flags=re.M | re.S)
        tuples = m.group(1).split("\n")
        assert tuples == [str(info[name]) for info in state["list_mx_info"].values()]
