"""GPU test of the binary sketch side-car: a sketch written with mxg_write_sketch_bin and read back with
mxg_add_assembly_bin gives the same minimizer lists, flags, graph and .mx.dot as the TSV route the reference takes
(read_minimizers, bin/ntjoin_utils.py:167-193)."""
import os

import numpy as np
import pytest

from oracle import graph_oracle as go
from tests.conftest import GOLDEN, golden_cases, load_case

pytestmark = pytest.mark.gpu
FASTA = os.path.join(GOLDEN, "fasta")
CASES = [m["name"] for m in golden_cases()][:8]


@pytest.mark.parametrize("name", CASES)
def test_sidecar_round_trip_equals_tsv_route(name, tmp_path):
    from ntjoin_amd.engine import MxEngine
    meta = load_case(name)["meta"]
    asms = meta["refs"] + [meta["target"]]
    cdir = os.path.join(GOLDEN, "cases", name)
    with MxEngine(k=meta["k"], w=meta["w"], variant=meta["variant"]) as eng:
        for a in asms:
            eng.add_fasta(a["tsv"], a["weight"], os.path.join(FASTA, a["fasta"]))
        eng.sketch()
        for i, a in enumerate(asms):
            eng.write_sketch_bin(i, tmp_path / (a["tsv"] + ".bin"))
    with MxEngine(k=meta["k"], w=1, variant=meta["variant"]) as via_bin, MxEngine(k=meta["k"], w=1, variant=meta["variant"]) as via_tsv:
        for a in asms:
            via_bin.add_bin(a["tsv"], a["weight"], tmp_path / (a["tsv"] + ".bin"))
            via_tsv.add_tsv(a["tsv"], a["weight"], os.path.join(cdir, a["tsv"]))
        for i in range(len(asms)):
            s1, s2 = via_bin.get_sketch(i), via_tsv.get_sketch(i)
            assert s1["record_ids"] == s2["record_ids"]
            for key in ("out_hash", "pos", "record", "record_first"):
                assert np.array_equal(s1[key], s2[key]), (i, key)
        via_bin.build_graph()
        via_tsv.build_graph()
        g1, g2 = via_bin.get_graph(), via_tsv.get_graph()
        for key in g1:
            assert np.array_equal(np.asarray(g1[key]), np.asarray(g2[key])), key
        via_bin.write_dot(tmp_path / "b.mx.dot")
    with open(os.path.join(cdir, "reference.mx.dot"), encoding="utf-8") as fh:
        want = go.canonical_dot_from_text(fh.read())
    assert go.canonical_dot_from_text((tmp_path / "b.mx.dot").read_text(encoding="utf-8")) == want


def test_sidecar_rejects_other_k_and_garbage(tmp_path):
    from ntjoin_amd.engine import MxEngine, MxError
    with MxEngine(k=32, w=100) as eng:
        eng.add_records("x", 1.0, [("r", "ACGTTGCA" * 200)])
        eng.sketch()
        eng.write_sketch_bin(0, tmp_path / "x.bin")
    with MxEngine(k=31, w=100) as eng:
        with pytest.raises(MxError):
            eng.add_bin("x", 1.0, tmp_path / "x.bin")
    (tmp_path / "junk.bin").write_bytes(b"not a sketch")
    with MxEngine(k=32, w=100) as eng:
        with pytest.raises(MxError):
            eng.add_bin("x", 1.0, tmp_path / "junk.bin")
        with pytest.raises(FileNotFoundError):
            eng.add_bin("x", 1.0, tmp_path / "missing.bin")
        assert eng.n_assemblies == 0   # failed adds leave nothing behind
