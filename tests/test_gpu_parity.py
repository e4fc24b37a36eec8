"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C-ABI, against
(1) the committed golden fixtures produced by the reference's own Python, and (2) the CPU oracle on
seeded inputs.  Bit-exact: integer hashes / positions / edge lists, repr-exact float weights."""
import filecmp
import os
import random

import numpy as np
import pytest

from oracle import graph_oracle as go
from tests import _oracle
from tests.conftest import GOLDEN, golden_cases, load_case

pytestmark = pytest.mark.gpu
FASTA = os.path.join(GOLDEN, "fasta")
CASES = [m["name"] for m in golden_cases()]


def _engine(**kw):
    from ntjoin_amd.engine import MxEngine
    return MxEngine(**kw)


@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("name", CASES)
def test_golden_case(name, dense, tmp_path):
    case = load_case(name)
    meta, ref = case["meta"], case["reference"]
    asms = meta["refs"] + [meta["target"]]
    cdir = os.path.join(GOLDEN, "cases", name)
    with _engine(k=meta["k"], w=meta["w"], variant=meta["variant"], dense_only=dense) as eng:
        for a in asms:
            eng.add_fasta(a["tsv"], a["weight"], os.path.join(FASTA, a["fasta"]))
        eng.sketch()
        # (1) TSV byte-identical to the committed sketch (oracle output, accepted by the reference parser)
        for i, a in enumerate(asms):
            out = tmp_path / a["tsv"]
            eng.write_tsv(i, str(out), with_pos=True, with_strand=False, with_seq=True)
            assert filecmp.cmp(str(out), os.path.join(cdir, a["tsv"]), shallow=False), a["tsv"]
        eng.build_graph()
        # (2) uniqueness / intersection flags vs the reference's mx_info and filtered lists
        for i, a in enumerate(asms):
            sk = eng.get_sketch(i)
            flags = eng.get_mx_flags(i)
            uniq = {str(h): [sk["record_ids"][r], int(p)] for h, p, r, f in
                    zip(sk["out_hash"].tolist(), sk["pos"].tolist(), sk["record"].tolist(), flags.tolist()) if f & 1}
            assert uniq == ref["mx_info"][a["tsv"]]
            # filtered lists: per record that had >= 1 minimizer, in order; empty lists kept
            filt = []
            for r in range(len(sk["record_ids"])):
                lo, hi = int(sk["record_first"][r]), int(sk["record_first"][r + 1])
                if hi > lo:
                    filt.append([str(h) for h, f in zip(sk["out_hash"][lo:hi].tolist(), flags[lo:hi].tolist()) if f & 2])
            assert filt == ref["filtered"][a["tsv"]]
        # (3) graph arrays vs the reference's igraph content
        g = eng.get_graph()
        names = [str(h) for h in g["vertex_hash"].tolist()]
        assert sorted(names, key=int) == ref["vertices"]
        tsvs = [a["tsv"] for a in asms]
        mine = {}
        for u, v, m, wt in zip(g["edge_u"].tolist(), g["edge_v"].tolist(), g["edge_support"].tolist(),
                               g["edge_weight"].tolist()):
            mine[frozenset((names[u], names[v]))] = ([tsvs[b] for b in range(len(tsvs)) if m >> b & 1], wt)
        theirs = {frozenset((s, t)): (sup, wt) for s, t, sup, wt in ref["edges"]}
        assert mine == theirs
        # (4) .mx.dot canonical form identical to the file the reference's print_graph wrote
        dot = tmp_path / "out.mx.dot"
        eng.write_dot(str(dot))
        with open(os.path.join(cdir, "reference.mx.dot"), encoding="utf-8") as fh:
            want = go.canonical_dot_from_text(fh.read())
        assert go.canonical_dot_from_text(dot.read_text(encoding="utf-8")) == want


def _check_records(eng, a, recs, oracle, k, w, variant):
    sk = eng.get_sketch(a)
    assert len(sk["record_ids"]) == len(recs)
    for r, (rid, seq) in enumerate(recs):
        lo, hi = int(sk["record_first"][r]), int(sk["record_first"][r + 1])
        want = oracle.sketch(seq, k, w, variant)
        got = list(zip(sk["out_hash"][lo:hi].tolist(), sk["pos"][lo:hi].tolist(), sk["forward"][lo:hi].tolist()))
        assert got == [(h, p, f) for h, p, f, _ in want], (rid, k, w)


@pytest.mark.parametrize("dense", [False, True])
def test_random_records_vs_oracle(oracle, dense):
    """ragged / N-heavy / tie-heavy / soft-masked records over a sweep of (k, w)."""
    rng = random.Random(1234)
    for trial in range(24):
        k = rng.choice([1, 4, 15, 16, 17, 31, 32, 33, 48, 64, 100])
        w = rng.choice([1, 2, 7, 10, 50, 100, 500, 1000])
        variant = rng.choice(["v2", "v1"])
        recs = []
        for r in range(rng.randint(1, 12)):
            alphabet = rng.choice(["ACGT", "ACGT", "ACGTN", "AC", "A", "ACGTacgtNnUuRY", "ACGT" * 50 + "N"])
            n = rng.choice([0, 1, k - 1, k, k + w - 2, k + w - 1, k + w, 300, 3000, 20000])
            recs.append((f"r{r}", "".join(rng.choice(alphabet) for _ in range(max(n, 0)))))
        with _engine(k=k, w=w, variant=variant, dense_only=dense) as eng:
            eng.add_records("x", 1.0, recs)
            eng.sketch()
            _check_records(eng, 0, recs, oracle, k, w, _oracle.V1_MIN if variant == "v1" else _oracle.V2_SUM)


def _synth_pair(seed, n_ref, sub=0.005):
    rng = np.random.default_rng(seed)
    ref = rng.integers(0, 4, size=n_ref, dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref_s = lut[ref].tobytes().decode()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    pieces, p, i = [], 0, 0
    while p < n_ref:
        ln = int(np.exp(rng.uniform(np.log(10_000), np.log(400_000))))
        seg = ref[p:p + ln].copy()
        p += ln + int(rng.integers(20, 500))
        mut = rng.random(seg.size) < sub
        seg[mut] = (seg[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
        s = lut[seg].tobytes()
        if rng.random() < 0.5:
            s = s.translate(comp)[::-1]
        pieces.append((f"ctg{i}", s.decode()))
        i += 1
    order = rng.permutation(len(pieces))
    return [("chr1", ref_s)], [pieces[j] for j in order]


@pytest.mark.parametrize("dense", [False, True])
def test_synthetic_3mbp_vs_oracle(oracle, dense, tmp_path):
    """config-2-shaped input at a size the oracle finishes in seconds; TSV + canonical .mx.dot identical."""
    k, w = 32, 1000
    ref, tgt = _synth_pair(7, 3_000_000)
    names = ["ref.fa.k32.w1000.tsv", "tgt.fa.k32.w1000.tsv"]
    os.chdir(tmp_path)
    with _engine(k=k, w=w, dense_only=dense) as eng:
        eng.add_records(names[0], 2.0, ref)
        eng.add_records(names[1], 1.0, tgt)
        eng.sketch()
        _check_records(eng, 0, ref, oracle, k, w, _oracle.V2_SUM)
        _check_records(eng, 1, tgt, oracle, k, w, _oracle.V2_SUM)
        for a in (0, 1):
            eng.write_tsv(a, names[a])
        eng.build_graph()
        eng.write_dot("o.mx.dot")
    state = go.load_and_build([names[0]], [2.0], names[1], 1.0)
    got = go.canonical_dot_from_text(open("o.mx.dot", encoding="utf-8").read())
    assert got == go.canonical_dot_from_state(state)
    assert len(state["vertices"]) > 3000
