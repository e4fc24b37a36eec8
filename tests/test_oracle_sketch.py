"""CPU tests: the C oracle of the sketch stage against the reference's golden vectors."""
import filecmp
import os
import random

import pytest

from tests import _oracle
from tests.conftest import GOLDEN, golden_cases

FASTA = os.path.join(GOLDEN, "fasta")
EXPECTED = os.path.join(GOLDEN, "reference_expected_outputs")


def _positions(oracle, fasta, rec_id, k, w, variant=_oracle.V2_SUM):
    for rid, seq in _oracle.read_fasta(os.path.join(FASTA, fasta)):
        if rid == rec_id:
            return [p for _, p, _, _ in oracle.sketch(seq, k, w, variant)]
    raise KeyError(rec_id)


@pytest.mark.parametrize("fasta", ["ref.fa", "scaf.f-f.fa"])
def test_v1_golden_tsv_bit_exact(oracle, fasta, tmp_path):
    """reference tests/expected_outputs/*.k32.w1000.tsv: 13 exact (out_hash,pos) pairs (variant V1)."""
    out = tmp_path / "o.tsv"
    oracle.fasta_to_tsv(os.path.join(FASTA, fasta), str(out), 32, 1000, _oracle.V1_MIN, pos=True, seq=False)
    assert filecmp.cmp(str(out), os.path.join(EXPECTED, f"{fasta}.k32.w1000.tsv"), shallow=False)


def test_v2_positions_pinned_by_head_tests(oracle):
    """reference tests/ntjoin_test.py:133 expects cuts 0-2232, 2110-4489, 0-1568, 2712-4379 (k=32, w=500):
    minimizers at 2200(+32), 2110, 1536(+32), 2712 -- only the V2 canonical form yields them."""
    p1 = _positions(oracle, "scaf.misassembled.f-f.r-r.fa", "1_1p-2_2n", 32, 500)
    p2 = _positions(oracle, "scaf.misassembled.f-f.r-r.fa", "2_1n-1_2p", 32, 500)
    assert 1536 in p1 and 2110 in p1
    assert 2200 in p2 and 2712 in p2
    # :148 expects 0-1624, 2058-4489, 0-2232, 2518-4379
    q1 = _positions(oracle, "scaf.misassembled.f-r.r-f.fa", "1_1p-2_2p", 32, 500)
    q2 = _positions(oracle, "scaf.misassembled.f-r.r-f.fa", "2_1n-1_2n", 32, 500)
    assert 1592 in q1 and 2058 in q1
    assert 2200 in q2 and 2518 in q2
    # V1 gives different positions there (SURVEY.md finding 0.3)
    v1 = _positions(oracle, "scaf.misassembled.f-f.r-r.fa", "1_1p-2_2n", 32, 500, _oracle.V1_MIN)
    assert 1536 not in v1


def test_v2_probe_vectors(oracle):
    """SURVEY.md Appendix C rows for tests/ref.fa and the termN fixture (V2, w=1000)."""
    seq = dict(_oracle.read_fasta(os.path.join(FASTA, "ref.fa")))["test"]
    got = [(h, p) for h, p, _, _ in oracle.sketch(seq, 32, 1000)]
    assert got == [(12115725118402243787, 438), (4639234455201901010, 1425), (15239283564189465324, 1437),
                   (13009180041705437851, 2413), (18425627561391081985, 3170), (7462327723943047921, 3608)]
    term = dict(_oracle.read_fasta(os.path.join(FASTA, "scaf.f-f.termN.fa")))["1_f"]
    assert [(h, p) for h, p, _, _ in oracle.sketch(term, 32, 1000)] == \
        [(12115725118402243787, 442), (4639234455201901010, 1429)]
    # config 1 sizes (k=32, w=100): 83 / 37 + 43 minimizers
    assert len(oracle.sketch(seq, 32, 100)) == 83
    ff = dict(_oracle.read_fasta(os.path.join(FASTA, "scaf.f-f.fa")))
    assert (len(oracle.sketch(ff["1_f"], 32, 100)), len(oracle.sketch(ff["2_f"], 32, 100))) == (37, 43)


def test_overlap_shared_minimizer_k15_w10(oracle):
    """reference tests/ntjoin_test.py:202 `1+:0-2033 ... 2+:34-2331` (k=15,w=10): the same k-mer is a
    minimizer at 2033 in contig 1 and at 34 in contig 2."""
    recs = dict(_oracle.read_fasta(os.path.join(FASTA, "scaf.f-f.overlapping.fa")))
    a = {p: h for h, p, _, _ in oracle.sketch(recs["1"], 15, 10)}
    b = {p: h for h, p, _, _ in oracle.sketch(recs["2"], 15, 10)}
    assert 2033 in a and 34 in b and a[2033] == b[34]


def test_rolling_equals_direct(oracle):
    rng = random.Random(7)
    for k in (1, 2, 15, 31, 32, 33, 64, 100):
        seq = "".join(rng.choice("ACGTacgtNU") for _ in range(400))
        mh, oh, fw, ok = oracle.kmer_hashes(seq, k)
        import ctypes
        for i in range(len(seq) - k + 1):
            f, r = ctypes.c_uint64(), ctypes.c_uint64()
            valid = oracle.lib.mxo_nthash_direct(seq[i:i + k].encode(), k, ctypes.byref(f), ctypes.byref(r))
            assert bool(valid) == bool(ok[i])
            if valid:
                assert int(mh[i]) == (f.value + r.value) & 0xFFFFFFFFFFFFFFFF
                assert int(oh[i]) == oracle.lib.mxo_ext_hash(int(mh[i]), k)
                assert bool(fw[i]) == (f.value <= r.value)


def test_srol_identities(oracle):
    rng = random.Random(3)
    for _ in range(200):
        x = rng.getrandbits(64)
        assert oracle.lib.mxo_sror(oracle.lib.mxo_srol(x)) == x
        y = x
        for n in range(0, 70):
            assert oracle.lib.mxo_srol_n(x, n) == y
            y = oracle.lib.mxo_srol(y)
    assert oracle.lib.mxo_srol_n(12345, 1023) == 12345  # period lcm(33,31)


@pytest.mark.parametrize("k,w", [(32, 1000), (32, 500), (32, 100), (15, 10), (20, 50), (32, 1)])
def test_stateful_equals_stateless_on_fixtures(oracle, k, w):
    for fa in sorted(os.listdir(FASTA)):
        for _, seq in _oracle.read_fasta(os.path.join(FASTA, fa)):
            for variant in (_oracle.V2_SUM, _oracle.V1_MIN):
                assert oracle.sketch(seq, k, w, variant) == oracle.sketch_stateless(seq, k, w, variant)


def test_stateful_equals_stateless_tie_heavy(oracle):
    """low-complexity / N-heavy random records: ties everywhere, windows spanning invalid k-mers."""
    rng = random.Random(11)
    for trial in range(300):
        alphabet = rng.choice(["A", "AC", "ACGT", "ACN", "AAAAAAAAN", "ACGTN"])
        n = rng.randint(0, 300)
        seq = "".join(rng.choice(alphabet) for _ in range(n))
        k = rng.randint(1, 12)
        w = rng.randint(1, 40)
        assert oracle.sketch(seq, k, w) == oracle.sketch_stateless(seq, k, w), (seq, k, w)


def test_short_and_empty_records(oracle, tmp_path):
    assert oracle.sketch("", 32, 10) == []
    assert oracle.sketch("ACGT", 32, 10) == []
    assert oracle.sketch("ACGT" * 10, 32, 10) == []        # 9 k-mers < w
    assert len(oracle.sketch("ACGTTGCA" * 6, 32, 17)) >= 1  # exactly w k-mers -> one window
    fa = tmp_path / "x.fa"
    fa.write_text(">empty\n>short desc\nACGT\n>ok\n" + "ACGTTGCATTGACCA" * 10 + "\n")
    out = tmp_path / "x.tsv"
    oracle.fasta_to_tsv(str(fa), str(out), 8, 4)
    lines = out.read_text().split("\n")
    assert lines[0] == "empty\t" and lines[1] == "short\t" and lines[2].startswith("ok\t") and lines[3] == ""


def test_case_tsvs_reproduce(oracle, tmp_path):
    """every committed case TSV is what the oracle produces from the committed FASTA."""
    for meta in golden_cases():
        variant = _oracle.V1_MIN if meta["variant"] == "v1" else _oracle.V2_SUM
        for asm in meta["refs"] + [meta["target"]]:
            out = tmp_path / "t.tsv"
            oracle.fasta_to_tsv(os.path.join(FASTA, asm["fasta"]), str(out), meta["k"], meta["w"], variant)
            assert filecmp.cmp(str(out), os.path.join(GOLDEN, "cases", meta["name"], asm["tsv"]), shallow=False)
