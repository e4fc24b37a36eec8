"""GPU test of the in-process Indexlr / SeqReader counterparts: what ntJoin's overlap stage reads from
btllib (reference bin/ntjoin_assemble.py:313-316,490-516) -- record ids in file order, (out_hash, pos) per record at
k=15, w=10 -- against the CPU oracle and the committed sketches."""
import os

import pytest

from tests import _oracle
from tests.conftest import GOLDEN, golden_cases, load_case

pytestmark = pytest.mark.gpu
FASTA = os.path.join(GOLDEN, "fasta")


def _fasta_records(path):
    recs, rid, chunks = [], None, []
    for line in open(path, encoding="ascii"):
        if line.startswith(">"):
            if rid is not None:
                recs.append((rid, "".join(chunks)))
            rid, chunks = line[1:].split()[0], []
        else:
            chunks.append(line.strip())
    if rid is not None:
        recs.append((rid, "".join(chunks)))
    return recs


@pytest.mark.parametrize("k,w", [(15, 10), (32, 100)])
def test_indexlr_iterator_matches_oracle(oracle, k, w):
    import ntjoin_amd.indexlr as btllib
    for fa in sorted(os.listdir(FASTA))[:6]:
        path = os.path.join(FASTA, fa)
        want = _fasta_records(path)
        with btllib.Indexlr(path, k, w, btllib.IndexlrFlag.LONG_MODE, 4) as minimizers:
            got = list(minimizers)
        assert [e.id for e in got] == [rid for rid, _ in want]
        assert [e.readlen for e in got] == [len(s) for _, s in want]
        for e, (rid, seq) in zip(got, want):
            exp = oracle.sketch(seq, k, w)
            assert [(m.out_hash, m.pos, int(m.forward)) for m in e.minimizers] == [(h, p, f) for h, p, f, _ in exp], (fa, rid)


def test_indexlr_usage_of_the_overlap_stage():
    """the loop of NtjoinScaffolder.tally_minimizers_overlap (bin/ntjoin_assemble.py:504-516) runs unchanged"""
    import ntjoin_amd.indexlr as btllib
    case = next(m for m in golden_cases() if m["name"].startswith("f-f"))
    path = os.path.join(FASTA, case["target"]["fasta"])
    n = 0
    with btllib.Indexlr(path, 15, 10, btllib.IndexlrFlag.LONG_MODE, 1) as minimizers:
        for mx_entry in minimizers:
            seen = {}
            for mx_pos_strand in mx_entry.minimizers:
                mx, pos = str(mx_pos_strand.out_hash), mx_pos_strand.pos
                seen.setdefault(mx, int(pos))
            assert mx_entry.id and len(seen) > 0
            n += 1
    assert n >= 1


def test_seqreader_counterpart(tmp_path):
    import ntjoin_amd.indexlr as btllib
    p = tmp_path / "x.fa"
    p.write_text(">a first record\nACGT\nacgtn\n>b\n\nTTTT\n>c\n", encoding="ascii")
    with btllib.SeqReader(str(p), btllib.SeqReaderFlag.LONG_MODE, 2) as fin:
        recs = [(r.id, r.comment, r.seq) for r in fin]
    assert recs == [("a", "first record", "ACGTacgtn"), ("b", "", "TTTT"), ("c", "", "")]
    with pytest.raises(FileNotFoundError):
        with btllib.SeqReader(str(tmp_path / "missing.fa"), btllib.SeqReaderFlag.LONG_MODE, 2) as fin:
            list(fin)


def test_fuzz_indexlr_iterator(oracle, tmp_path):
    """random FASTA files (ragged lines, lower case, N runs, empty and tiny records, hundreds of records) through the Indexlr
    counterpart at the overlap stage's parameters and others: ids, lengths and (out_hash, pos, strand) per record = the oracle"""
    import random
    import ntjoin_amd.indexlr as btllib
    trials = int(os.environ.get("MXG_FUZZ_TRIALS", "20"))
    rng = random.Random(int(os.environ.get("MXG_FUZZ_SEED", "31")))
    for t in range(trials):
        k, w = rng.choice([(15, 10), (15, 10), (32, 100), (21, 50), (11, 5), (32, 1000)])
        recs = []
        for i in range(rng.choice([1, 3, 30, 300])):
            n = rng.choice([0, 1, k - 1, k, k + w - 2, k + w - 1, 200, 5000, 40000])
            alphabet = rng.choice(["ACGT", "ACGT", "ACGTacgt", "ACGTN", "AC", "ACGTRYn"])
            recs.append((f"rec{i}", "".join(rng.choice(alphabet) for _ in range(n))))
        path = str(tmp_path / f"f{t}.fa")
        with open(path, "w", encoding="ascii") as fh:
            for rid, s in recs:
                fh.write(f">{rid} comment {t}\n")
                p = 0
                while p < len(s):
                    wd = rng.randint(1, 120)
                    fh.write(s[p:p + wd] + "\n")
                    p += wd
        with btllib.Indexlr(path, k, w, btllib.IndexlrFlag.LONG_MODE, rng.choice([1, 2, 6])) as minimizers:
            got = list(minimizers)
        assert [e.id for e in got] == [rid for rid, _ in recs], (t, k, w)
        assert [e.readlen for e in got] == [len(s) for _, s in recs], (t, k, w)
        for e, (rid, seq) in zip(got, recs):
            exp = oracle.sketch(seq, k, w)
            assert [(m.out_hash, m.pos, int(m.forward)) for m in e.minimizers] == [(h, p, f) for h, p, f, _ in exp], (t, k, w, rid)
