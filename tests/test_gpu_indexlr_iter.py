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
