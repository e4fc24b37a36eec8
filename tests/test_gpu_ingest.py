"""GPU tests of the file route: FASTA text classified and packed ON THE DEVICE (ntjoin_amd/csrc/ingest.hip) and the TSV text
formatted ON THE DEVICE, against (a) the CPU oracle's own FASTA -> TSV driver (byte-identical files) and (b) the host parser /
host writer of the library (MXG_HOST_INGEST=1 / MXG_HOST_TSV=1).  Shapes that stress the text handling: CRLF line ends
(also across the 4 MiB read buffer of the host parser and across the device's 4096-byte tiles), lower case, N-runs,
IUPAC codes, ragged line lengths, no final newline, '>' inside header lines, empty records, thousands of tiny records."""
import os
import random
import subprocess

import numpy as np
import pytest

from ntjoin_amd.engine import MxEngine
from tests import _oracle
from tests.conftest import BIN_DIR, REPO

pytestmark = pytest.mark.gpu


def _write_fasta(path, recs, width=60, eol="\n", ragged=None, final_newline=True):
    rng = random.Random(7)
    with open(path, "w", newline="") as fh:
        for i, (hdr, seq) in enumerate(recs):
            fh.write(">" + hdr + eol)
            p = 0
            while p < len(seq):
                wd = rng.randint(1, 2 * width) if ragged else width
                last = p + wd >= len(seq) and i == len(recs) - 1
                fh.write(seq[p:p + wd] + ("" if last and not final_newline else eol))
                p += wd


def _seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def _messy_records(seed):
    rng = random.Random(seed)
    recs = [("chr1 some comment > with a bracket", _seq(rng, 150_000)),
            ("lower", _seq(rng, 40_000).lower()),
            ("mixed\tcase", "".join(c.lower() if rng.random() < 0.3 else c for c in _seq(rng, 60_000))),
            ("empty", ""),
            ("tiny", "ACGTACGTAC"),
            ("withN", _seq(rng, 30_000) + "N" * 700 + _seq(rng, 25_000) + "n" * 3 + _seq(rng, 9_000)),
            ("iupac", _seq(rng, 20_000) + "RYKM" + _seq(rng, 20_000) + "-" + _seq(rng, 5_000)),
            ("rna", _seq(rng, 30_000, "ACGU")),
            ("startsN", "NNNN" + _seq(rng, 12_000) + "NN")]
    recs += [(f"frag{i}", _seq(rng, rng.randint(20, 2500))) for i in range(300)]
    recs.append(("last", _seq(rng, 70_000)))
    return recs


def _tsv_by_engine(fa, k, w, out, **env):
    saved = {k_: os.environ.get(k_) for k_ in env}
    os.environ.update(env)
    try:
        with MxEngine(k=k, w=w, threads=3) as eng:
            eng.add_fasta("x", 1.0, fa)
            eng.sketch()
            eng.write_tsv(0, out, with_pos=True, with_strand=False, with_seq=True)
            st = eng.stats()
    finally:
        for k_, v in saved.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v
    return st


@pytest.mark.parametrize("eol,ragged,final_nl", [("\n", False, True), ("\r\n", False, True), ("\n", True, False), ("\r\n", True, True)])
@pytest.mark.parametrize("k,w", [(32, 100), (15, 10)])
def test_device_route_equals_oracle_and_host_route(tmp_path, eol, ragged, final_nl, k, w):
    orc = _oracle.load()
    fa = str(tmp_path / "messy.fa")
    _write_fasta(fa, _messy_records(3), width=70, eol=eol, ragged=ragged, final_newline=final_nl)
    want = str(tmp_path / "oracle.tsv")
    orc.fasta_to_tsv(fa, want, k, w)
    dev, host = str(tmp_path / "dev.tsv"), str(tmp_path / "host.tsv")
    st_dev = _tsv_by_engine(fa, k, w, dev)
    st_host = _tsv_by_engine(fa, k, w, host, MXG_HOST_INGEST="1", MXG_HOST_TSV="1")
    assert open(dev, "rb").read() == open(want, "rb").read()
    assert open(host, "rb").read() == open(want, "rb").read()
    assert st_dev["bases"] == st_host["bases"] and st_dev["kmers"] == st_host["kmers"] and st_dev["minimizers"] > 500


def test_crlf_across_buffer_and_tile_borders(tmp_path):
    """a CRLF FASTA larger than the host parser's 4 MiB read buffer, with a line length that walks the "\\r\\n" pair over
    every alignment (a pair cut by the buffer border left the '\\r' in the sequence)"""
    orc = _oracle.load()
    rng = random.Random(5)
    recs = [("a", _seq(rng, 5_300_000)), ("b", _seq(rng, 3_100_000))]
    lf, crlf = str(tmp_path / "lf.fa"), str(tmp_path / "crlf.fa")
    _write_fasta(lf, recs, width=61)
    _write_fasta(crlf, recs, width=61, eol="\r\n")
    want = str(tmp_path / "want.tsv")
    orc.fasta_to_tsv(lf, want, 32, 1000)
    for tag, env in (("dev", {}), ("host", {"MXG_HOST_INGEST": "1", "MXG_HOST_TSV": "1"})):
        out = str(tmp_path / f"{tag}.tsv")
        _tsv_by_engine(crlf, 32, 1000, out, **env)
        assert open(out, "rb").read() == open(want, "rb").read(), tag


def test_cli_threads_and_strand_column(tmp_path):
    """`indexlr -t N` (reference ntJoin:205 passes -t $(t)): same bytes for any thread count, and with --strand / without --seq"""
    orc = _oracle.load()
    fa = str(tmp_path / "m.fa")
    _write_fasta(fa, _messy_records(9), width=80)
    exe = os.path.join(BIN_DIR, "indexlr")
    want = str(tmp_path / "want.tsv")
    orc.fasta_to_tsv(fa, want, 32, 100)
    for t in ("1", "4", "48"):
        out = subprocess.run([exe, "--seq", "--long", "--pos", "-k32", "-w100", f"-t{t}", fa], capture_output=True, check=True).stdout
        assert out == open(want, "rb").read(), t
    want2 = str(tmp_path / "want2.tsv")
    orc.fasta_to_tsv(fa, want2, 32, 100, pos=True, strand=True, seq=False)
    out = subprocess.run([exe, "--pos", "--strand", "-k", "32", "-w", "100", "-t", "2", "-o", str(tmp_path / "o.tsv"), fa], check=True)
    assert open(tmp_path / "o.tsv", "rb").read() == open(want2, "rb").read()


def test_tsv_of_packed_device_assembly_is_formatted_on_the_device(tmp_path, oracle):
    """no text anywhere (bases born packed in HBM): the k-mer column is decoded from the packed bases; several output windows"""
    from ntjoin_amd import synth
    cfg = synth.genome_config(40_000_000, 5, seed=4, min_len=2000, max_len=200_000)
    with MxEngine(k=32, w=20) as eng:   # w=20: ~4 M minimizers, > 250 MB of TSV text = several 64 MiB windows
        segs = cfg["tgt_segs"]
        d = synth.fill_device(segs, cfg["tgt_words"], cfg["seed"], cfg["sub_seed"], synth.SUB_PER_65536)
        eng.add_packed_device("t", 1.0, d.data_ptr(), segs[:, 0], segs[:, 2], keepalive=d)
        eng.sketch()
        out = str(tmp_path / "t.tsv")
        eng.write_tsv(0, out)
        sk = eng.get_sketch(0)
    assert os.path.getsize(out) > 3 * (64 << 20)
    first = sk["record_first"]
    with open(out, "rb") as fh:
        for r, line in enumerate(fh):
            rid, _, rest = line.rstrip(b"\n").partition(b"\t")
            assert rid == str(r).encode()
            lo, hi = int(first[r]), int(first[r + 1])
            if r % 97 == 0 or r < 3:   # every field of these records, the k-mers against the generator's numpy mirror
                ents = rest.split(b" ") if rest else []
                assert len(ents) == hi - lo
                codes = synth.segment_codes(segs[r], cfg["seed"], cfg["sub_seed"], synth.SUB_PER_65536)
                seq = synth.to_ascii(codes)
                for e, i in zip(ents, range(lo, hi)):
                    hsh, pos, kmer = e.split(b":")
                    assert int(hsh) == int(sk["out_hash"][i]) and int(pos) == int(sk["pos"][i])
                    assert kmer == seq[int(pos):int(pos) + 32]
            else:
                assert rest.count(b" ") + (1 if rest else 0) == hi - lo
    assert r == len(segs) - 1


def test_gzip_fasta_input(tmp_path):
    """`.gz` input (indexlr reads both; SURVEY.md 8b B1 lists it as optional): same TSV as the plain file, through the CLI and
    through the sub-record split loader"""
    import gzip
    orc = _oracle.load()
    fa = str(tmp_path / "m.fa")
    _write_fasta(fa, _messy_records(11), width=60)
    gz = fa + ".gz"
    with open(fa, "rb") as src, gzip.open(gz, "wb") as dst:
        dst.write(src.read())
    want = str(tmp_path / "want.tsv")
    orc.fasta_to_tsv(fa, want, 32, 100)
    exe = os.path.join(BIN_DIR, "indexlr")
    out = subprocess.run([exe, "--seq", "--long", "--pos", "-k32", "-w100", "-t2", gz], capture_output=True, check=True).stdout
    assert out == open(want, "rb").read()
    parts = []
    for s in range(3):
        with MxEngine(k=32, w=100) as eng:
            eng.add_fasta_split("x", 1.0, gz, s, 3)
            eng.sketch()
            parts.append(eng.get_sketch(0))
    with MxEngine(k=32, w=100) as eng:
        eng.add_fasta("x", 1.0, fa)
        eng.sketch()
        whole = eng.get_sketch(0)
    for key in ("out_hash", "pos", "record"):
        assert np.array_equal(np.concatenate([p[key] for p in parts]), whole[key]), key


def test_fuzz_fasta_text_three_ways(tmp_path):
    """random FASTA text -- line widths from 1 up, LF or CRLF, with or without the final line end, empty and tiny records next to
    records of several device tiles, lower case, N runs and IUPAC letters at random places, header lines with blanks, tabs and
    '>' -- through the device route, the host route and the oracle's own FASTA -> TSV driver: three byte-identical files"""
    orc = _oracle.load()
    trials = int(os.environ.get("MXG_FUZZ_TRIALS", "25"))
    rng = random.Random(int(os.environ.get("MXG_FUZZ_SEED", "99")))
    for t in range(trials):
        k, w = rng.choice([(32, 100), (32, 300), (32, 1000), (15, 10), (21, 50)])
        recs = []
        for i in range(rng.choice([1, 2, 5, 40, 400])):
            n = rng.choice([0, 1, k - 1, k, k + w - 2, k + w - 1, 200, 4095, 4096, 4097, 9000, 70000])
            alphabet = rng.choice(["ACGT", "ACGT", "ACGTacgt", "ACGTN", "ACGTRYKMn-", "ACGU", "A", "AC"])
            s = [rng.choice(alphabet) for _ in range(n)]
            if n > 3000 and rng.random() < 0.4:
                a = rng.randrange(n - 2000)
                s[a:a + rng.choice([1, 31, 32, 700, 1900])] = rng.choice("Nn") * len(s[a:a + rng.choice([1, 31, 32, 700, 1900])])
            hdr = rng.choice([f"r{i}", f"r{i} a comment", f"r{i}\twith tab", f"r{i} > bracket", f"{i}"])
            recs.append((hdr, "".join(s)))
        fa = str(tmp_path / f"f{t}.fa")
        eol = rng.choice(["\n", "\r\n"])
        width = rng.choice([1, 7, 60, 61, 80, 4096, 100000])
        _write_fasta(fa, recs, width=width, eol=eol, ragged=rng.random() < 0.3, final_newline=rng.random() < 0.7)
        want = str(tmp_path / f"want{t}.tsv")
        orc.fasta_to_tsv(fa, want, k, w)
        for tag, env in (("dev", {}), ("host", {"MXG_HOST_INGEST": "1", "MXG_HOST_TSV": "1"})):
            out = str(tmp_path / f"{tag}{t}.tsv")
            _tsv_by_engine(fa, k, w, out, **env)
            assert open(out, "rb").read() == open(want, "rb").read(), (t, tag, k, w, width, repr(eol), len(recs))
