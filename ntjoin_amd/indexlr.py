"""
In-process counterparts of the two btllib classes ntJoin's scaffolding stage uses:

    btllib.Indexlr(fasta, k, w, btllib.IndexlrFlag.LONG_MODE, threads)     reference bin/ntjoin_assemble.py:490-492
        iterating yields entries with  .id  and  .minimizers[i].out_hash / .pos           (:495-516)
    btllib.SeqReader(fasta, btllib.SeqReaderFlag.LONG_MODE, threads)       reference bin/ntjoin_assemble.py:313-316
        iterating yields records with  .id  and  .seq

Indexlr sketches the WHOLE file on the GPU when the iteration starts (one add_fasta + one mxg_sketch; at the overlap
stage's k=15, w=10 that is the dense kernel) and then hands the records out in file order: the order, the record ids
and the (out_hash, pos) lists are what `indexlr --long --pos` prints.  A maintainer switches with
`import ntjoin_amd.indexlr as btllib` in bin/ntjoin_assemble.py (only these two classes of btllib are used there).
"""
from .engine import MxEngine


class IndexlrFlag:
    NO_ID = 1
    BX = 2
    SEQ = 4
    FILTER_IN = 8
    FILTER_OUT = 16
    SHORT_MODE = 32
    LONG_MODE = 64


class SeqReaderFlag:
    FOLD_CASE = 1
    SHORT_MODE = 2
    LONG_MODE = 4


class Minimizer:
    __slots__ = ("out_hash", "pos", "forward", "seq")

    def __init__(self, out_hash, pos, forward, seq=None):
        self.out_hash, self.pos, self.forward, self.seq = out_hash, pos, forward, seq

    def __repr__(self):
        return f"Minimizer(out_hash={self.out_hash}, pos={self.pos}, forward={self.forward})"


class IndexlrRecord:
    __slots__ = ("num", "id", "barcode", "readlen", "minimizers")

    def __init__(self, num, rid, readlen, minimizers):
        self.num, self.id, self.barcode, self.readlen, self.minimizers = num, rid, "", readlen, minimizers


class Indexlr:
    """with Indexlr(path, k, w, IndexlrFlag.LONG_MODE, threads) as minimizers: for entry in minimizers: ..."""

    def __init__(self, seqfile, k, w, flags=IndexlrFlag.LONG_MODE, threads=1, verbose=False, variant="v2", device=-1):
        if flags & (IndexlrFlag.FILTER_IN | IndexlrFlag.FILTER_OUT | IndexlrFlag.BX):
            raise NotImplementedError("Bloom-filter and barcode modes are not used by ntJoin")
        self._path, self.k, self.w, self._flags = str(seqfile), int(k), int(w), int(flags)
        self._variant, self._device = variant, device
        self._eng = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        if self._eng is not None:
            self._eng.close()
            self._eng = None

    def __iter__(self):
        self.close()
        self._eng = eng = MxEngine(k=self.k, w=self.w, variant=self._variant, device=self._device)
        a = eng.add_fasta(self._path, 1.0, self._path)
        eng.sketch(a)
        sk = eng.get_sketch(a)
        ids, first = sk["record_ids"], sk["record_first"]
        out_hash, pos, fwd = sk["out_hash"].tolist(), sk["pos"].tolist(), sk["forward"].tolist()
        lens = eng.record_lengths(a)
        no_id = bool(self._flags & IndexlrFlag.NO_ID)
        for r, rid in enumerate(ids):
            lo, hi = int(first[r]), int(first[r + 1])
            yield IndexlrRecord(r, "" if no_id else rid, lens[r],
                                [Minimizer(out_hash[i], pos[i], bool(fwd[i])) for i in range(lo, hi)])
        self.close()


class SeqRecord:
    __slots__ = ("num", "id", "comment", "seq", "qual")

    def __init__(self, num, rid, comment, seq):
        self.num, self.id, self.comment, self.seq, self.qual = num, rid, comment, seq, ""


class SeqReader:
    """with SeqReader(path, SeqReaderFlag.LONG_MODE, threads) as fin: for rec in fin: rec.id, rec.seq
    Plain FASTA on the host (multi-line records, '>' header: id = first word, comment = the rest); this is I/O, not a
    hot path: the packed bases that feed the GPU are produced by the library's own ingest (host_io.cpp)."""

    def __init__(self, seqfile, flags=SeqReaderFlag.LONG_MODE, threads=1):
        self._path, self._flags, self._fh = str(seqfile), int(flags), None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        if self._fh is not None:
            self._fh.close()
            self._fh = None

    def __iter__(self):
        self.close()
        self._fh = fh = open(self._path, "r", encoding="ascii")  # FileNotFoundError propagates, as the reference expects
        fold = bool(self._flags & SeqReaderFlag.FOLD_CASE)
        num, rid, comment, chunks = 0, None, "", []
        for line in fh:
            if line.startswith(">"):
                if rid is not None:
                    yield SeqRecord(num, rid, comment, "".join(chunks))
                    num += 1
                head = line[1:].rstrip("\r\n").split(None, 1)
                rid, comment, chunks = (head[0] if head else ""), (head[1] if len(head) > 1 else ""), []
            elif rid is not None:
                s = line.strip()
                chunks.append(s.upper() if fold else s)
        if rid is not None:
            yield SeqRecord(num, rid, comment, "".join(chunks))
        self.close()
