"""ctypes loader for the C oracle (oracle/_build/libmx_oracle.so).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "libmx_oracle.so")
BIN_PATH = os.path.join(ORACLE_DIR, "_build", "mx_oracle")

V2_SUM, V1_MIN = 0, 1


class Minimizer(ctypes.Structure):
    _fields_ = [("out_hash", ctypes.c_uint64), ("min_hash", ctypes.c_uint64),
                ("pos", ctypes.c_uint32), ("forward", ctypes.c_uint8)]


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        for fn in (L.mxo_sketch_stateful, L.mxo_sketch_stateless):
            fn.restype = ctypes.c_size_t
            fn.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_uint, ctypes.c_int,
                           ctypes.POINTER(ctypes.POINTER(Minimizer))]
        L.mxo_free.argtypes = [ctypes.c_void_p]
        L.mxo_kmer_hashes.restype = ctypes.c_size_t
        L.mxo_kmer_hashes.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.mxo_nthash_direct.restype = ctypes.c_int
        L.mxo_nthash_direct.argtypes = [ctypes.c_char_p, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint64),
                                        ctypes.POINTER(ctypes.c_uint64)]
        L.mxo_sketch_fasta_to_tsv.restype = ctypes.c_int
        L.mxo_sketch_fasta_to_tsv.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint, ctypes.c_uint,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.POINTER(ctypes.c_uint64)]
        for fn in (L.mxo_srol, L.mxo_sror):
            fn.restype = ctypes.c_uint64
            fn.argtypes = [ctypes.c_uint64]
        L.mxo_srol_n.restype = ctypes.c_uint64
        L.mxo_srol_n.argtypes = [ctypes.c_uint64, ctypes.c_uint]
        L.mxo_ext_hash.restype = ctypes.c_uint64
        L.mxo_ext_hash.argtypes = [ctypes.c_uint64, ctypes.c_uint]

    def sketch_packed_mt(self, words, rec_start, rec_len, k, w, variant=V2_SUM, threads=1, chunk_kmers=0):
        """`indexlr -t threads` on 2-bit packed N-free records -> (out_hash u64[], pos u32[], record u32[])"""
        L = self.lib
        L.mxo_sketch_packed_mt.restype = ctypes.c_size_t
        L.mxo_sketch_packed_mt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint,
                                           ctypes.c_uint, ctypes.c_int, ctypes.c_uint, ctypes.c_uint64,
                                           ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                           ctypes.POINTER(ctypes.c_void_p)]
        words = np.ascontiguousarray(words, dtype=np.uint32)
        rs = np.ascontiguousarray(rec_start, dtype=np.uint64)
        rl = np.ascontiguousarray(rec_len, dtype=np.uint64)
        ph, pp, pr = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        n = L.mxo_sketch_packed_mt(words.ctypes.data, rs.ctypes.data, rl.ctypes.data, len(rs), k, w, variant, threads,
                                   chunk_kmers, ctypes.byref(ph), ctypes.byref(pp), ctypes.byref(pr))

        def take(p, ct, dt):
            a = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ct)), shape=(max(n, 1),))[:n].astype(dt, copy=True)
            L.mxo_free(p)
            return a
        return take(ph, ctypes.c_uint64, np.uint64), take(pp, ctypes.c_uint32, np.uint32), take(pr, ctypes.c_uint32, np.uint32)

    def graph(self, hashes, recs, weights, edges=False):
        """C restatement of uniqueness + intersection + build_graph on arrays (assemblies in the reference's order)
        -> dict(unique, vertices, edges[, eu, ev, esup, ew])"""
        L = self.lib
        A = len(hashes)
        hh = [np.ascontiguousarray(h, dtype=np.uint64) for h in hashes]
        rr = [np.ascontiguousarray(r, dtype=np.uint32) for r in recs]
        ph = (ctypes.c_void_p * A)(*[h.ctypes.data for h in hh])
        pr = (ctypes.c_void_p * A)(*[r.ctypes.data for r in rr])
        nn = (ctypes.c_uint64 * A)(*[len(h) for h in hh])
        ww = (ctypes.c_double * A)(*[float(x) for x in weights])
        counts = (ctypes.c_uint64 * 3)()
        L.mxo_graph.restype = ctypes.c_int
        L.mxo_graph.argtypes = [ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        eu, ev, es, ew = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        if edges:
            rc = L.mxo_graph(A, ph, pr, nn, ww, counts, ctypes.byref(eu), ctypes.byref(ev), ctypes.byref(es), ctypes.byref(ew))
        else:
            rc = L.mxo_graph(A, ph, pr, nn, ww, counts, None, None, None, None)
        if rc != 0:
            raise RuntimeError(f"mxo_graph failed ({rc})")
        out = {"unique": int(counts[0]), "vertices": int(counts[1]), "edges": int(counts[2])}
        if edges:
            ne = out["edges"]

            def take(p, ct, dt):
                a = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ct)), shape=(max(ne, 1),))[:ne].astype(dt, copy=True)
                L.mxo_free(p)
                return a
            out["eu"], out["ev"] = take(eu, ctypes.c_uint64, np.uint64), take(ev, ctypes.c_uint64, np.uint64)
            out["esup"], out["ew"] = take(es, ctypes.c_uint32, np.uint32), take(ew, ctypes.c_double, np.float64)
        return out

    def _sketch(self, fn, seq, k, w, variant):
        if isinstance(seq, str):
            seq = seq.encode("ascii")
        out = ctypes.POINTER(Minimizer)()
        n = fn(seq, len(seq), k, w, variant, ctypes.byref(out))
        res = [(out[i].out_hash, out[i].pos, out[i].forward, out[i].min_hash) for i in range(n)]
        self.lib.mxo_free(out)
        return res

    def sketch(self, seq, k, w, variant=V2_SUM):
        """[(out_hash, pos, forward, min_hash)] via the btllib-style stateful loop."""
        return self._sketch(self.lib.mxo_sketch_stateful, seq, k, w, variant)

    def sketch_stateless(self, seq, k, w, variant=V2_SUM):
        return self._sketch(self.lib.mxo_sketch_stateless, seq, k, w, variant)

    def kmer_hashes(self, seq, k, variant=V2_SUM):
        if isinstance(seq, str):
            seq = seq.encode("ascii")
        n = max(len(seq) - k + 1, 0)
        mh = np.zeros(n, dtype=np.uint64)
        oh = np.zeros(n, dtype=np.uint64)
        fw = np.zeros(n, dtype=np.uint8)
        ok = np.zeros(n, dtype=np.uint8)
        if n:
            self.lib.mxo_kmer_hashes(seq, len(seq), k, variant, mh.ctypes.data, oh.ctypes.data,
                                     fw.ctypes.data, ok.ctypes.data)
        return mh, oh, fw, ok

    def fasta_to_tsv(self, fasta, out, k, w, variant=V2_SUM, pos=True, strand=False, seq=True):
        stats = (ctypes.c_uint64 * 3)()
        rc = self.lib.mxo_sketch_fasta_to_tsv(fasta.encode(), out.encode(), k, w, variant,
                                              int(pos), int(strand), int(seq), stats)
        if rc != 0:
            raise RuntimeError(f"oracle failed on {fasta}")
        return {"records": stats[0], "bases": stats[1], "minimizers": stats[2]}


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def load():
    if not os.path.exists(LIB_PATH):
        build()
    return Oracle(ctypes.CDLL(LIB_PATH))


def read_fasta(path):
    """[(id, sequence)] -- id = first whitespace-delimited token after '>'."""
    recs = []
    with open(path, encoding="ascii") as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                recs.append([line[1:].split()[0] if line[1:].split() else "", []])
            elif recs and line:
                recs[-1][1].append(line)
    return [(rid, "".join(chunks)) for rid, chunks in recs]
