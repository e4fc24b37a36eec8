"""
MxEngine -- Python face of one libntjoin_mx handle (one k, w, hash variant, one GPU).

Order of use mirrors the reference pipeline:
    eng = MxEngine(k=32, w=1000)
    eng.add_fasta("ref.fa.k32.w1000.tsv", 2.0, "ref.fa")        # references first, CLI order
    eng.add_fasta("scaf.fa.k32.w1000.tsv", 1.0, "scaf.fa")      # target last
    eng.sketch()                                                # = indexlr              ntJoin:204-205
    eng.write_tsv(0, "ref.fa.k32.w1000.tsv")
    eng.build_graph()                                           # = read_minimizers' uniqueness + filter_minimizers + build_graph
    eng.write_dot("prefix.mx.dot")                              # = Ntjoin.print_graph   bin/ntjoin.py:25-62
All arrays returned are numpy COPIES (the library owns its buffers), hence picklable.
"""
import ctypes as C

import numpy as np

from . import capi


class MxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libntjoin_mx error {code}: {msg}")
        self.code = code


def _np(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).astype(dtype, copy=True)


class MxEngine:
    def __init__(self, k=32, w=1000, variant="v2", device=-1, stream=None, dense_only=False, drop_seq=False,
                 timing=False, cand_per_window=0, timing_fine=False, threads=0):
        self._lib = capi.load()
        self._h = C.c_void_p()
        cfg = capi.Config()
        cfg.struct_size = C.sizeof(capi.Config)
        cfg.k, cfg.w = int(k), int(w)
        cfg.variant = capi.VARIANT_V1_MIN if str(variant).lower() in ("v1", "min", "1") else capi.VARIANT_V2_SUM
        cfg.device = int(device)
        cfg.flags = ((capi.FLAG_DENSE_ONLY if dense_only else 0) | (capi.FLAG_DROP_SEQ if drop_seq else 0) |
                     (capi.FLAG_TIMING if timing else 0) | (capi.FLAG_TIMING_FINE if timing_fine else 0))
        cfg.stream = C.c_void_p(stream) if stream else None
        cfg.cand_per_window = int(cand_per_window)
        cfg.host_threads = int(threads)
        rc = self._lib.mxg_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise MxError(rc, (self._lib.mxg_last_error(None) or b"").decode())
        self.k, self.w = int(k), int(w)
        self._keep = []  # objects that must outlive the handle (borrowed device buffers)

    # -- plumbing -------------------------------------------------------------------------------------
    def _check(self, rc):
        if rc < 0:
            raise MxError(rc, (self._lib.mxg_last_error(self._h) or b"").decode())
        return rc

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.mxg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- assemblies -------------------------------------------------------------------------------------
    def add_fasta(self, name, weight, fasta_path):
        return self._check(self._lib.mxg_add_assembly_fasta(self._h, str(name).encode(), float(weight),
                                                            str(fasta_path).encode()))

    def add_fasta_shard(self, name, weight, fasta_path, shard, n_shards):
        """every record is registered (global indices) but only shard `shard` of `n_shards` is packed and sketched"""
        return self._check(self._lib.mxg_add_assembly_fasta_shard(self._h, str(name).encode(), float(weight),
                                                                  str(fasta_path).encode(), int(shard), int(n_shards)))

    def add_fasta_split(self, name, weight, fasta_path, shard, n_shards):
        """as add_fasta_shard, but the shards are equal BASE ranges: long records are sketched in pieces (with a halo) and the
        rank-ordered concatenation of the shards' sketches is the sketch of the whole file"""
        return self._check(self._lib.mxg_add_assembly_fasta_split(self._h, str(name).encode(), float(weight),
                                                                  str(fasta_path).encode(), int(shard), int(n_shards)))

    def assembly_continues(self, a):
        """split load: does this handle's first record continue a record begun on the shard before?"""
        rc = self._lib.mxg_assembly_continues(self._h, int(a))
        if rc < 0:
            raise ValueError("mxg_assembly_continues: bad arguments")
        return bool(rc)

    def xchg_pack(self, d_slot, head_bytes, caps):
        cc = np.ascontiguousarray(caps, dtype=np.uint64)
        return self._check(self._lib.mxg_xchg_pack(self._h, C.c_void_p(int(d_slot)), int(head_bytes),
                                                   cc.ctypes.data_as(C.POINTER(C.c_uint64))))

    def sketch_pack(self, d_slot, head_bytes, caps):
        """sketch every assembly and pack the sketches into the exchange slot without a host sync (finish with sketch_finish)"""
        cc = np.ascontiguousarray(caps, dtype=np.uint64)
        return self._check(self._lib.mxg_sketch_pack(self._h, C.c_void_p(int(d_slot)), int(head_bytes),
                                                     cc.ctypes.data_as(C.POINTER(C.c_uint64))))

    def sketch_pack_parts(self, d_parts, caps, rcaps):
        """sketch every assembly; each one's sketch is packed into its own buffer (d_parts[a]: 64-byte header + 12 * caps[a] bytes of
        hashes and positions + 4 * rcaps[a] bytes: the records' first entries) right behind its last kernel --
        part_packed_wait(a, stream) then lets a communication stream send it while the next assembly is sketched (finish with
        sketch_finish)"""
        cc = np.ascontiguousarray(caps, dtype=np.uint64)
        rc_ = np.ascontiguousarray(rcaps, dtype=np.uint64)
        pp = (C.c_void_p * len(d_parts))(*[int(x) for x in d_parts])
        return self._check(self._lib.mxg_sketch_pack_parts(self._h, pp, cc.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                           rc_.ctypes.data_as(C.POINTER(C.c_uint64))))

    def part_packed_wait(self, a, stream):
        """make `stream` (a raw hipStream_t) wait until assembly a's part is packed"""
        return self._check(self._lib.mxg_part_packed_wait(self._h, int(a), C.c_void_p(int(stream))))

    def xchg_unpack_graph_parts(self, d_all_parts, world, caps, rcaps, rec_offsets):
        """-> False when some rank's sketch did not fit its part (nothing usable), True: sketches unpacked + graph built"""
        cc = np.ascontiguousarray(caps, dtype=np.uint64)
        rc_ = np.ascontiguousarray(rcaps, dtype=np.uint64)
        ro = np.ascontiguousarray(rec_offsets, dtype=np.uint64)
        pp = (C.c_void_p * len(d_all_parts))(*[int(x) for x in d_all_parts])
        rc = self._lib.mxg_xchg_unpack_graph_parts(self._h, pp, int(world), cc.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                   rc_.ctypes.data_as(C.POINTER(C.c_uint64)), ro.ctypes.data_as(C.POINTER(C.c_uint64)))
        if rc == 1:
            return False
        self._check(rc)
        return True

    def sketch_finish(self):
        return self._check(self._lib.mxg_sketch_finish(self._h))

    def xchg_unpack_graph(self, d_all, world, slot_bytes, head_bytes, caps, rec_offsets):
        """-> False when some rank's sketch did not fit its slot (nothing usable), True: sketches unpacked + graph built"""
        cc = np.ascontiguousarray(caps, dtype=np.uint64)
        ro = np.ascontiguousarray(rec_offsets, dtype=np.uint64)
        rc = self._lib.mxg_xchg_unpack_graph(self._h, C.c_void_p(int(d_all)), int(world), int(slot_bytes), int(head_bytes),
                                             cc.ctypes.data_as(C.POINTER(C.c_uint64)), ro.ctypes.data_as(C.POINTER(C.c_uint64)))
        if rc == 1:
            return False
        self._check(rc)
        return True

    def assembly_shard(self, a):
        lo, hi = C.c_uint64(), C.c_uint64()
        self._check(self._lib.mxg_assembly_shard(self._h, int(a), C.byref(lo), C.byref(hi)))
        return int(lo.value), int(hi.value)

    def add_records(self, name, weight, records):
        """records: iterable of (id, sequence str/bytes)."""
        ids, seqs = [], []
        for rid, s in records:
            ids.append(str(rid).encode())
            seqs.append(s.encode("ascii") if isinstance(s, str) else bytes(s))
        blob = b"".join(seqs)
        offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        if seqs:
            offs[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
        arr_ids = (C.c_char_p * max(len(ids), 1))(*ids)
        buf = C.create_string_buffer(blob, len(blob) + 1)
        return self._check(self._lib.mxg_add_assembly_buffers(
            self._h, str(name).encode(), float(weight), C.cast(buf, C.c_void_p),
            offs.ctypes.data_as(C.POINTER(C.c_uint64)), arr_ids, len(seqs)))

    def add_packed_device(self, name, weight, d_ptr, rec_start, rec_len, ids=None, keepalive=None):
        """2-bit packed bases already in HBM (see include/ntjoin_mx.h); d_ptr is borrowed."""
        rs = np.ascontiguousarray(rec_start, dtype=np.uint64)
        rl = np.ascontiguousarray(rec_len, dtype=np.uint64)
        arr_ids = None
        if ids is not None:
            arr_ids = (C.c_char_p * len(ids))(*[str(i).encode() for i in ids])
        if keepalive is not None:
            self._keep.append(keepalive)
        return self._check(self._lib.mxg_add_assembly_packed_device(
            self._h, str(name).encode(), float(weight), C.c_void_p(int(d_ptr)),
            rs.ctypes.data_as(C.POINTER(C.c_uint64)), rl.ctypes.data_as(C.POINTER(C.c_uint64)), arr_ids, len(rs)))

    def plan_split(self, lengths, shard, n_shards):
        """pieces of shard `shard` of `n_shards` for N-free records of the given lengths -> (lo, hi, drop) arrays"""
        ln = np.ascontiguousarray(lengths, dtype=np.uint64)
        lo, hi = np.zeros(len(ln), dtype=np.uint64), np.zeros(len(ln), dtype=np.uint64)
        drop = np.zeros(len(ln), dtype=np.uint8)
        rc = self._lib.mxg_plan_split(ln.ctypes.data, len(ln), int(shard), int(n_shards), self.k, self.w, lo.ctypes.data, hi.ctypes.data,
                                      drop.ctypes.data)
        if rc != 0:
            raise ValueError("mxg_plan_split: bad arguments")
        return lo, hi, drop

    def add_packed_device_pieces(self, name, weight, d_ptr, rec_start, rec_len, lo, hi, drop, ids=None, keepalive=None):
        """sub-record shard of 2-bit packed bases in HBM (see include/ntjoin_mx.h); d_ptr is borrowed"""
        arrs = [np.ascontiguousarray(x, dtype=np.uint64) for x in (rec_start, rec_len, lo, hi)]
        dr = np.ascontiguousarray(drop, dtype=np.uint8)
        arr_ids = None
        if ids is not None:
            arr_ids = (C.c_char_p * len(ids))(*[str(i).encode() for i in ids])
        if keepalive is not None:
            self._keep.append(keepalive)
        return self._check(self._lib.mxg_add_assembly_packed_device_pieces(
            self._h, str(name).encode(), float(weight), C.c_void_p(int(d_ptr)), arrs[0].ctypes.data, arrs[1].ctypes.data,
            arrs[2].ctypes.data, arrs[3].ctypes.data, dr.ctypes.data, arr_ids, len(arrs[0])))

    def add_tsv(self, name, weight, tsv_path):
        return self._check(self._lib.mxg_add_assembly_tsv(self._h, str(name).encode(), float(weight),
                                                          str(tsv_path).encode()))

    def add_bin(self, name, weight, bin_path):
        """a sketch from the binary side-car written by write_sketch_bin (no text parsing)"""
        rc = self._lib.mxg_add_assembly_bin(self._h, str(name).encode(), float(weight), str(bin_path).encode())
        if rc < 0:
            msg = (self._lib.mxg_last_error(self._h) or b"").decode()
            if "cannot open" in msg:
                raise FileNotFoundError(msg)
            raise MxError(rc, msg)
        return rc

    def write_sketch_bin(self, a, path):
        self._check(self._lib.mxg_write_sketch_bin(self._h, int(a), str(path).encode()))

    def add_minimizers(self, name, weight, out_hash, pos, record, record_ids):
        hh = np.ascontiguousarray(out_hash, dtype=np.uint64)
        pp = np.ascontiguousarray(pos, dtype=np.uint32)
        rr = np.ascontiguousarray(record, dtype=np.uint32)
        ids = (C.c_char_p * max(len(record_ids), 1))(*[str(i).encode() for i in record_ids])
        return self._check(self._lib.mxg_add_assembly_minimizers(
            self._h, str(name).encode(), float(weight), hh.ctypes.data, pp.ctypes.data, rr.ctypes.data,
            len(hh), ids, len(record_ids)))

    @property
    def n_assemblies(self):
        return self._lib.mxg_num_assemblies(self._h)

    def assembly_name(self, a):
        return self._lib.mxg_assembly_name(self._h, a).decode()

    def n_records(self, a):
        return int(self._lib.mxg_num_records(self._h, int(a)))

    def assembly_weight(self, a):
        return float(self._lib.mxg_assembly_weight(self._h, int(a)))

    def record_ids(self, a, n_records):
        return [self._lib.mxg_record_id(self._h, a, r).decode() for r in range(int(n_records))]

    def record_lengths(self, a):
        return [int(self._lib.mxg_record_length(self._h, int(a), r)) for r in range(self.n_records(a))]

    # -- sketch stage -------------------------------------------------------------------------------------
    def sketch(self, assembly=-1):
        self._check(self._lib.mxg_sketch(self._h, int(assembly)))

    def sketch_graph(self):
        """sketch every assembly and build the graph in one call (one host sync in the common case)"""
        self._check(self._lib.mxg_sketch_graph(self._h))

    def get_sketch(self, a):
        v = capi.SketchView()
        self._check(self._lib.mxg_get_sketch(self._h, int(a), C.byref(v)))
        return {"out_hash": _np(v.out_hash, v.n, np.uint64), "pos": _np(v.pos, v.n, np.uint32),
                "record": _np(v.record, v.n, np.uint32), "forward": _np(v.forward, v.n, np.uint8),
                "record_first": _np(v.record_first, v.n_records + 1, np.uint64),
                "record_ids": self.record_ids(a, v.n_records)}

    def get_sketch_device(self, a):
        v = capi.SketchDView()
        self._check(self._lib.mxg_get_sketch_device(self._h, int(a), C.byref(v)))
        return {"n": int(v.n), "out_hash": v.out_hash or 0, "pos": v.pos or 0, "record": v.record or 0,
                "forward": v.forward or 0}

    def compute_strands(self, a):
        self._check(self._lib.mxg_compute_strands(self._h, int(a)))

    def sketch_size(self, a):
        v = capi.SketchDView()
        self._check(self._lib.mxg_get_sketch_device(self._h, int(a), C.byref(v)))
        return int(v.n)

    def pack_sketch_device(self, a, d_buf, nmax):
        self._check(self._lib.mxg_pack_sketch_device(self._h, int(a), C.c_void_p(int(d_buf)), int(nmax)))

    def set_sketch_gathered(self, a, d_allbuf, nmax, counts, rec_offsets):
        cc = np.ascontiguousarray(counts, dtype=np.uint64)
        ro = np.ascontiguousarray(rec_offsets, dtype=np.uint64)
        self._check(self._lib.mxg_set_sketch_gathered(self._h, int(a), C.c_void_p(int(d_allbuf)), len(cc), int(nmax),
                                                      cc.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                      ro.ctypes.data_as(C.POINTER(C.c_uint64))))

    def set_sketch_gathered_strided(self, a, d_allbuf, stride_bytes, nmax, counts, rec_offsets):
        """like set_sketch_gathered, rank r's packed buffer at d_allbuf + r * stride_bytes"""
        cc = np.ascontiguousarray(counts, dtype=np.uint64)
        ro = np.ascontiguousarray(rec_offsets, dtype=np.uint64)
        self._check(self._lib.mxg_set_sketch_gathered_strided(self._h, int(a), C.c_void_p(int(d_allbuf)), len(cc),
                                                              int(stride_bytes), int(nmax),
                                                              cc.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                              ro.ctypes.data_as(C.POINTER(C.c_uint64))))

    def set_sketch_device(self, a, d_hash, d_pos, d_record, d_forward, n):
        self._check(self._lib.mxg_set_sketch_device(self._h, int(a), C.c_void_p(int(d_hash)), C.c_void_p(int(d_pos)),
                                                    C.c_void_p(int(d_record)),
                                                    C.c_void_p(int(d_forward)) if d_forward else None, int(n)))

    def write_tsv(self, a, path, with_pos=True, with_strand=False, with_seq=True):
        self._check(self._lib.mxg_write_tsv(self._h, int(a), str(path).encode(), int(with_pos), int(with_strand),
                                            int(with_seq)))

    # -- graph stage ----------------------------------------------------------------------------------------
    def build_graph(self):
        self._check(self._lib.mxg_build_graph(self._h))

    def get_mx_flags(self, a):
        p = C.POINTER(C.c_uint8)()
        n = C.c_uint64()
        self._check(self._lib.mxg_get_mx_flags(self._h, int(a), C.byref(p), C.byref(n)))
        return _np(p, n.value, np.uint8)

    def get_graph(self):
        g = capi.GraphView()
        self._check(self._lib.mxg_get_graph(self._h, C.byref(g)))
        A, nv, ne = int(g.n_assemblies), int(g.n_vertices), int(g.n_edges)
        return {"n_assemblies": A,
                "vertex_hash": _np(g.vertex_hash, nv, np.uint64),
                "vertex_pos": _np(g.vertex_pos, A * nv, np.uint32).reshape(A, nv),
                "vertex_record": _np(g.vertex_record, A * nv, np.uint32).reshape(A, nv),
                "edge_u": _np(g.edge_u, ne, np.uint32), "edge_v": _np(g.edge_v, ne, np.uint32),
                "edge_support": _np(g.edge_support, ne, np.uint32),
                "edge_weight": _np(g.edge_weight, ne, np.float64)}

    def find_paths(self, n=1):
        """linear paths through the graph (ntJoin's -n = minimum edge weight): list of (component id, [vertex indices])"""
        v = capi.PathsView()
        self._check(self._lib.mxg_find_paths(self._h, int(n), C.byref(v)))
        npaths = int(v.n_paths)
        first = _np(v.path_first, npaths + 1, np.uint64)
        verts = _np(v.path_vertex, int(first[-1]) if npaths else 0, np.uint32)
        comp = _np(v.path_component, npaths, np.uint32)
        self.n_components = int(v.n_components)
        return [(int(comp[i]), verts[int(first[i]):int(first[i + 1])].tolist()) for i in range(npaths)]

    def path_segments(self, a):
        """runs of consecutive path vertices on the same record of assembly a (after find_paths): dict of arrays
        path, record, first (offset into the concatenated paths), n, min_pos, max_pos, inc, dec"""
        v = capi.SegmentsView()
        self._check(self._lib.mxg_path_segments(self._h, int(a), C.byref(v)))
        n = int(v.n_segments)
        st = _np(v.seg_stat, 5 * n, np.uint32).reshape(n, 5)
        return {"path": _np(v.seg_path, n, np.uint32), "record": _np(v.seg_record, n, np.uint32),
                "first": _np(v.seg_first, n, np.uint32), "n": st[:, 0], "min_pos": st[:, 1], "max_pos": st[:, 2],
                "inc": st[:, 3], "dec": st[:, 4]}

    def mx_extremes(self, a):
        """per record of assembly a: (min, max) position over its graph vertices; None for records without one"""
        mn, mx, n = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.c_uint64()
        self._check(self._lib.mxg_mx_extremes(self._h, int(a), C.byref(mn), C.byref(mx), C.byref(n)))
        lo, hi = _np(mn, n.value, np.uint32), _np(mx, n.value, np.uint32)
        return [None if l > h_ else (int(l), int(h_)) for l, h_ in zip(lo.tolist(), hi.tolist())]

    def dot_part_format(self, part, n_parts):
        """format part `part` of `n_parts` of the .mx.dot into memory -> (vertex segment bytes, edge segment bytes)"""
        b = (C.c_uint64 * 2)()
        self._check(self._lib.mxg_dot_part_format(self._h, int(part), int(n_parts), b))
        return int(b[0]), int(b[1])

    def dot_part_write(self, path, v_off, e_off, first, last):
        self._check(self._lib.mxg_dot_part_write(self._h, str(path).encode(), int(v_off), int(e_off), int(bool(first)), int(bool(last))))

    def write_dot(self, path):
        self._check(self._lib.mxg_write_dot(self._h, str(path).encode()))

    # -- stats ------------------------------------------------------------------------------------------------
    def stats(self):
        s = capi.Stats()
        s.struct_size = C.sizeof(capi.Stats)
        self._check(self._lib.mxg_get_stats(self._h, C.byref(s)))
        return {f: getattr(s, f) for f, _ in capi.Stats._fields_ if f not in ("struct_size", "reserved")}

    def knobs(self):
        """the MXG_* environment knobs this handle has read and found set: 'NAME=value ...' (parsed once per handle)"""
        buf = C.create_string_buffer(4096)
        self._lib.mxg_knobs(self._h, buf, 4096)
        return buf.value.decode()

    def reset_timers(self):
        self._check(self._lib.mxg_reset_timers(self._h))
